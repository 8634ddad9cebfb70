#!/bin/bash
# round 5, fourth GPU session: same-box A/B of the library's revisions (adsbdec_amd/lib_ab/*) on the sparse headline and the
# dense captures; where the host's time goes at the channel's capacity (tuning build, ADSB_DEBUG_HOST); file ingest probe.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out tools/bin
O=gpurun_out
bench_line() {
  python bench.py --steps 1000 --warmup 50 --no-extras --no-cpu-baseline 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']
        print({k: j.get(k) for k in ('value', 'ms_per_step')}, 'kernel us', round(2 * 268435440 / (r['achieved'] * 1e9) * 1e6 / 1, 2) if r.get('achieved') else None, 'frac', r.get('frac'))
"
}
{
for rep in 1 2; do
  for v in r4base v1 v2 cur_nopf new; do
    if [ $v = new ]; then unset ADSB_LIB_PATH; else export ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so; fi
    echo "== $v (rep $rep): bench.py --steps 1000 --no-extras --no-cpu-baseline"
    bench_line
  done
done
for v in r4base new cur_nopf; do
  if [ $v = new ]; then unset ADSB_LIB_PATH; else export ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so; fi
  echo "== $v: tools/dense_probe.py"
  python tools/dense_probe.py 2>&1 | grep -v "amdgpu.ids\|all_candidates"
done
export ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/cur_tuning/libadsbdec_amd.so
echo "== cur_tuning, ADSB_DEBUG_HOST=1: where the host's time goes (dense10, then sparse)"
ADSB_DEBUG_HOST=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -60
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from adsbdec_amd import capi
from bench import make_dense10, make_workload, bind_near_gpu
torch.cuda.set_device(0); bind_near_gpu(torch, 0)
n = (256 << 20); n -= n % 28
for name, x in (("dense10", make_dense10(torch, n, 101)), ("sparse", make_workload(torch, n, seed=1)[0])):
    torch.cuda.synchronize()
    for kw in (dict(), dict(host_threads=2)):
        d = capi.Decoder(df18=True, profile=True, **kw)
        for i in range(6):
            sys.stderr.write(f"--- {name} {kw} step {i}\n"); sys.stderr.flush()
            d.decode_device_raw(x.data_ptr(), x.numel())
        d.close()
PY
unset ADSB_LIB_PATH
} > $O/r5d_ab.txt 2>&1
tail -70 $O/r5d_ab.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/bin/ingest_probe tools/ingest_probe.hip -lpthread > $O/r5d_ingest.txt 2>&1
tools/bin/ingest_probe >> $O/r5d_ingest.txt 2>&1
grep -v warning $O/r5d_ingest.txt | grep -v "^ *[0-9]* |\|\^~" | tail -60
