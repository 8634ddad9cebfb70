#!/bin/bash
# round 5, third GPU session (exact reservations, reader prefetch, bounded waits): the default suite on the build with one record per run of copies; same-box A/B against round 4's
# library (adsbdec_amd/lib_ab/r4base) on the sparse headline and on the dense captures; the opt-in 2^32-4 test three times.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
t0=$(date +%s)
( time python -m adsbdec_amd._build --force ) > $O/r5c_build.txt 2>&1
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/r5c_gpu_tests.txt 2>&1
echo "suite rc=$? wall=$(( $(date +%s) - t0 )) s since the start of the from-source build" >> $O/r5c_gpu_tests.txt
tail -8 $O/r5c_gpu_tests.txt
{
for rep in 1 2; do
  for v in r4base new; do
    if [ $v = new ]; then unset ADSB_LIB_PATH; else export ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so; fi
    echo "== $v (rep $rep): bench.py --steps 1000 --no-extras --no-cpu-baseline"
    python bench.py --steps 1000 --warmup 50 --no-extras --no-cpu-baseline 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); print({k: j.get(k) for k in ('value', 'ms_per_step')}, 'kernel_ms', j['roofline'].get('kernel_ms'), 'frac', j['roofline'].get('frac'))
"
    echo "== $v (rep $rep): tools/dense_probe.py"
    python tools/dense_probe.py 2>&1 | grep -v amdgpu.ids
  done
done
unset ADSB_LIB_PATH
} > $O/r5c_ab.txt 2>&1
tail -40 $O/r5c_ab.txt
for i in 1 2 3; do
  ( time timeout 600 python -m pytest tests/test_gpu_full_configs.py -m gpu --gpu-big -k counter_limit -x -q ) > $O/r5c_big_$i.txt 2>&1
  echo "big run $i rc=$?" | tee -a $O/r5c_big_$i.txt
done
