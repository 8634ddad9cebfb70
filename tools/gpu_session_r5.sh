#!/bin/bash
# One gpurun session of round 5: tools/gpu_session_r5.sh <step> ...   (every output lands in gpurun_out/)
#   build        from-source build, timed
#   tests        the default -m gpu suite (cheap-first order, per-test limits: pytest.ini, tests/conftest.py)
#   big          the opt-in gpu_big test (2^32 - 4 samples), three times
#   bench        the default bench line -> r5_bench.json
#   shard        bench.py --mode shard runs that are in profiles/r5_bench_shard_*
#   prof         rocprofv3 evidence set (tools/profile_session.sh r5), plus the statistics and dense variants
#   ab           same-box A/B of the library builds under adsbdec_amd/lib_ab/ against the tree's (sparse headline + dense captures)
#   hostpath     tools/hostpath_bench.cpp on this box's CPU
#   ingest       tools/ingest_probe.hip
#   cli          the C host program's per-stage timing on a 510 MiB tmpfs capture
#   fuzz [s]     tools/fuzz_parity.py for s seconds (default 300)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O tools/bin; export TMPDIR=/tmp
run() { out=$1; shift; timeout 1500 python bench.py "$@" > $O/$out.json 2> $O/$out.err; echo "$out: exit $? $(python -c "import json; d=json.load(open('$O/$out.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])" 2>&1 | tail -1)"; }
bench_line() { python bench.py --steps 1000 --warmup 50 --no-extras --no-cpu-baseline 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print({k: j.get(k) for k in ('value', 'ms_per_step')}, 'kernel ms', r.get('launch_ms'), 'frac', r.get('frac'))
"; }
while [ $# -gt 0 ]; do
  case $1 in
    build) ( time python -m adsbdec_amd._build --force ) > $O/r5_build.txt 2>&1; tail -4 $O/r5_build.txt;;
    tests) t0=$(date +%s); ( time timeout 900 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/r5_gpu_tests.txt 2>&1
           echo "suite exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/r5_gpu_tests.txt; tail -24 $O/r5_gpu_tests.txt;;
    big) for i in 1 2 3; do ( time timeout 600 python -m pytest tests/test_gpu_full_configs.py -m gpu --gpu-big -k counter_limit -x -q ) > $O/r5_gpu_big_$i.txt 2>&1; echo "big run $i exit $?" | tee -a $O/r5_gpu_big_$i.txt; done;;
    bench) run r5_bench;;
    shard)
      run r5_bench_shard_N1_2Gi --mode shard --steps 20 --warmup 3
      run r5_bench_shard_N1_2Gi_stats --mode shard --steps 10 --warmup 2 --stats
      run r5_bench_shard_8handles_one_device_2Gi --mode shard --gpus 8 --one-device-test --steps 10 --warmup 2 --stats
      run r5_bench_shard_host_fed_N1_512Mi --mode shard --shard-source host --steps 5 --warmup 1 --stats
      run r5_bench_shard_file_fed_N1_512Mi --mode shard --shard-source file --steps 5 --warmup 1 --stats
      run r5_bench_shard_host_fed_4handles_512Mi --mode shard --gpus 4 --one-device-test --shard-source host --steps 5 --warmup 1 --stats
      run r5_bench_shard_file_fed_4handles_512Mi --mode shard --gpus 4 --one-device-test --shard-source file --steps 5 --warmup 1 --stats
      run r5_bench_stream_N8_one_device_plumbing --gpus 8 --one-device-test --samples 67108864 --steps 10 --warmup 2 --no-extras
      ;;
    prof) bash tools/profile_session.sh r5; bash tools/profile_session.sh r5_stats --stats; bash tools/profile_session.sh r5_dense --dense;;
    ab) { for rep in 1 2; do for v in $(ls adsbdec_amd/lib_ab 2>/dev/null) tree; do
            if [ $v = tree ]; then unset ADSB_LIB_PATH; else export ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so; fi
            echo "== $v (rep $rep): bench.py --steps 1000 --no-extras --no-cpu-baseline"; bench_line
            echo "== $v (rep $rep): tools/dense_probe.py"; python tools/dense_probe.py 2>&1 | grep -v "amdgpu.ids\|all_candidates"
          done; done; unset ADSB_LIB_PATH; } > $O/r5_ab.txt 2>&1; tail -40 $O/r5_ab.txt;;
    hostpath) { lscpu | grep -E "Model name|Socket|NUMA node|L3|L2"; g++ -O2 -std=c++17 tools/hostpath_bench.cpp -o tools/bin/hostpath_bench -pthread 2>/dev/null
                tools/bin/hostpath_bench | grep -v "rep [0-2]"; } > $O/r5_hostpath.txt 2>&1; tail -30 $O/r5_hostpath.txt;;
    ingest) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/bin/ingest_probe tools/ingest_probe.hip -lpthread > /dev/null 2>&1
            tools/bin/ingest_probe > $O/r5_ingest.txt 2>&1; tail -40 $O/r5_ingest.txt;;
    cli) { python - <<'PY'
import numpy as np
rng = np.random.default_rng(5)
x = (2048 + rng.normal(0, 20, 255 << 20)).clip(0, 4095).astype(np.uint16)
x.tofile("/dev/shm/r5_cap.u16")
PY
           for i in 1 2 3; do ADSB_CLI_TIMING=2 adsbdec_amd/lib/adsbdec_amd_cli -f /dev/shm/r5_cap.u16 2>&1 >/dev/null | grep "push\|timing"; done
           echo "-G 1"; for i in 1 2 3; do ADSB_CLI_TIMING=1 adsbdec_amd/lib/adsbdec_amd_cli -G 1 -f /dev/shm/r5_cap.u16 2>&1 >/dev/null | grep timing; done
           echo "-G 0,0"; for i in 1 2 3; do ADSB_CLI_TIMING=1 adsbdec_amd/lib/adsbdec_amd_cli -G 0,0 -f /dev/shm/r5_cap.u16 2>&1 >/dev/null | grep timing; done
           rm -f /dev/shm/r5_cap.u16; } > $O/r5_cli.txt 2>&1; grep timing $O/r5_cli.txt;;
    fuzz) shift; FZ=${1:-300}; timeout $((FZ + 300)) python tools/fuzz_parity.py --seconds $FZ --seed 920000 > $O/r5_fuzz.txt 2>&1; echo "fuzz exit $?"; tail -2 $O/r5_fuzz.txt | cut -c1-900;;
  esac
  shift
done
