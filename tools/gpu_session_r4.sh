#!/bin/bash
# One gpurun session of round 4: tools/gpu_session_r4.sh <step> ...   (every output lands in gpurun_out/)
#   tests        the whole -m gpu suite
#   bench        the default bench line -> r4_bench.json
#   shard        bench.py --mode shard: N = 1 device-resident 2 Gi, and the host-fed / multi-handle plumbing runs
#   prof         rocprofv3 evidence set (tools/profile_session.sh r4)
#   fuzz [s]     tools/fuzz_parity.py for s seconds (default 300)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
run() { out=$1; shift; timeout 1500 python bench.py "$@" > $O/$out.json 2> $O/$out.err; echo "$out: exit $? $(python -c "import json; d=json.load(open('$O/$out.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])" 2>&1 | tail -1)"; }
while [ $# -gt 0 ]; do
  case $1 in
    tests) timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/r4_gpu_tests.txt 2>&1; echo "tests exit $?"; tail -25 $O/r4_gpu_tests.txt;;
    bench) run r4_bench;;
    shard)
      run r4_bench_shard_N1_2Gi --mode shard --steps 20 --warmup 3
      run r4_bench_shard_N1_2Gi_stats --mode shard --steps 10 --warmup 2 --stats
      run r4_bench_shard_8handles_one_device_2Gi --mode shard --gpus 8 --one-device-test --steps 10 --warmup 2 --stats
      run r4_bench_shard_host_fed_N1_512Mi --mode shard --shard-source host --steps 5 --warmup 1 --stats
      run r4_bench_shard_host_fed_4handles_512Mi --mode shard --gpus 4 --one-device-test --shard-source host --steps 5 --warmup 1 --stats
      run r4_bench_shard_file_fed_4handles_512Mi --mode shard --gpus 4 --one-device-test --shard-source file --steps 5 --warmup 1 --stats
      run r4_bench_stream_N8_one_device_plumbing --gpus 8 --one-device-test --samples 67108864 --steps 10 --warmup 2 --no-extras
      ;;
    prof) bash tools/profile_session.sh r4;;
    fuzz) shift; FZ=${1:-300}; timeout $((FZ + 300)) python tools/fuzz_parity.py --seconds $FZ --seed 910000 > $O/r4_fuzz.txt 2>&1; echo "fuzz exit $?"; tail -2 $O/r4_fuzz.txt | cut -c1-900;;
  esac
  shift
done
