#!/bin/bash
# The bench records of a round's final build, in one gpurun call: every file lands in gpurun_out/ under the name it has
# in profiles/ (copy them over afterwards, then `python tools/design_numbers.py`).   tools/final_records.sh [fuzz seconds]
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
FZ=${1:-300}
run() { out=$1; shift; timeout 1200 python bench.py "$@" > $O/$out.json 2> $O/$out.err; echo "$out: exit $? $(python -c "import json; d=json.load(open('$O/$out.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])" 2>&1 | tail -1)"; }
run r3_bench
run r3_bench_shard_N1_2Gi --mode shard --steps 20 --warmup 3
run r3_bench_shard_N1_2Gi_gather_path --mode shard --shard-path gather --steps 10 --warmup 2
run r3_bench_shard_N2_one_device_plumbing --gpus 2 --one-device-test --mode shard --samples 536870912 --steps 10 --warmup 2
run r3_bench_stream_N2_one_device_plumbing --gpus 2 --one-device-test --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run r3_bench_stream_2Gi --samples 2147483632 --steps 20 --warmup 3 --no-cpu-baseline --no-extras
timeout $((FZ + 300)) python tools/fuzz_parity.py --seconds $FZ --seed 900000 > $O/r3_fuzz_final.txt 2>&1; echo "fuzz exit $?"; tail -2 $O/r3_fuzz_final.txt | cut -c1-600
