cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reader_thread" 2>&1 | tail -2
for rep in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline > $O/b27_$rep.json 2> $O/b27_$rep.err
  python -c "import json; d=json.load(open('$O/b27_$rep.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], 'stats', d['with_stats']['ms_per_step'], 'two threads', d['with_stats']['host_threads_2']['ms_per_step'])"
done | tee $O/b27_runs.txt
timeout 1200 python bench.py --gpus 2 --one-device-test --mode shard --samples 536870912 --steps 10 --warmup 2 > $O/r3_bench_shard_N2_one_device_plumbing.json 2> $O/r3_bench_shard_N2_one_device_plumbing.err; echo "N2 shard exit $?"
python -c "import json; d=json.load(open('$O/r3_bench_shard_N2_one_device_plumbing.json')); print(d['value'], d['ms_per_step'], d['config']['samples_total'])"
