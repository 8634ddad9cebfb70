cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python bench.py > $O/r3_bench.json 2> $O/r3_bench.err; echo "exit $?"
python -c "import json; d=json.load(open('$O/r3_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], 'stats', d['with_stats']['ms_per_step'], 'two threads', d['with_stats']['host_threads_2']['ms_per_step'], d['cpu_baseline']['value'])"
