O=gpurun_out
bash tools/gpu_session_r3.sh tests
timeout 900 python bench.py > $O/r3_bench.json 2> $O/r3_bench.err; echo "bench exit $?"
timeout 600 python bench.py --mode shard --steps 20 --warmup 3 > $O/r3_bench_shard_N1_2Gi.json 2> $O/shard1.err; echo "shard N=1 exit $?"
timeout 600 python bench.py --samples $((2<<30)) --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/r3_bench_stream_2Gi.json 2> $O/stream2gi.err; echo "stream 2Gi exit $?"
python - <<'PY'
import json,glob
for f in ('gpurun_out/r3_bench.json','gpurun_out/r3_bench_shard_N1_2Gi.json','gpurun_out/r3_bench_stream_2Gi.json'):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline'].get('launches_overlap'))
d=json.load(open('gpurun_out/r3_bench.json')); print(d['value_dropin']['ms_per_step'], d['value_cold']['ms_each_step'], d['roofline_valu']['clock_ghz'], d['roofline_valu']['frac'], d['roofline_valu']['frac_at_peak_clock'])
PY
