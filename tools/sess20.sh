O=gpurun_out
bash tools/gpu_session_r3.sh tests
timeout 900 python bench.py > $O/r3_bench.json 2> $O/r3_bench.err; echo "bench exit $?"
timeout 600 python bench.py --mode shard --steps 20 --warmup 3 > $O/r3_bench_shard_N1_2Gi.json 2> $O/shard1.err; echo "shard N=1 exit $?"
timeout 600 python bench.py --mode shard --shard-path gather --steps 10 --warmup 2 --no-cpu-baseline > $O/r3_bench_shard_N1_2Gi_gather_path.json 2> $O/shard1g.err; echo "gather exit $?"
timeout 600 python bench.py --gpus 2 --one-device-test --mode shard --samples $((512<<20)) --steps 10 --warmup 2 > $O/r3_bench_shard_N2_one_device_plumbing.json 2> $O/shard2.err; echo "shard N=2 exit $?"
timeout 600 python bench.py --gpus 2 --one-device-test --steps 20 --warmup 3 --no-extras > $O/r3_bench_stream_N2_one_device_plumbing.json 2> $O/stream2.err; echo "stream N=2 exit $?"
timeout 600 python bench.py --samples $((2<<30)) --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/r3_bench_stream_2Gi.json 2> $O/stream2gi.err; echo "stream 2Gi exit $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_bench*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['config'].get('rank0_serial_us'), d['config'].get('deqframe_calls_walked_by_rank0'))
    except Exception as e: print(f, 'ERR', e)
PY
