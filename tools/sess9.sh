O=gpurun_out
bash tools/gpu_session_r3.sh tests
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full.json'))
print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])
print('dropin', d['value_dropin']); print('with_stats', d['with_stats']['ms_per_step'], d['with_stats']['launch_ms'])
print('cold', d['value_cold']); print('e2e', d['e2e_host_fed']); print('multi', d['multi_stream_host_fed']); print('cli', d['cli_whole_process'])
PY
