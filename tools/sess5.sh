bash tools/kb_session.sh "pipeNoB_roles 1 5 - -DADSB_PIPE_ABLATE=1" "pipeNoB_fixed 1 5 - -DADSB_PIPE_ABLATE=1 -DADSB_PIPE_ROLES=0" "pipe_roles 1 5 -" "pipe_fixed 1 5 - -DADSB_PIPE_ROLES=0" "classic 0 7 -"
O=gpurun_out
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench exit $?"; tail -3 $O/bench_full.err; head -c 1500 $O/bench_full.json; echo
for alt in 1 0; do
  ADSB_ALT_STREAMS=$alt timeout 600 python bench.py --mode shard --steps 20 --warmup 3 --no-cpu-baseline > $O/shard1_alt$alt.json 2> $O/shard1_alt$alt.err; echo "shard N=1 resolved alt=$alt exit $?: $(python -c "import json; d=json.load(open('$O/shard1_alt$alt.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['config']['rank0_serial_us'])")"
done
ADSB_ALT_STREAMS=1 timeout 600 python bench.py --mode shard --shard-path gather --steps 20 --warmup 3 --no-cpu-baseline > $O/shard1_gather.json 2> $O/shard1_gather.err; echo "shard N=1 gather exit $?: $(python -c "import json; d=json.load(open('$O/shard1_gather.json')); print(d['value'], d['ms_per_step'])")"
for alt in 1 0; do
  ADSB_ALT_STREAMS=$alt timeout 600 python bench.py --samples $((2<<30)) --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/stream2Gi_alt$alt.json 2> $O/stream2Gi_alt$alt.err; echo "stream 2Gi alt=$alt exit $?: $(python -c "import json; d=json.load(open('$O/stream2Gi_alt$alt.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])")"
done
