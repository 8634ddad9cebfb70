O=gpurun_out
bash tools/gpu_session_r3.sh tests
ADSB_ALT_STREAMS=1 timeout 600 python bench.py --mode shard --steps 20 --warmup 3 > $O/shard1.json 2> $O/shard1.err; echo "shard N=1 exit $?"; python -c "import json; d=json.load(open('$O/shard1.json')); print(d['value'], d['ms_per_step'], d['config'])"
timeout 600 python bench.py --gpus 2 --one-device-test --mode shard --samples $((512<<20)) --steps 10 --warmup 2 > $O/shard2.json 2> $O/shard2.err; echo "shard N=2 one device exit $?"; python -c "import json; d=json.load(open('$O/shard2.json')); print(d['value'], d['ms_per_step'], d['config'])"
timeout 600 python bench.py --gpus 2 --one-device-test --steps 20 --warmup 3 --no-extras > $O/stream2.json 2> $O/stream2.err; echo "stream N=2 one device exit $?"; python -c "import json; d=json.load(open('$O/stream2.json')); print(d['value'], d['ms_per_step'], d['config'])"
