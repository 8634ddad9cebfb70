cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['launch_ms'], r['frac'], d['config'].get('cpu_binding_rank0'))" 2>&1 | tail -1; }
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head; numactl -H 2>/dev/null | head -5; lscpu | grep -i numa
for rep in 1 2 3; do
  for b in off on; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --bind-cpu $b > $O/bind_$b.json 2> $O/bind_$b.err
    echo "bind=$b: $(line $O/bind_$b.json)"
  done
done | tee $O/bind_runs.txt
timeout 600 python bench.py --gpus 2 --one-device-test --no-cpu-baseline --no-extras --steps 50 > $O/bind_n2.json 2> $O/bind_n2.err; echo "N=2 one device: $(line $O/bind_n2.json)" | tee -a $O/bind_runs.txt
timeout 600 python bench.py --gpus 2 --one-device-test --mode shard --steps 20 > $O/bind_n2s.json 2> $O/bind_n2s.err; python -c "import json; d=json.load(open('$O/bind_n2s.json')); print('N=2 shard:', d['value'], d['ms_per_step'], d['config'].get('cpu_binding_rank0'), d['config'].get('parity'))" | tee -a $O/bind_runs.txt
