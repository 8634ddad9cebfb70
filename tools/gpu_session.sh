#!/bin/bash
# One gpurun session: tests, bench (shipped build + A/B variants), CLI timing. Everything lands in gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
what=${1:-all}
if [[ $what == all || $what == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi
if [[ $what == all || $what == bench ]]; then
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cat $O/bench.json; tail -3 $O/bench.err
  for v in adsbdec_amd/lib_var/*/libadsbdec_amd.so; do
    [ -f "$v" ] || continue
    n=$(basename $(dirname $v))
    ADSB_LIB_PATH=$v timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
    echo "variant $n: $(python -c "import json,sys; d=json.load(open('$O/bench_$n.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])" 2>&1 | tail -1)"
  done
  timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_again.json 2>/dev/null
  echo "shipped again: $(python -c "import json; d=json.load(open('$O/bench_again.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])")"
fi
if [[ $what == all || $what == cli ]]; then
  python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from tools import gen_signal as G
x, _ = G.sparse_capture(256 << 20, 12000, seed=3)
x.tofile('/tmp/cap512.u16')
PY
  for i in 1 2 3; do
    for reg in 1 0; do
      cat /tmp/cap512.u16 > /dev/null; s=$(date +%s%N); ADSB_CLI_TIMING=1 ADSB_CLI_REGISTER=$reg adsbdec_amd/lib/adsbdec_amd_cli -f /tmp/cap512.u16 > /tmp/cli.out 2> /tmp/cli.err; e=$(date +%s%N); echo "cli register=$reg wall $(( (e - s) / 1000000 )) ms, $(wc -l < /tmp/cli.out) frames; $(grep timing /tmp/cli.err)"
    done
  done | tee $O/cli_timing.txt
fi
if [[ $what == all || $what == shard ]]; then
  timeout 900 python bench.py --mode shard --steps 20 --warmup 3 > $O/bench_shard1.json 2> $O/bench_shard1.err; echo "shard N=1 exit $?"; cat $O/bench_shard1.json; tail -3 $O/bench_shard1.err
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --mode shard --samples $((512<<20)) --steps 20 --warmup 3 --one-device-test > $O/bench_shard2.json 2> $O/bench_shard2.err; echo "shard N=2 one-device exit $?"; cat $O/bench_shard2.json; tail -3 $O/bench_shard2.err
fi
