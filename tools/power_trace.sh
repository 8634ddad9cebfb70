#!/bin/bash
# tools/power_trace.sh : socket power and shader clock (rocm-smi, 2 samples/s) while bench.py runs ~20 s of steps.
# Prints the samples with the time since the bench started, and the bench line's kernel time.
python bench.py --no-cpu-baseline --no-extras --steps 100000 --warmup 50 > /tmp/pt_bench.json 2>/dev/null &
bp=$!
t0=$(date +%s.%N)
while kill -0 $bp 2>/dev/null; do
  s=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1 MHz/; s/.*Power \(W\): ([0-9.]+).*/power \1 W/' | tr '\n' ' ')
  printf "%6.1f s  %s\n" "$(echo "$(date +%s.%N) $t0" | awk '{print $1-$2}')" "$s"
  sleep 0.4
done
python -c "import json; d=json.load(open('/tmp/pt_bench.json')); print('kernel ms', d['roofline']['launch_ms'], 'step ms', d['ms_per_step'])"
