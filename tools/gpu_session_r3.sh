#!/bin/bash
# Round-3 GPU session: tests with both kernels, then same-call A/B bench lines (classic vs pipelined kernel).
# Variant libraries are built ON the GPU box (hipcc is there; adsbdec_amd/lib_var/ does not travel).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
what=${1:-all}
line() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['launch_ms'], r['frac'])" 2>&1 | tail -1; }
if [[ $what == all || $what == tests ]]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -15 $O/pytest_gpu.log
fi
if [[ $what == all || $what == ab ]]; then
  tools/build_variant.sh pw4 -DADSB_PIPE_WAVES=4 > /dev/null 2>&1
  for rep in 1 2; do
    for cfg in "0 7 4 lib" "1 5 4 lib" "1 4 4 lib" "1 5 3 lib" "1 3 4 lib" "1 7 3 pw4" "1 5 3 pw4" "1 6 3 pw4"; do
      set -- $cfg
      lib=adsbdec_amd/lib/libadsbdec_amd.so; [ $4 != lib ] && lib=adsbdec_amd/lib_var/$4/libadsbdec_amd.so
      ADSB_LIB_PATH=$lib ADSB_PIPE=$1 ADSB_PASSES=$2 ADSB_PIPE_GROUPS_PER_CU=$3 timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/ab_p$1_k$2_g$3_$4.json 2> $O/ab_p$1_k$2_g$3_$4.err
      echo "pipe=$1 K=$2 groups/CU=$3 $4: $(line $O/ab_p$1_k$2_g$3_$4.json)"
    done
  done | tee $O/ab_runs.txt
fi
