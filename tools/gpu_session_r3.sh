#!/bin/bash
# Round-3 GPU session: tests with both kernels, then same-call A/B bench lines (classic vs pipelined kernel).
# Variant libraries are built ON the GPU box (hipcc is there; adsbdec_amd/lib_var/ does not travel).
#   tools/gpu_session_r3.sh tests | ab "<pipe> <K> <groups/CU> <lib|variant> [variant hipcc flags]" ...
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
what=${1:-all}; shift
line() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['launch_ms'], r['frac'])" 2>&1 | tail -1; }
if [[ $what == all || $what == tests ]]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
  tail -15 $O/pytest_gpu.log
fi
if [[ $what == all || $what == ab ]]; then
  cfgs=("$@")
  for cfg in "${cfgs[@]}"; do
    read -r p k g v flags <<< "$cfg"
    if [ "$v" != lib ] && [ ! -f adsbdec_amd/lib_var/$v/libadsbdec_amd.so ]; then
      tools/build_variant.sh $v $flags > /dev/null 2>&1 || echo "variant $v failed to build"
    fi
  done
  for rep in 1 2; do
    for cfg in "${cfgs[@]}"; do
      read -r p k g v flags <<< "$cfg"
      lib=adsbdec_amd/lib/libadsbdec_amd.so; [ "$v" != lib ] && lib=adsbdec_amd/lib_var/$v/libadsbdec_amd.so
      ADSB_LIB_PATH=$lib ADSB_PIPE=$p ADSB_PASSES=$k ADSB_PIPE_GROUPS_PER_CU=$g timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/ab_p${p}_k${k}_g${g}_$v.json 2> $O/ab_p${p}_k${k}_g${g}_$v.err
      echo "pipe=$p K=$k groups/CU=$g $v: $(line $O/ab_p${p}_k${k}_g${g}_$v.json)"
    done
  done | tee -a $O/ab_runs.txt
fi
