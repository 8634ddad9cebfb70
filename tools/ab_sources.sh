#!/bin/bash
# Same-box A/B of whole source trees: tools/ab_sources.sh [--rounds N] name=csrc_dir [name=csrc_dir ...]
# Builds every directory (a copy of adsbdec_amd/csrc, e.g. `git show <rev>:adsbdec_amd/csrc/scan_kernel.hip` over a copy, made
# BEFORE the gpurun call: the GPU box has no .git) into adsbdec_amd/lib_var/<name>/ and runs bench.py --steps 300 --no-extras on
# each (plus $BENCH_FLAGS, e.g. --stats), N rounds, order reversed every other round.  The directories must lie two levels below a directory that holds include/
# (the sources include "../../include/adsbdec_amd.h").  profiles/r4_ab_runs.txt section 7 was made with it.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rounds=4; [ "$1" = --rounds ] && { rounds=$2; shift 2; }
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None"
names=()
for spec in "$@"; do
  name=${spec%%=*}; src=${spec#*=}; names+=("$name")
  out=adsbdec_amd/lib_var/$name; mkdir -p "$out"
  for s in scan_kernel decoder; do /opt/rocm/bin/hipcc $FLAGS -c "$src/$s.hip" -o "$out/$s.o" || exit 1; done
  gcc -O2 -fPIC -c "$src/format.c" -o "$out/format.o" || exit 1
  for s in multi host_abi; do g++ -O2 -fPIC -std=c++17 -pthread -c "$src/$s.cpp" -o "$out/$s.o" || exit 1; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libadsbdec_amd.so" "$out"/*.o -lm -lpthread || exit 1
done
for round in $(seq 1 "$rounds"); do
  order=("${names[@]}")
  if [ $((round % 2)) = 0 ]; then order=(); for ((i=${#names[@]}-1; i>=0; i--)); do order+=("${names[i]}"); done; fi
  for v in "${order[@]}"; do
    ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_var/$v/libadsbdec_amd.so timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-extras $BENCH_FLAGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('round $round $v: ms_per_step', d['ms_per_step'], 'launch_ms', r['launch_ms'], 'frac', r['frac'])"
  done
done
