#!/bin/bash
# same-box A/B: the scan kernel before (fb379c8) and after the overflow-round restructuring
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -Iadsbdec_amd/csrc"
for v in old new; do
  out=adsbdec_amd/lib_var/$v; mkdir -p $out
  src=adsbdec_amd/csrc/scan_kernel.hip; [ $v = old ] && src=tmp_ab/scan_kernel_old.hip
  /opt/rocm/bin/hipcc $FLAGS -c $src -o $out/scan_kernel.o || exit 1
  /opt/rocm/bin/hipcc $FLAGS -c adsbdec_amd/csrc/decoder.hip -o $out/decoder.o || exit 1
  gcc -O2 -fPIC -c adsbdec_amd/csrc/format.c -o $out/format.o || exit 1
  for s in multi host_abi; do g++ -O2 -fPIC -std=c++17 -pthread -c adsbdec_amd/csrc/$s.cpp -o $out/$s.o || exit 1; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libadsbdec_amd.so $out/*.o -lm -lpthread || exit 1
done
for round in 1 2 3 4 5 6; do
  order="new old"; [ $((round % 2)) = 0 ] && order="old new"
  for v in $order; do
    ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_var/$v/libadsbdec_amd.so timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('round $round $v: ms_per_step', d['ms_per_step'], 'launch_ms', r['launch_ms'], 'frac', r['frac'])"
  done
done
for v in old new; do
  ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_var/$v/libadsbdec_amd.so timeout 300 python bench.py --mode shard --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shard 2Gi $v: ms_per_step', d['ms_per_step'])"
done
