#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> "<extra hipcc flags>"
#   -> adsbdec_amd/lib_var/<name>/libadsbdec_amd.so   (load it with ADSB_LIB_PATH=...)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
out=adsbdec_amd/lib_var/$name
mkdir -p "$out"
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -Xarch_host -mavx2 $*"
for s in scan_kernel decoder; do
  /opt/rocm/bin/hipcc $FLAGS -c adsbdec_amd/csrc/$s.hip -o "$out/$s.o" || exit 1
done
gcc -O2 -fPIC -c adsbdec_amd/csrc/format.c -o "$out/format.o" || exit 1
for s in multi host_abi numa; do
  g++ -O2 -fPIC -std=c++17 -pthread -c adsbdec_amd/csrc/$s.cpp -o "$out/$s.o" || exit 1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libadsbdec_amd.so" "$out"/scan_kernel.o "$out"/decoder.o "$out"/format.o "$out"/multi.o "$out"/host_abi.o "$out"/numa.o -lm -lpthread || exit 1
echo "$out/libadsbdec_amd.so"
