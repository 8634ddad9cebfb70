#!/bin/bash
# same-box A/B: workgroups per CU of the restructured kernel (96 VGPRs: 5 fit) against the old one (100 VGPRs: 4)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None"
build() { # name, source dir
  out=adsbdec_amd/lib_var/$1; mkdir -p $out
  /opt/rocm/bin/hipcc $FLAGS -c $2/scan_kernel.hip -o $out/scan_kernel.o || exit 1
  /opt/rocm/bin/hipcc $FLAGS -c $2/decoder.hip -o $out/decoder.o || exit 1
  gcc -O2 -fPIC -c $2/format.c -o $out/format.o || exit 1
  for s in multi host_abi; do g++ -O2 -fPIC -std=c++17 -pthread -c $2/$s.cpp -o $out/$s.o || exit 1; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libadsbdec_amd.so $out/*.o -lm -lpthread || exit 1
}
mkdir -p /tmp/ab/include; cp include/*.h /tmp/ab/include/; for v in old new4 new5s; do rm -rf /tmp/ab/x/src_$v; mkdir -p /tmp/ab/x; cp -r adsbdec_amd/csrc /tmp/ab/x/src_$v; done
cp tmp_ab/scan_kernel_old.hip /tmp/ab/x/src_old/scan_kernel.hip
sed -i 's/const size_t lds = lds_bytes(args.passes);/const size_t lds = lds_bytes(args.passes) + 2048;/' /tmp/ab/x/src_new4/scan_kernel.hip
grep -c "lds_bytes(args.passes) + 2048" /tmp/ab/x/src_new4/scan_kernel.hip
sed -i 's/constexpr int kMinWaves = 4; /constexpr int kMinWaves = 5; /' /tmp/ab/x/src_new5s/scan_kernel.h
grep -c "kMinWaves = 5" /tmp/ab/x/src_new5s/scan_kernel.h
build old /tmp/ab/x/src_old; build new adsbdec_amd/csrc; build new4 /tmp/ab/x/src_new4; build new5s /tmp/ab/x/src_new5s
for round in 1 2 3 4; do
  order="old new new4 new5s"; [ $((round % 2)) = 0 ] && order="new5s new4 new old"
  for v in $order; do
    ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_var/$v/libadsbdec_amd.so timeout 300 python bench.py --steps 300 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('round $round $v: ms_per_step', d['ms_per_step'], 'launch_ms', r['launch_ms'], 'frac', r['frac'])"
  done
done
