#!/bin/bash
# build kbench variants: tools/build_kbench.sh "<ablate> <minwaves>" ...
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/bin
for cfg in "$@"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -I adsbdec_amd/csrc \
     -DADSB_ABLATE=$1 -DADSB_MIN_WAVES=$2 -DADSB_FIR_GROUP=${3:-4} tools/kbench.hip -o tools/bin/kb_a$1_w$2_g${3:-4} 2>&1 | grep -E "error" &
done
wait
ls tools/bin
