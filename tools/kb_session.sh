#!/bin/bash
# kbench matrix on the GPU box: tools/kb_session.sh "<name> <KB_PIPE> <K> <groups/CU or -> <hipcc -D flags...>" ...
# Each variant is built there (tools/kbench.hip includes scan_kernel.hip) and timed over 400 launches of 256 Mi samples of
# noise with the hand-off stream on (nobody reading it): kernel time only, no host consumer.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O /tmp/kb
cfgs=("$@")
for cfg in "${cfgs[@]}"; do
  read -r name pipe k g flags <<< "$cfg"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -I adsbdec_amd/csrc \
     $flags tools/kbench.hip -o /tmp/kb/$name 2>&1 | grep -E "error" &
done
wait
for rep in 1 2; do
  for cfg in "${cfgs[@]}"; do
    read -r name pipe k g flags <<< "$cfg"
    [ "$g" = "-" ] && unset ADSB_PIPE_GROUPS_PER_CU || export ADSB_PIPE_GROUPS_PER_CU=$g
    hand=1; [[ $name == *_nohand ]] && hand=   # (names ending in _nohand: no hand-off stream, no reservation, no marker stores)
    echo "$name (pipe=$pipe K=$k groups=$g $flags): $(KB_HAND=$hand KB_PIPE=$pipe timeout 120 /tmp/kb/$name 256 400 $k 2>&1 | tail -1)"
  done
done | tee -a $O/kb_runs.txt
