#!/bin/bash
# A/B on one box: offsets per launch of the 256 Mi-sample step (a -DADSB_TUNING build reads ADSB_CHUNK_MI): one launch of 128 Mi
# offsets (what ships) against 2 / 3 / 4 launches on the two alternating compute streams.  Three rounds, interleaved.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
bash tools/build_variant.sh tuning -DADSB_TUNING > /dev/null || exit 1
L=$PWD/adsbdec_amd/lib_var/tuning/libadsbdec_amd.so
for round in 1 2 3; do
  for mi in 128 64 43 32; do
    ADSB_LIB_PATH=$L ADSB_CHUNK_MI=$mi timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('round $round chunk ${mi} Mi offsets: ms_per_step', d['ms_per_step'], 'value', d['value'], 'launches/step', r['launches_per_step'], 'launch_ms', r['launch_ms'])"
  done
done
