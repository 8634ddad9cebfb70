#!/usr/bin/env python3
"""How full are the CUs' four tile slots while a launch runs?  Reads the per-tile dump of a kbench built with
-DADSB_TILE_CLOCK=1 (tile begin_us end_us xcc cu se) and prints the mean number of resident tiles per CU over
tenths of the launch, and the gap between a tile's end and the next tile's begin on the same CU.
    KB_HAND=1 ADSB_CLOCK_OUT=/tmp/tc.txt tools/bin/kb_clock 256 50; python tools/tile_occupancy.py /tmp/tc.txt"""
import collections
import sys

rows = [l.split() for l in open(sys.argv[1])]
tiles = [(float(r[1]), float(r[2]), (int(r[3]), int(r[4]), int(r[5]))) for r in rows]
T = max(e for _, e, _ in tiles)
ncu = len({c for _, _, c in tiles})
print(f"{len(tiles)} tiles on {ncu} CUs, launch {T:.1f} us, mean tile life {sum(e - b for b, e, _ in tiles) / len(tiles):.2f} us")
for k in range(10):
    lo, hi = T * k / 10, T * (k + 1) / 10
    occ = sum(max(0.0, min(e, hi) - max(b, lo)) for b, e, _ in tiles) / (hi - lo) / ncu
    print(f"  {lo:6.1f} .. {hi:6.1f} us: {occ:.2f} tiles resident per CU")
print("mean life of the tiles that BEGIN in each tenth of the launch (us):",
      " ".join(f"{(lambda v: sum(v) / len(v) if v else 0.0)([e - b for b, e, _ in tiles if T * k / 10 <= b < T * (k + 1) / 10]):.1f}" for k in range(10)))
per = collections.defaultdict(list)
for b, e, c in tiles:
    per[c].append((b, e))
gaps = []
for c, ts in per.items():
    ends = sorted(e for _, e in ts)
    begins = sorted(b for b, _ in ts)[4:]          # the first four start with the launch
    for b in begins:                                 # a tile that begins takes the slot of the latest tile that ended before it
        prev = max((e for e in ends if e <= b + 0.005), default=None)
        if prev is not None:
            gaps.append(b - prev)
            ends.remove(prev)
gaps.sort()
print(f"end of a tile -> begin of the next on the same CU: median {gaps[len(gaps) // 2]:.2f} us, mean {sum(gaps) / len(gaps):.2f}, p90 {gaps[len(gaps) * 9 // 10]:.2f}  ({len(gaps)} hand-overs)")
