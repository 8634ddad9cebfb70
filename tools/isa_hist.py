#!/usr/bin/env python3
"""Histogram of instruction mnemonics per kernel in a gfx950 .s file (largest basic-block loop first)."""
import collections, re, sys
src = open(sys.argv[1]).read().split('\n')
kern = None
hist = collections.defaultdict(collections.Counter)
for line in src:
    m = re.match(r'^(_Z\w+):', line)
    if m: kern = m.group(1)
    if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
        kern = None
    m = re.match(r'^\t([a-z_0-9]+)\s', line)
    if kern and m and not m.group(1).startswith('.'):
        hist[kern][m.group(1)] += 1
for k, h in hist.items():
    tot = sum(h.values()); v = sum(c for n, c in h.items() if n.startswith('v_'))
    print(f"== {k[:60]} total {tot} valu {v}")
    print("   " + ", ".join(f"{n}:{c}" for n, c in h.most_common(28)))
