"""make_golden.py -- TEST INFRASTRUCTURE ONLY: mints tests/golden/*.

Run in the build container (needs /root/reference for oracle/_ref):

    python oracle/make_golden.py

For every case it writes <name>.npz (the uint16 input) and <name>.json (expected
records).  How the expectation is produced, precisely:

  input x --(oracle front end, restated from air.c:54-92: NOT the reference, which
             cannot be built here for lack of libairspy)--> power samples
          --(REAL reference deqframe/getdf/getabyte/validShort/validLong/CrcStep/
             formatpkt/print_stats, oracle/_ref/ref_demod)--> frames, ts, pw,
             AVR / AVR-MLAT / Beast bytes, Try/Ok table.

So the fixtures pin everything downstream of the power samples to the real
reference and everything upstream to the restatement (DESIGN.md "Oracle").  The
script asserts that the all-restatement oracle gives identical records.
The reference's own tests hold no vectors for this path; the
public CRC known answers it cites are in crc_kat.json.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_signal as G  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def cases():
    # BASELINE.json configs[0]: 1 Mi samples, ~100 DF17, the reference's own CPU case
    x, _ = G.sparse_capture(1 << 20, 100, seed=1)
    yield "config1_1Mi_100xDF17", "config1_1Mi_100xDF17", x, False
    # mixed DF11/17/18 with -a, frames back to back and overlapping, moderate noise
    x, _ = G.dense_capture(3 * (1 << 17), seed=7, sigma=40.0, n_frames=120, amp=(300, 1800))
    yield "mixed_df_a_384Ki", "mixed_df_384Ki", x, True
    # same input without -a: DF18 must vanish from Try and Ok
    yield "mixed_df_noa_384Ki", "mixed_df_384Ki", x, False
    # config-3 flavour: wide-band noise, ~7 % of offsets pass the preamble test
    x, _ = G.dense_capture(1 << 18, seed=11, sigma=300.0, n_frames=30)
    yield "dense_noise_256Ki", "dense_noise_256Ki", x, True
    # ragged length (not a multiple of 4) just above the first deqframe call
    x, _ = G.sparse_capture(4 * 40980 // 2 + 4 * 1300 + 3, 6, seed=13, dfs=(17, 11))
    yield "ragged_tail", "ragged_tail", x, True
    # too short for deqframe ever to fire: no output at all
    x, _ = G.sparse_capture(81000, 5, seed=17)
    yield "too_short", "too_short", x, False


def main():
    if not O.build_ref():
        raise SystemExit("oracle/_ref is not available (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    for name, input_name, x, df18 in cases():
        a = O.power(x)
        rf, rstats = O.ref_demod(a, df18=df18)
        of, ostats = O.decode(x, df18=df18)
        assert ostats == rstats, (name, ostats, rstats)
        assert [(f["ts"], f["pw"], f["frame"]) for f in of] == [(f["ts"], f["pw"], f["frame"]) for f in rf], name
        rec = dict(
            name=name, input=input_name + ".npz", df18=df18, n_samples=int(x.size),
            provenance="power samples from oracle front end; records from the real reference "
                       "demod.c/valid.c/output.c objects (oracle/_ref/ref_demod)",
            stats={k: {str(d): int(v) for d, v in rstats[k].items()} for k in rstats},
            frames=[dict(g=o["g"], ts=r["ts"], pw=r["pw"], frame=r["frame"].hex().upper(),
                         avr=r["avr"].decode(), mlat=r["mlat"].decode(), beast=r["beast"].hex().upper())
                    for r, o in zip(rf, of)],
        )
        np.savez_compressed(os.path.join(OUT, input_name + ".npz"), x=x)
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(rec, f, indent=0)
        print(f"{name}: {x.size} samples, {len(rf)} frames, stats {rstats}")

    # CRC known answers the survey verified through the reference's crc.h
    kat = dict(
        source="public DF17 frames with residual 0 and a DF11 with residual F8740F (SURVEY.md section 4)",
        vectors=[
            dict(frame="8D4840D6202CC371C32CE0576098", residual="000000"),
            dict(frame="8D40621D58C382D690C8AC2863A7", residual="000000"),
            dict(frame="5D4840D6000000", residual="F8740F"),
        ],
        table_first8=["000000", "FFF409", "001C1B", "FFE812", "003836", "FFCC3F", "00242D", "FFD024"],
        table_last=["FA0480"],
    )
    with open(os.path.join(OUT, "crc_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)


if __name__ == "__main__":
    main()
