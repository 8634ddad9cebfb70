"""The synthetic-capture generators (tools/gen_signal.py) pinned by digest.

The -m gpu tests at BASELINE's full sizes and bench.py build their captures with make_workload / make_dense /
make_dense10 / make_gate_storm; what they check is only as stable as those functions.  Each generator's output on a
fixed slice is hashed against tests/golden/generator_digests.json -- once with the noise drawn by torch's CPU generator
(runs everywhere) and once on the device (-m gpu: the Philox stream of the ROCm build, what the big tests really see).
`python tests/test_generators.py --write [cpu|gpu]` regenerates the file's section after a DELIBERATE generator change.
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DIGESTS = os.path.join(ROOT, "tests", "golden", "generator_digests.json")
N = 1 << 20


def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()[:32]


def compute(torch, dev):
    from tools import gen_signal as G
    out = {}
    t, truth = G.make_workload(torch, N, seed=1, device=dev)
    out["make_workload_seed1_1Mi"] = _sha(t)
    out["make_workload_truth"] = hashlib.sha256(repr([(s, fr.hex()) for s, fr in truth]).encode()).hexdigest()[:32]
    # a slice of a longer stream across a generation-block boundary (shard mode): must not depend on the slicing
    lo, hi = G.GEN_BLOCK - 300_000, G.GEN_BLOCK + 300_000
    whole, _ = G.make_workload(torch, 2 * G.GEN_BLOCK, seed=9, damage_share=0.2, device=dev)
    part, _ = G.make_workload(torch, 2 * G.GEN_BLOCK, seed=9, damage_share=0.2, lo=lo, hi=hi, device=dev)
    assert torch.equal(whole[lo:hi], part), "a slice of the stream differs from the same samples of the whole stream"
    out["make_workload_seed9_damaged_slice"] = _sha(part)
    out["make_dense_seed100_1Mi"] = _sha(G.make_workload(torch, N, seed=100, sigma=300.0, df11_share=0.0,
                                                         amp=(1200.0, 2000.0), device=dev)[0])
    out["make_dense10_seed101_1Mi"] = _sha(G.make_tiled(torch, N, 101, 300.0, 0.03, True, device=dev))
    out["make_gate_storm_seed102_1Mi"] = _sha(G.make_tiled(torch, N, 102, 30.0, 1.0, False, device=dev))
    return out


def _check(kind, got):
    with open(DIGESTS) as f:
        want = json.load(f)[kind]
    assert got == want, (f"tools/gen_signal.py no longer produces the captures the full-size tests were written against ({kind}); "
                         "if the change is deliberate: python tests/test_generators.py --write " + kind)


def test_generator_digests_on_cpu():
    import torch
    _check("cpu", compute(torch, torch.device("cpu")))


def test_make_dense_is_make_workload_with_its_parameters():
    import inspect
    from tools import gen_signal as G
    assert "sigma=300.0" in inspect.getsource(G.make_dense) and "amp=(1200.0, 2000.0)" in inspect.getsource(G.make_dense)
    assert "300.0, 0.03, True" in inspect.getsource(G.make_dense10)
    assert "30.0, 1.0, False" in inspect.getsource(G.make_gate_storm)


def test_no_gpu_test_imports_its_captures_from_bench():
    import re
    for name in os.listdir(os.path.join(ROOT, "tests")):
        if name.endswith(".py") and name not in ("test_bench_cpu.py", "test_distributed_cpu.py", "test_generators.py"):
            with open(os.path.join(ROOT, "tests", name)) as f:
                src = f.read()
            assert not re.search(r"^\s*(from bench import|import bench)", src, re.M), name


@pytest.mark.gpu
def test_generator_digests_on_the_device():
    import torch
    assert torch.cuda.is_available()
    _check("gpu", compute(torch, torch.device("cuda", 0)))


if __name__ == "__main__":
    import torch
    kind = sys.argv[2] if len(sys.argv) > 2 else "cpu"
    assert sys.argv[1] == "--write" and kind in ("cpu", "gpu")
    rec = json.load(open(DIGESTS)) if os.path.exists(DIGESTS) else {}
    rec[kind] = compute(torch, torch.device("cpu") if kind == "cpu" else torch.device("cuda", 0))
    with open(DIGESTS, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(rec[kind], indent=1))
