"""bench.py's host-side plumbing, on CPU: the self-launch of N ranks (the driver calls `python bench.py --gpus N` from a plain
shell), the one-JSON-line rule, and the guards around canned / sampled figures.  The numbers themselves need an MI355X."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_starts_one_rank_per_gpu_with_the_same_arguments(monkeypatch):
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "9", "--mode", "shard"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(4)
    assert e.value.code == 7                      # the workers' exit code is the script's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "9", "--mode", "shard"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_self_launches_before_anything_touches_the_gpu(monkeypatch):
    """--gpus N > 1 without WORLD_SIZE: main() must hand over to self_launch() before importing torch.cuda state."""
    called = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setattr(bench, "self_launch", lambda n: called.setdefault("n", n))
    bench.main()
    assert called == {"n": 2}


def test_emit_line_is_one_json_line(capsys):
    bench.emit_line({"metric": "x", "value": 1.5})
    out = capsys.readouterr().out
    assert out.endswith("\n") and out.count("\n") == 1 and json.loads(out) == {"metric": "x", "value": 1.5}


def test_clock_sampler_rejects_readings_that_cannot_be_the_clock_under_load():
    cs = bench.ClockSampler(index=10 ** 6)        # no such card: nothing sampled
    assert cs.ghz() == (None, 0)
    cs.samples = [95_000_000] * 9                 # what the file reads under rocprofv3
    assert cs.ghz()[0] is None
    cs.samples = [2_390_000_000] * 3
    assert cs.ghz()[0] is None                    # too few samples
    cs.samples = [2_390_000_000] * 9
    assert cs.ghz() == (2.39, 9)


def test_roofline_sources_are_named():
    p0 = dict(kernel_ms=0.0, big_offsets=134216525, big_launches=0, big_ms=0.0, offsets=0)
    p1 = dict(kernel_ms=28.0, big_offsets=134216525, big_launches=200, big_ms=28.0, offsets=200 * 134216525)
    roof, valu = bench.roofline_objects(p0, p1, 200, "", clock=(2.39, 15), step_ms=0.155, rounds=("r3",))
    assert roof["launch_ms"] == 0.14 and abs(roof["frac"] - 4 * 134216525 / 0.14e-3 / 1e9 / 8000) < 1e-4
    assert roof["traffic"] and "profiles/r3_pmc.json" in roof["traffic_source"] and "NOT collected in this run" in roof["traffic_source"]
    assert "live" in roof["launch_ms_source"] and "launches_overlap" not in roof
    assert valu["clock_ghz"] == 2.39 and "sysfs" in valu["clock_source"] and valu["peak_clock_ghz"] == 2.4
    assert 0.5 < valu["frac_at_peak_clock"] < 1.0 and "profiles/r3_pmc.json" in valu["pmc_source"]
    # a multi-launch step on two alternating streams: the rate is priced on the whole step
    p1b = dict(p1, big_launches=1600, big_ms=1600 * 0.26, kernel_ms=1600 * 0.26, offsets=1600 * 134216525)
    roof2, _ = bench.roofline_objects(p0, p1b, 200, "", clock=None, step_ms=1.2, rounds=("r3",))
    assert roof2["launches_overlap"] is True and abs(roof2["achieved"] - 8 * 4 * 134216525 / 1.2e-3 / 1e9) < 1


def test_bind_near_gpu_never_fails_a_run(monkeypatch, tmp_path):
    """The multi-rank CPU binding is best effort: whatever torch or sysfs say, it returns a description."""
    before = os.sched_getaffinity(0)

    class Raises:
        class cuda:
            @staticmethod
            def get_device_properties(_i):
                raise RuntimeError("no device")
    assert bench.bind_near_gpu(Raises, 0).startswith("none (RuntimeError")

    class NoPci:
        class cuda:
            @staticmethod
            def get_device_properties(_i):
                return object()
    assert bench.bind_near_gpu(NoPci, 0).startswith("none (torch does not report")

    class Props:
        pci_domain_id, pci_bus_id, pci_device_id = 0xFFFF, 0xFE, 0x1F   # no such device here

    class Absent:
        class cuda:
            @staticmethod
            def get_device_properties(_i):
                return Props()
    assert bench.bind_near_gpu(Absent, 0).startswith("none (")
    assert os.sched_getaffinity(0) == before


def test_main_workloads_name_their_evidence_sets():
    """--dense10 / --gate-storm / --dense are main workloads since round 6 (what tools/profile_session.sh puts under
    rocprofv3); each looks its committed PMC evidence up under its own tag, newest round first, and the sets exist."""
    import types
    tag = lambda **kw: bench.profile_tag(types.SimpleNamespace(**{"dense": False, "dense10": False, "gate_storm": False, **kw}))
    assert (tag(), tag(dense=True), tag(dense10=True), tag(gate_storm=True)) == ("", "_dense", "_dense10", "_storm")
    assert bench.PROFILE_ROUNDS[0] == "r6"
    for t in ("", "_stats", "_dense10", "_dense10_stats", "_storm", "_storm_stats"):
        for kind in ("_pmc.json", "_kernel_stats.csv", "_dispatches.csv", "_bench_under_rocprofv3.json"):
            assert os.path.exists(os.path.join(ROOT, "profiles", "r6" + t + kind)), t + kind
    p0 = dict(kernel_ms=0.0, big_offsets=134216525, big_launches=0, big_ms=0.0, offsets=0)
    p1 = dict(kernel_ms=17.5, big_offsets=134216525, big_launches=100, big_ms=17.5, offsets=100 * 134216525)
    roof, valu = bench.roofline_objects(p0, p1, 100, "_dense10", clock=(2.3, 9), step_ms=0.21)
    assert "profiles/r6_dense10_pmc.json" in roof["traffic_source"] and abs(roof["frac"] - 4 * 134216525 / 0.175e-3 / 1e9 / 8000) < 1e-3
    assert valu["valu_wave_instructions"] > 70e6     # a full channel's tiles issue more than the sparse capture's 65 M
