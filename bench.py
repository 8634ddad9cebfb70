#!/usr/bin/env python3
"""bench.py -- throughput of the adsbdec "-f" demodulation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode stream|shard]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--mode stream (default).  A "step" is one full pass of the hot path over one synthetic
capture that is already resident in HBM: adsb_decode_device = adsb_reset -> adsb_push_device_final (fused scan
kernel over every preamble offset, record gather, greedy resolution, end-of-file rule)
-> adsb_take (frames in reference order, in place).  Workload = BASELINE.json configs[1]:
256 Mi uint16 samples @ 20 MS/s, sparse frames (~1 k frames/s, DF17 with some DF11),
sigma = 8 noise.  With N > 1 every rank decodes its own independent stream of that size
(configs[3]); no data-path collective exists, so scaling is "weak".  The timed steps
rotate over three different captures resident in HBM, so that no step can pass on the
records the previous step left in the hand-off buffers.

--mode shard.  BASELINE.json configs[4]: ONE stream of --samples (default 2 Gi) samples
time-sharded over the N ranks.  Each rank holds only its halo'd slice
(adsb_plan_shards: 8 pairs before, one 1196-sample window after), scans the offsets it
owns (adsb_scan_shard), and the fixed-layout adsb_candidate arrays are gathered to rank 0
(one tensor gather per step over a gloo group: the records are host-resident, tens of
bytes per frame -- no RCCL on the data path), where ONE resolver replays the reference's
sequential rules.  Total work is fixed: scaling is "strong".  Gate: at N = 1 the frames
equal the oracle's on the whole stream; at N > 1 rank 0 afterwards decodes the whole
stream alone (the N = 1 path) and the sharded result must equal it.

Before the W warm-up steps the step is run untimed for --preroll-ms (default 60 ms, in
the JSON line as `preroll_ms`): the GPU's clock governor needs ~20 ms of load to settle,
and a service decoding captures back to back lives in that steady state.
`value_cold` is the same step timed right after an idle period, without pre-roll.

Rank 0 prints ONE JSON line.  `value` is whole-job Msamples/s, inputs resident in HBM.
`roofline` prices the scan kernel against HBM (algorithmic traffic = 2 B per input sample
= 4 B per preamble offset; kernel time = latest tile end - earliest tile start on the
device's own clock, taken inside the kernel over the timed steps: within 1 % of
rocprofv3); `roofline_valu` prices it against what actually binds it, VALU issue
(DESIGN.md 4).  `cpu_baseline` is the REAL reference chain (oracle/_ref/ref_adsbdec:
air.c decodeiq + demod.c + valid.c compiled from the reference, 1 thread) timed on this
host on the same capture when that binary travelled with the snapshot (kind
"reference"), else the oracle's C restatement (kind "port"); either way the run is gated:
every frame the GPU path returned must equal the CPU result.  `e2e_host_fed` is the
PCIe-inclusive rate from page-locked host memory (never `value`).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E peak (6.3 TB/s achievable)
from tools.gen_signal import (FRAME_GAP, GEN_BLOCK, TILE_BLOCK, _frame_plan, _frame_start_wave, make_dense,  # noqa: E402,F401
                              make_dense10, make_gate_storm, make_tiled, make_workload)  # the build's own generators

N_SIMD = 256 * 4       # MI355X: 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4        # peak shader clock (MI355X_MICROARCH.md)


def frames_key(frames, with_g=True):
    if with_g:
        return [(f["g"], f["ts"], f["pw"], f["frame"]) for f in frames]
    return [(f["ts"], f["pw"], f["frame"]) for f in frames]


def cpu_reference(x_host: np.ndarray, df18: bool):
    """Time the CPU path on capture x_host -> (cpu_baseline dict, the oracle's frames)."""
    from oracle import oracle as O
    O.build()
    ncpu = os.cpu_count() or 0
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except OSError:
        model = "?"
    t1 = time.perf_counter()
    want, _ = O.decode(x_host, df18=df18)
    port_dt = time.perf_counter() - t1
    port = round(x_host.size / port_dt / 1e6, 2)
    host = f"{model}, {ncpu} logical CPUs on the box, 1 used"
    if O.ref_available():
        path = ("/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp") + f"/adsb_bench_{os.getpid()}.u16"
        x_host.tofile(path)
        try:
            t1 = time.perf_counter()
            rf, _ = O.ref_decode(None, df18, path=path)
            ref_dt = time.perf_counter() - t1
        finally:
            os.unlink(path)
        if frames_key(rf, False) != frames_key(want, False):
            raise SystemExit("PARITY FAILURE: the oracle's restatement differs from the real reference chain on this capture")
        cpu = {"value": round(x_host.size / ref_dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "reference",
               "sample": f"the whole capture ({x_host.size} samples, {len(rf)} frames) from a tmpfs file through "
                         "oracle/_ref/ref_adsbdec = the reference's own air.c:29-101 decodeiq + demod.c + valid.c + "
                         "formatpkt (gcc -O2 -ffp-contract=off), wall time of the process, frames formatted to a pipe",
               "host": host, "port_value": port}
    else:
        cpu = {"value": port, "unit": "Msamples/s", "cores": 1, "kind": "port",
               "sample": f"the whole capture ({x_host.size} samples, {len(want)} frames), oracle/liboracle.so "
                         "(gcc -O2 -ffp-contract=off); oracle/_ref did not travel with this snapshot",
               "host": host}
    return cpu, want


def gate(got, want, what):
    a, b = frames_key(got), frames_key(want)
    if a != b:
        i = next((i for i, (p, q) in enumerate(zip(a, b)) if p != q), min(len(a), len(b)))
        raise SystemExit(f"PARITY FAILURE ({what}): GPU path returned {len(a)} frames, expected {len(b)}; first difference at index {i}")


PROFILE_ROUNDS = ("r6", "r5", "r4", "r3")   # committed rocprofv3 evidence sets, newest first (profiles/<round>[_<workload>]_pmc.json)


def profile_tag(args) -> str:
    """Suffix of the evidence set under profiles/ that belongs to the main workload of this run."""
    return "_dense" if args.dense else "_dense10" if args.dense10 else "_storm" if args.gate_storm else ""


def roofline_objects(p0, p1, steps, profiles_tag="", clock=None, step_ms=None, rounds=PROFILE_ROUNDS):
    """HBM roofline of the dominant launch + the VALU-issue roofline that actually binds it."""
    kernel_ms = p1["kernel_ms"] - p0["kernel_ms"]
    big_off = p1["big_offsets"]
    if p0["big_offsets"] == big_off:  # the warm-up already ran launches of the dominant size
        n_big, ms_big = p1["big_launches"] - p0["big_launches"], p1["big_ms"] - p0["big_ms"]
    else:
        n_big, ms_big = p1["big_launches"], p1["big_ms"]
    avg_ms = ms_big / n_big if n_big else 0.0
    achieved = 4.0 * big_off / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_source, valu = None, None, None
    for name in tuple(f"{r}{profiles_tag}_pmc.json" for r in rounds):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            pmc = json.load(f)
        if pmc.get("launch_offsets") != big_off:
            continue
        if "hbm_bytes_per_launch" in pmc:
            traffic = int(pmc["hbm_bytes_per_launch"])
            traffic_source = (f"profiles/{name}: rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE per dispatch of this kernel, "
                              "MI355X_MICROARCH.md's gfx950 correction) recorded by tools/profile_session.sh on a launch of the same "
                              "size; a committed figure, NOT collected in this run (counters cannot share a run with timing)")
        c = pmc["counters"]
        if "SQ_INSTS_VALU" in c and avg_ms > 0:
            # Floor of the launch if the VALU did nothing but issue: wave-instructions (PMC) x
            # cycles per wave-instruction / SIMDs / clock.  A SIMD-32 issues a wave64 VALU
            # instruction over 2 cycles, 4 for the packed-f32 and three-operand forms
            # (MI355X_MICROARCH.md; tools/valu_bench.hip measured 2.4 / 4.2-4.4 with 4 waves).
            # The mix comes from the ISA of the Stage A loop, ~94 % of the dynamic count
            # (profiles/<tag>_isa_mix.json, written by tools/isa_mix.py from the shipped code object).
            mix_path = os.path.join(ROOT, "profiles", name.replace("_pmc.json", "_isa_mix.json"))
            cyc, mix_src = 4.0, "assumed 4 cycles per instruction"
            if os.path.exists(mix_path):
                with open(mix_path) as f:
                    mix = json.load(f)
                cyc = float(mix["cycles_per_valu_instruction"])
                mix_src = f"profiles/{os.path.basename(mix_path)}"
            insts = float(c["SQ_INSTS_VALU"]["mean"])
            floor_peak_ms = insts * cyc / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
            # the chip does not hold its 2.4 GHz under this load (it runs at its power limit): price the floor at the
            # clock it did hold -- sampled from sysfs while the timed steps ran, else the figure of profiles/r2_power_trace.txt
            ghz, ghz_src = (clock[0], f"sysfs freq1_input, median of {clock[1]} samples taken every 2 ms during the timed steps "
                                      "(instantaneous DVFS readings: different boxes have read 2.0 and 2.4 GHz for the same kernel time, "
                                      "so frac_at_peak_clock is the conservative figure)") \
                if clock and clock[0] else (2.33, "profiles/r2_power_trace.txt (rocm-smi beside a 15 s bench run; sysfs was not readable in this run)")
            floor_ms = insts * cyc / N_SIMD / (ghz * 1e9) * 1e3
            valu = {"bound": "valu_issue", "valu_wave_instructions": int(insts), "cycles_per_instruction": round(cyc, 3),
                    "mix_source": mix_src, "simds": N_SIMD, "clock_ghz": round(ghz, 3), "clock_source": ghz_src,
                    "floor_ms": round(floor_ms, 5), "launch_ms": round(avg_ms, 5), "frac": round(floor_ms / avg_ms, 4),
                    "floor_ms_at_peak_clock": round(floor_peak_ms, 5), "peak_clock_ghz": CLOCK_GHZ,
                    "frac_at_peak_clock": round(floor_peak_ms / avg_ms, 4),
                    "pmc_source": f"profiles/{name} (a committed count for a launch of this size, not collected in this run)"}
        break
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "launch_ms_source": "the kernel's own clock (latest tile end - earliest tile start on the device's 100 MHz counter), "
                                    "averaged over the launches of the timed steps: live, this run",
                "kernel": "adsb::scan_kernel", "launch_offsets": big_off, "launch_bytes": 4 * big_off,
                "launch_ms": round(avg_ms, 5), "launches_per_step": round(n_big / steps, 2),
                "kernel_ms_per_step": round(kernel_ms / steps, 4),
                "limited_by": "VALU issue (every product and sum of the 14-tap FIR is rounded separately: "
                              "784 flops per 28 outputs), not HBM: see roofline_valu and DESIGN.md section 4"}
    if step_ms and n_big / steps > 1.5:
        # a multi-launch stream: consecutive launches run on two alternating streams and overlap, so a launch's own duration
        # (two resident) says nothing about the rate -- price the whole step instead (a lower bound: it includes the host's share)
        total_bytes = 4.0 * (p1["offsets"] - p0["offsets"]) / steps
        ach = total_bytes / (step_ms * 1e-3) / 1e9
        roofline.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "launches_overlap": True,
                         "achieved_what": "algorithmic bytes of ALL launches of a step / the step's wall time: consecutive launches "
                                          "overlap on two compute streams, launch_ms is the duration of one launch with two resident"})
    return roofline, valu


_REAL_STDOUT = None


def emit_line(line: dict):
    """Rank 0's ONE JSON line, on the process's real stdout (see main(): with N > 1 fd 1 is parked on stderr)."""
    text = json.dumps(line) + "\n"
    if _REAL_STDOUT is None:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, text.encode())


def preroll(step, ms, at_least=40):
    """Untimed steps for `ms` of wall time (and at least `at_least` of them): the clock governor's ramp, see main()."""
    t, i = time.perf_counter(), 0
    while i < at_least or (time.perf_counter() - t) * 1e3 < ms:
        step(i)
        i += 1


def timed_steps(step, steps, fence, sync=None):
    """K steps bracketed by barrier + synchronize on both sides.  The time is this rank's own: from the common
    start (behind the opening barrier) to the completion of its K-th step (device idle); the caller takes the
    max over ranks.  The closing barrier's own latency (gloo: a few hundred microseconds) is not part of it."""
    fence()
    t0 = time.perf_counter()
    raw = None
    for i in range(steps):
        raw = step(i)
    if sync is not None:
        sync()
    dt = time.perf_counter() - t0
    fence()
    return dt, raw


def bind_near_gpu(torch, device):
    """Multi-rank runs: keep this rank's host threads (the hand-off stream's consumer spins on device-written pinned
    memory) on the CPUs of the GPU's own NUMA node.  Best effort -- returns what was done, for the JSON line."""
    try:
        pr = torch.cuda.get_device_properties(device)
        dom, bus, dev = (getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is None or dev is None:
            return "none (torch does not report the PCI address)"
        base = f"/sys/bus/pci/devices/{int(dom or 0):04x}:{int(bus):02x}:{int(dev):02x}.0"
        with open(base + "/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return "none (numa_node = -1)"
        with open(base + "/local_cpulist") as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "none (no allowed CPU on the GPU's node)"
        os.sched_setaffinity(0, cpus)
        return f"NUMA node {node} of GPU {device} ({len(cpus)} CPUs)"
    except Exception as e:  # noqa: BLE001 -- never a reason to fail a bench run
        return f"none ({type(e).__name__}: {e})"


class Ranks:
    """What the ranks of a multi-process run (stream mode: one process and one independent stream per GPU, BASELINE
    configs[3]) share: nothing on the data path -- only the barriers around the timed region, the max of their
    times and the AND of their parity flags.  All of it is CPU-side, so the one process group is gloo; RCCL is not
    initialised at all (it would move nothing).  A rank that dies takes the job down: the others' next collective fails
    (connection reset) or times out, and they exit non-zero naming the step they were in (tests/test_distributed_cpu.py)."""

    def __init__(self, world=1, rank=0, timeout_s=900.0):
        self.world, self.rank, self.dist = world, rank, None
        if world > 1:
            import datetime
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
            self.dist = dist

    def _collective(self, what, fn):
        try:
            return fn()
        except Exception as e:      # gloo: "Connection closed by peer", a timeout ... -- another rank is gone
            raise SystemExit(f"rank {self.rank}: lost the other ranks at `{what}` ({type(e).__name__}: {str(e)[:200]}): "
                             "a rank of this job died or hung; this rank gives up") from e

    def fence(self, sync=None, what="barrier"):
        if sync is not None:
            sync()
        if self.world > 1:
            self._collective(what, self.dist.barrier)

    def max_over(self, dt, what="max of the ranks' times"):
        if self.world == 1:
            return dt
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        self._collective(what, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX))
        return float(t.item())

    def all_ok(self, ok, what="AND of the ranks' parity flags"):
        if self.world == 1:
            return bool(ok)
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        self._collective(what, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN))
        return bool(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell: run the same command line under torch.distributed.run."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


class ClockSampler:
    """Shader clock of GPU `index` while the timed steps run: sysfs freq1_input (Hz) read every 2 ms on a thread."""

    def __init__(self, index=0):
        import glob
        import threading
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        self.path = cards[index] if index < len(cards) else None
        self.samples, self._stop, self._t = [], False, None
        if self.path:
            self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop:
            try:
                with open(self.path) as f:
                    self.samples.append(int(f.read()))
            except (OSError, ValueError):
                return
            time.sleep(0.002)

    def __enter__(self):
        if self._t:
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._t:
            self._t.join(timeout=1)

    def ghz(self):
        """(median GHz, samples) -- or (None, n) when the figure cannot be the shader clock under load: too few samples, or a
        value outside 1.2 .. 2.6 GHz (under rocprofv3 the file reads 0.1-2.0: the profiler serialises the dispatches)."""
        v = [x for x in self.samples if x > 0]
        if len(v) < 5:
            return (None, len(v))
        g = float(np.median(v)) / 1e9
        return (g, len(v)) if 1.2 <= g <= 2.6 else (None, len(v))


def preamble_pass_fraction(x_host: np.ndarray, n=4 << 20):
    """Share of the preamble offsets of a prefix that pass demod.c:102-107 (p1 > 2 s1 && p2 > 2 s2), from the oracle's
    power samples: the density figure BASELINE configs[2] quotes ("~10 % of offsets above preamble threshold")."""
    from oracle import oracle as O
    a = O.power(np.ascontiguousarray(x_host[:n]))
    m = a.size - 1196
    c = np.trunc(a[:-10] + a[10:]).astype(np.int64)   # c[k] = (int)(a[k] + a[k+10]): all four sums have this form
    p1, s1, s2, p2 = c[0:m], c[5:m + 5], c[30:m + 30], c[35:m + 35]
    return float(np.mean((p1 > 2 * s1) & (p2 > 2 * s2)))


def cli_whole_process(x_host: np.ndarray, capi, df18: bool, sizes=(16 << 20, 64 << 20, 256 << 20)):
    """The C host program (the actual drop-in: reader thread, async pushes, AVR to stdout, Try/Ok table) on files of three
    sizes: wall time of the whole process, exec to exit, with the reference's own wall time on the same file beside it
    (oracle/_ref/ref_adsbdec, one core) -- and from the two, the file size below which the CPU finishes first."""
    import subprocess
    if not os.path.exists(capi.CLI_PATH):
        return None
    from oracle import oracle as O
    have_ref = O.ref_available()
    path = ("/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp") + f"/adsb_bench_cli_{os.getpid()}.u16"
    flags = ["-a"] if df18 else []

    def run_cli(extra_args, env_extra=None):
        walls, inits, decodes, frames = [], [], [], None
        for _ in range(3):
            t0 = time.perf_counter()
            p = subprocess.run([capi.CLI_PATH] + flags + extra_args + ["-f", path], capture_output=True,
                               env=dict(os.environ, ADSB_CLI_TIMING="1", **(env_extra or {})))
            walls.append(time.perf_counter() - t0)
            if p.returncode != 0:
                raise RuntimeError(p.stderr.decode()[-300:])
            frames = p.stdout.count(b"\n")
            for ln in p.stderr.decode().splitlines():
                if ln.startswith("timing:"):
                    f = ln.replace(",", " ").split()
                    inits.append(float(f[3]))
                    decodes.append(float(f[6]))
        med = lambda v: round(sorted(v)[len(v) // 2], 1) if v else None
        return {"wall_ms": round(sorted(walls)[1] * 1e3, 1), "runtime_init_ms": med(inits), "decode_ms": med(decodes), "frames": frames}

    out = {"what": "adsbdec_amd_cli -f <capture as a tmpfs file>: whole process, exec to exit, median of 3 (AVR lines to a pipe, "
                   "Try/Ok table on stderr); reference_wall_ms: oracle/_ref/ref_adsbdec (the real chain, one core) on the same file, "
                   "once", "unit": "Msamples/s", "files": {}}
    try:
        pts = []
        for n in sizes:
            n = min(n, x_host.size)
            n -= n % 4
            x_host[:n].tofile(path)
            rec = run_cli([])
            rec["value"] = round(n / rec["wall_ms"] / 1e3, 1)
            if have_ref:
                t0 = time.perf_counter()
                p = subprocess.run([O.REF_ADSBDEC] + flags + [path], capture_output=True)
                rec["reference_wall_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
                if p.stdout.count(b"\n") != rec["frames"]:
                    raise SystemExit(f"PARITY FAILURE: the C host program printed {rec['frames']} frames, the reference {p.stdout.count(chr(10).encode())}")
                rec["speedup_vs_reference"] = round(rec["reference_wall_ms"] / rec["wall_ms"], 2)
            out["files"][f"{n >> 20}Mi"] = rec
            pts.append((n, rec["wall_ms"], rec.get("reference_wall_ms")))
            if n == x_host.size:
                break
        big_n = pts[-1][0]
        last = out["files"][f"{big_n >> 20}Mi"]
        # the largest file again with every device of the node visible to the runtime (what round 3's program did), and through
        # the multi-GPU driver with two handles on this device (-G 0,0): same bytes, the driver's start-up beside the plain one
        out["largest_file_all_devices_visible"] = run_cli([], {"ADSB_CLI_ALL_DEVICES": "1"})
        out["largest_file_G_0_0"] = run_cli(["-G", "0,0"])
        out.update({"samples": big_n, "frames": last["frames"], "wall_ms": last["wall_ms"], "value": last["value"],
                    "runtime_init_ms": last["runtime_init_ms"], "decode_ms": last["decode_ms"]})
        if have_ref and len(pts) >= 2:
            (n0, g0, c0), (n1, g1, c1) = pts[0], pts[-1]
            b = (g1 - g0) / (n1 - n0)            # ms per sample once the process is up
            a = g0 - b * n0                      # what it costs to get there: loader + GPU runtime
            c = c1 / n1                          # the reference: proportional to the file
            if c > b:
                n_star = a / (c - b)
                out["crossover"] = {"samples": int(n_star), "file_MB": round(2 * n_star / 1e6), "seconds_of_signal": round(n_star / 20e6, 1),
                                    "startup_ms": round(a, 1), "gpu_ms_per_Mi_samples": round(b * (1 << 20), 4),
                                    "reference_ms_per_Mi_samples": round(c * (1 << 20), 4),
                                    "what": "below this file size the reference's single CPU thread finishes before this program: "
                                            "wall = startup + samples x rate for this program (fitted to the smallest and largest file), "
                                            "samples x rate for the reference"}
        out["init_share"] = ("the dynamic loader + the GPU runtime's start (runtime_init_ms, measured from main(); the loader adds "
                             "~0.1 s in front of it) are most of the wall time; decode_ms is reading, copying, scanning and writing "
                             "the whole capture once the runtime is up")
        return out
    except RuntimeError as e:
        return {"error": str(e)}
    finally:
        if os.path.exists(path):
            os.unlink(path)


def multi_stream_host_fed(torch, capi, x_dev, df18, counts=(1, 2, 4)):
    """configs[3]'s host side on ONE GPU: K handles driven by K host threads, each pushing its own stream from
    page-locked memory with adsb_push_async at the reference's call size (1 Mi samples).  What it shows: whether
    anything in the library or the runtime serialises the streams before the links do."""
    import threading
    n = min(x_dev.numel(), 64 << 20)
    n -= n % (1 << 20)
    L = capi.load()
    p = L.adsb_host_alloc(2 * n)
    if not p:
        return None
    out = {"samples_per_stream": n, "unit": "Msamples/s", "call": "adsb_push_async, 1 Mi samples per call, frames taken after "
           "every call; one host thread per handle; every handle reads the same page-locked capture; best of 3"}
    try:
        host = np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(p))
        host[:] = x_dev[:n].cpu().numpy().view(np.uint16)
        chunk = 1 << 20
        for k in counts:
            decs = [capi.Decoder(df18=df18) for _ in range(k)]
            got = [0] * k

            def run(i):
                d = decs[i]
                d.reset()
                c = 0
                for off in range(0, n, chunk):
                    d.push_async((p + 2 * off, min(chunk, n - off)))
                    c += d.take_raw()[1]
                d.finish()
                got[i] = c + d.take_raw()[1]

            best = 1e9
            for _ in range(3):
                ths = [threading.Thread(target=run, args=(i,)) for i in range(k)]
                t0 = time.perf_counter()
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                best = min(best, time.perf_counter() - t0)
            if len(set(got)) != 1:
                raise SystemExit(f"PARITY FAILURE (multi-stream host-fed, {k} handles): {got}")
            out[f"streams_{k}"] = {"aggregate": round(k * n / best / 1e6, 1), "host_threads": k, "frames_per_stream": got[0]}
            for d in decs:
                d.close()
    finally:
        L.adsb_host_free(p)
    return out


def host_fed_sharded(torch, capi, x_dev, df18, want_key, want_stats, counts=(1, 2, 4)):
    """The library's multi-GPU driver (adsb_multi_decode_host: one process, a worker thread and a handle per device) on ONE
    capture in page-locked host memory, with K handles on THIS device: what `adsbdec_amd_cli -G K -f file` runs, Try/Ok
    table included.  One GPU has one link, so K > 1 measures the driver's plumbing (plan, K copy/scan/resolve pipelines side
    by side, stitch, gather), not scaling; frames and statistics are gated against the stream decode of the same capture."""
    from adsbdec_amd import sharding
    n = x_dev.numel()
    host = torch.empty(n, dtype=torch.int16, pin_memory=True)
    host.copy_(x_dev)
    torch.cuda.synchronize()
    dev = torch.cuda.current_device()
    out = {"samples": n, "unit": "Msamples/s",
           "what": "adsb_multi_decode_host of one page-locked capture, collect_stats=1, K handles on one device; best of 3 calls "
                   "(PCIe-inclusive: never `value`); serial_us = stitch + gather on the calling thread behind the last worker"}
    for k in counts:
        md = sharding.MultiDecoder(k, [dev] * k, df18=df18, collect_stats=True)
        md.decode_host(host.data_ptr(), n)
        best, raw, inf = 1e9, None, None
        for _ in range(3):
            t0 = time.perf_counter()
            raw = md.decode_host(host.data_ptr(), n)
            dt = time.perf_counter() - t0
            if dt < best:
                best, inf = dt, md.info()
        if frames_key(capi._frames_to_dicts(raw[0], raw[1])) != want_key:
            raise SystemExit(f"PARITY FAILURE (host-fed sharded, {k} handles): frames differ from the stream decode")
        st = md.stats()
        if want_stats is not None and (st["try"] != want_stats["try"] or st["ok"] != want_stats["ok"]):
            raise SystemExit(f"PARITY FAILURE (host-fed sharded, {k} handles): Try/Ok {st} != {want_stats}")
        out[f"handles_{k}"] = {"value": round(n / best / 1e6, 1), "ms": round(best * 1e3, 3), "shards": int(inf["shards"]),
                               "serial_us": round(inf["serial_us"], 1), "stitch_us": round(inf["stitch_us"], 1),
                               "slowest_worker_ms": round(inf["workers_ms"], 3), "fallback": int(inf["fallback"]),
                               "frames": int(raw[1]), "create_ms": round(inf["create_ms"], 1)}
        md.close()
    out["parity"] = "frames and Try/Ok table equal to the one-handle stream decode of the same capture" + (
        "" if want_stats is None else " (itself gated against the oracle)")
    return out


def host_fed_rates(torch, capi, x_dev, df18):
    """PCIe-inclusive decode rates from page-locked host memory (128 Mi samples)."""
    n = min(x_dev.numel(), 128 << 20)
    n -= n % (1 << 20)
    out = {"samples": n, "unit": "Msamples/s",
           "what": "adsb_reset .. adsb_finish of n samples that start in page-locked HOST memory, frames taken "
                   "after every call; best of 3"}
    L = capi.load()
    p = L.adsb_host_alloc(2 * n)
    if not p:
        return None
    try:
        host = np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(p))
        host[:] = x_dev[:n].cpu().numpy().view(np.uint16)
        dec_plain, dec_overlap = capi.Decoder(df18=df18), capi.Decoder(df18=df18, push_overlap=True)
        ref_frames = None

        def run(dec, chunk, asyn):
            dec.reset()
            got = 0
            for i in range(0, n, chunk):
                piece = (p + 2 * i, min(chunk, n - i))
                if asyn:
                    dec.push_async(piece)
                elif L.adsb_push(dec._h, piece[0], piece[1]) != 0:
                    raise SystemExit("adsb_push failed in the host-fed leg")
                got += dec.take_raw()[1]
            dec.finish()
            return got + dec.take_raw()[1]

        # push_*_overlap: adsb_push with cfg.push_overlap (ONE buffer, the call returns when the copy is done, the scan
        # stays in flight): the three-line drop-in of INTEGRATION.md at the reference's own call site (air.c:230-239)
        for label, chunk, asyn, dec in (("push_1Mi_sync", 1 << 20, False, dec_plain), ("push_1Mi_overlap", 1 << 20, False, dec_overlap),
                                        ("push_1Mi_async", 1 << 20, True, dec_plain), ("push_16Mi_sync", 16 << 20, False, dec_plain),
                                        ("push_16Mi_overlap", 16 << 20, False, dec_overlap), ("push_16Mi_async", 16 << 20, True, dec_plain)):
            best, frames = 1e9, None
            for _ in range(3):
                t0 = time.perf_counter()
                frames = run(dec, chunk, asyn)
                best = min(best, time.perf_counter() - t0)
            if ref_frames is None:
                ref_frames = frames
            elif frames != ref_frames:
                raise SystemExit(f"PARITY FAILURE (host-fed {label}): {frames} frames vs {ref_frames}")
            out[label] = round(n / best / 1e6, 1)
        out["frames"] = ref_frames
        dec_plain.close()
        dec_overlap.close()
    finally:
        L.adsb_host_free(p)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["stream", "shard"], default="stream")
    ap.add_argument("--preroll-ms", type=float, default=60.0,
                    help="run the step untimed for this long before the warm-up steps: the GPU's clock governor "
                         "needs ~20 ms of load to settle (tools/kernel_time_course.py); 0 = off")
    ap.add_argument("--samples", type=int, default=0,
                    help="input samples per GPU per step (stream mode, default 256 Mi) / in the whole stream (shard mode, default 2 Gi)")
    ap.add_argument("--dense", action="store_true", help="configs[2] as the main workload: wide-band noise, ~7%% preamble hits, -a")
    ap.add_argument("--dense10", action="store_true",
                    help="BASELINE configs[2] at its stated density as the main workload: 112-bit frames packed back to back in "
                         "sigma=300 noise, ~10%% of the offsets pass the preamble test, ~106 k frames per 256 Mi samples, -a")
    ap.add_argument("--gate-storm", action="store_true",
                    help="the adversarial capture as the main workload: nothing but frame starts, every tile overflows its survivor queue, -a")
    ap.add_argument("--stats", action="store_true", help="also reproduce valid.c's Try counters (collect_stats=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cold / dense / host-fed legs (profiling runs)")
    ap.add_argument("--shard-source", choices=["device", "host", "file"], default="device",
                    help="shard mode: the slices are resident in their devices' HBM (default), or one capture in page-locked host "
                         "memory / in a tmpfs file is fed to the devices by the library's workers (PCIe-inclusive)")
    ap.add_argument("--bind-cpu", choices=["on", "off"], default="on",
                    help="bind this rank's host threads to the CPUs of its GPU's NUMA node (on a two-socket host an unbound "
                         "run is sometimes 20 %% slower: profiles/r3_ab_runs.txt)")
    ap.add_argument("--die-rank", type=int, default=-1, help=argparse.SUPPRESS)   # test hook: that rank vanishes (os._exit) in front of
    #                                                                                the timed region; the job must end non-zero, not hang
    ap.add_argument("--one-device-test", action="store_true",
                    help="plumbing test only: every rank uses GPU 0 and gloo (numbers are meaningless)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.mode != "shard":
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) and relay rank 0's
        # line.  This parent has not touched the GPU and never does: the workers are children, not an exec.
        return self_launch(args.gpus)
    if world != args.gpus and not (args.mode == "shard" and world == 1):   # (shard mode is one process however many devices)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    global _REAL_STDOUT
    if world > 1:
        # Only rank 0's JSON line may reach stdout: gloo announces its connections on std::cout ("[Gloo] Rank 0 is
        # connected to 1 peer ranks ..."), from every rank.  Everything this process prints from here on goes to stderr;
        # emit_line() writes the one line to the real stdout.
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)

    import torch
    from adsbdec_amd import _build, capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    if args.one_device_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    cpu_binding = bind_near_gpu(torch, local_rank) if args.bind_cpu == "on" else "none (--bind-cpu off: the scheduler's choice)"
    args.cpu_binding = cpu_binding
    ranks = Ranks(world, rank)   # (no data-path collective exists in either mode: see the class)
    dist = ranks.dist
    if not os.path.exists(capi.LIB_PATH):
        if rank == 0:
            _build.build()
        ranks.fence(what="waiting for rank 0's build")

    def fence(sync=True):
        ranks.fence(torch.cuda.synchronize if sync else None)

    max_over_ranks, all_ranks_ok = ranks.max_over, ranks.all_ok

    if args.mode == "shard":
        return run_shard(args, torch, capi, rank, world, fence)

    n = (args.samples or (256 << 20))
    n -= n % 28
    if args.dense + args.dense10 + args.gate_storm > 1:
        raise SystemExit("--dense, --dense10 and --gate-storm are three different main workloads: choose one")
    df18 = bool(args.dense or args.dense10 or args.gate_storm)
    if args.dense:
        xs = [make_dense(torch, n, 100 + 10 * rank + j) for j in range(2)]
        workload = (f"dense noise sigma=300 + one 112-bit frame per ms, {n} uint16 samples/GPU, -a (what rounds 1-3 called BASELINE configs[2]: "
                    "7 % of the offsets pass the preamble test); the steps rotate over 2 captures")
    elif args.dense10:
        xs = [make_dense10(torch, n, 101 + 10 * rank + j) for j in range(2)]
        workload = (f"BASELINE configs[2] at its stated density: 112-bit frames packed back to back in sigma=300 noise, 3 % of the 1 ms slots "
                    f"full of frame starts (~10 % of the offsets pass the preamble test), {n} uint16 samples/GPU, -a; the steps rotate over 2 captures")
    elif args.gate_storm:
        xs = [make_gate_storm(torch, n, 102 + 10 * rank + j) for j in range(2)]
        workload = (f"adversarial: frame starts (preamble + DF17's five bits) packed back to back, {n} uint16 samples/GPU, -a: ~30 % of the offsets "
                    "pass the preamble test, ~7.6 % of ALL offsets the DF gate (every tile's survivor queue overflows); the steps rotate over 2 captures")
    else:
        caps = [make_workload(torch, n, seed=1 + rank + 1000 * j) for j in range(3)]
        xs = [c[0] for c in caps]
        workload = (f"{n} uint16 samples @20MSPS per GPU, {len(caps[0][1])} frames (~1k frames/s, DF17+DF11), "
                    f"sigma=8, device-resident (BASELINE configs[1]" + ("; one stream per GPU, configs[3])" if world > 1 else ")")
                    + "; the steps rotate over 3 different captures")
        del caps
    torch.cuda.synchronize()

    dec = capi.Decoder(df18=df18, device=local_rank, profile=True, collect_stats=args.stats)
    ptrs = [(x.data_ptr(), x.numel()) for x in xs]

    def step(i=0):
        p, m = ptrs[i % len(ptrs)]
        return dec.decode_device_raw(p, m)  # adsb_decode_device == adsb_reset + adsb_push_device_final + adsb_take, one call;
        #                                     the frames stay where the library queued them and are converted after timing

    # ---- cold figure: the first steps after an idle period, no pre-roll (a one-shot `adsbdec -f` user lives here)
    value_cold = None
    if not args.no_extras:
        step(0)                       # loads the code object, allocates the slot buffers
        torch.cuda.synchronize()
        time.sleep(0.5)               # let the clocks fall back
        cold_steps = 3
        pk0 = dec.profile()
        with ClockSampler(local_rank) as cold_clock:
            tc0 = time.perf_counter()
            cold_each = []
            for i in range(cold_steps):
                ts0 = time.perf_counter()
                step(i)
                cold_each.append(round((time.perf_counter() - ts0) * 1e3, 4))
            torch.cuda.synchronize()
            cold_dt = time.perf_counter() - tc0
        cold_dt = max_over_ranks(cold_dt)
        pk1 = dec.profile()
        value_cold = {"value": round(world * n * cold_steps / cold_dt / 1e6, 1), "unit": "Msamples/s", "steps": cold_steps,
                      "ms_per_step": round(cold_dt / cold_steps * 1e3, 4),
                      "kernel_ms_per_step": round((pk1["kernel_ms"] - pk0["kernel_ms"]) / cold_steps, 4),
                      "ms_each_step": cold_each,
                      "sclk_ghz_samples": [round(v / 1e9, 2) for v in cold_clock.samples[:8]],
                      "what": "the same step right after 0.5 s of idle: no pre-roll, no warm-up (the clock governor needs ~20 ms "
                              "of load to leave its low state: tools/kernel_time_course.py, DESIGN.md 5)"}

    # Clock pre-roll (disclosed in the JSON line): on MI355X the first ~40 steps after an idle
    # period run 15 % slower than the steady state while the clock governor settles -- 185 us
    # per kernel against 159 us from ~20 ms of continuous load on (tools/kernel_time_course.py).
    t_pre = time.perf_counter()
    i = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        step(i)
        i += 1
    for i in range(args.warmup):
        step(i)
    if args.die_rank == rank and world > 1:
        os._exit(9)

    fence()
    p0 = dec.profile()  # counters accumulate over the handle's life: take differences
    with ClockSampler(local_rank) as sampler:
        dt, raw = timed_steps(step, args.steps, fence, torch.cuda.synchronize)
    clock = sampler.ghz()
    dt = max_over_ranks(dt)
    last = (args.steps - 1) % len(ptrs)

    value = world * n * args.steps / dt / 1e6  # Msamples/s, whole job
    frames = capi._frames_to_dicts(raw[0], raw[1])
    p1 = dec.profile()
    roofline, roofline_valu = roofline_objects(p0, p1, args.steps, profile_tag(args), clock, dt / args.steps * 1e3)

    # ---- the same step 1000 more times (the driver's default is 20 steps = 3 ms of measurement; box-to-box the kernel
    # spreads by 7 %): a sturdier sample of the same quantity, every rank, same fences
    steady = None
    if not args.no_extras:
        q0 = dec.profile()
        with ClockSampler(local_rank) as sampler2:
            dt2, _ = timed_steps(step, 1000, fence, torch.cuda.synchronize)
        dt2 = max_over_ranks(dt2)
        q1 = dec.profile()
        roof2, _ = roofline_objects(q0, q1, 1000, profile_tag(args), sampler2.ghz(), dt2 / 1000 * 1e3)
        steady = {"steps": 1000, "value": round(world * n * 1000 / dt2 / 1e6, 1), "unit": "Msamples/s",
                  "ms_per_step": round(dt2 / 1000 * 1e3, 4), "launch_ms": roof2["launch_ms"], "roofline_frac": roof2["frac"],
                  "what": "the timed region repeated with 1000 steps right behind the K steps of `value` (same captures in rotation, "
                          "same fences): 0.15 s of measurement instead of 3 ms"}

    # every capture of the rotation, decoded once more and kept for the gate
    per_capture = []
    for j in range(len(ptrs)):
        pj = step(j)
        per_capture.append(capi._frames_to_dicts(pj[0], pj[1]))
    if frames_key(per_capture[last]) != frames_key(frames):
        raise SystemExit("PARITY FAILURE: the last timed step and a repeat of the same capture differ")

    # ---- correctness gate on EVERY rank (each decodes its own stream: demod.c:86,99,125-141 and air.c:94-99 are
    # sequential rules that nothing else would check at N > 1) + the CPU baseline, timed on rank 0 only ----
    cpu = None
    parity = None
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        if world > 1:        # one rank (re)builds the checker's shared object, the others load it afterwards
            if rank == 0:
                O.build()
            ranks.fence(what="waiting for rank 0's build of the checker")
        ok, why = True, ""
        try:
            first = 0
            if rank == 0:
                cpu, want = cpu_reference(xs[0].cpu().numpy().view(np.uint16), df18)
                gate(per_capture[0], want, "capture 0 vs the CPU path")
                first = 1
            for j in range(first, len(xs)):  # the other captures of the rotation: against the oracle
                wj, _ = O.decode(xs[j].cpu().numpy().view(np.uint16), df18=df18)
                gate(per_capture[j], wj, f"rank {rank}, capture {j} vs the oracle")
        except SystemExit as e:  # a mismatch on one rank must fail the job, not hang the others in the all-reduce
            ok, why = False, str(e)
        if not all_ranks_ok(ok):
            raise SystemExit(why or "PARITY FAILURE on another rank")
        parity = True

    # ---- extra legs, after the timed region (rank 0, N == 1): dense sub-record, host-fed rates ----
    dense = None
    e2e = None
    with_stats = None
    multi = None
    sharded = None
    cli = None
    if rank == 0 and world == 1 and not args.no_extras:
        e2e = host_fed_rates(torch, capi, xs[0], df18)
        multi = multi_stream_host_fed(torch, capi, xs[0], df18)
        dsx = capi.Decoder(df18=df18, device=local_rank, collect_stats=True)
        px, mx = ptrs[0]
        want_x = frames_key(capi._frames_to_dicts(*dsx.decode_device_raw(px, mx)))
        sharded = host_fed_sharded(torch, capi, xs[0], df18, want_x, dsx.stats())
        dsx.close()
        cli = cli_whole_process(xs[0].cpu().numpy().view(np.uint16), capi, df18)
        if not args.stats:
            # the same step with valid.c's Try/Ok table reproduced too (the reference always keeps it and prints
            # it at exit): tries counted on the device beside the next scan; the table is read once, after the loop
            ds = capi.Decoder(df18=df18, device=local_rank, profile=True, collect_stats=True)

            def sstep(i=0):
                p, m = ptrs[i % len(ptrs)]
                return ds.decode_device_raw(p, m)
            preroll(sstep, args.preroll_ms)
            torch.cuda.synchronize()
            s0 = ds.profile()
            sdt, _ = timed_steps(sstep, 50, torch.cuda.synchronize)
            s1 = ds.profile()
            sroof, _ = roofline_objects(s0, s1, 50, profile_tag(args))
            with_stats = {"what": "collect_stats=1: the step above + the Try table of valid.c:84-100", "steps": 50, "preroll_ms": args.preroll_ms,
                          "value": round(n * 50 / sdt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(sdt / 50 * 1e3, 4),
                          "launch_ms": sroof["launch_ms"]}
            if not args.no_cpu_baseline:
                from oracle import oracle as O
                sstep(0)
                _, wst = O.decode(xs[0].cpu().numpy().view(np.uint16), df18=df18)
                got = ds.stats()
                if got["try"] != wst["try"] or got["ok"] != wst["ok"]:
                    raise SystemExit(f"PARITY FAILURE: Try/Ok table {got} != oracle {wst}")
                with_stats["table"] = {"try": got["try"], "ok": got["ok"]}
                with_stats["parity_vs_oracle"] = True
            ds.close()
            # the same statistics step with cfg.host_threads = 2 (a second host thread reads the hand-off stream)
            d2 = capi.Decoder(df18=df18, device=local_rank, profile=True, collect_stats=True, host_threads=2)

            def s2step(i=0):
                p, m = ptrs[i % len(ptrs)]
                return d2.decode_device_raw(p, m)
            preroll(s2step, args.preroll_ms)
            torch.cuda.synchronize()
            s2dt, s2raw = timed_steps(s2step, 50, torch.cuda.synchronize)
            with_stats["host_threads_2"] = {"value": round(n * 50 / s2dt / 1e6, 1), "ms_per_step": round(s2dt / 50 * 1e3, 4),
                                            "frames": int(s2raw[1]),
                                            "what": "opt-in: the reader thread shares the caller's L3 (placed by the library)"}
            d2.close()
        if not df18:
            del xs[1:], ptrs[1:]
            torch.cuda.empty_cache()
            dense = {}
            for key, ptag, make, what in (
                    ("noise", "_dense", lambda: make_dense(torch, n, 100),
                     "sigma=300 noise + one 112-bit frame per ms at amplitude 1200-2000 (what rounds 1-3 called configs[2]: the noise "
                     "alone makes 7 % of the offsets pass the preamble test)"),
                    ("target_10_percent", "_dense10", lambda: make_dense10(torch, n, 101),
                     "BASELINE configs[2] at its stated density: 112-bit frames packed back to back in sigma=300 noise, 3 % of the "
                     "1 ms slots filled with frame starts: ~10 % of the offsets pass the preamble test, ~100 k frames decode (123 k records per "
                     "launch: the handle's reader thread and its gang of four -- batches decided ahead, frames written by the gang -- "
                     "start by themselves behind the first launch, cfg.host_threads = 0)"),
                    ("gate_storm", "_storm", lambda: make_gate_storm(torch, n, 102),
                     "adversarial: frame starts (preamble + DF17's five bits) packed back to back: ~30 % of the offsets pass the "
                     "preamble test, ~7 % the DF gate -- every tile's survivor queue overflows and the tile is redone in ranges of chunks -- no CRC matches")):
                xd = make()
                rec = {"workload": f"{n} samples, -a: " + what, "steps": 20, "unit": "Msamples/s"}
                for stats in (False, True):
                    dd = capi.Decoder(df18=True, device=local_rank, profile=True, collect_stats=stats)

                    def dstep(_i=0):
                        return dd.decode_device_raw(xd.data_ptr(), xd.numel())
                    preroll(dstep, min(args.preroll_ms, 30.0), at_least=3)
                    torch.cuda.synchronize()
                    q0 = dd.profile()
                    ddt, draw = timed_steps(dstep, 20, torch.cuda.synchronize)
                    q1 = dd.profile()
                    droof, _ = roofline_objects(q0, q1, 20, profiles_tag=ptag)
                    r = {"value": round(n * 20 / ddt / 1e6, 1), "ms_per_step": round(ddt / 20 * 1e3, 4), "launch_ms": droof["launch_ms"],
                         "roofline_frac": droof["frac"], "frames": int(draw[1]), "relaunches": int(q1["relaunches"] - q0["relaunches"])}
                    if not args.no_cpu_baseline and not stats:
                        from oracle import oracle as O
                        xdh = xd.cpu().numpy().view(np.uint16)
                        wd, wds = O.decode(xdh, df18=True)
                        gate(capi._frames_to_dicts(*dstep()), wd, f"dense capture ({key}) vs the oracle")
                        rec["parity_vs_oracle"] = True
                        # the density BASELINE configs[2] is about, measured on this capture (not assumed)
                        rec["preamble_pass_fraction"] = round(preamble_pass_fraction(xdh), 5)
                        rec["df_gate_pass_fraction_of_visited"] = round(sum(wds["try"].values()) / max(1, n // 2), 6)
                        rec["oracle_stats"] = wds
                    if stats and "oracle_stats" in rec:
                        got = dd.stats()
                        if got["try"] != rec["oracle_stats"]["try"] or got["ok"] != rec["oracle_stats"]["ok"]:
                            raise SystemExit(f"PARITY FAILURE: Try/Ok table of the dense capture ({key}) {got} != oracle {rec['oracle_stats']}")
                        r["table_equals_oracle"] = True
                    rec["with_stats" if stats else "plain"] = r
                    dd.close()
                rec.pop("oracle_stats", None)
                dense[key] = rec
                del xd
                torch.cuda.empty_cache()
            dense["density_what"] = ("preamble_pass_fraction: share of the first 2 Mi offsets with p1 > 2 s1 && p2 > 2 s2 "
                                     "(demod.c:102-107), from the oracle's power samples; df_gate_pass_fraction_of_visited: "
                                     "the oracle's Try total (valid.c:46,68) over all offsets of the capture")
            # (the keys round 3's line carried at the top of `dense`, for the noise workload)
            dense.update({k: dense["noise"]["plain"][k] for k in ("value", "ms_per_step", "launch_ms", "roofline_frac", "frames", "relaunches")})

    if rank == 0:
        line = {
            "metric": "Msamples/s demodulated (20MSPS uint16 real), whole job",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "preroll_ms": args.preroll_ms, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "samples_per_gpu": n, "frames_decoded_rank0": len(frames),
                       "parity_vs_cpu": parity, "ranks_gated": world if parity else 0, "cpu_binding_rank0": cpu_binding},
            "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu,
            "value_dropin": None if with_stats is None else {
                "value": with_stats["value"], "unit": "Msamples/s", "ms_per_step": with_stats["ms_per_step"],
                "what": "the like-for-like figure of the drop-in: the same step with collect_stats=1, which is what the C host "
                        "program and the INTEGRATION.md patch run (the reference always keeps and prints valid.c's Try/Ok table); "
                        "`value` is the step without that table"},
            "value_cold": value_cold, "value_1000_steps": steady, "with_stats": with_stats, "dense": dense, "e2e_host_fed": e2e,
            "multi_stream_host_fed": multi, "e2e_host_fed_sharded": sharded, "cli_whole_process": cli,
        }
        emit_line(line)
    dec.close()
    ranks.close()


def pinned_capture(torch, total, seed, chunk=64 << 20):
    """The synthetic stream in page-locked host memory (built on the device slice by slice and copied over)."""
    host = torch.empty(total, dtype=torch.int16, pin_memory=True)
    for lo in range(0, total, chunk):
        hi = min(total, lo + chunk)
        host[lo:hi].copy_(make_workload(torch, total, seed=seed, lo=lo, hi=hi)[0])
    torch.cuda.synchronize()
    return host


def run_shard(args, torch, capi, rank, world, fence):
    """BASELINE configs[4]: ONE stream, time-sharded over N devices by the library's own multi-GPU driver
    (adsb_multi_*, csrc/multi.cpp): one process, a worker thread and a handle per device, no collective."""
    from adsbdec_amd import sharding
    n_dev = args.gpus
    if world > 1 and rank != 0:
        # Started as N ranks (torchrun): the driver of this mode is ONE process with a thread per device, so rank 0 runs
        # it over all N devices and the other ranks only keep the job's shape (they never touch a GPU).
        fence(sync=False)
        return
    devices = [0] * n_dev if args.one_device_test else list(range(n_dev))
    source = args.shard_source
    total = args.samples or ((2 << 30) if source == "device" else (512 << 20))
    total -= total % 28
    md = sharding.MultiDecoder(n_dev, devices, df18=True, profile=True, collect_stats=args.stats)
    plan = md.plan(total)
    keep = []
    dense10 = bool(args.dense10)   # BASELINE configs[2]'s traffic through the multi-GPU driver (configs[4] x configs[2])
    if args.dense or args.gate_storm:
        raise SystemExit("--mode shard takes the sparse capture (default) or --dense10")

    def capture(lo, hi):
        """Samples [lo, hi) of THE capture of this run, on the current device."""
        if not dense10:
            return make_workload(torch, total, seed=9, lo=lo, hi=hi)[0]
        if "whole" not in capture.__dict__:    # (the tiled generator makes a capture whole: once, on device 0)
            with torch.cuda.device(devices[0]):
                capture.whole = make_dense10(torch, total, 101)
        return capture.whole[lo:hi].to(torch.cuda.current_device(), copy=True)

    if source == "device":
        ptrs = []
        for i, p in enumerate(plan):
            with torch.cuda.device(devices[i]):
                t = capture(p["first_sample"], p["first_sample"] + p["n_samples"])
                torch.cuda.synchronize()
            keep.append(t)
            ptrs.append(t.data_ptr())

        def step(_i=0):
            return md.decode_device(total, ptrs)
    else:
        # The capture in page-locked host memory laid out shard by shard on the NUMA node of the device that pulls it
        # (adsb_multi_host_alloc, csrc/numa.cpp: where the reference has `iqbuff = malloc`, air.c:230).
        host_np, host_addr = md.host_alloc(total)
        host = torch.from_numpy(host_np.view(np.int16))
        for lo in range(0, total, 64 << 20):
            hi = min(total, lo + (64 << 20))
            host[lo:hi].copy_(capture(lo, hi))
        torch.cuda.synchronize()
        keep.append(host)
        if source == "file":
            import tempfile
            tf = tempfile.NamedTemporaryFile(suffix=".u16", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
            host.numpy().tofile(tf.name)
            keep.append(tf)

            def step(_i=0):
                return md.decode_file(tf.name)
        else:
            def step(_i=0):
                return md.decode_host(host.data_ptr(), total)

    step()
    preroll(step, args.preroll_ms, at_least=1)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    prof0 = [md.worker_profile(i) for i in range(len(plan))]
    serial, stitch, workers, each = [], [], [], []
    t0 = time.perf_counter()
    raw = None
    for _ in range(args.steps):
        ts0 = time.perf_counter()
        raw = step()
        each.append(round((time.perf_counter() - ts0) * 1e3, 4))
        inf = md.info()
        serial.append(inf["serial_us"])
        stitch.append(inf["stitch_us"])
        workers.append(inf["workers_ms"])
    dt = time.perf_counter() - t0            # (every call returns behind its workers: nothing is left in flight)
    prof1 = [md.worker_profile(i) for i in range(len(plan))]
    info = md.info()
    roofline, roofline_valu = roofline_objects(prof0[0], prof1[0], args.steps, step_ms=dt / args.steps * 1e3)
    value = total * args.steps / dt / 1e6
    frames = capi._frames_to_dicts(raw[0], raw[1])
    stats_table = md.stats() if args.stats else None

    cpu, parity = None, None
    if not args.no_cpu_baseline:
        if source == "device":
            whole = capture(0, total)
            x_host = None
        else:
            whole = keep[0].cuda()
            x_host = keep[0].numpy().view(np.uint16)
        # the N = 1 answer: the whole stream through one handle as one ordinary stream
        d1 = capi.Decoder(df18=True, device=devices[0], collect_stats=args.stats)
        d1.push_device_final(whole.data_ptr(), whole.numel())
        gate(frames, d1.drain(), f"{len(plan)}-way sharded stream vs the single-handle decode of the same stream")
        if args.stats and d1.stats() != stats_table:
            raise SystemExit(f"PARITY FAILURE: Try/Ok table of the sharded stream {stats_table} != single-handle decode {d1.stats()}")
        d1.close()
        parity = "equal to the single-handle decode of the same stream (one device, after the timed region)"
        if total <= (512 << 20):             # ... and, where the CPU gets through it in reasonable time, to the CPU path
            if x_host is None:
                x_host = whole.cpu().numpy().view(np.uint16)
            cpu, want = cpu_reference(x_host, True)
            gate(frames, want, "sharded stream vs the CPU path")
            parity += "; equal to the CPU path on the whole stream"
        del whole
    line = {
        "metric": "Msamples/s demodulated (20MSPS uint16 real), whole job",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": n_dev, "steps": args.steps,
        "warmup": args.warmup, "preroll_ms": args.preroll_ms, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[4] x configs[2] (112-bit frames packed back to back, ~10 % of the offsets pass the preamble "
                                "test: every worker's handle runs its reader thread and gang of four): " if dense10 else "BASELINE configs[4]: ")
                               + f"ONE stream of {total} uint16 samples time-sharded over {len(plan)} "
                               f"device handle(s) by adsb_multi_decode_{source} (one process, one worker thread per handle), halo 8 "
                               "pairs + one 1196-sample window, -a, 1-bit repair off; "
                               + {"device": "every slice resident in its device's HBM",
                                  "host": "the capture in page-locked host memory, every worker copies its slice over its device's "
                                          "link in 32 MiB pieces (PCIe-inclusive: never the headline `value`)",
                                  "file": "the capture in a tmpfs file, every worker preads its slice into page-locked buffers "
                                          "(read + PCIe inclusive: never the headline `value`)"}[source]
                               + "; every shard is resolved while its kernels run, the calling thread repairs the seams, hands out "
                                 "ts offsets, applies the end-of-file horizon and gathers the frames",
                   "shard_source": source, "devices": devices, "shards": len(plan), "samples_total": total,
                   "samples_shard0": plan[0]["n_samples"], "frames_decoded": len(frames), "parity": parity,
                   "statistics": stats_table, "fallback_steps": int(info["fallback"]),
                   "serial_us": round(float(np.median(serial)), 1),
                   "serial_what": "per step (median), on the calling thread behind the last worker: adsb_stitch_shards -- seam repair "
                                  "O(shards) + the walk of the deqframe call chain for the end-of-file horizon -- and the gather of "
                                  "the frames into one array (ts offsets applied on the way)",
                   "stitch_us": round(float(np.median(stitch)), 1),
                   "slowest_worker_ms": round(float(np.median(workers)), 4),
                   "ms_each_step": each,
                   "deqframe_calls_walked": int(info["calls_walked"]), "deqframe_calls_jumped": int(info["calls_jumped"]),
                   "create_ms": round(info["create_ms"], 1),
                   "helper_threads": int(info.get("helper_threads", 0)),
                   "helper_threads_per_worker": [int(p.get("host_threads_running", 0)) for p in prof1],
                   "gang_launches_per_worker": [int(p1.get("gang_launches", 0) - p0.get("gang_launches", 0)) for p0, p1 in zip(prof0, prof1)],
                   "helper_threads_what": "reader / gang threads the workers' handles own after the timed steps (adsb_multi_info.helper_threads, "
                                          "adsb_profile.host_threads_running per worker): 0 under sparse traffic, 5 per worker on a full channel",
                   "placement": ([md.placement(i) for i in range(len(plan))] if source == "host" else None),
                   "placement_what": "per worker: the NUMA node of its device, whether its thread runs on that node's CPUs, the node "
                                     "most pages of its slice of the capture live on and the share of them on the device's node "
                                     "(adsb_multi_worker_placement; the capture came from adsb_multi_host_alloc)"},
        "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu,
    }
    emit_line(line)
    md.close()
    if world > 1:
        fence(sync=False)


if __name__ == "__main__":
    main()
