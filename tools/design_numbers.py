#!/usr/bin/env python3
"""Regenerates the two measured tables of DESIGN.md (section 7 results, section 6 shard runs) from the files under profiles/,
between the <!-- r6-...:begin/end --> markers: every number in them is read from a committed file, none is typed.

    python tools/design_numbers.py          # rewrites DESIGN.md in place
"""
import csv
import json
import os
import re
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda name: os.path.join(ROOT, "profiles", name)
J = lambda name: json.load(open(P(name)))
B = 536866100  # algorithmic bytes of the dominant launch (2 B x 268 435 440 samples ... 4 B x 134 216 525 offsets)


def durations(name):
    return [int(r["Duration_ns"]) / 1e3 for r in csv.DictReader(open(P(name)))
            if "scan_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 600000]


def frac(us):
    return B / us / 1e3 / 8000  # bytes per microsecond = 1e-3 GB/s


def ab_vs_r5():
    """'sparse -4.6 % / -3.6 %; ...' from profiles/r6_ab_vs_r5.txt (the tree's line of every workload)."""
    out, wl = [], None
    for ln in open(P("r6_ab_vs_r5.txt")):
        m = re.search(r"workload ([^,]+),", ln)
        if m:
            wl = m.group(1).replace("dense10", "`configs[2]`").replace("gate_storm", "storm").replace("+stats", "+ table")
        m = re.match(r"\s+tree\s.*ratio to r5 \(median of rounds\) ([0-9.]+)\s+\| call [0-9.]+ ms, ratio ([0-9.]+)", ln)
        if m and wl:
            k, c = float(m.group(1)), float(m.group(2))
            # (with the table the other handle's count pass runs beside this one's scan: the call figures of those rows say
            # nothing about a process that owns the device alone, so only the kernel is quoted)
            out.append(f"{wl} {100 * (k - 1):+.1f} %" + ("" if "table" in wl else f" / {100 * (c - 1):+.1f} %"))
    return "; ".join(out)


def results():
    import ast
    fz = ast.literal_eval([ln for ln in open(P('r6_fuzz_600s.txt')) if ln.startswith('fuzz ok: ')][-1][len('fuzz ok: '):].strip())
    fz2 = ast.literal_eval([ln for ln in open(P('r6_fuzz.txt')) if ln.startswith('fuzz ok: ')][-1][len('fuzz ok: '):].strip())
    fz3 = ast.literal_eval([ln for ln in open(P('r6_fuzz_900s.txt')) if ln.startswith('fuzz ok: ')][-1][len('fuzz ok: '):].strip())
    fz4 = ast.literal_eval([ln for ln in open(P('r6_fuzz_1500s.txt')) if ln.startswith('fuzz ok: ')][-1][len('fuzz ok: '):].strip())
    d, u, pmc = J("r6_bench.json"), J("r6_bench_under_rocprofv3.json"), J("r6_pmc.json")
    dur, sdur = durations("r6_dispatches.csv"), durations("r6_stats_dispatches.csv")
    c = pmc["counters"]
    wc = c["SQ_WAVE_CYCLES"]["mean"]
    cnt = [r for r in csv.reader(open(P("r6_stats_kernel_stats.csv"))) if "count_tries_kernel" in r[0]][0]
    e, k = d["e2e_host_fed"], d["value_1000_steps"]
    sh, cli, dn = d["e2e_host_fed_sharded"], d["cli_whole_process"], d["dense"]
    x = cli.get("crossover") or {}
    files = " / ".join(f"{n}: {r['wall_ms']:.0f} ms against {r.get('reference_wall_ms', float('nan')):.0f} ms" for n, r in cli["files"].items())
    return f"""| what | value | file |
|---|---|---|
| `bench.py`, default line (one box) | **{d['value'] / 1e6:.3f} Tsamples/s**, {d['ms_per_step']:.4f} ms per 256 Mi-sample step; scan kernel {d['roofline']['launch_ms'] * 1e3:.1f} µs on its own clock = {d['roofline']['achieved']:.0f} GB/s = **{d['roofline']['frac']:.3f} of 8 TB/s** | `r6_bench.json` |
| ... the same region with 1000 steps instead of the driver's 20 (`value_1000_steps`) | {k['value'] / 1e6:.3f} Tsamples/s, {k['ms_per_step']:.4f} ms per step, kernel {k['launch_ms'] * 1e3:.1f} µs = {k['roofline_frac']:.3f} | `r6_bench.json` |
| the same command under `rocprofv3 --kernel-trace --stats`, 1000 timed steps (another box of the pool, a faster one: the bench line printed under the profiler is in the row) | `scan_kernel<false>`: {len(dur)} dispatches, average **{statistics.mean(dur):.1f} µs** (pre-roll and warm-up included) = {frac(statistics.mean(dur)):.3f}; last 1000: {statistics.mean(dur[-1000:]):.1f} µs = **{frac(statistics.mean(dur[-1000:])):.3f}**; minimum {min(dur):.1f} µs; the bench line of that very run read {u['roofline']['launch_ms'] * 1e3:.1f} µs in-kernel (1 % below the trace, as in every round) | `r6_kernel_stats.csv`, `r6_dispatches.csv`, `r6_bench_under_rocprofv3.json` |
| spread of the kernel | box to box 0.127–0.140 ms (0.48–0.53) for this build: the boxes' clocks differ (sysfs reads 2.23–2.39 GHz under the same load); rounds 3–5's builds 0.132–0.148 | `r6_bench.json` `roofline_valu.clock_ghz`, `r6_ab_*.txt` |
| against round 5's library, launches of both interleaved in ONE process (`tools/ab_interleaved.py`: kernel on its own clock / `adsb_decode_device` as the caller sees it) | {ab_vs_r5()} | `r6_ab_vs_r5.txt` |
| HBM traffic per launch | {pmc['hbm_bytes_per_launch'] / 1e6:.1f} MB = {pmc['hbm_bytes_per_launch'] / B:.3f} × the 536.9 MB of algorithmic input (FETCH_SIZE × 2 + WRITE_SIZE, separate passes) | `r6_pmc.json` |
| VALU wave-instructions per launch | {c['SQ_INSTS_VALU']['mean'] / 1e6:.2f} M; Stage A's pass {J('r6_isa_mix.json')['valu_instructions_per_pass']} instructions = {J('r6_isa_mix.json')['issue_cycles_per_pass']} issue cycles ({J('r6_isa_mix.json')['cycles_per_valu_instruction']:.2f} per instruction; rounds 1–5: 659–661 = 2 558) | `r6_pmc.json`, `r6_isa_mix.json` |
| wave time | issuing {c['SQ_ACTIVE_INST_ANY']['mean'] / wc * 100:.0f} %, stalled wanting to issue {c['SQ_WAIT_INST_ANY']['mean'] / wc * 100:.0f} %, parked on `s_waitcnt` / `s_barrier` {c['SQ_WAIT_ANY']['mean'] / wc * 100:.0f} % | `r6_pmc.json` |
| with the Try/Ok table (`collect_stats=1`: what the drop-in runs) | {d['with_stats']['ms_per_step']:.4f} ms per step = **{d['with_stats']['value'] / 1e6:.3f} Tsamples/s** (`value_dropin`); under rocprofv3 `scan_kernel<true>` {statistics.mean(sdur[-1000:]):.1f} µs (last 1000 of {len(sdur)}), `count_tries_kernel` {float(cnt[3]) / 1e3:.1f} µs on its own stream | `r6_bench.json`, `r6_stats_kernel_stats.csv`, `r6_stats_dispatches.csv` |
| dense, σ = 300 noise + one 112-bit frame per ms, `-a` (rounds 1–3's `configs[2]`) | {dn['noise']['plain']['ms_per_step']:.4f} ms per step, kernel {dn['noise']['plain']['launch_ms'] * 1e3:.1f} µs = {dn['noise']['plain']['roofline_frac']:.3f}; {dn['noise']['preamble_pass_fraction'] * 100:.2f} % of the offsets pass the preamble test, {dn['noise']['df_gate_pass_fraction_of_visited'] * 100:.3f} % the DF gate; with the table {dn['noise']['with_stats']['ms_per_step']:.4f} ms | `r6_bench.json` `dense.noise` |
| dense, **`configs[2]` at its stated density** (112-bit frames packed back to back in σ = 300 noise, 3 % of the ms slots full of frame starts) | **{dn['target_10_percent']['preamble_pass_fraction'] * 100:.2f} %** of the offsets pass the preamble test; {dn['target_10_percent']['plain']['frames']} frames per step; {dn['target_10_percent']['plain']['ms_per_step']:.3f} ms per step = {dn['target_10_percent']['plain']['value'] / 1e3:.1f} Gsamples/s (kernel {dn['target_10_percent']['plain']['launch_ms']:.3f} ms: {dn['target_10_percent']['plain']['ms_per_step'] / dn['target_10_percent']['plain']['launch_ms']:.2f} × — the handle's reader thread and gang of four at work, see below); with the table {dn['target_10_percent']['with_stats']['ms_per_step']:.3f} ms; frames and table equal to the oracle's | `r6_bench.json` `dense.target_10_percent` |
| ... the same capture as the MAIN workload under `rocprofv3` (`bench.py --dense10`, 1000 steps) | `scan_kernel<false>` average **{statistics.mean(durations("r6_dense10_dispatches.csv")):.1f} µs** = {frac(statistics.mean(durations("r6_dense10_dispatches.csv"))):.3f}; VALU wave-instructions {J("r6_dense10_pmc.json")["counters"]["SQ_INSTS_VALU"]["mean"] / 1e6:.1f} M per launch (sparse: {c['SQ_INSTS_VALU']['mean'] / 1e6:.1f} M), HBM traffic {J("r6_dense10_pmc.json")["hbm_bytes_per_launch"] / B:.3f} × algorithmic; with the table `scan_kernel<true>` {statistics.mean(durations("r6_dense10_stats_dispatches.csv")):.1f} µs, `count_tries_kernel` {float([r for r in csv.reader(open(P("r6_dense10_stats_kernel_stats.csv"))) if "count_tries_kernel" in r[0]][0][3]) / 1e3:.1f} µs | `r6_dense10_*`, `r6_dense10_stats_*` |
| adversarial: nothing but frame starts (`gate_storm`) | {dn['gate_storm']['preamble_pass_fraction'] * 100:.1f} % pass the preamble test, {dn['gate_storm']['df_gate_pass_fraction_of_visited'] * 100:.2f} % of ALL offsets the DF gate: every tile overflows its survivor queue; kernel {dn['gate_storm']['plain']['launch_ms']:.3f} ms = **{dn['gate_storm']['plain']['launch_ms'] / d['roofline']['launch_ms']:.1f} × the sparse launch**, step {dn['gate_storm']['plain']['ms_per_step']:.3f} ms; with the table step {dn['gate_storm']['with_stats']['ms_per_step']:.3f} ms (kernel {dn['gate_storm']['with_stats']['launch_ms']:.2f} ms, {dn['gate_storm']['with_stats']['relaunches']} relaunches); table equal to the oracle's | `r6_bench.json` `dense.gate_storm` |
| ... under `rocprofv3` (`bench.py --gate-storm`) | `scan_kernel<false>` average {statistics.mean(durations("r6_storm_dispatches.csv")):.1f} µs, VALU {J("r6_storm_pmc.json")["counters"]["SQ_INSTS_VALU"]["mean"] / 1e6:.1f} M; with the table `scan_kernel<true>` {statistics.mean(durations("r6_storm_stats_dispatches.csv")):.1f} µs and `count_tries_kernel` **{float([r for r in csv.reader(open(P("r6_storm_stats_kernel_stats.csv"))) if "count_tries_kernel" in r[0]][0][3]) / 1e3:.1f} µs** (round 5's build: 479 µs, `r5build_storm_stats_kernel_stats.csv`) | `r6_storm_*`, `r6_storm_stats_*` |
| cold (first steps after 0.5 s of idle) | {d['value_cold']['ms_each_step']} ms: only the FIRST step is slow; `value_cold` = {d['value_cold']['value'] / 1e6:.2f} Tsamples/s | `r6_bench.json` `value_cold` |
| host-fed, PCIe-inclusive (never `value`) | 1 Mi-sample calls: `adsb_push` {e['push_1Mi_sync'] / 1e3:.1f}, with `push_overlap` {e['push_1Mi_overlap'] / 1e3:.1f}, `adsb_push_async` {e['push_1Mi_async'] / 1e3:.1f} GS/s; 16 Mi: {e['push_16Mi_sync'] / 1e3:.1f} / {e['push_16Mi_overlap'] / 1e3:.1f} / {e['push_16Mi_async'] / 1e3:.1f}; 1 / 2 / 4 streams on one GPU: {d['multi_stream_host_fed']['streams_1']['aggregate'] / 1e3:.1f} / {d['multi_stream_host_fed']['streams_2']['aggregate'] / 1e3:.1f} / {d['multi_stream_host_fed']['streams_4']['aggregate'] / 1e3:.1f} GS/s | `r6_bench.json` `e2e_host_fed`, `multi_stream_host_fed` |
| the multi-GPU driver on ONE page-locked capture, Try/Ok table included, 1 / 2 / 4 handles on this one device (`e2e_host_fed_sharded`) | {sh['handles_1']['value'] / 1e3:.1f} / {sh['handles_2']['value'] / 1e3:.1f} / {sh['handles_4']['value'] / 1e3:.1f} GS/s (one link: plumbing, not scaling); calling thread's share {sh['handles_1']['serial_us']:.0f} / {sh['handles_2']['serial_us']:.0f} / {sh['handles_4']['serial_us']:.0f} µs per call | `r6_bench.json` |
| the C host program, whole process, exec to exit, against the reference's wall time on the same file | {files}; below **{x.get('samples', 0) / 1e6:.0f} M samples ({x.get('file_MB', 0)} MB, {x.get('seconds_of_signal', 0)} s of signal) the reference's one CPU thread finishes first** (start-up {x.get('startup_ms', 0):.0f} ms, then {x.get('gpu_ms_per_Mi_samples', 0):.2f} against {x.get('reference_ms_per_Mi_samples', 0):.2f} ms per Mi samples); every device visible to the runtime: {cli['largest_file_all_devices_visible']['runtime_init_ms']:.0f} ms of runtime start against {cli['runtime_init_ms']:.0f} (a one-GPU box: no difference to see); `-G 0,0`: {cli['largest_file_G_0_0']['wall_ms']:.0f} ms | `r6_bench.json` `cli_whole_process` |
| CPU beside it (the REAL reference chain, 1 core, EPYC 9575F) | {d['cpu_baseline']['value']:.0f} Msamples/s (the oracle's restatement: {d['cpu_baseline']['port_value']:.0f}) | `r6_bench.json` `cpu_baseline` |
| fuzz | {fz['seconds']:.0f} s, {fz['captures']} random captures ({fz.get('with_frame_start_storms', 0)} with stretches of nothing but frame starts, {fz['with_shrunken_record_buffers']} on handles with shrunken record buffers, {fz['also_checked_against_real_reference_chain']} also against the real reference chain) over ten feeding modes (the multi-GPU driver and resolved shards with statistics among them), {fz['frames']} frames: {fz['mismatches']} mismatches ({fz.get('multi_driver_captures_with_storms', 0)} storm captures through the multi-GPU driver, {fz['multi_driver_fallbacks']} fallbacks); the final build (the FIR's fused forms, measured passes per tile) for {fz2['seconds']:.0f} s + {fz3['seconds']:.0f} s + {fz4['seconds']:.0f} s (the last on other seeds, with the AVX2 host side like the first): {fz2['captures'] + fz3['captures'] + fz4['captures']} captures, {fz2['frames'] + fz3['frames'] + fz4['frames']} frames, {fz2['mismatches'] + fz3['mismatches'] + fz4['mismatches']} mismatches | `r6_fuzz_600s.txt` (the build before the FIR work), `r6_fuzz.txt`, `r6_fuzz_900s.txt`, `r6_fuzz_1500s.txt` |"""


def shards():
    rows = []
    for name, label in (("r6_bench_shard_N1_2Gi", "1 handle, 2 Gi samples resident in HBM"),
                        ("r6_bench_shard_N1_2Gi_stats", "... with the Try/Ok table"),
                        ("r6_bench_shard_8handles_one_device_2Gi", "8 handles on ONE device, 2 Gi samples resident, Try/Ok table"),
                        ("r6_bench_shard_host_fed_N1_512Mi", "1 handle, 512 Mi samples in page-locked host memory placed by `adsb_multi_host_alloc`, Try/Ok table"),
                        ("r6_bench_shard_file_fed_N1_512Mi", "1 handle, the capture as a tmpfs file (eight helper threads read ahead into a page-locked ring)"),
                        ("r6_bench_shard_host_fed_4handles_512Mi", "4 handles on ONE device (one link), the capture in host memory"),
                        ("r6_bench_shard_file_fed_4handles_512Mi", "4 handles, the capture as a tmpfs file"),
                        ("r6_bench_shard_dense10_1handle_512Mi", "**configs[4] × configs[2]**: 1 handle, 512 Mi samples of a full channel (112-bit frames back to back) resident in HBM, Try/Ok table"),
                        ("r6_bench_shard_dense10_4handles_512Mi", "... 4 handles on ONE device"),
                        ("r6_bench_shard_dense10_8handles_512Mi", "... 8 handles on ONE device"),
                        ("r6_bench_shard_dense10_8handles_2Gi", "... 8 handles on ONE device, 2 Gi samples (850 k frames)")):
        j = J(name + ".json")
        c = j["config"]
        rows.append(f"| {label} | {j['ms_per_step']:.3f} | {j['value'] / 1e3:.1f} | {c['slowest_worker_ms']:.3f} ms | {c['serial_us']:.0f} µs (stitch {c['stitch_us']:.0f}) | "
                    f"{c['deqframe_calls_walked']} / {c['deqframe_calls_jumped']} | {c['frames_decoded']} frames; {c['parity']} |")
    st8 = J("r6_bench_stream_N8_one_device_plumbing.json")
    rows.append(f"| `configs[3]` in the driver's command shape: 8 PROCESSES on one device, 64 Mi samples each (`bench.py --gpus 8 --one-device-test`) | {st8['ms_per_step']:.3f} | "
                f"{st8['value'] / 1e3:.1f} | — | — | — | every one of the {st8['config']['ranks_gated']} ranks gated against the oracle |")
    return ("| run (`bench.py --mode shard`, one process) | ms per call | Gsamples/s | slowest worker | calling thread behind it | deqframe calls walked / jumped | parity |\n"
            "|---|---|---|---|---|---|---|\n" + "\n".join(rows)).replace("\\n", "\n")


def dense_shards():
    rows = []
    for name, label in (("r6_bench_shard_dense10_1handle_512Mi", "1 handle, 512 Mi samples (212 k frames)"),
                        ("r6_bench_shard_dense10_4handles_512Mi", "4 handles, 512 Mi samples"),
                        ("r6_bench_shard_dense10_8handles_512Mi", "8 handles, 512 Mi samples"),
                        ("r6_bench_shard_dense10_8handles_2Gi", "8 handles, 2 Gi samples (850 k frames)")):
        j = J(name + ".json")
        c, r = j["config"], j["roofline"]
        med = statistics.median(c['ms_each_step'])
        rows.append(f"| {label} | **{med:.2f} ms** ({min(c['ms_each_step']):.2f}–{max(c['ms_each_step']):.2f}) | {j['config']['samples_total'] / med / 1e6:.0f} | "
                    f"{r['kernel_ms_per_step']:.2f} ms | {c['slowest_worker_ms']:.2f} ms | {c['serial_us']:.0f} µs | "
                    f"{'/'.join(str(t) for t in sorted(set(c['helper_threads_per_worker'])))} | {c['fallback_steps']} |")
    return ("| `bench.py --mode shard --dense10 --stats`, one process, every handle on ONE device | per call: median (range over the steps: a step in which a worker's helper threads are started or moved takes ~14 ms) | Gsamples/s at the median | "
            "kernels of worker 0 per call | slowest worker | calling thread behind it | helper threads per worker | fallbacks |\n"
            "|---|---|---|---|---|---|---|---|\n" + "\n".join(rows))


def file_source():
    h1, f1 = J("r6_bench_shard_host_fed_N1_512Mi.json"), J("r6_bench_shard_file_fed_N1_512Mi.json")
    h4, f4 = J("r6_bench_shard_host_fed_4handles_512Mi.json"), J("r6_bench_shard_file_fed_4handles_512Mi.json")
    return (f"**{f1['value'] / 1e3:.1f} of the host source's {h1['value'] / 1e3:.1f} Gsamples/s with one handle ({f1['value'] / h1['value']:.2f}), "
            f"{f4['value'] / 1e3:.1f} of {h4['value'] / 1e3:.1f} with four on the one link ({f4['value'] / h4['value']:.2f})** "
            "(`profiles/r6_bench_shard_file_fed_*`, `…host_fed_*`: one box, one call; run to run the file source spreads by 10 %, `r5_file_readers.txt`)")


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    t = open(path).read()
    for tag, text in (("results", results()), ("shards", shards()), ("file", file_source()), ("denseshards", dense_shards())):
        a, b = f"<!-- r6-{tag}:begin -->", f"<!-- r6-{tag}:end -->"
        assert a in t and b in t, tag
        t = t[: t.index(a) + len(a)] + "\n" + text + "\n" + t[t.index(b):]
    open(path, "w").write(t)
    print("DESIGN.md: tables regenerated from profiles/")


if __name__ == "__main__":
    main()
