"""Where does the host's time go in a step on the sparse headline capture, with and without the Try/Ok table?  Tuning builds
only (ADSB_DEBUG_HOST=1 makes the collect print its own breakdown per launch); this script runs itself as a child per
library and averages the lines of 200 steps."""
import os, re, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import torch
    from adsbdec_amd import capi
    from bench import make_workload, bind_near_gpu
    torch.cuda.set_device(0); bind_near_gpu(torch, 0)
    n = (256 << 20); n -= n % 28
    x = make_workload(torch, n, seed=1)[0]
    torch.cuda.synchronize()
    for stats in (False, True):
        d = capi.Decoder(df18=False, collect_stats=stats, profile=True)
        for i in range(100):
            d.decode_device_raw(x.data_ptr(), x.numel())
        t0 = time.perf_counter()
        for i in range(500):
            d.decode_device_raw(x.data_ptr(), x.numel())
        dt = (time.perf_counter() - t0) / 500 * 1e3
        sys.stderr.write(f"=== stats={int(stats)} {dt:.4f}\n"); sys.stderr.flush()
        os.environ["ADSB_DEBUG_HOST"] = "1"
        for i in range(200):
            d.decode_device_raw(x.data_ptr(), x.numel())
        del os.environ["ADSB_DEBUG_HOST"]
        d.close()
    sys.exit(0)
for rep in range(2):
    for v in ("r4base_tuning", "tree_tuning"):
        env = dict(os.environ, ADSB_LIB_PATH=os.path.join(os.getcwd(), "adsbdec_amd", "lib_ab", v, "libadsbdec_amd.so"))
        err = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stderr
        mode, acc = None, {}
        for ln in err.splitlines():
            m = re.match(r"=== stats=(\d) ([\d.]+)", ln)
            if m:
                mode = int(m.group(1)); acc[mode] = {"step_ms": float(m.group(2)), "collect": [], "resolve": [], "waits": [], "batches": [], "submit": [], "drain": []}
            m = re.match(r"stream collect: ([\d.]+) us in all, resolve ([\d.]+) us in (\d+) batches, waits ([\d.]+) us", ln)
            if m and mode is not None:
                a = acc[mode]; a["collect"].append(float(m.group(1))); a["resolve"].append(float(m.group(2))); a["batches"].append(int(m.group(3))); a["waits"].append(float(m.group(4)))
            m = re.match(r"push_device_final: submit ([\d.]+) us, drain ([\d.]+) us", ln)
            if m and mode is not None:
                acc[mode]["submit"].append(float(m.group(1))); acc[mode]["drain"].append(float(m.group(2)))
        for mode, a in acc.items():
            med = lambda k: sorted(a[k])[len(a[k]) // 2] if a[k] else float("nan")
            print(f"{v:14s} rep {rep} stats={mode}: step {a['step_ms']:.4f} ms (no debug) | medians of {len(a['collect'])} debug steps: collect {med('collect'):6.1f} us, "
                  f"resolve {med('resolve'):6.1f}, waits {med('waits'):5.1f}, batches {med('batches')}, submit {med('submit'):5.1f}, drain {med('drain'):6.1f}", flush=True)
