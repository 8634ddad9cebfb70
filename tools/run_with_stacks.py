#!/usr/bin/env python3
"""Run a command; if it is still alive after --after seconds, attach rocgdb and print every thread's NATIVE stack
(with --gdb; without it: every thread's kernel wait channel, syscall and CPU times from /proc) (then again every --every seconds, at most --times times); kill it at --kill seconds.

    python tools/run_with_stacks.py --after 240 --kill 900 -- python -m pytest tests -m gpu --gpu-big -k counter_limit -x -q

The child allows any process of this user to ptrace it (Yama: PR_SET_PTRACER_ANY), so the debugger -- a sibling, not an
ancestor -- can attach.  For the question "WHERE is a test that never returns stuck: Python, the library, the runtime?".
Never used by the product or the test suite itself.
"""
import argparse
import ctypes
import os
import shutil
import signal
import subprocess
import sys
import time

PR_SET_PTRACER = 0x59616D61


def allow_ptrace():
    try:
        ctypes.CDLL(None).prctl(PR_SET_PTRACER, ctypes.c_ulong(-1 & 0xFFFFFFFFFFFFFFFF), 0, 0, 0)
    except Exception:
        pass


def proc_summary(pid):
    out = []
    for tid in sorted(os.listdir(f"/proc/{pid}/task"), key=int):
        rec = {"tid": tid}
        for name in ("comm", "wchan", "syscall"):
            try:
                with open(f"/proc/{pid}/task/{tid}/{name}") as f:
                    rec[name] = f.read().strip()
            except OSError as e:
                rec[name] = f"<{e.strerror}>"
        try:
            with open(f"/proc/{pid}/task/{tid}/stat") as f:
                st = f.read().rsplit(")", 1)[1].split()
            rec["state"], rec["utime"], rec["stime"] = st[0], st[11], st[12]
        except OSError:
            pass
        out.append(rec)
    return out


def children(pid):
    kids = []
    for p in os.listdir("/proc"):
        if p.isdigit():
            try:
                with open(f"/proc/{p}/stat") as f:
                    if int(f.read().rsplit(")", 1)[1].split()[1]) == pid:
                        kids.append(int(p))
            except (OSError, ValueError, IndexError):
                pass
    return kids


def dump(pid, gdb):
    for p in [pid] + children(pid):
        print(f"\n===== pid {p}: threads (/proc) =====", flush=True)
        try:
            for rec in proc_summary(p):
                print("  ", rec, flush=True)
        except OSError as e:
            print("  ", e)
        if gdb:
            print(f"===== pid {p}: native stacks ({gdb}) =====", flush=True)
            try:
                r = subprocess.run([gdb, "-p", str(p), "-batch", "-nx", "-ex", "set pagination off",
                                    "-ex", "thread apply all bt 30"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   timeout=120, text=True)
                print(r.stdout[-30000:], flush=True)
            except Exception as e:  # noqa: BLE001
                print("   debugger failed:", e, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--after", type=float, default=240)
    ap.add_argument("--every", type=float, default=180)
    ap.add_argument("--times", type=int, default=3)
    ap.add_argument("--kill", type=float, default=900)
    ap.add_argument("--gdb", action="store_true", help="also attach rocgdb (off by default: attaching a debugger to a process "
                    "that holds a GPU queue switches the device into debug mode)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    gdb = None if not a.gdb else shutil.which("rocgdb") or ("/opt/rocm/bin/rocgdb" if os.path.exists("/opt/rocm/bin/rocgdb") else shutil.which("gdb"))
    t0 = time.time()
    child = subprocess.Popen(cmd, preexec_fn=allow_ptrace)
    next_dump, dumps = a.after, 0
    while child.poll() is None:
        time.sleep(1.0)
        el = time.time() - t0
        if el >= next_dump and dumps < a.times:
            print(f"\n##### still running after {el:.0f} s: {' '.join(cmd)}", flush=True)
            dump(child.pid, gdb)
            dumps += 1
            next_dump += a.every
        if el >= a.kill:
            print(f"\n##### killing after {el:.0f} s", flush=True)
            child.send_signal(signal.SIGABRT)   # faulthandler prints the Python stacks on the way out
            try:
                child.wait(20)
            except subprocess.TimeoutExpired:
                child.kill()
            return 124
    print(f"##### ended with {child.returncode} after {time.time() - t0:.0f} s", flush=True)
    return child.returncode


if __name__ == "__main__":
    sys.exit(main())
