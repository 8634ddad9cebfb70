"""Run N bench-sized steps with ADSB_DEBUG_HOST=1 and summarise the host's stream-collect lines."""
import os, re, subprocess, sys
env = dict(os.environ, ADSB_DEBUG_HOST="1")
out = subprocess.run([sys.executable, "tools/breakdown.py"], env=env, capture_output=True, text=True)
rows = [tuple(map(float, m.groups())) for m in re.finditer(
    r"stream collect: ([\d.]+) us in all, resolve ([\d.]+) us in (\d+) batches, waits ([\d.]+) us; ([\d.]+) us after", out.stderr)]
rows = rows[2:]
push = [float(m.group(1)) for m in re.finditer(r"push ([\d.]+) finish", out.stdout)][2:]
kern = [float(m.group(1)) for m in re.finditer(r"kernel ([\d.]+) ms", out.stdout)][2:]
import statistics as st
print(f"steps {len(rows)}: collect {st.mean(r[0] for r in rows):.1f} us, resolve {st.mean(r[1] for r in rows):.1f}, "
      f"waits {st.mean(r[3] for r in rows):.1f}, after-last-wait {st.mean(r[4] for r in rows):.1f} "
      f"(min {min(r[4] for r in rows):.1f} max {max(r[4] for r in rows):.1f}) | push {1e3*st.mean(push):.1f} us kernel {1e3*st.mean(kern):.1f} us")
