#!/bin/bash
# One gpurun session of round 6: tools/gpu_session_r6.sh <step> ...   (every output lands in gpurun_out/)
#   build        from-source build, timed
#   tests        the default -m gpu suite
#   bench        the default bench line -> r6_bench.json
#   phase        per-phase tile clocks of the stamps build (tools/phase_probe.py) -> r6_phase_stamps.txt
#   prof         rocprofv3 evidence sets (tools/profile_session.sh): r6 (sparse headline), r6_stats, r6_dense10, r6_dense10_stats,
#                r6_storm, r6_storm_stats
#   profdense    only the dense10 / storm sets
#   abparity     the parity tests that exercise the kernel's every path, once per library build under adsbdec_amd/lib_ab/
#   ab           same-box A/B of the library builds under adsbdec_amd/lib_ab/ against the tree's: sparse headline + dense captures
#   shard        bench.py --mode shard runs -> r6_bench_shard_*
#   cli          the C host program's per-stage timing on a 510 MiB tmpfs capture
#   fuzz [s]     tools/fuzz_parity.py for s seconds (default 300)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O tools/bin; export TMPDIR=/tmp
run() { out=$1; shift; timeout 1500 python bench.py "$@" > $O/$out.json 2> $O/$out.err; echo "$out: exit $? $(python -c "import json; d=json.load(open('$O/$out.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])" 2>&1 | tail -1)"; }
bench_line() { python bench.py --steps 1000 --warmup 50 --no-extras --no-cpu-baseline "$@" 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print({k: j.get(k) for k in ('value', 'ms_per_step')}, 'kernel ms', r.get('launch_ms'), 'frac', r.get('frac'))
"; }
while [ $# -gt 0 ]; do
  case $1 in
    build) ( time python -m adsbdec_amd._build --force ) > $O/r6_build.txt 2>&1; tail -4 $O/r6_build.txt;;
    tests) t0=$(date +%s); ( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/r6_gpu_tests.txt 2>&1
           echo "suite exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/r6_gpu_tests.txt; tail -24 $O/r6_gpu_tests.txt;;
    bench) run r6_bench;;
    phase) bash tools/build_variant.sh stamps -DADSB_PHASE_STAMPS > /dev/null 2>&1
           ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_var/stamps/libadsbdec_amd.so timeout 900 python tools/phase_probe.py > $O/r6_phase_stamps.txt 2> $O/r6_phase_stamps.err
           echo "phase exit $?"; grep -v "amdgpu.ids" $O/r6_phase_stamps.txt | head -120;;
    prof) bash tools/profile_session.sh r6; bash tools/profile_session.sh r6_stats --stats;&
    profdense)
          bash tools/profile_session.sh r6_dense10 --dense10; bash tools/profile_session.sh r6_dense10_stats --dense10 --stats
          bash tools/profile_session.sh r6_storm --gate-storm; bash tools/profile_session.sh r6_storm_stats --gate-storm --stats;;
    ab) { for rep in $(seq 1 ${AB_REPS:-2}); do for v in $(ls adsbdec_amd/lib_ab 2>/dev/null | grep -v -x "${AB_SKIP:-none}") tree; do
            if [ $v = tree ]; then unset ADSB_LIB_PATH; else export ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so; fi
            echo "== $v (rep $rep): bench.py --steps 1000 --no-extras --no-cpu-baseline"; bench_line
            echo "== $v (rep $rep): --dense10"; bench_line --dense10
            if [ "$AB_SHORT" = 2 ] || [ -z "$AB_SHORT" ]; then echo "== $v (rep $rep): --stats"; bench_line --stats; fi
            if [ -z "$AB_SHORT" ]; then
            echo "== $v (rep $rep): --dense10 --stats"; bench_line --dense10 --stats
            echo "== $v (rep $rep): --gate-storm"; bench_line --gate-storm --steps 300
            echo "== $v (rep $rep): --gate-storm --stats"; bench_line --gate-storm --stats --steps 300
            fi
          done; done; unset ADSB_LIB_PATH; } > $O/r6_ab.txt 2>&1; grep -v "amdgpu.ids" $O/r6_ab.txt | tail -60;;
    abparity) { for v in $(ls adsbdec_amd/lib_ab 2>/dev/null | grep -v "^r5$"); do
            echo "== $v"; ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q \
              -k "golden_device or seeded_vs or full_range or statistics_read or try_counting or back_to_back or queue_overflow or overflow_rounds or staged_list or one_bit or exhaustive or at_ten_percent" 2>&1 | tail -3
          done; } > $O/r6_abparity.txt 2>&1; cat $O/r6_abparity.txt;;
    why) shift; v=$1; ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_device" 2>&1 | grep -v amdgpu.ids | tail -40 > $O/r6_why_$v.txt; cat $O/r6_why_$v.txt;;
    shard)
      run r6_bench_shard_N1_2Gi --mode shard --steps 20 --warmup 3
      run r6_bench_shard_N1_2Gi_stats --mode shard --steps 10 --warmup 2 --stats
      run r6_bench_shard_8handles_one_device_2Gi --mode shard --gpus 8 --one-device-test --steps 10 --warmup 2 --stats
      run r6_bench_shard_host_fed_N1_512Mi --mode shard --shard-source host --steps 5 --warmup 1 --stats
      run r6_bench_shard_file_fed_N1_512Mi --mode shard --shard-source file --steps 5 --warmup 1 --stats
      run r6_bench_shard_host_fed_4handles_512Mi --mode shard --gpus 4 --one-device-test --shard-source host --steps 5 --warmup 1 --stats
      run r6_bench_shard_file_fed_4handles_512Mi --mode shard --gpus 4 --one-device-test --shard-source file --steps 5 --warmup 1 --stats
      run r6_bench_shard_dense10_1handle_512Mi --mode shard --dense10 --samples 536870912 --steps 10 --warmup 2 --stats
      run r6_bench_shard_dense10_4handles_512Mi --mode shard --dense10 --samples 536870912 --gpus 4 --one-device-test --steps 10 --warmup 2 --stats
      run r6_bench_shard_dense10_8handles_512Mi --mode shard --dense10 --samples 536870912 --gpus 8 --one-device-test --steps 10 --warmup 2 --stats
      run r6_bench_shard_dense10_8handles_2Gi --mode shard --dense10 --gpus 8 --one-device-test --steps 5 --warmup 1 --stats --no-cpu-baseline
      run r6_bench_stream_N8_one_device_plumbing --gpus 8 --one-device-test --samples 67108864 --steps 10 --warmup 2 --no-extras
      ;;
    cli) { python - <<'PY'
import numpy as np
rng = np.random.default_rng(5)
x = (2048 + rng.normal(0, 20, 255 << 20)).clip(0, 4095).astype(np.uint16)
x.tofile("/dev/shm/r6_cap.u16")
PY
           for i in 1 2 3; do ADSB_CLI_TIMING=2 adsbdec_amd/lib/adsbdec_amd_cli -f /dev/shm/r6_cap.u16 2>&1 >/dev/null | grep "push\|timing"; done
           echo "-G 1"; for i in 1 2 3; do ADSB_CLI_TIMING=1 adsbdec_amd/lib/adsbdec_amd_cli -G 1 -f /dev/shm/r6_cap.u16 2>&1 >/dev/null | grep timing; done
           echo "-G 0,0"; for i in 1 2 3; do ADSB_CLI_TIMING=1 adsbdec_amd/lib/adsbdec_amd_cli -G 0,0 -f /dev/shm/r6_cap.u16 2>&1 >/dev/null | grep timing; done
           rm -f /dev/shm/r6_cap.u16; } > $O/r6_cli.txt 2>&1; grep timing $O/r6_cli.txt;;
    clitrace) { export ADSB_CLI_CLEAN_EXIT=1 ADSB_CLI_TIMING=2; python - <<'PY'
import numpy as np
rng = np.random.default_rng(5)
x = (2048 + rng.normal(0, 20, 255 << 20)).clip(0, 4095).astype(np.uint16)
x.tofile("/dev/shm/r6_cap.u16")
PY
           cd /tmp; rm -rf /tmp/clitrace; timeout 300 rocprofv3 --hip-runtime-trace --hsa-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/clitrace -o run -- $OLDPWD/adsbdec_amd/lib/adsbdec_amd_cli -f /dev/shm/r6_cap.u16 > /dev/null 2> /tmp/clitrace.err; cd $OLDPWD
           python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/clitrace/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
rows.sort()
tid = {}
for f in glob.glob("/tmp/clitrace/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tid[(int(r["Start_Timestamp"]), r["Function"])] = r.get("Thread_Id", "?")
if rows:
    # the longest hipMemcpyAsync behind the first scan kernel = the push that stalls: every call of every thread around it
    t_first = rows[0][0]
    t_ready = max([b for a, b, f in rows if f.startswith("hipStreamCreate")] or [t_first])  # adsb_create is over
    late = [(b - a, a, b) for a, b, f in rows if f == "hipMemcpyAsync" and a > t_ready + 2e6]
    if late:
        d, a0, b0 = max(late)
        print(f"every HIP call from 3 ms before the slow hipMemcpyAsync ({d / 1e6:.2f} ms) to 1 ms behind it (start ms, duration ms, thread, call):")
        for a, b, f in rows:
            if a0 - 3e6 <= a <= b0 + 1e6:
                print(f"  {(a - t_first) / 1e6:9.3f} ms  {(b - a) / 1e6:8.3f} ms  {tid.get((a, f), '?'):>8}  {f}")
        hs = []
        for f in glob.glob("/tmp/clitrace/**/*hsa_api_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                hs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "?")))
        hs.sort()
        print(f"HSA calls inside that hipMemcpyAsync (all threads), 20 us and longer, and a count of the shorter ones:")
        short = {}
        for a, b, f, t in hs:
            if a0 <= a <= b0:
                if b - a >= 20000:
                    print(f"  {(a - t_first) / 1e6:9.3f} ms  {(b - a) / 1e6:8.3f} ms  {t:>8}  {f}")
                else:
                    short[f] = short.get(f, 0) + 1
        print("  shorter:", short)
if not rows:
    import subprocess
    print("no hip_api_trace rows; files under /tmp/clitrace:")
    print(subprocess.run("find /tmp/clitrace -type f | head -20; echo; tail -20 /tmp/clitrace.err", shell=True, capture_output=True, text=True).stdout)
    for f in glob.glob("/tmp/clitrace/**/*.csv", recursive=True)[:3]:
        print(f, open(f).readline().strip())
t0 = rows[0][0] if rows else 0
print("HIP calls longer than 0.4 ms (start ms, duration ms, call), and every call between 300 ms and the end that follows a hipMemcpyAsync:")
for a, b, f in rows:
    if b - a > 400000:
        print(f"  {(a - t0) / 1e6:9.2f} ms  {(b - a) / 1e6:8.2f} ms  {f}")
kr = []
for f in glob.glob("/tmp/clitrace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob("/tmp/clitrace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
kr.sort()
print("device timeline (start ms, duration ms):")
for a, b, n in kr[:120]:
    print(f"  {(a - t0) / 1e6:9.2f} ms  {(b - a) / 1e6:8.3f} ms  {n}")
PY
           echo "the program's own timing lines of the traced run:"; grep "push\|timing" /tmp/clitrace.err | head -40
           rm -f /dev/shm/r6_cap.u16; } > $O/r6_cli_trace.txt 2>&1; head -60 $O/r6_cli_trace.txt;;
    fuzz) shift; FZ=${1:-300}; timeout $((FZ + 300)) python tools/fuzz_parity.py --seconds $FZ --seed 960000 > $O/r6_fuzz.txt 2>&1; echo "fuzz exit $?"; tail -2 $O/r6_fuzz.txt | cut -c1-900;;
  esac
  shift
done
