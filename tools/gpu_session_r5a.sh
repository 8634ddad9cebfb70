#!/bin/bash
# round 5, first GPU session: (1) device-side generator digests, (2) the default -m gpu suite from a from-source build, timed,
# (3) the driver's round-4 sequence that never ended (test_gpu_full_configs.py in file order WITH the opt-in 2^32-4 test) under
# a watcher that prints where every thread is if it does not end.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
t0=$(date +%s)
( time python -m adsbdec_amd._build --force ) > $O/r5a_build.txt 2>&1
python tests/test_generators.py --write gpu > $O/r5a_gpu_digests.txt 2>&1; cp tests/golden/generator_digests.json $O/
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/r5a_gpu_tests_run1.txt 2>&1
echo "suite rc=$? wall=$(( $(date +%s) - t0 )) s since the start of the build" >> $O/r5a_gpu_tests_run1.txt
tail -5 $O/r5a_gpu_tests_run1.txt
free -g > $O/r5a_box.txt; nproc >> $O/r5a_box.txt; rocm-smi --showmeminfo vram >> $O/r5a_box.txt 2>&1
python tools/run_with_stacks.py --after 240 --every 240 --times 3 --kill 1100 -- \
    python -X faulthandler -m pytest tests/test_gpu_full_configs.py -m gpu --gpu-big -x -q -s --durations=10 -o faulthandler_timeout=200 \
    > $O/r5a_big_sequence.txt 2>&1
echo "big rc=$?" >> $O/r5a_big_sequence.txt
tail -30 $O/r5a_big_sequence.txt
