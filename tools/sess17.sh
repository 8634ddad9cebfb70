O=gpurun_out
timeout 1500 python tools/fuzz_parity.py --seconds 900 --seed 700000 > $O/fuzz_900s.txt 2>&1; tail -2 $O/fuzz_900s.txt | cut -c1-900
timeout 900 python -m pytest tests -m gpu -q -x -k "two_rank or decode_device or accepted_frame_log" 2>&1 | tail -2
