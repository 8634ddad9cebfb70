timeout 1500 python -m pytest tests/test_gpu_full_configs.py -m gpu -x -q 2>&1 | tail -3
timeout 400 python tools/fuzz_parity.py --seconds 150 --seed 90001 2>&1 | tail -2
timeout 300 python tools/async_race.py 150 2>&1 | cut -c1-120 | tail -13
