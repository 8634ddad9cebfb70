O=gpurun_out
timeout 1200 python tools/async_race.py 60 > $O/async_race.txt 2>&1; tail -22 $O/async_race.txt
timeout 900 python tools/fuzz_parity.py --seconds 420 --seed 500000 > $O/fuzz_420s.txt 2>&1; tail -3 $O/fuzz_420s.txt
