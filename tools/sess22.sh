cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['launch_ms'], r['frac'])" 2>&1 | tail -1; }
nproc; lscpu | grep -E "Model name|Thread|Core|Socket|L3" 
# quick parity of both modes first
ADSB_HOST_THREADS=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for cfg in "1 1" "2 1" "2 0"; do
    read -r t pl <<< "$cfg"
    ADSB_HOST_THREADS=$t ADSB_READER_PLACE=$pl timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/ht_${t}_${pl}.json 2> $O/ht_${t}_${pl}.err
    echo "host_threads=$t place=$pl: $(line $O/ht_${t}_${pl}.json)"
    ADSB_HOST_THREADS=$t ADSB_READER_PLACE=$pl timeout 300 python bench.py --no-cpu-baseline --no-extras --stats > $O/hts_${t}_${pl}.json 2> $O/hts_${t}_${pl}.err
    echo "host_threads=$t place=$pl stats: $(line $O/hts_${t}_${pl}.json)"
  done
done | tee $O/ht_runs.txt
for t in 1 2; do echo "== host_threads=$t"; ADSB_HOST_THREADS=$t ADSB_DEBUG_HOST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>&1 >/dev/null | grep -E "stream collect|reader thread|push_device_final" | tail -9; done | tee -a $O/ht_runs.txt
