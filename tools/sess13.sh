O=gpurun_out
bash tools/gpu_session_r3.sh ab "0 7 4 lib" "0 7 4 nofin -DADSB_PARALLEL_FIN=0" "0 7 4 lib" "0 7 4 nofin -DADSB_PARALLEL_FIN=0"
tools/build_variant.sh clk3 -DADSB_TILE_CLOCK=3 > /dev/null 2>&1
tools/build_variant.sh clk3nofin -DADSB_TILE_CLOCK=3 -DADSB_PARALLEL_FIN=0 > /dev/null 2>&1
for v in clk3 clk3nofin; do
echo "== $v"; ADSB_CLOCK_OUT=1 ADSB_LIB_PATH=adsbdec_amd/lib_var/$v/libadsbdec_amd.so ADSB_PIPE=0 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --preroll-ms 0 2>&1 >/dev/null | tail -2
done | tee $O/clk3_parallel_fin.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "classic" 2>&1 | tail -2
