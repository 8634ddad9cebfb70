O=gpurun_out
bash tools/gpu_session_r3.sh ab "0 7 4 lib" "0 7 4 nores -DADSB_EARLY_RESERVE=0" "0 7 4 lib" "0 7 4 nores -DADSB_EARLY_RESERVE=0"
tools/build_variant.sh clk3 -DADSB_TILE_CLOCK=3 > /dev/null 2>&1
tools/build_variant.sh clk3nores -DADSB_TILE_CLOCK=3 -DADSB_EARLY_RESERVE=0 > /dev/null 2>&1
for v in clk3 clk3nores; do
echo "== $v"; ADSB_CLOCK_OUT=1 ADSB_LIB_PATH=adsbdec_amd/lib_var/$v/libadsbdec_amd.so ADSB_PIPE=0 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --preroll-ms 0 2>&1 >/dev/null | tail -2
done | tee $O/clk3_early_reserve.txt
bash tools/gpu_session_r3.sh tests
