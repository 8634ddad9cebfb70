bash tools/kb_session.sh "classic 0 7 -" "classicA 0 7 - -DADSB_ABLATE=2" "pipe 1 5 -" "pipeNoB 1 5 - -DADSB_PIPE_ABLATE=1" "pipeNoB_g3 1 5 3 -DADSB_PIPE_ABLATE=1" "pipeNoB_g2 1 5 2 -DADSB_PIPE_ABLATE=1" "pipeNoB_w4k7 1 7 - -DADSB_PIPE_ABLATE=1 -DADSB_PIPE_WAVES=4" "pipeNoB_w4k5 1 5 - -DADSB_PIPE_ABLATE=1 -DADSB_PIPE_WAVES=4" "pipeNoB_nost 1 5 - -DADSB_PIPE_ABLATE=1 -DADSB_SLEEP_STAGGER=0" "pipe_g3 1 5 3" "pipe_g2 1 5 2"
tools/build_variant.sh clk3 -DADSB_TILE_CLOCK=3 > /dev/null 2>&1
for p in 1 0; do
ADSB_CLOCK_OUT=1 ADSB_LIB_PATH=adsbdec_amd/lib_var/clk3/libadsbdec_amd.so ADSB_PIPE=$p timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --preroll-ms 0 2>&1 >/dev/null | tail -4 | tee -a gpurun_out/clk3_phases.txt
done
