O=gpurun_out
tools/build_variant.sh nores -DADSB_EARLY_RESERVE=0 > /dev/null 2>&1
tools/build_variant.sh nofin -DADSB_PARALLEL_FIN=0 > /dev/null 2>&1
for v in lib nores nofin; do
lib=adsbdec_amd/lib/libadsbdec_amd.so; [ $v != lib ] && lib=adsbdec_amd/lib_var/$v/libadsbdec_amd.so
echo "== $v"; ADSB_DEBUG_HOST=1 ADSB_LIB_PATH=$lib ADSB_PIPE=0 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --preroll-ms 0 2>&1 >/dev/null | grep -E "stream collect|push_device_final" | tail -6
done | tee $O/debug_host.txt
