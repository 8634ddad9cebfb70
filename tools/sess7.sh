O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "two_rank" > $O/pytest_two_rank.log 2>&1; tail -3 $O/pytest_two_rank.log
tools/build_variant.sh gil -DADSB_GATE_IN_LOOP=1 > /dev/null 2>&1 || echo "variant gil failed to build"
ADSB_LIB_PATH=adsbdec_amd/lib_var/gil/libadsbdec_amd.so timeout 1500 python -m pytest tests -m gpu -q -x -k "classic and not two_rank and not cli" > $O/pytest_gil.log 2>&1; tail -3 $O/pytest_gil.log
bash tools/gpu_session_r3.sh ab "0 7 4 lib" "0 7 4 gil -DADSB_GATE_IN_LOOP=1" "0 7 4 lib" "0 7 4 gil -DADSB_GATE_IN_LOOP=1"
