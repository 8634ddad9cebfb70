cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
bash tools/gpu_session_r3.sh ab "0 7 4 lib" "0 7 4 sw -DADSB_STAGED_WAIT=1"
ADSB_LIB_PATH=adsbdec_amd/lib_var/sw/libadsbdec_amd.so ADSB_PIPE=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "classic and (golden or seeded or full_range or beyond or real_reference)" 2>&1 | tail -3
bash tools/kb_session.sh "classicA 0 7 - -DADSB_ABLATE=2" "classicA_sw 0 7 - -DADSB_ABLATE=2 -DADSB_STAGED_WAIT=1" "classic 0 7 -" "classic_sw 0 7 - -DADSB_STAGED_WAIT=1" 2>&1 | grep -E "^classic"
