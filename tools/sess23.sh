cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reader_thread or fuzz" 2>&1 | tail -5
