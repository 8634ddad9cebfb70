"""Re-export of the build's synthetic-capture generator (tools/gen_signal.py) so that
tests can keep writing `from oracle import gen_signal`.  The generator is not part of
the oracle: bench.py imports it from tools/ directly."""
from tools.gen_signal import (FRAME_SAMPLES_LONG, FRAME_SAMPLES_SHORT, POLY, crc24, dense_capture,  # noqa: F401
                              frame_envelope, make_frame, sparse_capture, synth)
