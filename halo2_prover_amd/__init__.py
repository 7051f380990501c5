"""MI355X (gfx950) MSM / NTT backend for the halo2 hot path of 0xWOLAND/halo2-prover.

Python host layer over the C-ABI library libh2hip.so (include/h2hip.h).  Importing the
package does not touch the GPU; the first compute call (or `init()`) binds the process to
one device.  There is no CPU fallback.
"""
from .api import (Bases, ParamsKZG, best_fft, best_fft_batch, best_fft_group, best_multiexp, init, ntt_device)  # noqa: F401
from .lib import CURVES, H2Error, LIB_PATH, SYMBOLS, load  # noqa: F401
