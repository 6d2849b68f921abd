"""MI355X-native engine for the thz-image-explorer recompute path.

The product is `libthzgpu.so` (C ABI, include/thzgpu.h) built from `csrc/`.
This package only carries the ctypes binding used by tests and bench.py.
"""
from .binding import (Engine, DevBuf, ThzError, load_library, LIB_PATH, SYMBOLS,  # noqa: F401
                      host_frequency_axis, host_fft_window, host_adapted_blackman,
                      host_td_bandpass, host_fd_bandpass, host_tilt_plan)
from . import binding  # noqa: F401
