"""MI355X-native engine for the thz-image-explorer recompute path.

The product is `libthzgpu.so` (C ABI, include/thzgpu.h) built from `csrc/`.
This package only carries the ctypes binding used by tests and bench.py.
"""
from .binding import (Engine, DevBuf, ThzError, load_library, LIB_PATH, SYMBOLS,  # noqa: F401
                      host_frequency_axis, host_fft_window, host_adapted_blackman,
                      host_td_bandpass, host_fd_bandpass, host_tilt_plan,
                      host_water_line_mask, host_wiener_filter, host_psf_eval, host_filter_bank, host_band_psf, psf_from_npz, Psf, DeconvCfg, ChainCfg, Session, chain_cfg_default,
                      BUF_RAW, BUF_FFT, BUF_AMPLITUDES, BUF_PHASES, BUF_DATA, BUF_IMG, BUF_AVG_FFT,
                      BUF_AVG_AMPLITUDES, BUF_AVG_PHASES, BUF_OPACITY, VoxelCfg, VOXEL_INSTANCE, VOXEL_MAX_INSTANCES,
                      voxel_cfg_default, Group, GroupSession, host_slab, group_unique_id, GATHER_SMALL, GATHER_TIME, GATHER_ALL, PipelineIo,
                      host_optical_properties, host_align_reference, PlotOut, host_gaussian_kernel1d, host_select_step, host_select_value)
from . import binding  # noqa: F401
