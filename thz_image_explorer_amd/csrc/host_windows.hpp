// host_windows.hpp — see host_windows.cpp
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace thz {
void adapted_blackman(const float *axis, size_t len, float lower, float upper, float *out);
void fft_window(int type, const float *time, size_t nt, float lower, float upper, float *out);
void td_bandpass(const float *time, size_t nt, double *low, double *high, double width, float *out,
                 int64_t *lower_out, int64_t *upper_out);
void fd_bandpass(const float *freq, size_t nf, double low, double high, double width, float *out,
                 int64_t *lower_out, int64_t *upper_out);
void water_line_mask(const float *freq, size_t nf, const float *lines, size_t n_lines, float sigma,
                     float *out);
void wiener_filter(const float *ref_fft, size_t nf, float eps_rel, float *out);
size_t tilt_plan(const float *time, size_t nt, size_t nx, size_t ny, double tilt_x_deg,
                 double tilt_y_deg, float dx, float dy, float *new_time, int32_t *insert_index);
int align_reference(const float *scan_time, size_t nt, const float *ref_time, const float *ref_signal, size_t nref,
                    float *out);
bool reference_window(int type, const float *ref_time, size_t nref, float lower, float upper, size_t nt, float *win);
void optical_properties(const float *sample_amp, const float *sample_phase, const float *ref_amp,
                        const float *ref_phase, const float *freq, size_t nf, float thickness, float *n_out,
                        float *alpha_out, float *kappa_out);
void gaussian_kernel1d(float sigma, int radius, float *out);
struct VoxelLayout {
    float cube_width, cube_height, cube_depth;
    float spacing_w, spacing_h, spacing_d, half_w, half_h, half_d;
};
VoxelLayout voxel_layout(float time_span, size_t gw, size_t gh, size_t gd, size_t ow, size_t oh, size_t od);
int select_step(const unsigned long long *hist, int nbins, unsigned long long k, int *bin,
                unsigned long long *k_rem);
}  // namespace thz
