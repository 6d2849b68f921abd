// deconv_host.hpp — host-side parts of the frequency-dependent deconvolution
// (FIR bank design, PSF spline evaluation, 2-D PSF construction): O(bands)
// scalar work the reference also does on the host side of its band loop.
#pragma once
#include "../../include/thzgpu.h"

#include <vector>

namespace thz {

constexpr int kDeconvTaps = 499;  // deconvolution.rs:167

float spline_eval(const thz_spline &s, float x);
float spline_eval_const(const thz_spline &s, float x);
float hybrid_eval(const thz_hybrid_fit &h, float f);

// create_filter_bank, deconvolution.rs:160-211
void filter_bank(int n_filters, double start_freq, double end_freq, double win_width,
                 const float *time, std::vector<float> &filters, std::vector<float> &centers);

struct BandPsf {
    int rows = 0, cols = 0;   // (x, y) extents, odd
    float wx = 0.0f;
    std::vector<float> v;     // rows*cols, row-major [x][y]: v[i][j] = fx[i] * fy[j]
    std::vector<float> fx, fy;  // the two profiles the array is the outer product of (rows / cols floats)
};
// per-band PSF, deconvolution.rs:906-960 + psf.rs:228-313
BandPsf band_psf(const thz_psf &P, float center_freq, float dx, float dy, int img_rows, int img_cols);

}  // namespace thz
