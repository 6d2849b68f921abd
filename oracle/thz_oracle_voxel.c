/* thz_oracle_voxel.c — CPU restatement of the 3-D voxel envelope the data thread
 * rebuilds after every recompute (update_intensity_image, data_thread.rs:48-101 ->
 * gui/threed_plot.rs:80-276).  Included by thz_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY (see thz_oracle.c header).
 *
 * PARITY STATUS: the reference holds no test for this function; third-party pieces
 * restated from their published definitions: InstanceData {position[3], scale,
 * color[4]} of bevy_voxel_plot 5.0 (Cargo.toml:44; fields as used at
 * threed_plot.rs:260-264) and bevy_color's Srgba -> LinearRgba conversion
 * (bevy 0.19, Cargo.toml:18: x <= 0 -> x; x <= 0.04045 -> x/12.92; else
 * ((x+0.055)/1.055)^2.4, alpha unchanged).  "parity unpinned".
 */

/* gaussian_kernel1d, threed_plot.rs:80-101 */
void thz_oracle_gaussian_kernel1d(float sigma, int radius, float *kernel)
{
    const int size = 2 * radius + 1;
    const float sigma2 = 2.0f * sigma * sigma;
    float sum = 0.0f;
    for (int i = 0; i < size; ++i) {
        const float x = (float)i - (float)radius;
        const float value = expf(-x * x / sigma2);
        sum += value;
        kernel[i] = value;
    }
    for (int i = 0; i < size; ++i) kernel[i] /= sum;
}

/* The two rayon passes of instance_from_data, threed_plot.rs:163-200, per trace:
 * square (powi(2)), convolve1d with powf(contrast) on the input (103-120), then the
 * max / min rule.  `out` may not alias `data`. */
void thz_oracle_voxel_opacity(const float *data, size_t npix, int nt, float sigma, int radius,
                              float contrast, float opacity_threshold, float *out)
{
    const int size = 2 * radius + 1;
    float *kernel = (float *)malloc(sizeof(float) * (size_t)size);
    thz_oracle_gaussian_kernel1d(sigma, radius, kernel);
#pragma omp parallel
    {
        float *sq = (float *)malloc(sizeof(float) * (size_t)(nt > 0 ? nt : 1));
#pragma omp for schedule(static)
        for (long long p = 0; p < (long long)npix; ++p) {
            const float *x = data + (size_t)p * nt;
            float *line = out + (size_t)p * nt;
            for (int i = 0; i < nt; ++i) sq[i] = x[i] * x[i];
            for (int i = 0; i < nt; ++i) {
                float acc = 0.0f;
                for (int k = 0; k < size; ++k) {
                    const long j = (long)i + k - radius;
                    if (j >= 0 && j < nt) acc += powf(sq[j], contrast) * kernel[k];
                }
                line[i] = acc;
            }
            float mx = -INFINITY;
            for (int i = 0; i < nt; ++i) mx = fmaxf(mx, line[i]); /* f32::max ignores NaN like fmaxf */
            if (mx < opacity_threshold) {
                for (int i = 0; i < nt; ++i) line[i] = 0.0f;
            } else {
                float mn = INFINITY;
                for (int i = 0; i < nt; ++i) mn = fminf(mn, line[i]);
                if (fabsf(mx - mn) > 1e-6f) {
                    for (int i = 0; i < nt; ++i) line[i] = (line[i] - mn) / (mx - mn);
                } else {
                    for (int i = 0; i < nt; ++i) line[i] = 0.0f;
                }
            }
        }
        free(sq);
    }
    free(kernel);
}

static int cmp_desc_f32(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x < y) - (x > y);
}

/* effective threshold, threed_plot.rs:205-214: element max_instances-1 of the values in
 * descending order when there are more than max_instances of them, else 0.0 */
float thz_oracle_voxel_threshold(const float *opacity, size_t n, size_t max_instances)
{
    if (n <= max_instances) return 0.0f;
    float *tmp = (float *)malloc(sizeof(float) * n);
    memcpy(tmp, opacity, sizeof(float) * n);
    qsort(tmp, n, sizeof(float), cmp_desc_f32);
    const float t = tmp[max_instances - 1];
    free(tmp);
    return t;
}

typedef struct thz_oracle_instance {
    float position[3];
    float scale;
    float color[4];
} thz_oracle_instance;

static float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

/* bevy_color Srgba -> LinearRgba (see header) */
static float srgb_to_linear(float x)
{
    if (x <= 0.0f) return x;
    if (x <= 0.04045f) return x / 12.92f;
    return powf((x + 0.055f) / 1.055f, 2.4f);
}

/* instance loop, threed_plot.rs:141-160 and 221-271.  Returns the number of instances
 * (written up to `cap`); cube_dims = {cube_width, cube_height, cube_depth}. */
size_t thz_oracle_voxel_instances(const float *opacity, size_t gw, size_t gh, size_t gd, float threshold,
                                  float time_span, int scaling, size_t ow, size_t oh, size_t od,
                                  thz_oracle_instance *out, size_t cap, float *cube_dims)
{
    const float base = 1.0f / 4.0f;
    const float c = 300000000.0f;
    const float cube_depth = base / (time_span * c / 1.0e9f * 2.0f);
    const float spacing_w = ((float)ow * base) / (float)gw;
    const float spacing_h = ((float)oh * base) / (float)gh;
    const float spacing_d = ((float)od * cube_depth) / (float)gd;
    const float half_w = ((float)ow * base) / 2.0f;
    const float half_h = ((float)oh * base) / 2.0f;
    const float half_d = ((float)od * cube_depth) / 2.0f;
    if (cube_dims) { cube_dims[0] = base; cube_dims[1] = base; cube_dims[2] = cube_depth; }
    size_t n = 0;
    for (size_t x = 0; x < gw; ++x)
        for (size_t y = 0; y < gh; ++y)
            for (size_t z = 0; z < gd; ++z) {
                const float o = opacity[x * gh * gd + y * gd + z];
                if (o < threshold) continue;
                if (n < cap) {
                    const float v = (o - threshold) / (1.0f - threshold);
                    const float four = 4.0f * v;
                    const float r = clamp01(four - 1.5f);
                    const float g = clamp01(four - 0.5f) - clamp01(four - 2.5f);
                    const float b = 1.0f - clamp01(four - 1.5f);
                    thz_oracle_instance *q = out + n;
                    q->position[0] = (float)y * spacing_h - half_h;
                    q->position[1] = half_w - (float)x * spacing_w;
                    q->position[2] = half_d - (float)z * spacing_d;
                    q->scale = (float)scaling;
                    q->color[0] = srgb_to_linear(r);
                    q->color[1] = srgb_to_linear(g);
                    q->color[2] = srgb_to_linear(b);
                    q->color[3] = o;
                }
                ++n;
            }
    return n;
}
