/* CPU oracle for the NeRF volume-rendering weights -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the algorithm of the reference's native kernels
 * (reference src/cuda.cu:3-30 forward, :32-58 backward).  The reference has no CPU
 * implementation (its host wrappers reject CPU tensors, cuda.cu:62-64) and cannot be
 * built here (needs nvcc + a GPU), so this file is a restatement, pinned by
 * tests/test_oracle_weights.py: independent vectorised formulation, fp64 gradcheck,
 * hand known-answers.
 *
 * One ray = one (start,count) pair of `info`; rays own disjoint sample ranges.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may call this.
 */
#include <math.h>
#include <stdint.h>

/* cuda.cu:14-28.  T is fp32; `T * (1. - alpha)` has double literals in the reference, so
 * the product is formed in fp64 and rounded once to fp32 (cuda.cu:25).  Samples after the
 * first k with T <= threshold keep the caller's zero fill (cuda.cu:84 zeros_like). */
void oracle_weights_fwd(const float *sigmas, const float *steps, const int32_t *info,
                        float threshold, float *weights, int64_t n_rays)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const int32_t start = info[2 * r], count = info[2 * r + 1];
        if (count == 0) continue;
        const int32_t end = start + count;
        float T = 1.0f;
        int32_t k = start;
        while (T > threshold && k < end) {
            const float alpha = expf(-sigmas[k] * steps[k]);
            weights[k] = (float)((double)T * (1.0 - (double)alpha));
            T *= alpha;
            ++k;
        }
    }
}

/* cuda.cu:49-56.  Two passes, all fp32, NO early termination: terminated samples still get
 * grad = step * T * g. */
void oracle_weights_bwd(const float *sigmas, const float *steps, const int32_t *info,
                        const float *weights, const float *grad_weights, float *grad_sigmas,
                        int64_t n_rays)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const int32_t start = info[2 * r], count = info[2 * r + 1];
        if (count == 0) continue;
        const int32_t end = start + count;
        float acc = 0.0f, T = 1.0f;
        for (int32_t k = start; k < end; ++k) acc -= weights[k] * grad_weights[k];
        for (int32_t k = start; k < end; ++k) {
            acc += weights[k] * grad_weights[k];
            T *= expf(-sigmas[k] * steps[k]);
            grad_sigmas[k] = steps[k] * (acc + T * grad_weights[k]);
        }
    }
}

/* reference core.py:256-265: rendered[r] = sum_k w_k rgb_k (+ bg (1 - sum_k w_k)), summed
 * sequentially in sample order (what index_add_ does on CPU). */
void oracle_composite(const float *rgbs, const float *weights, const int32_t *info,
                      const float *bg, float *out_rgb, float *out_opacity, int64_t n_rays,
                      int has_bg)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const int32_t start = info[2 * r], count = info[2 * r + 1];
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, op = 0.f;
        for (int32_t k = start; k < start + count; ++k) {
            c0 += rgbs[3 * k + 0] * weights[k];
            c1 += rgbs[3 * k + 1] * weights[k];
            c2 += rgbs[3 * k + 2] * weights[k];
            op += weights[k];
        }
        if (has_bg) {
            c0 = c0 + bg[0] * (1.f - op);
            c1 = c1 + bg[1] * (1.f - op);
            c2 = c2 + bg[2] * (1.f - op);
        }
        out_rgb[3 * r + 0] = c0; out_rgb[3 * r + 1] = c1; out_rgb[3 * r + 2] = c2;
        if (out_opacity) out_opacity[r] = op;
    }
}
