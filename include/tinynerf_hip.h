/* tinynerf_hip.h -- C ABI of libtinynerf_hip.so, the MI355X (gfx950) implementation of the
 * tinynerf ray-marching hot path.
 *
 * This is the drop-in boundary.  In the reference the native boundary is a pybind11 module
 * JIT-built at import time (reference src/core.py:7) with exactly two entry points,
 *     compute_weights_fwd(sigmas, steps, info, threshold) -> weights      (src/cuda.cu:66-95)
 *     compute_weights_bwd(sigmas, steps, info, weights, grad) -> grad_sig (src/cuda.cu:97-132)
 * Everything else on the path is ATen calls made from src/core.py and src/models.py; this
 * library gives each of those call sites one entry point as well (the cited file:line is
 * what the function replaces).
 *
 * Conventions
 *  - plain C: raw device pointers + sizes, no torch types; `stream` is a hipStream_t passed as
 *    void* (NULL = the null stream).  All work is enqueued asynchronously on that stream.
 *  - every function returns 0 on success, a negative TN_E_* code for a rejected argument, or a
 *    positive hipError_t; tn_last_error_string() describes the last failure of the calling thread.
 *  - the caller owns every buffer.  The library never allocates, frees or retains a pointer and
 *    keeps no mutable global state, so calls are re-entrant (the autograd engine calls the
 *    backward entry points from its own thread, reference src/core.py:203-207).
 *  - all tensors are dense row-major fp32 unless stated; `info` is int32 [n_rays,2] = (start,count)
 *    per ray, the reference's packing_info (src/core.py:165-188); rays own disjoint ranges.
 */
#ifndef TINYNERF_HIP_H
#define TINYNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TN_ABI_VERSION 6

enum {
    TN_OK = 0,
    TN_E_NULL = -1,      /* a required pointer is NULL */
    TN_E_SIZE = -2,      /* negative / inconsistent / unsupported size */
    TN_E_CONFIG = -3,    /* unsupported enum or layer configuration */
    TN_E_ALIGN = -4      /* pointer not aligned as the entry point requires */
};

const char *tn_last_error_string(void);
/* Most recent performance warning of the process ("" if none): a VALID call that landed on a general-shape fallback kernel -- a wide
 * stack whose first layer / layer shapes are not those of the reference's configurations (src/run.py:131-150) runs several times
 * slower than they do.  Said once per fallback on stderr as well (TN_QUIET=1 silences that).  Never an error: the results are right. */
const char *tn_last_warning_string(void);
int tn_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * a17  NeRF volume-rendering weights            (reference src/cuda.cu:3-58, core.py:192-207)
 * fwd: per ray T=1; w_k = T*(1-exp(-sigma_k*step_k)); T *= exp(..) while T > threshold; every
 *      sample of the ray after termination gets 0 (the kernel writes the whole range, `weights`
 *      does not need to be zeroed).  bwd: cuda.cu:49-56, no early termination.
 * ------------------------------------------------------------------------------------------ */
int tn_weights_fwd(const float *sigmas, const float *steps, const int32_t *info, float threshold,
                   float *weights, int64_t n_samples, int64_t n_rays, void *stream);
/* tn_weights_fwd that also raises gate[0] to 1.0 when any weight is > 0 (never lowers it: the caller zeroes it).  The
 * harness' device-side form of the reference's "Empty iteration" test (core.py:251-254: `(weights > 0).any()`). */
int tn_weights_fwd_gate(const float *sigmas, const float *steps, const int32_t *info, float threshold,
                        float *weights, float *gate, int64_t n_samples, int64_t n_rays, void *stream);
int tn_weights_bwd(const float *sigmas, const float *steps, const int32_t *info,
                   const float *weights, const float *grad_weights, float *grad_sigmas,
                   int64_t n_samples, int64_t n_rays, void *stream);

/* ------------------------------------------------------------------------------------------
 * a18  per-ray compositing                                     (reference core.py:256-265)
 * rendered[r] = sum_k w_k rgb_k (+ bg*(1-sum_k w_k) if bg != NULL); opacity[r] optional.
 * bwd: grad_rgbs[k] = w_k*g[r];  grad_weights[k] = <rgb_k, g[r]> - <bg, g[r]>.
 * ------------------------------------------------------------------------------------------ */
int tn_composite_fwd(const float *rgbs, const float *weights, const int32_t *info, const float *bg,
                     float *rendered, float *opacity, int64_t n_samples, int64_t n_rays, void *stream);
int tn_composite_bwd(const float *rgbs, const float *weights, const int32_t *info, const float *bg,
                     const float *grad_rendered, float *grad_rgbs, float *grad_weights,
                     int64_t n_samples, int64_t n_rays, void *stream);

/* ------------------------------------------------------------------------------------------
 * a5/a6  occupancy grid                                        (reference core.py:93-156)
 * grid is fp32 [D,H,W]; coords[...,0] indexes W (x), 1 -> H (y), 2 -> D (z), in [-1,1].
 * ------------------------------------------------------------------------------------------ */
/* core.py:147-156: out[i] = trilinear(grid, coords[i]) > threshold, bit-exact w.r.t. ATen's
 * grid_sampler_3d (align_corners=True, zeros padding).  values (optional) receives the taps. */
/* coarse[bz,by,bx] = max of grid over z in [4bz, 4bz+4], y, x likewise (clipped): every tap a point whose floor cell lies
 * in block (bz,by,bx) can read.  Rebuild after every change of the grid. */
int tn_occupancy_coarsen(const float *grid, int D, int H, int W, float *coarse, void *stream);
int tn_occupancy_query(const float *grid, int D, int H, int W, const float *coords, int64_t n,
                       float threshold, uint8_t *out, float *values, void *stream);
/* core.py:136: jittered voxel centres of depth slice `slice`: out[h*W+w] =
 * -1 + 2*((w,h,slice) + jitter)/(D,H,W) (the reference divides the flipped (x,y,z) index by the
 * un-flipped size vector; reproduced).  jitter [H,W,3] may be NULL: then a counter-based RNG
 * keyed by (seed, slice, h, w, c) supplies U[0,1). */
int tn_occupancy_slice_coords(int D, int H, int W, int slice, const float *jitter, uint64_t seed,
                              float *coords, void *stream);
/* core.py:138-143: alpha = 1-exp(-sigma*step); grid = alpha > thr ? 1 : decay*grid, over `n`
 * consecutive cells starting at grid_cells. */
int tn_occupancy_apply(float *grid_cells, const float *sigmas, int64_t n, float step_size,
                       float threshold, float decay, void *stream);
/* core.py:121-123,144: stats[0] = sum(grid) (fp64), stats[1] = #(grid > threshold) as fp64.
 * stats is a device buffer of 2 doubles, overwritten. */
int tn_occupancy_stats(const float *grid, int64_t n, float threshold, double *stats, void *stream);

/* ------------------------------------------------------------------------------------------
 * a1-a4, a7  ray marching + sample packing                    (reference core.py:11-88,158-188)
 * ------------------------------------------------------------------------------------------ */
enum { TN_MARCH_AABB = 0, TN_MARCH_UNBOUNDED = 1 };
enum { TN_CONTRACT_AABB = 0, TN_CONTRACT_MIP360_INF = 1, TN_CONTRACT_MIP360_L2 = 2 };

typedef struct tn_sampler_desc {
    int32_t marcher;          /* TN_MARCH_*    (core.py:36-88)                               */
    int32_t contraction;      /* TN_CONTRACT_* (core.py:11-31)                               */
    int32_t n_samples;        /* candidates per ray S                                        */
    int32_t grid_d, grid_h, grid_w;
    float aabb[6];            /* lo xyz, hi xyz (marcher AABB and/or contraction AABB)       */
    float near, far;          /* AABB marcher clamp (core.py:81)                             */
    float step_size;          /* AABB marcher: ||hi-lo||/S as computed by the caller (fp32)  */
    float threshold;          /* occupancy threshold min(base, mean) (core.py:125-127)       */
    const float *t_table;     /* unbounded marcher: t[S] and delta[S] (core.py:52-58)        */
    const float *delta_table;
    const float *grid;        /* occupancy grid [D,H,W]                                      */
    const float *jitter;      /* training: U[0,1) [R,S] (core.py:173) or NULL                */
    uint64_t seed;            /* training with jitter==NULL && use_rng: counter-based RNG    */
    int32_t use_rng;
    int32_t reserved;
    const float *coarse;      /* optional [ceil(D/4),ceil(H/4),ceil(W/4)] block maxima written by tn_occupancy_coarsen:
                               * candidates whose 2x2x2 taps all lie under a block whose maximum is safely below the
                               * threshold are rejected without reading the taps (same mask, bit for bit); NULL = off */
} tn_sampler_desc;

/* Stand-alone marcher / contraction calls (core.py:47-59,72-88 and core.py:15-31) with the same
 * arithmetic as the fused sampler: t,delta [R,S]; coords_out [n,3], mask [n] (AABB only, else NULL). */
int tn_march_rays(const tn_sampler_desc *desc, const float *rays_o, const float *rays_d, int64_t n_rays,
                  float *t_values, float *step_sizes, void *stream);
int tn_contract(const tn_sampler_desc *desc, const float *coords, int64_t n, float *coords_out,
                uint8_t *mask, void *stream);

/* pass 1: per-ray occupancy bitmask ([R, ceil(S/64)] uint64) and kept-sample count. */
int tn_sample_mask(const tn_sampler_desc *desc, const float *rays_o, const float *rays_d,
                   int64_t n_rays, uint64_t *maskbits, int32_t *counts, void *stream);
/* pass 2: info[r] = (exclusive_scan(counts)[r] + base_offset[0], counts[r]); total[0] = sum.
 * base_offset (device int32, may be NULL = 0) implements run.py:231 `info[:,0] += current_size`. */
int tn_sample_scan(const int32_t *counts, int64_t n_rays, const int32_t *base_offset,
                   int32_t *info, int32_t *total, void *stream);
/* a8 (reference run.py:215-244), device side: the harness draws loader batches of `batch_size` rays until
 * int(cur*(1+1/k)) >= target.  Given the kept-sample counts of n_rays = n_batches*batch_size candidate rays,
 * plan[0] = k (loader batches consumed), plan[1] = N (packed samples), plan[2] = R = k*batch_size rays,
 * plan[3] = 1 if the rule tripped within the supplied rays (else every supplied batch is used). */
int tn_batch_plan(const int32_t *counts, int64_t n_rays, int32_t batch_size, int64_t target, int32_t *plan,
                  void *stream);
/* tn_batch_plan and tn_sample_scan(counts, n_rays, NULL, info, NULL) over ALL candidate rays as one multi-workgroup launch
 * (the training step's form: the harness uses the first plan[2] rows of info once the plan is on the host, so that nothing but
 * tn_sample_pack is left behind the step's read-back; reference run.py:215-244 + core.py:179-181). */
int tn_batch_plan_scan(const int32_t *counts, int64_t n_rays, int32_t batch_size, int64_t target, int32_t *plan,
                       int32_t *info, void *stream);
/* pass 3: packed[start_r - base + j] = (contracted xyz, ray dir, step) for the j-th set bit
 * (core.py:182-186).  ray_ids / steps (optional) receive the ray index and the step size (column 6) of every packed
 * sample as contiguous arrays (what the weights kernels and the per-ray colour-head table index). */
int tn_sample_pack(const tn_sampler_desc *desc, const float *rays_o, const float *rays_d,
                   int64_t n_rays, const uint64_t *maskbits, const int32_t *info,
                   const int32_t *base_offset, float *packed, int32_t *ray_ids, float *steps, int64_t capacity,
                   void *stream);

/* ------------------------------------------------------------------------------------------
 * a9  positional encoding                                      (reference models.py:30-39)
 * out[n, c*2F + f] = sin(x[n,c]*freqs[f]), out[n, c*2F + F + f] = cos(..)
 * ------------------------------------------------------------------------------------------ */
int tn_posenc_fwd(const float *x, int64_t n, int n_channels, const float *freqs, int n_freqs,
                  float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * a10-a14  fused MLP heads on fp32 MFMA                        (reference models.py:7-89)
 * One launch evaluates  act_out( W_L ... relu(W_1 relu(W_0 enc(x) + b_0) + b_1) ... + b_L ).
 * Weights are torch nn.Linear layout [out,in] row-major.
 * ------------------------------------------------------------------------------------------ */
enum { TN_ACT_NONE = 0, TN_ACT_EXP_M1 = 1 /* exp(y-1), models.py:74 */, TN_ACT_SIGMOID = 2,
       TN_ACT_EXP = 3 /* tn_basis_dot_* only: exp(y) with the clamped backward of models.py:42-53 */ };
enum { TN_ENC_NONE = 0,
       TN_ENC_POSENC = 1,       /* input = PE_F(x[:, :3])                      (models.py:67)    */
       TN_ENC_DIR_CAT = 2,      /* input = cat[PE_F(dirs), dirs, x]            (models.py:87)    */
       TN_ENC_AUX_CAT = 3 };    /* input = cat[aux_table[aux_index[row]], x]: the same colour-head input with
                                 * cat[PE_F(d), d] evaluated once per RAY (tn_dir_encode) instead of per sample */
#define TN_MLP_MAX_LAYERS 12
#define TN_MLP_ACCUM_GRAD_X 1   /* tn_mlp_bwd: grad_x += instead of = (two heads sharing one feature tensor) */
#define TN_MLP_STASHED 2        /* tn_mlp_bwd: workspace holds the activations written by tn_mlp_fwd_stash
                                 * (same desc, x, aux, n): no forward recomputation */
#define TN_MLP_CHAIN_ONLY 4     /* tn_mlp_bwd_pair: data gradients only (grad_x and the G rows of the workspace) ...   */
#define TN_MLP_WGRAD_ONLY 8     /* ... weight / bias gradients only, from a workspace a CHAIN_ONLY call completed.  The split
                                 * lets a caller start consuming grad_x (e.g. scatter it and launch a gradient all-reduce)
                                 * while the weight gradients are still being computed. */
#define TN_MLP_GRAD_Y_ROWS 16   /* tn_mlp_bwd of a layer-by-layer configuration (tn_mlp_rows_view): d loss / d y already sits in the
                                 * workspace as [feature][32-sample] rows (a consumer's grad_x_rows); grad_y is ignored */

#define TN_MLP_BF16X3 32        /* wide stacks evaluated layer by layer (width 128 / 256): matrix products on the bf16 matrix cores with
                                 * EXACT three-way operand splits (x = hi + mid + lo in bf16, six of the nine partial products,
                                 * fp32 accumulate: every product good to 2^-23, the error of an fp32 product's own rounding) instead
                                 * of v_mfma_f32_32x32x2_f32 -- 2.67 x the matrix rate, results equal to fp32 rounding.  Ignored by
                                 * configurations without such a form. */

#define TN_MLP_F16X2 64         /* as TN_MLP_BF16X3, with the hidden layers on the fp16 matrix cores instead: two-term operand splits
                                 * (x s = hi + lo in fp16: 22 of fp32's 24 significand bits), three partial products, power-of-two scales
                                 * -- per sample column and per layer in the forward / data gradient, per operand and launch in the
                                 * weight gradient -- that are taken out of the fp32 accumulators again: no overflow whatever the
                                 * data, the same distance from an fp64 evaluation as the fp32 MFMA, half the matrix time of bf16x3
                                 * (csrc/mlp_f2_layers.hip).  The workspace's last 256 bytes carry the per-layer maxima between
                                 * the launches of a step.  Default of tinynerf_amd.models. */

#define TN_MLP_ROWS_ONLY 128    /* tn_mlp_fwd_stash of a layer-by-layer stack that offers row views (tn_mlp_rows_view) under TN_MLP_F16X2:
                                 * y is NOT written -- the output exists only as the workspace's [feature][32-sample] rows, which the
                                 * consumer reads through tn_mlp_desc::x_rows (TN_MLP_X_FROM_ROWS).  Saves the 4 * out bytes per
                                 * sample of the row-major copy.  `y` must still be a valid pointer. */
#define TN_MLP_X_FROM_ROWS 256  /* tn_mlp_fwd_stash of a width-64 head: x is read from x_rows ONLY (the producer ran with
                                 * TN_MLP_ROWS_ONLY; `x` is not dereferenced).  TN_E_CONFIG unless the launch is one that reads the row
                                 * view: TN_MLP_F16X2, in_dim 128 or 256, <= 4 outputs.  Without the flag such a launch still
                                 * prefers x_rows when they are set (coalesced 128-byte rows instead of 16 bytes per lane and sample). */

#define TN_MLP_SKIP_LAST 1024   /* a wide stack evaluated layer by layer whose LAST layer is a plain Linear feeding only Linear first layers of
                                 * its consumers (reference models.py:59-89: Linear(256, 256) then Linear(256, 64) twice with nothing in
                                 * between): W_head (W_last h + b_last) + b_head = (W_head W_last) h + (W_head b_last + b_head), so the caller
                                 * merges the two layers' parameters and the stack stops at its last HIDDEN activation.  tn_mlp_fwd_stash
                                 * then runs layers 0 .. L - 2 only (y is not written), tn_mlp_rows_view_hidden says where that activation,
                                 * the slot for its gradient and its ReLU bit rows are, and tn_mlp_bwd (with TN_MLP_STASHED |
                                 * TN_MLP_GRAD_Y_ROWS) starts from that gradient and leaves the last layer's parameter gradients untouched
                                 * (they follow from the merged parameters' gradients by the chain rule).  One layer launch less in the
                                 * forward pass, two less in the backward pass, and the feature tensor never exists. */
#define TN_MLP_LAYERWISE 2048   /* a TN_MLP_F16X2 stack in tn_mlp_fwd_ws / tn_mlp_fwd_stash / tn_mlp_bwd_layers: one launch per layer and direction
                                 * with the activations / gradients crossing HBM as workspace rows (the form of rounds 3 - 5) instead of the
                                 * cross-layer persistent launches of round 6 (csrc/mlp_fused_f2.hip: inference forward, training forward and
                                 * data-gradient chain with the sample columns in registers for the whole stack, weights streamed through
                                 * LDS).  Same results to fp32 rounding, same workspace layout; kept as the parity partner of the fused
                                 * launches in tests and for A / B timing.  The weight-gradient launches are per layer in both forms. */
#define TN_MLP_LEAN 512        /* the paired width-64 heads (tn_mlp_fwd_stash_pair / tn_kplanes_mlp_fwd_pair and their backward twins), round 5:
                                 * the training forward writes only the ReLU bit masks, the last pre-activation and the feature rows --
                                 * NOT the hidden activations H_l (1.3 KB per sample that crossed HBM twice) -- and the weight-gradient
                                 * half of the backward (TN_MLP_WGRAD_ONLY or the full call) rebuilds them from the feature row: a
                                 * wave recomputes its 32 samples' forward on the fp16 matrix cores (the f16x2 arithmetic of the
                                 * forward) in BOTH operand orientations -- out[feature][sample] to feed the next layer,
                                 * out[sample][feature] as the weight gradient's MFMA operand: no transposition through memory --
                                 * and multiplies with the G rows of the data-gradient chain as exact three-way bf16 splits
                                 * (csrc/mlp_wgrad_rc.hip).  Set on BOTH descriptors of BOTH calls; the workspace layout is unchanged
                                 * (its H rows stay unwritten).  Reference: the autograd of src/models.py:7-28,70-89. */

typedef struct tn_mlp_desc {
    int32_t n_layers;                         /* number of Linear layers (>= 1)               */
    int32_t in_dim;                           /* width of x (before encoding)                 */
    int32_t dims[TN_MLP_MAX_LAYERS + 1];      /* dims[0] = encoded input width, dims[l+1] = out of layer l */
    int32_t encoding;                         /* TN_ENC_*                                     */
    int32_t n_freqs;                          /* F for the encodings                          */
    int32_t out_activation;                   /* TN_ACT_*                                     */
    int32_t flags;                            /* TN_MLP_*                                     */
    const float *freqs;                       /* [n_freqs] encoding frequencies (models.py:34); NULL = 2^j*pi */
    const float *weights[TN_MLP_MAX_LAYERS];  /* [dims[l+1], dims[l]]                         */
    const float *biases[TN_MLP_MAX_LAYERS];   /* [dims[l+1]]                                  */
    /* TN_ENC_AUX_CAT: `aux` is a table [n_aux, aux_stride] whose rows hold dims[0]-in_dim values followed by
     * zeros (aux_stride and in_dim multiples of 4); row i of x uses table row aux_index[i] (NULL: row i). */
    const int32_t *aux_index;
    int32_t aux_stride;
    int32_t reserved;
    /* tn_mlp_fwd only: optional per-row gate [n] (the renderer passes the volume-rendering weights, core.py:246-251: the
     * colour head only matters where w > 0).  A 32-row tile whose gates are all 0 is not evaluated and yields 0. */
    const float *row_gate;
    /* Row views (optional, all NULL / 0 by default): x and grad_x of tn_mlp_bwd as [feature][32-sample] rows per 32-sample
     * tile -- the layout in which a wide stack's layer-by-layer kernels keep activations and gradients in their workspace
     * (tn_mlp_rows_view), so that the heads behind a width-256 feature stack (reference models.py:59-89, core.py:239-249)
     * exchange both with it without a row-major round trip:
     *   x_rows       value of feature f of sample 32 t + j at x_rows[t * x_rows_tile_stride + 32 f + j] (samples >= n: 0).
     *                tn_mlp_bwd (two-pass form, TN_MLP_STASHED) then takes the first layer's weight gradient over the x
     *                columns from these rows; needs in_dim % 32 == 0.  tn_mlp_fwd_stash (f16x2 heads, in_dim 128 / 256) reads
     *                its first-layer operands from them.  x itself is still required unless TN_MLP_X_FROM_ROWS says otherwise.
     *   grad_x_rows  tn_mlp_bwd writes (TN_MLP_ACCUM_GRAD_X: adds) d loss / d x there, same layout, INSTEAD of grad_x. */
    const float *x_rows;
    float *grad_x_rows;
    int64_t x_rows_tile_stride;               /* floats */
    int64_t grad_x_rows_tile_stride;
    /* ReLU bit rows for grad_x_rows (optional; tn_mlp_rows_view_hidden): when the rows a head reads are a HIDDEN activation of the
     * producer (TN_MLP_SKIP_LAST), d loss / d x leaves multiplied by relu'(x) -- bit r of the dword at
     * grad_x_mask_rows[t * grad_x_mask_tile_stride + 64 b + lane] says whether feature 32 b + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of
     * sample 32 t + (lane & 31) was positive (the layer kernels' bit rows, two 128-byte rows per 32-feature block). */
    const void *grad_x_mask_rows;
    int64_t grad_x_mask_tile_stride;          /* dwords */
} tn_mlp_desc;

/* Merged parameters for TN_MLP_SKIP_LAST (reference models.py:59-89, 239-247: Linear(F, F) -> Linear(F [+ other columns], rows), nothing
 * in between).  Per consumer ("head"): weight [rows][ld] whose columns [col0, col0 + F) multiply the stack's output, bias [rows].
 *   tn_linear_merge_fwd: out_weight[:, col0 .. col0 + F) = weight[:, col0 .. col0 + F) w_last, the other columns copied;
 *                        out_bias = bias + weight[:, col0 .. col0 + F) b_last.
 *   tn_linear_merge_bwd: grad_merged_* = gradients of those merged parameters; ADDS the gradients of the original parameters to
 *                        out_weight / out_bias (now: gradient buffers of weight / bias) and to grad_w_last [F][F] / grad_b_last [F]. */
#define TN_MERGE_MAX_HEADS 4
typedef struct tn_merge_head {
    const float *weight, *bias;
    int32_t rows, ld, col0, reserved;
    float *out_weight, *out_bias;
    const float *grad_merged_weight, *grad_merged_bias;       /* tn_linear_merge_bwd only */
} tn_merge_head;
int tn_linear_merge_fwd(int32_t n_heads, const tn_merge_head *heads, const float *w_last, const float *b_last, int32_t F, void *stream);
int tn_linear_merge_bwd(int32_t n_heads, const tn_merge_head *heads, const float *w_last, const float *b_last, int32_t F,
                        float *grad_w_last, float *grad_b_last, void *stream);

/* One plain Linear (reference src/models.py:183-191: KPlanesExplicitOpacityDecoder.net = torch.nn.Linear(96, 96), the only
 * Linear on the reference's path outside an MLP stack): y [n, out] = x [n, in] weight^T + bias (bias may be NULL), weight
 * [out, in] row-major as torch.nn.Linear holds it, 1 <= in, out <= 128, fp32 MFMA. */
int tn_linear_fwd(const float *x, const float *weight, const float *bias, int64_t n, int32_t in_features,
                  int32_t out_features, float *y, void *stream);
/* ... and its backward: grad_x [n, in] = grad_y weight (written; NULL: skipped), grad_weight [out, in] += grad_y^T x and
 * grad_bias [out] += column sums of grad_y (both ACCUMULATED, as autograd does into .grad; NULL: skipped -- each on its own:
 * x may be NULL when grad_weight is). */
int tn_linear_bwd(const float *x, const float *weight, const float *grad_y, int64_t n, int32_t in_features,
                  int32_t out_features, float *grad_x, float *grad_weight, float *grad_bias, void *stream);

/* Row views into the workspace of tn_mlp_fwd_stash(desc, ..., n) for a layer-by-layer configuration without output
 * activation (the Vanilla 256 x 10 and Cobafa 128 x 6 feature stacks): offsets in floats from the workspace base of
 *   y_rows       y as [feature][32-sample] rows -- valid until this stack's tn_mlp_bwd runs;
 *   grad_y_rows  where tn_mlp_bwd with TN_MLP_GRAD_Y_ROWS expects d loss / d y in that layout;
 * and the tile stride (floats per 32 samples) of both.  TN_E_CONFIG when the configuration has no such views. */
int tn_mlp_rows_view(const tn_mlp_desc *desc, int64_t n, int64_t *y_rows, int64_t *grad_y_rows, int64_t *tile_stride);
/* ... of a stack that ran / will run with TN_MLP_SKIP_LAST: the last hidden activation h (offsets in floats from the workspace
 * base, [feature][32-sample] rows, tile stride as above), the slot where tn_mlp_bwd expects d loss / d h -- ALREADY multiplied by
 * relu'(h): consumers apply tn_mlp_desc::grad_x_mask_rows --, and h's ReLU bit rows (offset in floats = dwords; the bit rows' tile
 * stride equals tile_stride). */
int tn_mlp_rows_view_hidden(const tn_mlp_desc *desc, int64_t n, int64_t *h_rows, int64_t *grad_h_rows, int64_t *mask_rows,
                            int64_t *tile_stride);

/* y [n, dims[n_layers]] = MLP(x [n,in_dim], aux [n,3] (dirs for TN_ENC_DIR_CAT, else NULL)).
 * pre_act (optional, [n, dims[n_layers]]) receives the last layer's output before out_activation. */
int tn_mlp_fwd(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y,
               float *pre_act, void *stream);
/* Inference forward with scratch: wide stacks on positional-encoding inputs (the Vanilla feature MLP, models.py:59-68) or on
 * <= 64 plain inputs (Cobafa's 36 gathered features into its 128-wide stack, models.py:239-247: staged as zero-padded rows) run
 * layer by layer through the weight-in-register kernels of the training forward, the activations ping-ponging between two
 * [feature][32-sample] row buffers in `workspace` (tn_mlp_fwd_workspace_bytes(desc, n) bytes: 2.3 KB per sample at width
 * 256; 0 = this configuration has no such form, use tn_mlp_fwd).  Same MFMA steps in the same order as the training
 * forward: identical y.  Used by infer() (run.py:15-50) and the occupancy refresh (core.py:133-145). */
int64_t tn_mlp_fwd_workspace_bytes(const tn_mlp_desc *desc, int64_t n);
int tn_mlp_fwd_ws(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y, void *workspace,
                  int64_t workspace_bytes, void *stream);
/* Training forward: y as tn_mlp_fwd, plus the hidden activations, their ReLU bit masks and the last layer's
 * pre-activation into `workspace` (tn_mlp_bwd_workspace_bytes(desc, n) bytes) in the layout tn_mlp_bwd's
 * backward (two-pass or layer-by-layer form) uses; pass the same workspace to tn_mlp_bwd with TN_MLP_STASHED set.
 * Returns TN_E_CONFIG for configurations whose backward takes no workspace (tn_mlp_bwd_workspace_bytes == 0). */
int tn_mlp_fwd_stash(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y,
                     void *workspace, int64_t workspace_bytes, void *stream);
/* Training forward of TWO heads on the same x in one launch (x is read from HBM once): `desc` (TN_ENC_NONE or
 * TN_ENC_AUX_CAT) and `partner` (TN_ENC_NONE, same in_dim), both of hidden width 64; workspaces as for tn_mlp_fwd_stash. */
int tn_mlp_fwd_stash_pair(const tn_mlp_desc *desc, const tn_mlp_desc *partner, const float *x, const float *aux, int64_t n,
                          float *y, float *partner_y, void *workspace, int64_t workspace_bytes, void *partner_workspace,
                          int64_t partner_workspace_bytes, void *stream);
/* out[r, :] = [PE_F(dirs[r]) (6F, models.py:36-39 order), dirs[r] (3), 0 ...] with row stride `stride` >= 6F+3:
 * the aux table of TN_ENC_AUX_CAT for the colour head (models.py:87). */
int tn_dir_encode(const float *dirs, int64_t n, const float *freqs, int n_freqs, float *out, int stride, void *stream);
/* Backward of tn_mlp_fwd: recomputes the hidden activations, accumulates (+=) weight/bias
 * gradients into grad_weights[l]/grad_biases[l] (same shapes; must be initialised by the caller)
 * and writes (flags & TN_MLP_ACCUM_GRAD_X: adds to) grad_x [n,in_dim] when non-NULL (TN_ENC_POSENC: no grad_x, coords carry no grad;
 * TN_ENC_DIR_CAT: gradient w.r.t. the feature part x only).
 * workspace (optional, tn_mlp_bwd_workspace_bytes(desc, n) bytes, uninitialised) selects the two-pass
 * form (data-gradient chain with LDS-resident weights, then a sample-reducing weight-gradient kernel);
 * without it, or for configurations the two-pass form does not cover (it returns 0 bytes), the
 * single-kernel form runs. */
int64_t tn_mlp_bwd_workspace_bytes(const tn_mlp_desc *desc, int64_t n);
int tn_mlp_bwd(const tn_mlp_desc *desc, const float *x, const float *aux, const float *grad_y,
               int64_t n, float *const *grad_weights, float *const *grad_biases, float *grad_x,
               void *workspace, int64_t workspace_bytes, void *stream);
/* Backward of TWO heads that read the same x in one data-gradient pass (the K-Planes colour head `desc` and its
 * 2-layer sigma head `partner`, models.py:70-89): grad_x = d/dx of both, written once.  Both descriptors carry
 * TN_MLP_STASHED (workspaces written by tn_mlp_fwd_stash); partner: 2 layers, TN_ENC_NONE, same in_dim (multiple of
 * 32) and hidden width 64.  Parameter gradients accumulate (+=) as in tn_mlp_bwd.
 * With row views on both descriptors (the heads behind a wide stack: the same x_rows / grad_x_rows [/ grad_x_mask_rows], TN_ENC_AUX_CAT on
 * `desc`, grad_x == NULL) and TN_MLP_F16X2, both chains stop at their G_0 rows, grad_x_rows = W_0[:, x]^T G_0 of both heads is one launch
 * on the fp16 matrix cores (two-term splits, 2^-22 relative; the fp32 / bf16x3 modes keep the fp32 MFMA), and both first layers' x-column
 * weight gradients share one launch. */
int tn_mlp_bwd_pair(const tn_mlp_desc *desc, const tn_mlp_desc *partner, const float *x, const float *aux,
                    const float *grad_y, const float *partner_grad_y, int64_t n, float *const *grad_weights,
                    float *const *grad_biases, float *const *partner_grad_weights, float *const *partner_grad_biases,
                    float *grad_x, void *workspace, int64_t workspace_bytes, void *partner_workspace,
                    int64_t partner_workspace_bytes, void *stream);

/* 1 when the pair (`desc` = the 5-layer head, `partner` = the 2-layer head on the same x) can run under TN_MLP_LEAN: the reference's
 * decoders as f16x2 heads (src/run.py:133-139 + TN_MLP_F16X2: colour 147 -> 64 x 4 -> 3 on [PE_8(d), d, x (96)] through a per-ray table,
 * sigma 96 -> 64 -> 1), else 0.  Callers choose the training forward's form with it. */
int tn_mlp_lean_supported(const tn_mlp_desc *desc, const tn_mlp_desc *partner);

/* ------------------------------------------------------------------------------------------
 * a15/a16  K-Planes feature field                              (reference models.py:93-163)
 * Planes are stored channel-last: plane[s][p] is fp32 [H_s, W_s, C] (torch memory_format
 * channels_last of the reference's [1,C,H,W] parameter, so the state_dict shape is unchanged).
 * feat[n, s*C + c] = prod_p bilinear(plane[s][p], x[n, pair_p])[c], pairs (0,1),(0,2),(1,2).
 * planes[s][1] / planes[s][2] may be NULL: that factor is 1 (KPlanesFeaturePlane.forward, models.py:105-113).
 * ------------------------------------------------------------------------------------------ */
#define TN_KPLANES_MAX_SCALES 4
typedef struct tn_kplanes_desc {
    int32_t n_scales;
    int32_t channels;                               /* C, multiple of 4, <= 32              */
    int32_t height[TN_KPLANES_MAX_SCALES];
    int32_t width[TN_KPLANES_MAX_SCALES];
    const float *planes[TN_KPLANES_MAX_SCALES][3];  /* [H,W,C] each                          */
} tn_kplanes_desc;

/* x has row stride x_stride floats (7 when reading packed_samples directly, 3 for [n,3]). */
int tn_kplanes_fwd(const tn_kplanes_desc *desc, const float *x, int64_t x_stride, int64_t n,
                   float *feat, void *stream);
/* grad_planes[s][p] ([H,W,C], += with fp32 atomics; caller initialises) from grad_feat [n, S*C]. */
int tn_kplanes_bwd(const tn_kplanes_desc *desc, const float *x, int64_t x_stride, int64_t n,
                   const float *grad_feat, float *const (*grad_planes)[3], void *stream);

/* ------------------------------------------------------------------------------------------
 * a19  K-Planes regularisers                         (reference models.py:115-121,165-181)
 * One pass per plane ([H,W,C] channel-last) instead of the 4 strided torch passes per plane of
 * the reference.  fwd: sums[0] += sum (p[y+1]-p[y])^2, sums[1] += sum (p[x+1]-p[x])^2,
 * sums[2] += sum |p|  (fp64 device accumulators, caller zeroes them).
 * bwd: grad += upstream[0] * ( cy * d/dp sum_dy + cx * d/dp sum_dx + cl1 * sign(p) ), where
 * upstream is a device scalar (dLoss/dRegulariser) and cy, cx, cl1 fold the means and weights. */
int tn_plane_reg_fwd(const float *plane, int H, int W, int C, double *sums, void *stream);
int tn_plane_reg_bwd(const float *plane, int H, int W, int C, float cy, float cx, float cl1,
                     const float *upstream, float *grad, void *stream);

/* Every plane of the field in one launch, forward and backward fused for a constant upstream gradient
 * (the harness: d(loss * grad_scale)/d(regulariser) is a host scalar): sums[3*i + {0,1,2}] += the three sums
 * of tn_plane_reg_fwd for item i (sums may be NULL), grad_i += upstream * (cy, cx, cl1 terms) (grad may be NULL).
 * `items` is a HOST array. */
#define TN_MULTI_MAX 32
typedef struct tn_plane_reg_item {
    const float *plane;       /* [H,W,C] channel-last */
    float *grad;              /* same layout, += ; NULL = value only */
    int32_t H, W, C;
    float cy, cx, cl1;
} tn_plane_reg_item;
int tn_plane_reg_multi(const tn_plane_reg_item *items, int32_t n_items, float upstream, double *sums, void *stream);

/* ------------------------------------------------------------------------------------------
 * a20  Cobafa factorised field                                  (reference models.py:209-266)
 * feat[n, off_i + c] = trilinear(basis_i, saw_i(x))[c] * trilinear(coef, x)[i],
 * saw_i(x) = 2*((f_i*x) mod 1) - 1 (models.py:213), levels concatenated (models.py:263-265).
 * f_i <= 0 selects saw_i(x) = x: with a 1x1x1 coefficient grid holding 1 this is the plain trilinear
 * lookup of CobafaGrid.forward (models.py:223-232).
 * Grids are channel-last [D,H,W,C] (torch channels_last_3d of the reference's [1,C,D,H,W]).
 * ------------------------------------------------------------------------------------------ */
#define TN_COBAFA_MAX_LEVELS 8
typedef struct tn_cobafa_desc {
    int32_t n_levels;
    int32_t coef_res[3];                           /* D,H,W of the coefficient grid (C = n_levels) */
    int32_t res[TN_COBAFA_MAX_LEVELS][3];          /* D,H,W of basis grid i                        */
    int32_t channels[TN_COBAFA_MAX_LEVELS];        /* <= 8                                         */
    float freqs[TN_COBAFA_MAX_LEVELS];
    const float *coef;                             /* [D,H,W,n_levels]                             */
    const float *basis[TN_COBAFA_MAX_LEVELS];      /* [D,H,W,channels[i]]                          */
} tn_cobafa_desc;
int tn_cobafa_fwd(const tn_cobafa_desc *desc, const float *x, int64_t n, float *feat, void *stream);
/* grad_coef / grad_basis[i]: same layouts as the grids, += with fp32 atomics (caller initialises) */
int tn_cobafa_bwd(const tn_cobafa_desc *desc, const float *x, int64_t n, const float *grad_feat, float *grad_coef,
                  float *const *grad_basis, void *stream);

/* north star: "the K-Planes bilinear grid sample ... fused into the same launch" as the persistent MLP.  tn_kplanes_fwd +
 * tn_mlp_fwd_stash_pair in ONE launch (reference call chain core.py:239-249 -> models.py:153-163 -> models.py:70-89): every
 * wave gathers the 3 x 3 planes x 4 taps of its 32 samples straight into the first-layer MFMA operand registers of both
 * heads; `feat` [n, 96] is written once for the backward (weight gradient of the first layers, plane scatter) and never read
 * here.  Requirements: 3 scales x 32 channels, all planes present, both heads as in tn_mlp_fwd_stash_pair with in_dim 96.
 * Results are bit-identical to the two-launch sequence.
 * Inference form (round 4): workspace == partner_workspace == NULL -- nothing is stashed and `feat` is not written (it may be
 * NULL; coords must then be 16-byte aligned): gather + sigma + colour of EVERY sample in one launch, for renders in which most
 * samples carry weight (the caller composites with tn_render_rays_fwd; samples with w == 0 contribute exactly 0 either way,
 * core.py:243-249).  The gated sequence tn_kplanes_mlp_fwd -> tn_weights_fwd -> tn_mlp_fwd(row_gate) wins when most tiles are dead. */
int tn_kplanes_mlp_fwd_pair(const tn_kplanes_desc *kdesc, const float *coords, int64_t coord_stride, const tn_mlp_desc *desc,
                            const tn_mlp_desc *partner, const float *aux, int64_t n, float *feat, float *y, float *partner_y,
                            void *workspace, int64_t workspace_bytes, void *partner_workspace, int64_t partner_workspace_bytes,
                            void *stream);

/* Inference form: the gather inside the launch of ONE head without encoding (the sigma head; tn_kplanes_fwd + tn_mlp_fwd):
 * y [n, dims[last]] and the feature rows `feat` [n, 96], which the colour head then reads where the weight is not 0. */
int tn_kplanes_mlp_fwd(const tn_kplanes_desc *kdesc, const float *coords, int64_t coord_stride, const tn_mlp_desc *desc, int64_t n,
                       float *feat, float *y, void *stream);

/* Backward twin: tn_mlp_bwd_pair with the plane scatter (tn_kplanes_bwd) inside the data-gradient chain launch.  d(loss)/d(feat)
 * never leaves the registers of the wave that computed it: each wave scatters its 32 samples' three 32-channel blocks into
 * grad_planes[s][p] (+=, fp32 atomics; NULL entries are skipped) while the other waves of its SIMD run their MFMA chains.
 * `grad_feat` may be NULL (it is only written on request).  `feat` is the row-major [n, 96] feature tensor the forward wrote
 * (the weight-gradient kernels read it).  desc->flags: TN_MLP_STASHED required; TN_MLP_CHAIN_ONLY / TN_MLP_WGRAD_ONLY split the
 * call as in tn_mlp_bwd_pair (CHAIN_ONLY leaves the plane gradients final). */
int tn_kplanes_mlp_bwd_pair(const tn_kplanes_desc *kdesc, const float *coords, int64_t coord_stride,
                            float *const (*grad_planes)[3], const tn_mlp_desc *desc, const tn_mlp_desc *partner,
                            const float *feat, const float *aux, const float *grad_y, const float *partner_grad_y, int64_t n,
                            float *const *grad_weights, float *const *grad_biases, float *const *partner_grad_weights,
                            float *const *partner_grad_biases, float *grad_feat, void *workspace, int64_t workspace_bytes,
                            void *partner_workspace, int64_t partner_workspace_bytes, void *stream);

/* Product stage of the explicit K-Planes decoders (models.py:183-205, exercised by the reference's tests/test_models.py:35-69;
 * train() itself uses the Vanilla decoders, run.py:135-139):
 *   out[n,k] = act(sum_c f[n,c] * basis[n,k,c]),   f [n,C], basis [n,K,C] row-major, 1 <= K <= 4.
 * Opacity decoder: basis = Linear(f) (K = 1), act = TN_ACT_EXP_M1 (exp(v - 1) with the truncated exponential's backward clamp,
 * models.py:42-55); colour decoder: basis = MLP([PE(d), d, f]).view(n, 3, C), act = TN_ACT_SIGMOID.
 * Backward: grad_basis [n,K,C] (may be NULL) is written, grad_f [n,C] is written or (accumulate_f) added to. */
int tn_basis_dot_fwd(const float *f, const float *basis, int64_t n, int32_t channels, int32_t n_out, int32_t activation,
                     float *out, void *stream);
int tn_basis_dot_bwd(const float *f, const float *basis, const float *grad_out, int64_t n, int32_t channels, int32_t n_out,
                     int32_t activation, float *grad_f, float *grad_basis, int32_t accumulate_f, void *stream);

/* From (packed [N,7], info [R,2]): ray_ids[i] = ray of sample i, steps[i] = packed[i,6] (contiguous, what tn_weights_*
 * take), dirs[r] = packed[start_r, 3:6] (the ray direction every sample of ray r carries, core.py:182-186; 0 for empty
 * rays).  One launch; feeds tn_dir_encode / TN_ENC_AUX_CAT when the sampler's own by-products are not at hand. */
int tn_ray_aux(const float *packed, const int32_t *info, int64_t n_rays, int32_t *ray_ids, float *steps, float *dirs,
               void *stream);

/* Loss of the harness (run.py:252,259): grad[i] = scale * (scale_dev ? scale_dev[0] : 1) * (rendered[i] - target[i]) and
 * sumsq[0] += sum (rendered - target)^2 (fp64, caller zeroes) over n = 3 * rays elements, one pass. */
int tn_mse_grad(const float *rendered, const float *target, int64_t n, float scale, const float *scale_dev, float *grad,
                double *sumsq, void *stream);
/* a17 + a18 of one batch as ONE launch each way (a wave per ray does both): tn_weights_fwd(_gate) + tn_composite_fwd -> weights
 * and rendered (bit-identical to the two calls; gate may be NULL), and tn_composite_bwd + tn_weights_bwd -> grad_rgbs and
 * grad_sigmas without the grad_weights round trip (reference cuda.cu:3-58 + core.py:256-265 and their autograd). */
int tn_render_rays_fwd(const float *sigmas, const float *steps, const float *rgbs, const int32_t *info, const float *bg,
                       float threshold, float *weights, float *rendered, float *gate, int64_t n_samples, int64_t n_rays,
                       void *stream);
int tn_render_rays_bwd(const float *sigmas, const float *steps, const float *rgbs, const int32_t *info, const float *bg,
                       const float *weights, const float *grad_rendered, float *grad_rgbs, float *grad_sigmas,
                       int64_t n_samples, int64_t n_rays, void *stream);
/* tn_mse_grad with the "Empty iteration" gate applied at the source: grad = 0 when !(gate[0] > 0) (core.py:251-254: the
 * image loss then reaches no parameter); sumsq as in tn_mse_grad. */
int tn_mse_grad_gated(const float *rendered, const float *target, int64_t n, float scale, const float *scale_dev,
                      const float *gate, float *grad, double *sumsq, void *stream);
/* The harness' batch draw (run.py:225-229, the DataLoader's index_select): rows idx[i] of the [N, 3] ray tables -> out_* [n, 3]
 * in one launch; rgbs / out_rgb may be NULL. */
int tn_gather_rays(const float *rays_o, const float *rays_d, const float *rgbs, const int32_t *idx, int64_t n,
                   float *out_o, float *out_d, float *out_rgb, void *stream);

/* ------------------------------------------------------------------------------------------
 * optimizer step of the harness                        (reference run.py:186,258-260: torch.optim.Adam)
 * One pass per parameter tensor: g' = g + wd*p (coupled L2, as torch), m = b1 m + (1-b1) g',
 * v = b2 v + (1-b2) g'^2, p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps); optionally zeroes g for the
 * next step.  28 B/element instead of torch's multi-kernel foreach path. */
int tn_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad, void *stream);

/* the same update for many tensors in one launch (`items` is a HOST array; all tensors share the step count) */
typedef struct tn_adam_item {
    float *param, *grad, *exp_avg, *exp_avg_sq;
    int64_t n;
} tn_adam_item;
int tn_adam_multi(const tn_adam_item *items, int32_t n_items, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, int32_t zero_grad, void *stream);

/* tn_adam_multi for parameters that receive NO gradient in an "Empty iteration" (core.py:251-254: every sample masked ->
 * rgbs / weights become fresh leaves, param.grad stays None and torch.optim.Adam skips the parameter, run.py:258-260,
 * including its per-parameter step count).  `gate` is a device scalar (e.g. max_i w_i of the step): gate > 0 -> `step_dev[0]`
 * (device int32 counter of the updates these tensors have received) is incremented and the update of tn_adam_multi runs with
 * that count; gate <= 0 -> parameters, moments and counter are left untouched (gradients are still zeroed if asked).
 * zero_grad: bit 0 = zero the gradients; bit 1 (round 5) = `step_dev` has a SECOND int32, step_dev[1], which is set to 1 when an updated
 * parameter is not finite (never cleared here).  The kernels' ReLU is v_max_f32: a NaN pre-activation becomes 0 where torch.relu
 * (models.py:7-28) hands it on, so a diverged run would train on silently; the harness turns this flag -- and, for K-Planes planes, the
 * regulariser sums of tn_adam_reg_multi, which any non-finite plane value poisons -- into the NaN loss the reference reports. */
int tn_adam_multi_gated(const tn_adam_item *items, int32_t n_items, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int32_t *step_dev, const float *gate, int32_t zero_grad, void *stream);

/* tn_adam_multi with the K-Planes regulariser folded in (run.py:254-260 in one pass): items with H > 0 are [H,W,C]
 * planes whose TV / L1 gradient (coefficients as in tn_plane_reg_multi, times `upstream`) is added to grad before the
 * update; it is built from the values in `param` while the update goes to `param_out` (a second buffer: the caller swaps
 * them), and sums[3*sum_slot + {0,1,2}] receive the regulariser's three sums (may be NULL).  Items with H == 0 are plain
 * Adam (param_out may equal param). */
typedef struct tn_adam_reg_item {
    const float *param;
    float *param_out, *grad, *exp_avg, *exp_avg_sq;
    int64_t n;
    int32_t H, W, C, sum_slot;
    float cy, cx, cl1;
    /* Sharded optimizer pass (round 5, N > 1 with the recipe's batch split over the ranks): rows [row0, row1) of an [H,W,C] plane are this
     * rank's -- only they are updated (their TV gradient still reads the neighbouring rows of `param`, which every rank holds), only their
     * terms go into `sums`; outside them nothing but the gradient zeroing happens (4 B per element instead of 32).  row1 == 0: every row. */
    int32_t row0, row1, reserved;
} tn_adam_reg_item;
int tn_adam_reg_multi(const tn_adam_reg_item *items, int32_t n_items, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int32_t step, int32_t zero_grad, float upstream, double *sums, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TINYNERF_HIP_H */
