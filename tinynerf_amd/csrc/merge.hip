// Two Linear layers with nothing in between are one Linear layer (TN_MLP_SKIP_LAST, include/tinynerf_hip.h): the reference's feature
// stacks end in Linear(F, F) and both decoders begin with Linear(F [+ direction columns], 64) (src/models.py:59-89, 239-247), so
//     W_head[:, x] (W_last h + b_last) + b_head  =  (W_head[:, x] W_last) h + (W_head[:, x] b_last + b_head).
// These two launches build the merged parameters of up to four consumers and take their gradients back to the original ones -- what
// five torch matmuls and their autograd did in ~35 launches of 5 - 40 us each (0.26 ms per Vanilla step); 8 MFLOP of work.
#include "tn_common.h"

namespace {

struct MergeArgs {
    int n_heads, F;
    const float *w_last, *b_last;            // [F][F], [F]
    const float *w[TN_MERGE_MAX_HEADS], *b[TN_MERGE_MAX_HEADS];      // [rows][ldw], [rows]
    int rows[TN_MERGE_MAX_HEADS], ldw[TN_MERGE_MAX_HEADS], col0[TN_MERGE_MAX_HEADS];
    float *w_out[TN_MERGE_MAX_HEADS], *b_out[TN_MERGE_MAX_HEADS];    // forward: merged; backward: gradients of the ORIGINAL head parameters (+=)
    const float *gw_m[TN_MERGE_MAX_HEADS], *gb_m[TN_MERGE_MAX_HEADS]; // backward: gradients of the merged parameters
    float *gw_last, *gb_last;                // backward: += gradients of the last layer
};

// one workgroup per head row: merged x columns (thread = output column), the other columns copied, the bias by a block reduction
__global__ __launch_bounds__(256) void merge_fwd_kernel(MergeArgs a)
{
    __shared__ float wrow[256];
    __shared__ float red[256];
    int hd = 0, i = blockIdx.x;
    while (hd < a.n_heads && i >= a.rows[hd]) { i -= a.rows[hd]; ++hd; }
    if (hd >= a.n_heads) return;
    const int F = a.F, ldw = a.ldw[hd], c0 = a.col0[hd], t = threadIdx.x;
    const float *w = a.w[hd] + (int64_t)i * ldw;
    float *o = a.w_out[hd] + (int64_t)i * ldw;
    if (t < F) wrow[t] = w[c0 + t];
    __syncthreads();
    if (t < F) {
        float acc = 0.0f;
        for (int j = 0; j < F; ++j) acc = fmaf(wrow[j], a.w_last[(int64_t)j * F + t], acc);
        o[c0 + t] = acc;
    }
    for (int c = t; c < ldw; c += 256)
        if (c < c0 || c >= c0 + F) o[c] = w[c];
    red[t] = t < F ? wrow[t] * a.b_last[t] : 0.0f;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    if (t == 0) a.b_out[hd][i] = a.b[hd][i] + red[0];
}

// blocks [0, total head rows): gradients of a head row; blocks behind them: row j of the last layer's weight gradient
__global__ __launch_bounds__(256) void merge_bwd_kernel(MergeArgs a, int total_rows)
{
    __shared__ float grow[256];
    const int F = a.F, t = threadIdx.x;
    if ((int)blockIdx.x < total_rows) {
        int hd = 0, i = blockIdx.x;
        while (hd < a.n_heads && i >= a.rows[hd]) { i -= a.rows[hd]; ++hd; }
        const int ldw = a.ldw[hd], c0 = a.col0[hd];
        const float *g = a.gw_m[hd] + (int64_t)i * ldw;
        float *o = a.w_out[hd] + (int64_t)i * ldw;
        if (t < F) grow[t] = g[c0 + t];
        __syncthreads();
        const float gb = a.gb_m[hd][i];
        if (t < F) {          // d / d W_head[i][c0 + t] = sum_k gW_m[i][c0 + k] W_last[t][k] + gb_m[i] b_last[t]
            const float *wl = a.w_last + (int64_t)t * F;
            float acc = gb * a.b_last[t];
            for (int k = 0; k < F; ++k) acc = fmaf(grow[k], wl[k], acc);
            o[c0 + t] += acc;
        }
        for (int c = t; c < ldw; c += 256)
            if (c < c0 || c >= c0 + F) o[c] += g[c];
        if (t == 0) a.b_out[hd][i] += gb;
        return;
    }
    // d / d W_last[j][k] = sum over all head rows i of W_head[i][c0 + j] gW_m[i][c0 + k];  d / d b_last[j] = sum_i W_head[i][c0 + j] gb_m[i]
    const int j = blockIdx.x - total_rows;
    float acc = 0.0f, accb = 0.0f;
    for (int hd = 0; hd < a.n_heads; ++hd) {
        const int ldw = a.ldw[hd], c0 = a.col0[hd];
        for (int i = 0; i < a.rows[hd]; ++i) {
            const float wij = a.w[hd][(int64_t)i * ldw + c0 + j];
            if (t < F) acc = fmaf(wij, a.gw_m[hd][(int64_t)i * ldw + c0 + t], acc);
            accb = fmaf(wij, a.gb_m[hd][i], accb);
        }
    }
    if (t < F) a.gw_last[(int64_t)j * F + t] += acc;
    if (t == 0) a.gb_last[j] += accb;
}

int fill(MergeArgs &a, int n_heads, int F, const float *w_last, const float *b_last, const tn_merge_head *heads)
{
    TN_REQUIRE(heads && w_last && b_last, TN_E_NULL, "tn_linear_merge: null pointer");
    TN_REQUIRE(n_heads >= 1 && n_heads <= TN_MERGE_MAX_HEADS && F >= 1 && F <= 256, TN_E_CONFIG, "tn_linear_merge: 1 .. 4 heads, F <= 256");
    a.n_heads = n_heads; a.F = F; a.w_last = w_last; a.b_last = b_last;
    for (int h = 0; h < n_heads; ++h) {
        const tn_merge_head &m = heads[h];
        TN_REQUIRE(m.weight && m.bias && m.out_weight && m.out_bias, TN_E_NULL, "tn_linear_merge: null head pointer");
        TN_REQUIRE(m.rows >= 1 && m.col0 >= 0 && m.col0 + F <= m.ld, TN_E_SIZE, "tn_linear_merge: the x columns [col0, col0 + F) must lie inside a weight row");
        a.w[h] = m.weight; a.b[h] = m.bias; a.rows[h] = m.rows; a.ldw[h] = m.ld; a.col0[h] = m.col0;
        a.w_out[h] = m.out_weight; a.b_out[h] = m.out_bias; a.gw_m[h] = m.grad_merged_weight; a.gb_m[h] = m.grad_merged_bias;
    }
    return TN_OK;
}

}  // namespace

extern "C" int tn_linear_merge_fwd(int32_t n_heads, const tn_merge_head *heads, const float *w_last, const float *b_last, int32_t F, void *stream)
{
    MergeArgs a;
    if (int rc = fill(a, n_heads, F, w_last, b_last, heads)) return rc;
    int total = 0;
    for (int h = 0; h < n_heads; ++h) total += a.rows[h];
    merge_fwd_kernel<<<dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream>>>(a);
    return tn::check_launch("merge_fwd_kernel");
}

extern "C" int tn_linear_merge_bwd(int32_t n_heads, const tn_merge_head *heads, const float *w_last, const float *b_last, int32_t F,
                                   float *grad_w_last, float *grad_b_last, void *stream)
{
    MergeArgs a;
    if (int rc = fill(a, n_heads, F, w_last, b_last, heads)) return rc;
    TN_REQUIRE(grad_w_last && grad_b_last, TN_E_NULL, "tn_linear_merge_bwd: null gradient pointer");
    for (int h = 0; h < n_heads; ++h) TN_REQUIRE(a.gw_m[h] && a.gb_m[h], TN_E_NULL, "tn_linear_merge_bwd: null merged gradient");
    a.gw_last = grad_w_last; a.gb_last = grad_b_last;
    int total = 0;
    for (int h = 0; h < n_heads; ++h) total += a.rows[h];
    merge_bwd_kernel<<<dim3((unsigned)(total + F)), dim3(256), 0, (hipStream_t)stream>>>(a, total);
    return tn::check_launch("merge_bwd_kernel");
}
