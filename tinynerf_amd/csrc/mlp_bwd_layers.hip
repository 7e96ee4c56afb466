// Backward of WIDE / DEEP fused MLPs (Vanilla feature stack 60->256x9->256, Cobafa 36->128x6), layer by
// layer.  The register-resident form (mlp_bwd2.hip) needs every hidden activation of a tile on
// chip; a 9 x 256 stack does not fit.  This path keeps the same transposed MFMA building blocks but runs
//
//   1 launch   forward that streams every hidden activation H_l to the workspace ([feature][32 samples]
//              rows, as in mlp_bwd2.hip) and leaves the output gradient g = dL/dy * act'(pre) there too;
//   per layer  wgrad (dW_l += G_l H_l^T, sample-reducing MFMA tiles) and dgrad
//              (G_{l-1} = relu'(H_l) * W_l^T G_l, or grad_x for the first layer),
//
// i.e. 2L+1 launches with all activations crossing HBM exactly once in each direction (the reference's
// autograd does the same through cuBLAS with NHW-major activations and separate bias / ReLU kernels).
// Weights are read from L2 (256x256 fp32 = 256 KiB per layer does not fit LDS next to anything else).
#include "mlp_layers.h"
#include <algorithm>

extern "C" int tn_mlp_wgrad_rows(const float *g_rows, int64_t g_stride, int ng, const float *a_rows, int64_t a_stride, int na, float *gW,
                                 int ldw, int col0, int kmax, float *gB, int64_t n, void *stream);

namespace {

using tn::f32x16;
using tn::f32x4;
using namespace tn::mlp;
using namespace tn::layers;

__device__ __forceinline__ float act_grad(float pre, int act) {
    if (act == TN_ACT_EXP_M1) return expf(fminf(fmaxf(pre - 1.0f, -15.0f), 15.0f));     // models.py:50-53
    if (act == TN_ACT_SIGMOID) { const float s = 1.0f / (1.0f + expf(-pre)); return s * (1.0f - s); }
    return 1.0f;
}

__device__ __forceinline__ void store_rows(float *__restrict__ rows, const f32x16 &t, int ob, int j, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rows[(32 * ob + frow(r, h)) * 32 + j] = t[r];
}

// behind the last tile of a training workspace: the largest |value| of every layer's input rows ([0, 16): activations, reported
// by the f16x2 forward of layer l; [16, 32): gradients, by its data gradient) -- the scales of the f16x2 weight gradient
constexpr int64_t WS_TAIL_BYTES = 256;
__host__ inline float *ws_tail(float *stash, int64_t total_rows) { return stash + total_rows * 32; }
// behind the tail: the packed weight stream of the cross-layer forward (mlp_fused_f2.hip), 256-byte aligned
__host__ inline int64_t fused_pack_offset(int64_t total_rows) { return (total_rows * 128 + WS_TAIL_BYTES + 255) & ~(int64_t)255; }
__host__ inline void *fused_pack_area(float *stash, int64_t total_rows) { return reinterpret_cast<unsigned char *>(stash) + fused_pack_offset(total_rows); }

__host__ __device__ inline bool plain_x_rows(int H, int n_layers, int enc, int K0_pad, int out_dim) {
    return enc == TN_ENC_NONE && K0_pad <= 64 && H >= 128 && n_layers >= 3 && out_dim > 4 && out_dim <= H;     // (= layer_kernel_path)
}

struct Layout {            // rows (of 32 floats) per 32-sample tile
    int rowsH, rowsE, rowsG, rowsM, total;
    int xs;                // first-layer slots that are plain x columns
};
__host__ __device__ inline Layout make_layout(int H, int n_layers, int enc, int in_dim, int K0_pad, int out_dim) {
    Layout L;
    L.xs = enc == TN_ENC_POSENC ? 0 : in_dim;
    L.rowsH = (n_layers - 1) * H;
    L.rowsE = enc == TN_ENC_NONE ? 0 : K0_pad - L.xs;
    // plain inputs of a stack that runs layer by layer (Cobafa: 36 features into 128 x 6, models.py:239-247): the training forward
    // stages x^T as 64 zero-padded rows (x_rows_kernel) so that the first layer and its weight gradient take the row-operand
    // kernels like every other layer; `xs` keeps its meaning for the generic kernels, which read x row-major
    if (plain_x_rows(H, n_layers, enc, K0_pad, out_dim)) L.rowsE = 64;
    const int outp = (out_dim + 31) & ~31;
    L.rowsG = H > outp ? H : outp;
    // wide stacks (layer kernels): ReLU bit masks of every hidden activation, one dword per lane and 32-feature block = 2 rows
    // per (activation, block), behind the two gradient buffers -- the data-gradient kernel reads ONE dword per lane where the
    // float mask rows cost it 16 row loads per tile (each vector-memory instruction stalls the matrix pipe for ~40 cycles)
    L.rowsM = H >= 128 ? 2 * (H / 32) * (n_layers - 1) : 0;
    L.total = L.rowsH + L.rowsE + 2 * L.rowsG + L.rowsM;
    return L;
}

__device__ __forceinline__ int col0(const MlpArgs &a, int q) { return layer0_col(a, q); }

// ------------------------------------------------------------------------------------------------
// forward with activation stash + output gradient
// ------------------------------------------------------------------------------------------------
// FWD_ONLY (training forward, tn_mlp_fwd_stash on a wide / deep stack): y is written, and buffer A receives the last
// layer's PRE-ACTIVATION instead of the output gradient; out_grad_kernel turns it into the gradient when the backward runs.
template <int H, int WPB, bool FWD_ONLY = false, bool FIRST_ONLY = false>
__global__ __launch_bounds__(WPB * 64) void fwd_stash_kernel(MlpArgs a, const float *__restrict__ x, const float *__restrict__ aux,
                                                             const float *__restrict__ gy, int64_t n, float *__restrict__ stash,
                                                             float *__restrict__ y = nullptr)
{
    constexpr int T = H / 32;
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int L = a.n_layers, G0 = a.K0_pad >> 3, out = a.out_dim;
    const Layout lay = make_layout(H, L, a.enc, a.in_dim, a.K0_pad, out);
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        const float *xrow = x + (valid ? row : 0) * a.in_dim;
        float aux3[3] = {0.f, 0.f, 0.f};
        if (valid) {
            if (a.enc == TN_ENC_POSENC) { aux3[0] = xrow[0]; aux3[1] = xrow[1]; aux3[2] = xrow[2]; }
            else if (a.enc == TN_ENC_DIR_CAT) { aux3[0] = aux[3 * row]; aux3[1] = aux[3 * row + 1]; aux3[2] = aux[3 * row + 2]; }
        }
        float *st = stash + tile * (int64_t)lay.total * 32;
        float *stE = st + lay.rowsH * 32;
        float *stG = stE + lay.rowsE * 32;               // buffer A: receives the output gradient
        f32x16 act[T];
#pragma unroll
        for (int ob = 0; ob < T; ++ob) act[ob] = tn::bias_tile(a.B[0], ob, h);
#pragma clang loop unroll(disable)
        for (int g = 0; g < G0; ++g) {
            const f32x4 b = fetch_input(a, xrow, aux3, valid, g, h);
            f32x4 w[T];
#pragma unroll
            for (int ob = 0; ob < T; ++ob)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = 8 * g + 4 * h + u;
                    w[ob][u] = q < a.K0 ? a.W[0][(int64_t)(32 * ob + j) * a.K0 + col0(a, q)] : 0.0f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ob = 0; ob < T; ++ob) act[ob] = tn::mfma32(w[ob][u], b[u], act[ob]);
            if (a.enc != TN_ENC_NONE && 8 * g + 4 * h + 3 >= lay.xs) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = 8 * g + 4 * h + u;
                    if (q >= lay.xs) stE[(q - lay.xs) * 32 + j] = b[u];
                }
            }
        }
#pragma unroll
        for (int ob = 0; ob < T; ++ob) {
            tn::pin16(act[ob]);
            act[ob] = tn::relu16(act[ob]);
            store_rows(st, act[ob], ob, j, h);
        }
        if constexpr (FIRST_ONLY) {                // the remaining layers run as layer-kernel launches
            // ReLU bits of this activation where the layer kernels keep them (two rows per 32-feature block), so that the data
            // gradient of layer 1 can take its mask from 256 B instead of 32 activation rows
            unsigned *bits = reinterpret_cast<unsigned *>(st + (lay.rowsH + lay.rowsE + 2 * lay.rowsG) * 32);
#pragma unroll
            for (int ob = 0; ob < T; ++ob) bits[2 * ob * 32 + lane] = relu_bits(act[ob]);
            continue;
        }
        for (int l = 1; l + 1 < L; ++l) {
            tn::hidden_layer<H>(a.W[l], a.B[l], H, act, j, h);
#pragma unroll
            for (int ob = 0; ob < T; ++ob) store_rows(st + l * H * 32, act[ob], ob, j, h);
        }
        // output layer -> g = gy * act'(pre) as rows [out feature][32 samples]
        const float *Wf = a.W[L - 1];
        const float *Bf = a.B[L - 1];
        if (out <= 4) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float g = 0.0f;
                if (o < out) {
                    const float pre = tn::small_out<H>(Wf + o * H, Bf[o], act, h);
                    if constexpr (FWD_ONLY) {
                        g = valid ? pre : 0.0f;
                        if (valid && h == 0) y[row * out + o] = tn::apply_act(pre, a.out_act);
                    } else g = valid ? gy[row * out + o] * act_grad(pre, a.out_act) : 0.0f;
                }
                if (h == 0) stG[o * 32 + j] = g;
            }
            // rows 4..31 of the first tile are read as zeros by the consumers (they clamp to `out`)
        } else {
            const int n_ob = (out + 31) >> 5;
            for (int ob = 0; ob < n_ob; ++ob) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int f = 32 * ob + frow(r, h); acc[r] = f < out ? Bf[f] : 0.f; }
                const int arow = 32 * ob + j;
#pragma unroll
                for (int kb = 0; kb < T; ++kb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 w = {0.f, 0.f, 0.f, 0.f};
                        if (arow < out) w = *reinterpret_cast<const f32x4 *>(Wf + (int64_t)arow * H + 32 * kb + 8 * q + 4 * h);
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc = tn::mfma32(w[u], act[kb][4 * q + u], acc);
                    }
                tn::pin16(acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * ob + frow(r, h);
                    if constexpr (FWD_ONLY) {
                        stG[f * 32 + j] = (valid && f < out) ? acc[r] : 0.0f;
                        if (valid && f < out) y[row * out + f] = tn::apply_act(acc[r], a.out_act);
                    } else stG[f * 32 + j] = (valid && f < out) ? gy[row * out + f] * act_grad(acc[r], a.out_act) : 0.0f;
                }
            }
        }
    }
}

// buffer A: pre-activation rows (written by the FWD_ONLY forward) -> output gradient  g = gy * act'(pre)
__global__ __launch_bounds__(256) void out_grad_kernel(const float *__restrict__ gy, int64_t n, int out, int out_act, int rows_total,
                                                       int64_t off_g, float *__restrict__ stash)
{
    // [32 samples][out] block of gy -> [out][32 samples] rows through LDS: both sides of the transposition are coalesced
    // (16-byte loads along a sample's row when `out` allows it, no integer division in the index arithmetic)
    constexpr int TS = 260;                                 // row stride (floats): 16-byte aligned, 4 (mod 64) banks apart
    __shared__ __attribute__((aligned(16))) float t[32 * TS];
    const int64_t n_tiles = (n + 31) >> 5;
    const int j = threadIdx.x & 31, sub = threadIdx.x >> 5;                 // 8 row slots per block
    const bool vec = (out & 3) == 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        float *g = stash + (tile * (int64_t)rows_total + off_g) * 32;
        const int64_t r0 = tile * 32;
        for (int c0 = 0; c0 < out; c0 += 256) {                   // 256 output columns per pass through the LDS tile
            const int cw = min(out - c0, 256);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = sub + 8 * rr;
                const bool rok = r0 + r < n;
                const float *src = gy + (r0 + (rok ? r : 0)) * out + c0;
                for (int c = 4 * j; c < cw; c += 128) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (vec) { if (rok) v = *reinterpret_cast<const f32x4 *>(src + c); }
                    else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (rok && c + u < cw) v[u] = src[c + u];
                    }
                    *reinterpret_cast<f32x4 *>(t + r * TS + c) = v;
                }
            }
            __syncthreads();
            for (int f = sub; f < cw; f += 8) {
                float v = t[j * TS + f];
                if (out_act != TN_ACT_NONE) v *= act_grad(g[(c0 + f) * 32 + j], out_act);       // buffer A holds the pre-activation
                g[(c0 + f) * 32 + j] = v;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// data gradient of one layer:  Gout = relu'(Hmask) * (W^T Gin)      (FIRST: grad_x, no mask, row-major out)
// ------------------------------------------------------------------------------------------------

template <int H, bool FIRST, int WPB>
__global__ __launch_bounds__(WPB * 64) void dgrad_layer_kernel(DgradArgs a, int64_t n, float *__restrict__ stash, float *__restrict__ gx)
{
    constexpr int T = H / 32;
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        float *st = stash + tile * (int64_t)a.rows_total * 32;
        // an output layer wider than H (N > 32 T, e.g. the 3 x 96 basis of KPlanesExplicitColorDecoder on a 128-wide stack) is
        // consumed in chunks of H rows; the chunks' contributions to the input gradient add up
        for (int n0 = 0; n0 < a.N; n0 += H) {
        const float *gin = st + (a.off_gin + n0) * 32;
        const int Nc = min(a.N - n0, H);
        const int TN = (Nc + 31) >> 5, ng = (Nc + 7) >> 3;
        f32x16 G[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * t + frow(r, h);
                G[t][r] = (t < TN && nn < Nc) ? gin[nn * 32 + j] : 0.0f;
            }
        const int n_kt = FIRST ? (a.in_dim + 31) >> 5 : T;
#pragma clang loop unroll(disable)
        for (int kt = 0; kt < n_kt; ++kt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const int k = 32 * kt + j;
            int kc = k;
            bool kok = true;
            if (FIRST) {
                kok = k < a.in_dim;
                kc = a.enc == TN_ENC_DIR_CAT ? 6 * a.n_freqs + 3 + k : k;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (4 * t + q < ng) {
                        float w[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int nn = 32 * t + 8 * q + 4 * h + u;
                            w[u] = (nn < Nc && kok) ? a.W[(int64_t)(n0 + nn) * a.K + kc] : 0.0f;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc = tn::mfma32(w[u], G[t][4 * q + u], acc);
                    }
                }
            }
            tn::pin16(acc);
            if (FIRST) {
                const int64_t row = tile * 32 + j;
                if (row < n) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * kt + frow(r, h);
                        if (f < a.in_dim) gx[row * a.in_dim + f] = (a.accum_gx || n0 > 0) ? gx[row * a.in_dim + f] + acc[r] : acc[r];
                    }
                }
            } else {
                const float *hm = st + a.off_mask * 32;
                float *gout = st + a.off_gout * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * kt + frow(r, h);
                    const float v = hm[f * 32 + j] > 0.0f ? acc[r] : 0.0f;
                    gout[f * 32 + j] = n0 > 0 ? gout[f * 32 + j] + v : v;
                }
            }
        }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// data gradient of a wide hidden layer (H = 128 / 256, K == H) with the weights in LDS.
// The generic kernel above reads W from L2 once per MFMA group (5.5 ms per 256x256 layer and 2^20 samples where the
// MFMAs need 0.9).  Here a workgroup stages the 32*NKT columns of W it is responsible for (all of them for H = 128, one
// half for H = 256: 133 KB) once, TRANSPOSED, and every wave walks its 32-sample tiles with the input gradient in registers
// and one conflict-free ds_read_b128 per four MFMAs.  blockIdx.y selects the column group.
// ------------------------------------------------------------------------------------------------
template <int H, int NKT, int WPB>
__global__ __launch_bounds__(WPB * 64) void dgrad_lds_kernel(DgradArgs a, int64_t n, float *__restrict__ stash)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = H / 32;
    constexpr int SW = H + 4;                             // LDS row stride (floats): W^T, one row per input feature k
    const int kt0 = blockIdx.y * NKT;
    for (int e = threadIdx.x; e < H * 32 * NKT; e += blockDim.x) {      // coalesced read of W, transposed into LDS (once)
        const int nn = e / (32 * NKT), c = e - nn * (32 * NKT);
        lds[c * SW + nn] = nn < a.N ? a.W[(int64_t)nn * a.K + 32 * kt0 + c] : 0.0f;
    }
    __syncthreads();
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        float *st = stash + tile * (int64_t)a.rows_total * 32;
        const float *gin = st + a.off_gin * 32;
        const float *hm = st + a.off_mask * 32;
        float *gout = st + a.off_gout * 32;
        f32x16 G[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * t + frow(r, h);
                const float v = gin[(nn < a.N ? nn : a.N - 1) * 32 + j];     // unconditional load, zeroed afterwards
                G[t][r] = nn < a.N ? v : 0.0f;
            }
#pragma clang loop unroll(disable)
        for (int ktl = 0; ktl < NKT; ++ktl) {
            const int kt = kt0 + ktl;
            float m[16];                                   // ReLU mask source of this output block: requested before the MFMAs
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = hm[(32 * kt + frow(r, h)) * 32 + j];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const float *wl = lds + (32 * ktl + j) * SW + 4 * h;
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(wl + 32 * t + 8 * q);     // W[n .. n+3][k]: one ds_read_b128
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = tn::mfma32(w[u], G[t][4 * q + u], acc);
                }
            }
            tn::pin16(acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) gout[(32 * kt + frow(r, h)) * 32 + j] = m[r] > 0.0f ? acc[r] : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// training forward of the FIRST layer of a wide stack (positional-encoding inputs, K0_pad <= 64) with the weights in LDS:
// a workgroup stages all H weight rows (zero-padded to 64 columns) once, every wave walks its tiles with the encoded
// inputs (workspace rows written by enc_rows_kernel, already in B-operand layout) in registers and one ds_read_b128 per
// four MFMAs.  The hidden and output layers run as fwd_wreg_kernel launches (below).
// ------------------------------------------------------------------------------------------------
// positional-encoding inputs of a tile as workspace rows [slot][32 samples] (the E rows of the layout)
// (slots: rows written -- K0_pad for the encodings, 64 for plain inputs: slots past the input width are zero rows)
__global__ __launch_bounds__(256) void enc_rows_kernel(MlpArgs a, const float *__restrict__ x, int64_t n, float *__restrict__ stash,
                                                       int rows_total, int64_t off_e, int slots)
{
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n + 31) >> 5;
    const int G0 = slots >> 3;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        const float *xrow = x + (valid ? row : 0) * a.in_dim;
        float aux3[3] = {0.f, 0.f, 0.f};
        if (a.enc == TN_ENC_POSENC) { aux3[0] = xrow[0]; aux3[1] = xrow[1]; aux3[2] = xrow[2]; }
        float *stE = stash + (tile * (int64_t)rows_total + off_e) * 32 + j;
        for (int g = 0; g < G0; ++g) {
            const f32x4 b = fetch_input(a, xrow, aux3, valid, g, h);
#pragma unroll
            for (int u = 0; u < 4; ++u) stE[(8 * g + 4 * h + u) * 32] = b[u];
        }
    }
}


// T = 32-row blocks of the layer input: 2 for the positional-encoding first layer, whose encoded inputs (<= 64 slots)
// enc_rows_kernel has written as workspace rows (the rows the weight gradient reads anyway)
template <int H, int T, int NOT, int WPB>
__global__ __launch_bounds__(WPB * 64) void fwd_lds_kernel(FwdLayerArgs a, int64_t n, float *__restrict__ stash, float *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SW = 32 * T + 4;                         // LDS row stride (floats): b128 reads of 16 rows hit 64 banks
    const int ot0 = blockIdx.y * NOT;
    if (a.K == 32 * T) {
        for (int e = threadIdx.x; e < 32 * NOT * (8 * T); e += blockDim.x) {
            const int o = e / (8 * T), c = (e - o * (8 * T)) * 4;
            const int nn = 32 * ot0 + o;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (nn < a.N) v = *reinterpret_cast<const f32x4 *>(a.W + (int64_t)nn * a.K + c);
            *reinterpret_cast<f32x4 *>(lds + o * SW + c) = v;
        }
    } else {                                               // ragged K (first layer): zero padded
        for (int e = threadIdx.x; e < 32 * NOT * 32 * T; e += blockDim.x) {
            const int o = e / (32 * T), c = e - o * (32 * T);
            const int nn = 32 * ot0 + o;
            lds[o * SW + c] = (nn < a.N && c < a.K) ? a.W[(int64_t)nn * a.K + c] : 0.0f;
        }
    }
    __syncthreads();
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int n_ot = min(NOT, ((a.N + 31) >> 5) - ot0);
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        float *st = stash + tile * (int64_t)a.rows_total * 32;
        const float *in = st + a.off_in * 32;
        float *outp = st + a.off_out * 32;
        f32x16 A[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * t + frow(r, h);
                if constexpr (32 * T == H) A[t][r] = in[k * 32 + j];          // hidden layer: K == H rows, immediate offsets
                else {
                    const float v = in[(k < a.Kp ? k : 0) * 32 + j];
                    A[t][r] = k < a.Kp ? v : 0.0f;                            // rows past Kp belong to another buffer
                }
            }
#pragma clang loop unroll(disable)
        for (int otl = 0; otl < n_ot; ++otl) {
            const int ot = ot0 + otl;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int f = 32 * ot + frow(r, h); acc[r] = a.B[f < a.N ? f : 0]; }
            const float *wl = lds + (32 * otl + j) * SW + 4 * h;
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(wl + 32 * t + 8 * q);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = tn::mfma32(w[u], A[t][4 * q + u], acc);
                }
            }
            tn::pin16(acc);
            acc = tn::relu16(acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) outp[(32 * ot + frow(r, h)) * 32 + j] = acc[r];
            if (a.off_bits >= 0) reinterpret_cast<unsigned *>(st + (a.off_bits + 2 * ot) * 32)[lane] = relu_bits(acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Hidden layers of the wide stacks (N == K == H = 128 / 256), weights in REGISTERS, activations through LDS.
//
// The LDS-weight form (dgrad_lds_kernel above; its forward twin was removed) keeps half a layer's weights in LDS and a
// tile's 128 input values per lane in registers: every wave reads its own tile from HBM (two column groups -> every input
// row twice), pays ~3 VALU instructions of address arithmetic per row load, and stalls on a full memory latency per tile
// whenever its SIMD partner does too (measured: matrix pipes 71-77 % busy; this form 78-88 %).  Here the roles are swapped: wave `ob` of a workgroup owns output block `ob` of the
// layer and holds W[32 ob .. 32 ob + 31][0 .. H) as A operands in 4 H / 32 registers for the whole launch; the H / 32 waves
// of a tile stream share each 32-sample input tile through LDS, [sample][feature] with a 4-float pad, so that the B operand
// of four consecutive MFMAs is one conflict-free ds_read_b128.  A wave brings in 32 of the tile's H rows (16 row loads per
// lane with immediate offsets, 4 ds_write_b128).  Three LDS tile buffers and ONE barrier per tile, placed in the middle of
// the tile's MFMAs: tile it+1 is written at the top of iteration it (requested a whole iteration earlier), read from the top
// of iteration it+1, and its buffer is reused for tile it+4 only after the barrier of iteration it+2 -- no wave ever waits
// at the barrier unless it is half a tile ahead of the slowest one.  Same MFMA steps in the same order as the LDS-weight
// kernels: bit-identical results.  What is left of the idle time is the vector-memory instructions themselves (ablation in
// scripts/microbench/wreg_layer.hip: no loads and no stores -> 97.6 % busy; staggering the two waves of a SIMD, spreading
// the instructions between the MFMAs, LDS-direct loads with ds_read_b32 operands and wide accesses through LDS transposes
// were all measured and are not faster).
// ------------------------------------------------------------------------------------------------
template <int H> struct WregGeom {
    static constexpr int T = H / 32;               // 32-row blocks of the layer's input and of its output
    static constexpr int STREAMS = 8 / T;          // tile streams per 8-wave workgroup
    static constexpr int SW = H + 4;               // LDS tile row stride (floats)
    static constexpr int TILE = 32 * SW;           // floats per tile buffer
    static constexpr size_t lds_bytes = (size_t)STREAMS * 3 * TILE * sizeof(float);
};

// MFMAs of reduction groups [G0, G1) (group g = 8 in-features: B operand = one ds_read_b128 at bt + 8 g) with the operands of
// the next PAIR of groups requested before the current pair's eight MFMAs are issued (left alone, hipcc sinks each read to
// just in front of its first use: read -> wait -> MFMAs -> read ...; both waves of a SIMD then wait at the same time)
template <int G0, int G1, int T>
__device__ __forceinline__ void wreg_mfma_run(f32x16 &acc, const f32x4 (&W)[T][4], const float *__restrict__ bt, f32x4 (&b)[2]) {
#pragma unroll
    for (int g = G0; g < G1; g += 2) {
        f32x4 nb[2];
        if (g + 2 < G1) {
            nb[0] = *reinterpret_cast<const f32x4 *>(bt + 8 * (g + 2));
            nb[1] = *reinterpret_cast<const f32x4 *>(bt + 8 * (g + 3));
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = tn::mfma32(W[(g + e) >> 2][(g + e) & 3][u], b[e][u], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < G1) { b[0] = nb[0]; b[1] = nb[1]; }
    }
}
__device__ __forceinline__ void wreg_first_pair(const float *__restrict__ bt, int g, f32x4 (&b)[2]) {
    b[0] = *reinterpret_cast<const f32x4 *>(bt + 8 * g);
    b[1] = *reinterpret_cast<const f32x4 *>(bt + 8 * (g + 1));
}

template <int H, bool LAST>
__global__ __launch_bounds__(512) void fwd_wreg_kernel(FwdLayerArgs a, int64_t n, float *__restrict__ stash, float *__restrict__ y)
{
    using G = WregGeom<H>;
    constexpr int T = G::T, SW = G::SW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stream = wave / T, ob = wave % T;
    float *buf = lds + stream * 3 * G::TILE;
    const int64_t n_tiles = (n + 31) >> 5;
    const int64_t stride = (int64_t)gridDim.x * G::STREAMS;
    const int64_t first = (int64_t)blockIdx.x * G::STREAMS;          // the workgroup's lowest tile: defines its iteration count
    const int64_t iters = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    if (iters == 0) return;
    // A operands: W[32 ob + j][32 t + 8 q + 4 h .. + 3]
    f32x4 W[T][4];
    {
        const int nn = 32 * ob + j;
        const float *wrow = a.W + (int64_t)(nn < a.N ? nn : 0) * a.K + 4 * h;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + 32 * t + 8 * q);
                W[t][q] = nn < a.N ? w : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    f32x16 bias;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int f = 32 * ob + frow(r, h); bias[r] = a.B[f < a.N ? f : 0]; }
    auto tile_of = [&](int64_t it) { const int64_t t = first + stream + it * stride; return t < n_tiles ? t : n_tiles - 1; };
    float stage[16];
    wreg_load_rows(urow(stash, tile_of(0) * a.rows_total + a.off_in), ob, j, h, stage);
    wreg_write_rows(buf, SW, ob, j, h, stage);
    wreg_load_rows(urow(stash, tile_of(1) * a.rows_total + a.off_in), ob, j, h, stage);
    __syncthreads();
    // results of a tile: [feature][32 samples] rows (+ y for the last layer).  A stream past the end recomputes the last tile
    // (same inputs -> same values) rather than branching around its stores.
    auto emit = [&](int64_t tile, f32x16 acc) {
        if (!LAST || 32 * ob < a.N) {
            if constexpr (!LAST) {
                wreg_store_block(urow(stash, tile * a.rows_total + a.off_out), ob, j, h, acc);
                if (a.off_bits >= 0) {                  // (wave-uniform) ReLU bits of this block for the data-gradient kernel
                    unsigned *bits = reinterpret_cast<unsigned *>(urow(stash, tile * a.rows_total + a.off_bits + 2 * ob));
                    bits[lane] = relu_bits(acc);
                }
            } else {
                float *outp = stash + (tile * a.rows_total + a.off_out + 32 * ob + 4 * h) * 32 + j;
                const int64_t row = tile * 32 + j;
                const bool valid = row < n;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = 32 * ob + 8 * q + 4 * h;
                    f32x4 v;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = valid && f + u < a.N;
                        outp[(u + 8 * q) * 32] = ok ? acc[4 * q + u] : 0.0f;
                        v[u] = tn::apply_act(acc[4 * q + u], a.out_act);
                    }
                    if (valid) {
                        if ((a.N & 3) == 0) { if (f < a.N) *reinterpret_cast<f32x4 *>(y + row * a.N + f) = v; }
                        else {
#pragma unroll
                            for (int u = 0; u < 4; ++u) if (f + u < a.N) y[row * a.N + f + u] = v[u];
                        }
                    }
                }
            }
        }
    };
    int cur = 0;
    f32x16 res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.0f;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it < iters; ++it) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        wreg_write_rows(buf + nxt * G::TILE, SW, ob, j, h, stage);              // tile it+1 (requested a whole iteration ago)
        // VMEM queue order: the PREVIOUS tile's stores, then the requests for tile it+2 -- the s_waitcnt vmcnt in front of the
        // next iteration's ds_write then waits for exactly these loads (stores issued after them would have to drain first:
        // vmcnt counts loads and stores in one in-order queue on gfx9)
        if (it > 0) emit(tile_of(it - 1), res);
        wreg_load_rows(urow(stash, tile_of(it + 2) * a.rows_total + a.off_in), ob, j, h, stage);
        __builtin_amdgcn_sched_barrier(0);
        const float *bt = buf + cur * G::TILE + j * SW + 4 * h;
        f32x16 acc = bias;
        f32x4 b[2];
        wreg_first_pair(bt, 0, b);
        wreg_mfma_run<0, 2 * T, T>(acc, W, bt, b);
        __syncthreads();
        wreg_first_pair(bt, 2 * T, b);
        wreg_mfma_run<2 * T, 4 * T, T>(acc, W, bt, b);
        tn::pin16(acc);
        if constexpr (!LAST) res = tn::relu16(acc); else res = acc;
        cur = nxt;
    }
    emit(tile_of(iters - 1), res);
}

// data gradient twin: wave `kb` owns input-feature block kb and holds W[0 .. H)[32 kb .. 32 kb + 31] (W^T rows) in registers
template <int H>
__global__ __launch_bounds__(512) void dgrad_wreg_kernel(DgradArgs a, int64_t n, float *__restrict__ stash)
{
    using G = WregGeom<H>;
    constexpr int T = G::T, SW = G::SW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stream = wave / T, kb = wave % T;
    float *buf = lds + stream * 3 * G::TILE;
    const int64_t n_tiles = (n + 31) >> 5;
    const int64_t stride = (int64_t)gridDim.x * G::STREAMS;
    const int64_t first = (int64_t)blockIdx.x * G::STREAMS;
    const int64_t iters = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    if (iters == 0) return;
    // A operands: W[32 t + 8 q + 4 h + u][32 kb + j]   (N == K == H)
    f32x4 W[T][4];
    {
        const float *wcol = a.W + (int64_t)(4 * h) * a.K + 32 * kb + j;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) W[t][q][u] = wcol[(int64_t)(32 * t + 8 * q + u) * a.K];
    }
    auto tile_of = [&](int64_t it) { const int64_t t = first + stream + it * stride; return t < n_tiles ? t : n_tiles - 1; };
    float stage[16];
    wreg_load_rows(urow(stash, tile_of(0) * a.rows_total + a.off_gin), kb, j, h, stage);
    wreg_write_rows(buf, SW, kb, j, h, stage);
    wreg_load_rows(urow(stash, tile_of(1) * a.rows_total + a.off_gin), kb, j, h, stage);
    __syncthreads();
    auto emit = [&](int64_t tile, const f32x16 &g) {
        wreg_store_block(urow(stash, tile * a.rows_total + a.off_gout), kb, j, h, g);
    };
    int cur = 0;
    f32x16 res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.0f;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it < iters; ++it) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        wreg_write_rows(buf + nxt * G::TILE, SW, kb, j, h, stage);
        // VMEM queue order (see fwd_wreg_kernel): previous tile's stores, this tile's ReLU-mask rows, requests for tile it+2
        if (it > 0) emit(tile_of(it - 1), res);
        const int64_t tile = tile_of(it);
        float m[16];
        unsigned mbits = 0;
        if (a.off_bits >= 0) mbits = reinterpret_cast<const unsigned *>(urow(stash, tile * a.rows_total + a.off_bits + 2 * kb))[lane];
        else wreg_load_block(urow(stash, tile * a.rows_total + a.off_mask), kb, j, h, m);
        wreg_load_rows(urow(stash, tile_of(it + 2) * a.rows_total + a.off_gin), kb, j, h, stage);
        __builtin_amdgcn_sched_barrier(0);
        const float *bt = buf + cur * G::TILE + j * SW + 4 * h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        f32x4 b[2];
        wreg_first_pair(bt, 0, b);
        wreg_mfma_run<0, 2 * T, T>(acc, W, bt, b);
        __syncthreads();
        wreg_first_pair(bt, 2 * T, b);
        wreg_mfma_run<2 * T, 4 * T, T>(acc, W, bt, b);
        tn::pin16(acc);
        if (a.off_bits >= 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = mask_keep(acc[r], mbits, r);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = m[r] > 0.0f ? acc[r] : 0.0f;
        }
        cur = nxt;
    }
    emit(tile_of(iters - 1), res);
}

template <int H, bool LAST>
int launch_fwd_wreg(const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s)
{
    using G = WregGeom<H>;
    auto kern = fwd_wreg_kernel<H, LAST>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_fwd: cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + G::STREAMS - 1) / G::STREAMS, 256));
    kern<<<dim3((unsigned)bl), dim3(512), G::lds_bytes, s>>>(f, n, stash, y);
    return tn::check_launch("fwd_wreg_kernel");
}

template <int H>
int launch_dgrad_wreg(const DgradArgs &d, int64_t n, float *stash, hipStream_t s)
{
    using G = WregGeom<H>;
    auto kern = dgrad_wreg_kernel<H>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + G::STREAMS - 1) / G::STREAMS, 256));
    kern<<<dim3((unsigned)bl), dim3(512), G::lds_bytes, s>>>(d, n, stash);
    return tn::check_launch("dgrad_wreg_kernel");
}

// ------------------------------------------------------------------------------------------------
// weight gradient of one layer: dW[N][K] += G[N][s] A[K][s]^T over all samples; db[N] += sum_s G
// ------------------------------------------------------------------------------------------------

template <int MAXS>
__global__ __launch_bounds__(512) void wgrad_layer_kernel(WgradArgs a, const float *__restrict__ x, int64_t n,
                                                          const float *__restrict__ stash)
{
    const int lane = tn::lane_id(), i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int Tn = (a.N + 31) >> 5, Tk = (a.K_pad + 31) >> 5;
    const int total = Tn * Tk;
    f32x16 acc[MAXS];
    float dbacc[MAXS];
    int ttn[MAXS], ttk[MAXS];
#pragma unroll
    for (int m = 0; m < MAXS; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
        dbacc[m] = 0.0f;
        const int id = wave + 8 * m;
        ttn[m] = id < total ? id / Tk : -1;
        ttk[m] = id < total ? id - (id / Tk) * Tk : 0;
    }
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const float *st = stash + tile * (int64_t)a.rows_total * 32;
#pragma unroll
        for (int m = 0; m < MAXS; ++m) {
            if (ttn[m] < 0) continue;
            const int nrow = 32 * ttn[m] + i;
            f32x4 gv[4], av[4];
            {
                const f32x4 *p = reinterpret_cast<const f32x4 *>(st + (a.off_g + (nrow < a.N ? nrow : 0)) * 32 + 16 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const f32x4 v = p[e]; gv[e] = nrow < a.N ? v : f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            const int q0 = 32 * ttk[m], q = q0 + i;
            if (a.first) {
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (q0 < a.xs) {
                    const int qx = q < a.xs ? q : 0;
                    const int64_t r0 = tile * 32 + 16 * h;
                    float xv[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) { const int64_t rc = r0 + t < n ? r0 + t : n - 1; xv[t] = x[rc * a.in_dim + qx]; }
#pragma unroll
                    for (int t = 0; t < 16; ++t) av[t >> 2][t & 3] = (q < a.xs && r0 + t < n) ? xv[t] : 0.0f;
                }
                if (a.enc != TN_ENC_NONE && q0 + 31 >= a.xs) {
                    const bool ise = q >= a.xs && q < a.K_pad;
                    const f32x4 *p = reinterpret_cast<const f32x4 *>(st + (a.off_e + (ise ? q - a.xs : 0)) * 32 + 16 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const f32x4 v = p[e]; if (ise) av[e] = v; }
                }
            } else {
                const f32x4 *p = reinterpret_cast<const f32x4 *>(st + (a.off_a + (q < a.K ? q : 0)) * 32 + 16 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const f32x4 v = p[e]; av[e] = q < a.K ? v : f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m] = tn::mfma32(gv[e][u], av[e][u], acc[m]);
            if (ttk[m] == 0) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s += (gv[e][0] + gv[e][1]) + (gv[e][2] + gv[e][3]);
                dbacc[m] += s;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MAXS; ++m) {
        if (ttn[m] < 0) continue;
        tn::pin16(acc[m]);
        const int k = 32 * ttk[m] + i;
        const bool kok = k < a.K;
        int kc = k;
        if (a.first && a.enc == TN_ENC_DIR_CAT) { const int pe = 6 * a.n_freqs + 3; kc = k < a.in_dim ? pe + k : k - a.in_dim; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = 32 * ttn[m] + frow(r, h);
            if (kok && nn < a.N) atomicAdd(&a.gW[(int64_t)nn * a.K + kc], acc[m][r]);
        }
        if (ttk[m] == 0) {
            float s = dbacc[m];
            s += __shfl_xor(s, 32, 64);
            const int nn = 32 * ttn[m] + i;
            if (h == 0 && nn < a.N) atomicAdd(&a.gB[nn], s);
        }
    }
}

// Hidden layers of the wide stacks (N == K == H = 128 / 256: H/32 x H/32 output tiles of 32 x 32): a wave owns a BN x BK block
// of tiles (2 x 4 for H = 256: 128 accumulator registers), so a 32-sample tile costs it BN + BK operand blocks for
// 16 BN BK MFMAs.  The workgroup brings a tile's 2 H rows (G_l and H_{l-1}, [feature][32 samples]) in ONCE, LDS-direct
// (global_load_lds_dwordx4: no staging registers, no ds_write; H / 32 instructions per wave and tile), double-buffered with
// one barrier per tile, and the operands are ds_read_b128.  (Each wave requesting its own operand blocks from L2 -- 24
// 16-byte loads per lane and tile, every block fetched by 2-4 waves -- ran the matrix pipes at 74 %: a vector-memory
// instruction costs its SIMD ~40-90 pipe cycles, scripts/microbench/wreg_layer.hip; this form: 84 %.)  The LDS image of an
// LDS-direct load is lane-linear (no row padding possible), so the 16-byte chunks of a row are XOR-swizzled through the
// SOURCE address: slot p of row r holds chunk p ^ ((r >> 1) & 7), which spreads the 16 rows of a ds_read_b128 phase over
// all 64 banks.
// global_load_lds_dwordx4: 16 bytes per lane from `src` (per lane) to LDS at `dst` (wave-uniform) + 16 * lane; counted in vmcnt.
// (Kept in a __device__ function: the builtin inside a __global__ template makes hipcc's host pass drop the kernel's stub.)
__device__ __forceinline__ void glds16(const float *src, float *dst) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

template <int H, int BN, int BK>
__global__ __launch_bounds__(512) void wgrad_lds_kernel(WgradArgs a, int64_t n, const float *__restrict__ stash)
{
    constexpr int TILE = 2 * H * 32;                      // floats per buffer: G rows [0, H), A rows [H, 2H)
    constexpr int NI = H / 32;                            // LDS-direct loads per wave and tile (8 rows each)
    constexpr int WK = (H / 32) / BK;                     // waves along k
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = tn::lane_id(), i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int tn0 = (wave / WK) * BN, tk0 = (wave % WK) * BK;
    f32x16 acc[BN][BK];
    float dbacc[BN];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) {
        dbacc[bn] = 0.0f;
#pragma unroll
        for (int bk = 0; bk < BK; ++bk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bn][bk][r] = 0.0f;
    }
    // this lane's part of a tile fetch: rows 8 (wave NI + e) + (lane >> 3), chunk slot lane & 7
    int src_off[NI];
#pragma unroll
    for (int e = 0; e < NI; ++e) {
        const int row = 8 * (wave * NI + e) + (lane >> 3);
        src_off[e] = (row < H ? row : row - H) * 32 + 4 * ((lane & 7) ^ ((row >> 1) & 7));       // relative to the G rows / the A rows of the tile
    }
    auto fetch = [&](int64_t tile, float *buf) {
        const float *stg = urow(stash, tile * a.rows_total + a.off_g), *sta = urow(stash, tile * a.rows_total + a.off_a);
#pragma unroll
        for (int e = 0; e < NI; ++e)
            glds16((8 * (wave * NI + e) < H ? stg : sta) + src_off[e], buf + 256 * (wave * NI + e));      // (wave-uniform choice: 8 rows per request)
    };
    // operand chunk e (samples 16 h + 4 e .. + 3) of row r: slot (4 h + e) ^ ((r >> 1) & 7)
    int g_off[BN], a_off[BK];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) g_off[bn] = (32 * (tn0 + bn) + i) * 32;
#pragma unroll
    for (int bk = 0; bk < BK; ++bk) a_off[bk] = (H + 32 * (tk0 + bk) + i) * 32;
    const int swz = (i >> 1) & 7;                          // rows 32 b + i: (r >> 1) & 7 == (i >> 1) & 7
    if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x, lds);
    __syncthreads();
    int cur = 0;
#pragma clang loop unroll(disable)
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const float *buf = lds + cur * TILE;
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x, lds + (cur ^ 1) * TILE);
        f32x4 gv[BN][4];
#pragma unroll
        for (int bn = 0; bn < BN; ++bn)
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[bn][e] = *reinterpret_cast<const f32x4 *>(buf + g_off[bn] + 4 * ((4 * h + e) ^ swz));
        if (tk0 == 0) {
#pragma unroll
            for (int bn = 0; bn < BN; ++bn) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s += (gv[bn][e][0] + gv[bn][e][1]) + (gv[bn][e][2] + gv[bn][e][3]);
                dbacc[bn] += s;
            }
        }
        // (measured and not kept: the barrier in front of the LAST k block's MFMAs, whose operands are in registers by then, so
        // that an early wave still has MFMAs to issue behind it -- 1.043 ms per layer against 1.015 with the barrier at the end)
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            f32x4 av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = *reinterpret_cast<const f32x4 *>(buf + a_off[bk] + 4 * ((4 * h + e) ^ swz));
#pragma unroll
            for (int bn = 0; bn < BN; ++bn)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[bn][bk] = tn::mfma32(gv[bn][e][u], av[e][u], acc[bn][bk]);
        }
        __syncthreads();                                   // (drains this wave's LDS-direct loads: the next tile is in place)
        cur ^= 1;
    }
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) {
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            tn::pin16(acc[bn][bk]);
            const int k = 32 * (tk0 + bk) + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * (tn0 + bn) + frow(r, h);
                atomicAdd(&a.gW[(int64_t)nn * a.K + k], acc[bn][bk][r]);
            }
        }
        if (tk0 == 0) {
            float s = dbacc[bn];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&a.gB[32 * (tn0 + bn) + i], s);
        }
    }
}

template <int H, int BN, int BK>
int launch_wgrad_lds(const WgradArgs &w, int64_t n, const float *stash, hipStream_t s)
{
    constexpr size_t lds_bytes = (size_t)2 * 2 * H * 32 * sizeof(float);
    auto kern = wgrad_lds_kernel<H, BN, BK>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    kern<<<dim3((unsigned)std::min<int64_t>(n_tiles, 256)), dim3(512), lds_bytes, s>>>(w, n, stash);
    return tn::check_launch("wgrad_lds_kernel");
}

// ------------------------------------------------------------------------------------------------
// d loss / d x of the FIRST layer of a wide stack on narrow plain inputs (in_dim <= 64: Cobafa's 36 gathered features,
// reference models.py:239-247, whose gradient goes on into the grid scatter):  gx[s][k] (+)= sum_n W_0[n][k] G_0[n][s].
// The layer-kernel form of fwd_lds_kernel with the roles swapped: W_0^T ([64 slots][H], zero padded) is staged in LDS once
// per workgroup, a wave holds its tile's H gradient rows as B operands (16 H / 32 registers) and produces the <= 2 blocks of 32
// input slots with one ds_read_b128 per four MFMAs; results leave as 16-byte stores into the row-major gx the grid kernels
// read.  Replaces dgrad_layer_kernel<H, true, 4> (every weight operand a scalar global load, 256 VGPRs + 188 AGPRs, one wave
// per SIMD: 0.55 ms per 2^20 samples at H = 128).
// ------------------------------------------------------------------------------------------------
template <int H, int WPB>
__global__ __launch_bounds__(WPB * 64) void dgrad_first_kernel(DgradArgs a, int64_t n, const float *__restrict__ stash, float *__restrict__ gx)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = H / 32, SW = H + 4;
    for (int e = threadIdx.x; e < 64 * H; e += blockDim.x) {          // lds[slot k][hidden n] = W_0[n][k]
        const int k = e / H, nn = e - k * H;
        lds[k * SW + nn] = k < a.K ? a.W[(int64_t)nn * a.K + k] : 0.0f;
    }
    __syncthreads();
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int n_kt = (a.in_dim + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        const float *gin = urow(stash, tile * (int64_t)a.rows_total + a.off_gin);
        f32x16 G[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) G[t][r] = gin[(32 * t + frow(r, h)) * 32 + j];
        const int64_t row = tile * 32 + j;
#pragma clang loop unroll(disable)
        for (int kt = 0; kt < n_kt; ++kt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const float *wl = lds + (32 * kt + j) * SW + 4 * h;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(wl + 32 * t + 8 * q);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = tn::mfma32(w[u], G[t][4 * q + u], acc);
                }
            tn::pin16(acc);
            if (row < n) {
                float *gr = gx + row * a.in_dim + 32 * kt + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * kt + 8 * q + 4 * h;
                    if (f0 + 3 < a.in_dim && (a.in_dim & 3) == 0) {
                        f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                        if (a.accum_gx) v += *reinterpret_cast<const f32x4 *>(gr + 8 * q);
                        *reinterpret_cast<f32x4 *>(gr + 8 * q) = v;
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (f0 + u < a.in_dim) gr[8 * q + u] = a.accum_gx ? gr[8 * q + u] + acc[4 * q + u] : acc[4 * q + u];
                    }
                }
            }
        }
    }
}

template <int H>
int launch_dgrad_first(const DgradArgs &d, int64_t n, const float *stash, float *gx, hipStream_t s)
{
    constexpr int WPB = 8;
    constexpr size_t lds_bytes = (size_t)64 * (H + 4) * sizeof(float);
    auto kern = dgrad_first_kernel<H, WPB>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int per_cu = std::max(1, std::min<int>((int)(160 * 1024 / lds_bytes), 2048 / (WPB * 64)));
    kern<<<dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * per_cu))), dim3(WPB * 64), lds_bytes, s>>>(d, n, stash, gx);
    return tn::check_launch("dgrad_first_kernel");
}

template <int H, int T, int NOT>
int launch_fwd_lds(const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s)
{
    constexpr int WL = 8;
    constexpr size_t lds_bytes = (size_t)32 * NOT * (32 * T + 4) * 4;
    auto kern = fwd_lds_kernel<H, T, NOT, WL>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_fwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int groups = (((f.N + 31) / 32) + NOT - 1) / NOT;
    const int per_cu = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + WL - 1) / WL, (256 * per_cu) / groups));
    kern<<<dim3((unsigned)bl, (unsigned)groups), dim3(WL * 64), lds_bytes, s>>>(f, n, stash, y);
    return tn::check_launch("fwd_lds_kernel");
}

// the layer-kernel training forward (run_fwd_only): one launch per layer, [feature][32-sample] rows between them
__host__ inline bool layer_kernel_path(int H, int L, int out) { return H >= 128 && L >= 3 && out > 4 && out <= H; }

// `inference`: no backward follows, so the hidden activations need not survive: two ping-pong buffers of H rows each (plus the
// encoded-input rows) instead of one buffer per layer -- (2 H + rowsE) * 128 B per 32 samples (2.3 KB per sample for the
// Vanilla stack against 11.5 KB of training workspace).
__host__ inline int infer_rows_total(int H, const Layout &lay) { return 2 * H + lay.rowsE; }
__host__ inline int64_t infer_ws_rows_bytes(int H, const Layout &lay, int64_t n_tiles) {       // (256-byte multiple: the packed stream of the fused form follows)
    return ((n_tiles * (int64_t)std::max(infer_rows_total(H, lay), 64) * 32 * (int64_t)sizeof(float)) + 255) & ~(int64_t)255;
}

// Where the row sets of a training workspace live; every kernel addresses  row = tile * rows_total + off.
//   * tile-major (make_layout; what the general-shape kernels compute for themselves): a tile holds all its row sets back to back,
//     rows_total = lay.total, off = the set's offset inside the tile.
//   * slab (round 5; the layer-kernel path whose first layer takes row operands, i.e. every launch gets its offsets as arguments):
//     every row set is contiguous over the tiles -- slab s = rows [s n_tiles H, (s + 1) n_tiles H), tile t of it at t H -- so
//     rows_total = H and off = s n_tiles H (+ the set's place inside a shared slab: the E rows and the bit rows are packed together).
//     A launch then streams two or three dense arrays instead of 32 KB pieces 300 - 400 KB apart: the same copy of 1 KB + 1 KB per
//     sample runs at 5.5 TB/s (6.0 with nt hints) instead of 4.98 (scripts/microbench/row_copy.hip) -- the layer kernels sat AT
//     that 4.98.
struct RowMap {
    bool slab;
    int rows_total, H, L;
    int64_t sl;                                  // rows per slab
    Layout lay;
    int64_t e_off, m_off[TN_MLP_MAX_LAYERS];
    int64_t gc_off;                              // slab layout: first of the L - 1 slabs of the fused data-gradient chain (round 6)
    int n_slabs;
    int64_t offH(int l) const { return slab ? l * sl : (int64_t)l * H; }                 // activation l (output of layer l), l < L - 1
    int64_t offE() const { return slab ? e_off : lay.rowsH; }
    int64_t offGA() const { return slab ? (L - 1) * sl : lay.rowsH + lay.rowsE; }
    int64_t offGB() const { return slab ? L * sl : lay.rowsH + lay.rowsE + lay.rowsG; }
    int64_t offM(int l) const { return slab ? m_off[l] : lay.rowsH + lay.rowsE + 2 * lay.rowsG + 2 * (H / 32) * l; }   // bit rows of activation l
    int64_t offGC(int l) const { return gc_off + l * sl; }      // slab layout: d loss / d (pre-activation of hidden layer l), kept for its weight gradient
    int64_t total_rows(int64_t n_tiles) const { return slab ? n_slabs * sl : n_tiles * lay.total; }
};
__host__ inline bool slab_eligible(int H, int L, int enc, int in_dim, int K0_pad, int out) {
    if (!(H == 128 || H == 256) || !layer_kernel_path(H, L, out) || L > TN_MLP_MAX_LAYERS) return false;
    const Layout lay = make_layout(H, L, enc, in_dim, K0_pad, out);
    const bool plain = plain_x_rows(H, L, enc, K0_pad, out) && lay.rowsE == 64;
    return ((enc == TN_ENC_POSENC && K0_pad <= 64) || plain) && lay.rowsG == H && lay.rowsE <= H;
}
__host__ inline RowMap make_rowmap(int H, int L, int enc, int in_dim, int K0_pad, int out, int64_t n_tiles, bool slab) {
    RowMap m;
    m.lay = make_layout(H, L, enc, in_dim, K0_pad, out);
    m.slab = slab; m.H = H; m.L = L; m.sl = n_tiles * H;
    m.rows_total = slab ? H : m.lay.total;
    m.e_off = 0; m.n_slabs = 0; m.gc_off = 0;
    for (int l = 0; l < TN_MLP_MAX_LAYERS; ++l) m.m_off[l] = 0;
    if (slab) {
        int64_t sidx = L + 1;                      // slabs 0 .. L - 2: activations, L - 1 / L: buffers A / B
        int within = 0;
        m.e_off = sidx * m.sl; within = m.lay.rowsE;
        const int mr = 2 * (H / 32);
        for (int l = 0; l + 1 < L; ++l) {
            if (within + mr > H) { ++sidx; within = 0; }
            m.m_off[l] = sidx * m.sl + within;
            within += mr;
        }
        m.n_slabs = (int)sidx + 1;
        // the cross-layer data-gradient chain (mlp_fused_f2.hip) writes every layer's gradient once and the weight-gradient launches read
        // them afterwards: one slab per hidden layer instead of the two ping-pong buffers
        m.gc_off = (int64_t)m.n_slabs * m.sl;
        m.n_slabs += L - 1;
    }
    return m;
}
// rows of a training workspace (without its tail): room for either layout (a backward pass without a stashed forward runs tile-major)
__host__ inline int64_t ws_rows(int H, int L, int enc, int in_dim, int K0_pad, int out, int64_t n_tiles) {
    const int64_t tm = make_rowmap(H, L, enc, in_dim, K0_pad, out, n_tiles, false).total_rows(n_tiles);
    if (!slab_eligible(H, L, enc, in_dim, K0_pad, out)) return tm;
    return std::max(tm, make_rowmap(H, L, enc, in_dim, K0_pad, out, n_tiles, true).total_rows(n_tiles));
}

template <int H>
int run_fwd_only(const MlpArgs &a, const float *x, const float *aux, int64_t n, float *y, float *stash, hipStream_t s, bool inference = false)
{
    const int64_t n_tiles = (n + 31) / 32;
    constexpr int WPB = H <= 64 ? 8 : 4;
    const int64_t blocks = std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * 2);
    if constexpr (H >= 128) {
        const int L = a.n_layers, out = a.out_dim;
        if (layer_kernel_path(H, L, out)) {       // first layer, then one launch per layer with W in LDS
            const RowMap rm = make_rowmap(H, L, a.enc, a.in_dim, a.K0_pad, out, n_tiles, !inference && slab_eligible(H, L, a.enc, a.in_dim, a.K0_pad, out));
            const Layout &lay = rm.lay;
            const int total = inference ? infer_rows_total(H, lay) : rm.rows_total;
            const int64_t offE = inference ? 2 * H : rm.offE();
            auto off_in = [&](int l) -> int64_t { return inference ? ((l - 1) & 1) * H : rm.offH(l - 1); };
            auto off_out = [&](int l) -> int64_t { return inference ? (l & 1) * H : (l + 1 < L ? rm.offH(l) : rm.offGA()); };
            auto off_bits = [&](int l) -> int64_t { return (inference || l + 1 >= L) ? -1 : rm.offM(l); };      // activation l = output of layer l
            float *tail = (a.f2 && !inference && L <= 16) ? ws_tail(stash, ws_rows(H, L, a.enc, a.in_dim, a.K0_pad, out, n_tiles)) : nullptr;
            if (tail) {
                hipError_t me = hipMemsetAsync(tail, 0, WS_TAIL_BYTES, s);
                if (me != hipSuccess) { tn::set_error("mlp_fwd(f16x2): cannot clear the workspace tail: %s", hipGetErrorString(me)); return (int)me; }
            }
            const bool plain = plain_x_rows(H, L, a.enc, a.K0_pad, out) && lay.rowsE == 64;
            if (inference && !a.layerwise && fused_fwd_ok(H, a) && ((a.enc == TN_ENC_POSENC && a.K0_pad <= 64) || plain)) {
                // round 6: the whole stack as ONE persistent launch (mlp_fused_f2.hip) -- the workspace holds the encoded input rows (64 per
                // tile, contiguous over the tiles) and, behind the layer-wise form's area, the packed weight stream
                enc_rows_kernel<<<dim3((unsigned)std::min<int64_t>((n_tiles + 3) / 4, 256 * 8)), dim3(256), 0, s>>>(a, x, n, stash, 64, 0, plain ? 64 : a.K0_pad);
                if (int rc = tn::check_launch("enc_rows_kernel")) return rc;
                if (!plain && a.K0_pad < 64) {               // rows K0_pad .. 63 of every tile are operands too (zero weights, but 0 x NaN = NaN)
                    hipError_t me = hipMemset2DAsync(stash + (size_t)a.K0_pad * 32, 64 * 128, 0, (size_t)(64 - a.K0_pad) * 128, (size_t)n_tiles, s);
                    if (me != hipSuccess) { tn::set_error("mlp_fwd(fused): cannot clear the padding rows: %s", hipGetErrorString(me)); return (int)me; }
                }
                void *pack = reinterpret_cast<unsigned char *>(stash) + infer_ws_rows_bytes(H, lay, n_tiles);
                return launch_fused_fwd_f2(H, a, n, stash, y, pack, s);
            }
            if (!inference && !a.layerwise && rm.slab && a.f2 && tail && lay.rowsE == 64 && fused_fwd_ok(H, a) &&
                ((a.enc == TN_ENC_POSENC && a.K0_pad == 64) || plain) && (!a.skip_last || out == H)) {
                // round 6: the training forward as ONE persistent launch (mlp_fused_f2.hip): activations stay in registers across the layers and
                // reach the workspace as the rows, bit rows and maxima the layer-wise backward reads -- written once, never read back here
                enc_rows_kernel<<<dim3((unsigned)std::min<int64_t>((n_tiles + 3) / 4, 256 * 8)), dim3(256), 0, s>>>(a, x, n, stash, total, offE, 64);
                if (int rc = tn::check_launch("enc_rows_kernel")) return rc;
                FusedStash sp;
                sp.rows = stash; sp.rows_total = total; sp.off_e = offE; sp.n_run = L - (a.skip_last ? 1 : 0); sp.tail = tail;
                for (int l = 0; l < TN_MLP_MAX_LAYERS; ++l) { sp.off_out[l] = 0; sp.off_bits[l] = 0; }
                for (int l = 0; l < sp.n_run; ++l) { sp.off_out[l] = off_out(l); sp.off_bits[l] = std::max<int64_t>(off_bits(l), 0); }
                float *y_f = (a.skip_last || (a.rows_only && out == H && a.out_act == TN_ACT_NONE)) ? nullptr : y;
                return launch_fused_fwd_f2(H, a, n, stash, y_f, fused_pack_area(stash, ws_rows(H, L, a.enc, a.in_dim, a.K0_pad, out, n_tiles)), s, &sp);
            }
            if ((a.enc == TN_ENC_POSENC && a.K0_pad <= 64) || plain) {      // (encoded) inputs as rows, then the first layer like any other
                enc_rows_kernel<<<dim3((unsigned)std::min<int64_t>((n_tiles + 3) / 4, 256 * 8)), dim3(256), 0, s>>>(a, x, n, stash, total, offE,
                                                                                                                  plain ? 64 : a.K0_pad);
                if (int rc = tn::check_launch("enc_rows_kernel")) return rc;
                FwdLayerArgs f;
                f.W = a.W[0]; f.B = a.B[0]; f.N = a.N[0]; f.K = a.K0; f.Kp = plain ? 64 : a.K0_pad; f.rows_total = total;
                f.off_in = offE; f.off_out = inference ? 0 : rm.offH(0); f.out_act = a.out_act; f.off_bits = off_bits(0);
                if (a.f2) { if (int rc = launch_fwd_first_f2(H, f, n, stash, s)) return rc; }
                else if (int rc = launch_fwd_lds<H, 2, H / 32>(f, n, stash, y, s)) return rc;
            } else {
                if (inference) return tn::fail(TN_E_CONFIG, "tn_mlp_fwd_ws: the layer-by-layer inference forward needs positional-encoding inputs");
                tn::warn_once(0, "mlp forward: the first layer of this width-%d stack (encoding %d, %d inputs) runs on the general-shape kernel "
                              "(fast forms: positional encoding with <= 64 slots, or <= 64 plain inputs)", H, a.enc, a.in_dim);
                fwd_stash_kernel<H, WPB, true, true><<<dim3((unsigned)std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * 4)), dim3(WPB * 64), 0, s>>>(
                    a, x, aux, nullptr, n, stash, y);
                if (int rc = tn::check_launch("fwd_stash_kernel(first layer)")) return rc;
            }
            if (a.skip_last && (inference || !rm.slab || !a.f2 || out != H))
                return tn::fail(TN_E_CONFIG, "tn_mlp_fwd_stash: TN_MLP_SKIP_LAST needs the f16x2 training forward of a slab-layout stack whose output is as wide as its hidden layers");
            for (int l = 1; l < L - (a.skip_last ? 1 : 0); ++l) {       // (TN_MLP_SKIP_LAST: the caller merged the last layer into its consumers)
                FwdLayerArgs f;
                f.W = a.W[l]; f.B = a.B[l]; f.N = a.N[l]; f.K = a.K[l]; f.Kp = a.K[l]; f.rows_total = total;
                f.off_in = off_in(l); f.off_out = off_out(l); f.out_act = a.out_act; f.off_bits = off_bits(l);
                f.max_in = tail ? tail + l : nullptr;
                // hidden layers (K == N == H) and the output layer (K == H, N <= H): weights in registers
                int rc;
                // TN_MLP_ROWS_ONLY: the last layer leaves its output as workspace rows only (out == H, no output activation: the
                // rows ARE y -- the conditions of tn_mlp_rows_view)
                float *y_l = (l + 1 == L && a.rows_only && !inference && a.f2 && out == H && a.out_act == TN_ACT_NONE) ? nullptr : y;
                if (a.f2) rc = launch_fwd_f2(H, l + 1 == L, f, n, stash, y_l, s);      // fp16 matrix cores, two-term splits, scaled
                else if (a.b3) rc = launch_fwd_b3(H, l + 1 == L, f, n, stash, y, s);   // bf16 matrix cores, exact 3-way splits
                else rc = l + 1 < L ? launch_fwd_wreg<H, false>(f, n, stash, y, s) : launch_fwd_wreg<H, true>(f, n, stash, y, s);
                if (rc) return rc;
            }
            return TN_OK;
        }
    }
    if (inference) return tn::fail(TN_E_CONFIG, "tn_mlp_fwd_ws: configuration without a layer-by-layer inference forward");
    if (H >= 128) tn::warn_once(1, "mlp forward: this width-%d stack (%d layers, %d outputs) is outside the layer-by-layer kernels and runs on "
                                   "the general-shape kernel", H, a.n_layers, a.out_dim);
    fwd_stash_kernel<H, WPB, true><<<dim3((unsigned)blocks), dim3(WPB * 64), 0, s>>>(a, x, aux, nullptr, n, stash, y);
    return tn::check_launch("fwd_stash_kernel(forward)");
}

template <int H>
int run_layers(const MlpArgs &a, const float *x, const float *aux, const float *gy, int64_t n, float *const *gw,
               float *const *gb, float *gx, float *stash, hipStream_t s, bool stashed = false, bool gy_rows = false)
{
    const int L = a.n_layers;
    const int64_t n_tiles = (n + 31) / 32;
    // (the slab layout is what the stashing forward of the same configuration left behind, run_fwd_only)
    const RowMap rm = make_rowmap(H, L, a.enc, a.in_dim, a.K0_pad, a.out_dim, n_tiles, stashed && slab_eligible(H, L, a.enc, a.in_dim, a.K0_pad, a.out_dim));
    const Layout &lay = rm.lay;
    const int64_t tail_rows = ws_rows(H, L, a.enc, a.in_dim, a.K0_pad, a.out_dim, n_tiles);
    constexpr int WPB = H <= 64 ? 8 : 4;
    if (a.skip_last) TN_REQUIRE(gy_rows && stashed && rm.slab && a.f2, TN_E_CONFIG, "tn_mlp_bwd: TN_MLP_SKIP_LAST needs TN_MLP_STASHED | TN_MLP_GRAD_Y_ROWS on the f16x2 slab-layout stack");
    if (gy_rows) {
        // TN_MLP_GRAD_Y_ROWS: the consumers of y (the heads' data-gradient chains) have written d loss / d y as rows into
        // buffer B (tn_mlp_rows_view): nothing to transpose, the walk below starts from B and ping-pongs into A
        TN_REQUIRE(stashed && a.out_act == TN_ACT_NONE && (a.out_dim & 31) == 0, TN_E_CONFIG,
                   "tn_mlp_bwd: TN_MLP_GRAD_Y_ROWS needs TN_MLP_STASHED, no output activation and out_dim % 32 == 0");
    } else if (stashed) {        // activations and the last pre-activation are in the workspace already (tn_mlp_fwd_stash)
        out_grad_kernel<<<dim3((unsigned)std::min<int64_t>(n_tiles, 256 * 8)), dim3(256), 0, s>>>(gy, n, a.out_dim, a.out_act, rm.rows_total,
                                                                                         rm.offGA(), stash);
        if (int rc = tn::check_launch("out_grad_kernel")) return rc;
    } else {
        const int64_t blocks = std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * 2);
        fwd_stash_kernel<H, WPB><<<dim3((unsigned)blocks), dim3(WPB * 64), 0, s>>>(a, x, aux, gy, n, stash);
        if (int rc = tn::check_launch("fwd_stash_kernel")) return rc;
    }
    const int64_t offE = rm.offE(), offGA = rm.offGA(), offGB = rm.offGB();
    int64_t cur = gy_rows ? offGB : offGA, nxt = gy_rows ? offGA : offGB;
    const int top = L - 1 - (a.skip_last ? 1 : 0);
    // round 6: every data gradient of the stack in ONE persistent launch (mlp_fused_f2.hip fused_chain_kernel): the gradients stay in
    // registers between the layers and reach the workspace once, as the rows the weight-gradient launches below read
    bool chain = false;
    if constexpr (H == 128 || H == 256) {
        bool square = top >= 1;
        for (int l = 1; l <= top; ++l) square = square && a.N[l] == H && a.K[l] == H;
        if (stashed && rm.slab && a.f2 && !a.layerwise && square && L <= 16 && layer_kernel_path(H, L, a.out_dim)) {
            FusedChain fc;
            fc.rows = stash; fc.rows_total = rm.rows_total; fc.off_in = cur; fc.tail = ws_tail(stash, tail_rows);
            for (int i = 0; i < TN_MLP_MAX_LAYERS; ++i) { fc.off_out[i] = 0; fc.off_bits[i] = 0; fc.tail_idx[i] = -1; }
            for (int i = 0; i < top; ++i) {
                const int l = top - i;
                fc.off_out[i] = rm.offGC(l - 1); fc.off_bits[i] = rm.offM(l - 1); fc.tail_idx[i] = 16 + l;
            }
            if (int rc = launch_fused_chain_f2(H, a, top, n, fused_pack_area(stash, tail_rows), s, &fc)) return rc;
            chain = true;
        }
    }
    const int64_t top_off = cur;
    auto g_rows = [&](int l) -> int64_t { return chain ? (l == top ? top_off : rm.offGC(l)) : cur; };       // d loss / d (pre-activation of layer l)
    // (TN_MLP_SKIP_LAST: buffer B holds d loss / d (pre-activation of layer L - 2) -- the walk starts one layer lower)
    for (int l = top; l >= 0; --l) {
        WgradArgs w;
        w.gW = gw[l]; w.gB = gb[l]; w.N = a.N[l]; w.K = a.K[l]; w.K_pad = l == 0 ? a.K0_pad : a.K[l];
        w.rows_total = rm.rows_total; w.off_g = g_rows(l); w.off_a = l > 0 ? rm.offH(l - 1) : 0; w.off_e = offE;
        w.first = l == 0; w.enc = a.enc; w.in_dim = a.in_dim; w.n_freqs = a.n_freqs; w.xs = lay.xs;
        const int tiles = ((w.N + 31) / 32) * ((w.K_pad + 31) / 32);
        const int64_t wblocks = std::min<int64_t>(n_tiles, 256 * 2);
        bool staged = false, dgrad_done = false;
        if constexpr (H == 256 || H == 128) {
            // f16x2: the layer's data gradient first -- it reports the largest |gradient| of the rows both kernels read -- then the
            // weight gradient with that scale and the one the forward pass left for the layer's input rows
            if (!w.first && w.N == H && w.K == H && a.f2 && stashed && layer_kernel_path(H, L, a.out_dim) && L <= 16) {
                float *tail = ws_tail(stash, tail_rows);
                DgradArgs d;
                d.W = a.W[l]; d.N = a.N[l]; d.K = a.K[l]; d.rows_total = rm.rows_total;
                d.off_gin = cur; d.off_gout = nxt; d.off_mask = rm.offH(l - 1);
                d.off_bits = rm.offM(l - 1);
                d.enc = a.enc; d.in_dim = a.in_dim; d.n_freqs = a.n_freqs; d.accum_gx = a.accum_gx;
                d.max_in = tail + 16 + l;
                if (!chain) { if (int rc = launch_dgrad_f2(H, d, n, stash, s)) return rc; }       // (chain: its launch has written the rows and the maximum)
                dgrad_done = true;
                w.g_max = tail + 16 + l; w.a_max = tail + l;
                if (int rc = launch_wgrad_f2(H, w, n, stash, s)) return rc;
                staged = true;
            }
        }
        if constexpr (H == 256 || H == 128) {
            if (!staged && !w.first && w.N == H && w.K == H) {
                if (a.b3) { if (int rc = launch_wgrad_b3(H, w, n, stash, s)) return rc; }
                else if (int rc = launch_wgrad_lds<H, 2, H == 256 ? 4 : 1>(w, n, stash, s)) return rc;
                staged = true;
            }
        }
        if constexpr (H == 256 || H == 128) {
            // plain inputs staged as 64 rows by the training forward (Cobafa's 36 features): same kernel, 128 x 64 / 256 x 64
            if (!staged && w.first && stashed && lay.rowsE == 64 && plain_x_rows(H, L, a.enc, a.K0_pad, a.out_dim) && w.N == H) {
                if (int rc = tn_mlp_wgrad_rows(stash + g_rows(l) * 32, (int64_t)rm.rows_total * 32, H, stash + offE * 32,
                                               (int64_t)rm.rows_total * 32, 64, w.gW, w.K, 0, w.K, w.gB, n, s)) return rc;
                staged = true;
            }
        }
        if constexpr (H == 256) {
            // first layer of the width-256 stack on positional-encoding inputs (<= 64 slots, all of them E rows of this workspace):
            // the row-operand kernel of mlp_wgrad_rows.hip (LDS-direct tiles, 2 x 1 accumulator tiles per wave) instead of
            // per-wave operand loads from L2
            if (!staged && w.first && lay.xs == 0 && w.K_pad == 64 && w.N == H) {
                if (int rc = tn_mlp_wgrad_rows(stash + g_rows(l) * 32, (int64_t)rm.rows_total * 32, H, stash + offE * 32,
                                               (int64_t)rm.rows_total * 32, 64, w.gW, w.K, 0, w.K, w.gB, n, s)) return rc;
                staged = true;
            }
        }
        if (staged) {}
        else if (tiles <= 16) wgrad_layer_kernel<2><<<dim3((unsigned)wblocks), dim3(512), 0, s>>>(w, x, n, stash);
        else if (tiles <= 64) {
            if (H >= 128) tn::warn_once(2, "mlp backward: weight gradient of a %d x %d layer on the general-shape kernel (8 tiles per wave, spills)", w.N, w.K);
            wgrad_layer_kernel<8><<<dim3((unsigned)wblocks), dim3(512), 0, s>>>(w, x, n, stash);
        }
        else return tn::fail(TN_E_CONFIG, "mlp_bwd: layer too large for the wgrad tiling");
        if (int rc = tn::check_launch("wgrad_layer_kernel")) return rc;

        DgradArgs d;
        d.W = a.W[l]; d.N = a.N[l]; d.K = a.K[l]; d.rows_total = rm.rows_total;
        d.off_gin = g_rows(l); d.off_gout = nxt; d.off_mask = l > 0 ? rm.offH(l - 1) : 0;
        // ReLU bit rows exist for every activation the training forward of the layer-kernel path wrote (run_fwd_only)
        const bool bits = stashed && layer_kernel_path(H, L, a.out_dim) && l >= 1;
        d.off_bits = bits ? rm.offM(l - 1) : -1;
        d.enc = a.enc; d.in_dim = a.in_dim; d.n_freqs = a.n_freqs; d.accum_gx = a.accum_gx;
        const int64_t blocks = std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * 4);
        if (l > 0) {
            bool done = dgrad_done || chain;
            if constexpr (H >= 128) {
                if (done) {}
                else if (a.K[l] == H && a.N[l] == H) {
                    if (a.f2 && d.off_bits >= 0) { if (int rc = launch_dgrad_f2(H, d, n, stash, s)) return rc; }
                    else if (a.b3 && d.off_bits >= 0) { if (int rc = launch_dgrad_b3(H, d, n, stash, s)) return rc; }
                    else if (int rc = launch_dgrad_wreg<H>(d, n, stash, s)) return rc;
                    done = true;
                } else if (a.K[l] == H && a.N[l] <= H) {          // weights of this layer's column group in LDS
                    constexpr int NKT = 4, WL = 8;
                    constexpr size_t lds_bytes = (size_t)32 * NKT * (H + 4) * 4;
                    auto kern = dgrad_lds_kernel<H, NKT, WL>;
                    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
                    const int groups = (H / 32) / NKT;
                    const int per_cu = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;
                    const int64_t bl = std::min<int64_t>((n_tiles + WL - 1) / WL, (256 * per_cu) / groups);
                    kern<<<dim3((unsigned)bl, (unsigned)groups), dim3(WL * 64), lds_bytes, s>>>(d, n, stash);
                    if (int rc = tn::check_launch("dgrad_lds_kernel")) return rc;
                    done = true;
                }
            }
            if (!done) {
                if (H >= 128) tn::warn_once(3, "mlp backward: data gradient of a %d x %d layer of a width-%d stack on the general-shape kernel", d.N, d.K, H);
                dgrad_layer_kernel<H, false, WPB><<<dim3((unsigned)blocks), dim3(WPB * 64), 0, s>>>(d, n, stash, nullptr);
                if (int rc = tn::check_launch("dgrad_layer_kernel")) return rc;
            }
            std::swap(cur, nxt);
        } else if (gx != nullptr && a.enc != TN_ENC_POSENC) {
            bool done = false;
            if constexpr (H == 128 || H == 256) {
                if (a.enc == TN_ENC_NONE && a.in_dim <= 64 && a.N[0] == H) {      // d loss / d x of a narrow plain input: W_0^T in LDS
                    if (int rc = launch_dgrad_first<H>(d, n, stash, gx, s)) return rc;
                    done = true;
                }
            }
            if (!done) {
                if (H >= 128) tn::warn_once(4, "mlp backward: d loss / d x of a width-%d stack with %d inputs (encoding %d) on the general-shape kernel", H, a.in_dim, a.enc);
                dgrad_layer_kernel<H, true, WPB><<<dim3((unsigned)blocks), dim3(WPB * 64), 0, s>>>(d, n, stash, gx);
                if (int rc = tn::check_launch("dgrad_layer_kernel(first)")) return rc;
            }
        }
    }
    return TN_OK;
}

}  // namespace

// internal entry points used by the dispatcher in mlp_bwd2.hip
extern "C" __attribute__((visibility("hidden"))) int64_t tn_mlp_bwd_layers_workspace_bytes(const tn_mlp_desc *desc, int64_t n)
{
    if (!desc || n <= 0) return 0;
    const int L = desc->n_layers, H = desc->dims[1];
    if (L < 2 || L > TN_MLP_MAX_LAYERS) return 0;
    if (H != 32 && H != 64 && H != 128 && H != 256) return 0;
    for (int l = 1; l < L; ++l) if (desc->dims[l] != H) return 0;
    if (((desc->dims[L] + 31) / 32) * (H / 32) > 64) return 0;          // weight-gradient tiling of the output layer (run_layers)
    const int64_t rows = ws_rows(H, L, desc->encoding, desc->in_dim, (desc->dims[0] + 7) & ~7, desc->dims[L], (n + 31) / 32);
    if (H == 128 || H == 256) return fused_pack_offset(rows) + tn::layers::fused_pack_bytes(H, L);       // (+ the packed weights of the cross-layer forward)
    return rows * 32 * (int64_t)sizeof(float) + WS_TAIL_BYTES;
}

extern "C" __attribute__((visibility("hidden"))) int tn_mlp_bwd_layers(const tn_mlp_desc *desc, const float *x, const float *aux,
                                                                      const float *grad_y, int64_t n, float *const *grad_weights,
                                                                      float *const *grad_biases, float *grad_x, float *workspace,
                                                                      void *stream)
{
    MlpArgs a;
    int H = 0;
    if (int rc = plan(desc, a, H)) return rc;
    TN_REQUIRE(x && (grad_y || (desc->flags & TN_MLP_GRAD_Y_ROWS)) && grad_weights && grad_biases && workspace, TN_E_NULL,
               "tn_mlp_bwd(layers): null pointer");
    TN_REQUIRE(a.enc != TN_ENC_DIR_CAT || aux, TN_E_NULL, "tn_mlp_bwd(layers): dir_cat needs aux");
    for (int l = 0; l < a.n_layers; ++l)
        TN_REQUIRE(grad_weights[l] && grad_biases[l], TN_E_NULL, "tn_mlp_bwd(layers): null gradient pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool stashed = (desc->flags & TN_MLP_STASHED) != 0, gy_rows = (desc->flags & TN_MLP_GRAD_Y_ROWS) != 0;
    switch (H) {
    case 32: return run_layers<32>(a, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, workspace, s, stashed, gy_rows);
    case 64: return run_layers<64>(a, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, workspace, s, stashed, gy_rows);
    case 128: return run_layers<128>(a, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, workspace, s, stashed, gy_rows);
    default: return run_layers<256>(a, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, workspace, s, stashed, gy_rows);
    }
}

extern "C" int tn_mlp_rows_view(const tn_mlp_desc *desc, int64_t n, int64_t *y_rows, int64_t *grad_y_rows, int64_t *tile_stride)
{
    TN_REQUIRE(desc && y_rows && grad_y_rows && tile_stride, TN_E_NULL, "tn_mlp_rows_view: null pointer");
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_mlp_rows_view: negative n");
    const int L = desc->n_layers;
    TN_REQUIRE(L >= 2 && L <= TN_MLP_MAX_LAYERS, TN_E_CONFIG, "tn_mlp_rows_view: n_layers must be in [2, 12]");
    const int H = desc->dims[1], out = desc->dims[L];
    // the configurations whose training forward runs layer by layer and leaves y^T behind (run_fwd_only), see tn_mlp_fwd_stash
    TN_REQUIRE(!two_pass_supported(desc) && tn_mlp_bwd_layers_workspace_bytes(desc, 32) > 0 && layer_kernel_path(H, L, out) &&
                   (out & 31) == 0 && desc->out_activation == TN_ACT_NONE && desc->encoding != TN_ENC_AUX_CAT, TN_E_CONFIG,
               "tn_mlp_rows_view: only wide stacks evaluated layer by layer (width 128 / 256, >= 3 layers, out % 32 == 0, no output activation) keep row views");
    const int K0p = (desc->dims[0] + 7) & ~7;
    const RowMap rm = make_rowmap(H, L, desc->encoding, desc->in_dim, K0p, out, (n + 31) / 32, slab_eligible(H, L, desc->encoding, desc->in_dim, K0p, out));
    *y_rows = rm.offGA() * 32;                 // buffer A: the last layer's (pre-)activation = y
    *grad_y_rows = rm.offGB() * 32;            // buffer B
    *tile_stride = (int64_t)rm.rows_total * 32;
    return TN_OK;
}

extern "C" int tn_mlp_rows_view_hidden(const tn_mlp_desc *desc, int64_t n, int64_t *h_rows, int64_t *grad_h_rows, int64_t *mask_rows,
                                       int64_t *tile_stride)
{
    TN_REQUIRE(desc && h_rows && grad_h_rows && mask_rows && tile_stride, TN_E_NULL, "tn_mlp_rows_view_hidden: null pointer");
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_mlp_rows_view_hidden: negative n");
    const int L = desc->n_layers;
    TN_REQUIRE(L >= 3 && L <= TN_MLP_MAX_LAYERS, TN_E_CONFIG, "tn_mlp_rows_view_hidden: n_layers must be in [3, 12]");
    const int H = desc->dims[1], out = desc->dims[L], K0p = (desc->dims[0] + 7) & ~7;
    TN_REQUIRE((desc->flags & TN_MLP_SKIP_LAST) && (desc->flags & TN_MLP_F16X2) && out == H && desc->out_activation == TN_ACT_NONE &&
                   !two_pass_supported(desc) && slab_eligible(H, L, desc->encoding, desc->in_dim, K0p, out), TN_E_CONFIG,
               "tn_mlp_rows_view_hidden: TN_MLP_SKIP_LAST | TN_MLP_F16X2 on a slab-layout stack (width 128 / 256, output as wide, no output activation)");
    const RowMap rm = make_rowmap(H, L, desc->encoding, desc->in_dim, K0p, out, (n + 31) / 32, true);
    *h_rows = rm.offH(L - 2) * 32;
    *grad_h_rows = rm.offGB() * 32;
    *mask_rows = rm.offM(L - 2) * 32;
    *tile_stride = (int64_t)rm.rows_total * 32;
    return TN_OK;
}

// inference forward of a wide stack on positional-encoding inputs through the layer kernels (88 % of the fp32 matrix rate where the
// register-resident kernel with weights from L2 reaches 67-70 %): workspace = two ping-pong activation buffers + encoded inputs
extern "C" int64_t tn_mlp_fwd_workspace_bytes(const tn_mlp_desc *desc, int64_t n)
{
    if (!desc || n <= 0) return 0;
    const int L = desc->n_layers;
    if (L < 2 || L > TN_MLP_MAX_LAYERS) return 0;
    const int H = desc->dims[1], out = desc->dims[L];
    const int K0p = (desc->dims[0] + 7) & ~7;         // positional-encoding inputs, or plain inputs staged as rows (Cobafa's 36 features)
    if ((H != 128 && H != 256) || !layer_kernel_path(H, L, out) || K0p > 64 ||
        !(desc->encoding == TN_ENC_POSENC || plain_x_rows(H, L, desc->encoding, K0p, out))) return 0;
    for (int l = 1; l < L; ++l) if (desc->dims[l] != H) return 0;
    const Layout lay = make_layout(H, L, desc->encoding, desc->in_dim, (desc->dims[0] + 7) & ~7, out);
    // the layer-wise form's ping-pong rows, then the packed weight stream of the cross-layer form (mlp_fused_f2.hip; 2.4 MB for Vanilla)
    return infer_ws_rows_bytes(H, lay, (n + 31) / 32) + fused_pack_bytes(H, L);
}

extern "C" int tn_mlp_fwd_ws(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y, void *workspace,
                             int64_t workspace_bytes, void *stream)
{
    TN_REQUIRE(desc, TN_E_NULL, "tn_mlp_fwd_ws: null descriptor");
    if (n <= 0) return n == 0 ? TN_OK : tn::fail(TN_E_SIZE, "tn_mlp_fwd_ws: negative n");
    const int64_t need = tn_mlp_fwd_workspace_bytes(desc, n);
    TN_REQUIRE(need > 0, TN_E_CONFIG, "tn_mlp_fwd_ws: this configuration has no workspace form (tn_mlp_fwd_workspace_bytes == 0): use tn_mlp_fwd");
    TN_REQUIRE(x && y && workspace && workspace_bytes >= need, TN_E_NULL, "tn_mlp_fwd_ws: null pointer or workspace too small");
    TN_REQUIRE((((uintptr_t)workspace | (uintptr_t)y) & 15) == 0, TN_E_ALIGN, "tn_mlp_fwd_ws: workspace / y must be 16-byte aligned");
    MlpArgs a;
    int H = 0;
    if (int rc = plan(desc, a, H)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (H == 128) return run_fwd_only<128>(a, x, aux, n, y, (float *)workspace, s, true);
    return run_fwd_only<256>(a, x, aux, n, y, (float *)workspace, s, true);
}

// training forward of a stack the layer-by-layer form covers: y + activations + last pre-activation into the workspace
extern "C" __attribute__((visibility("hidden"))) int tn_mlp_fwd_stash_layers(const tn_mlp_desc *desc, const float *x, const float *aux,
                                                                            int64_t n, float *y, float *workspace, void *stream)
{
    MlpArgs a;
    int H = 0;
    if (int rc = plan(desc, a, H)) return rc;
    TN_REQUIRE(x && y && workspace, TN_E_NULL, "tn_mlp_fwd_stash(layers): null pointer");
    TN_REQUIRE(a.enc != TN_ENC_DIR_CAT || aux, TN_E_NULL, "tn_mlp_fwd_stash(layers): dir_cat needs aux");
    TN_REQUIRE(a.enc != TN_ENC_AUX_CAT, TN_E_CONFIG, "tn_mlp_fwd_stash(layers): aux_cat is only implemented for the two-pass form");
    hipStream_t s = (hipStream_t)stream;
    switch (H) {
    case 32: return run_fwd_only<32>(a, x, aux, n, y, workspace, s);
    case 64: return run_fwd_only<64>(a, x, aux, n, y, workspace, s);
    case 128: return run_fwd_only<128>(a, x, aux, n, y, workspace, s);
    default: return run_fwd_only<256>(a, x, aux, n, y, workspace, s);
    }
}
