// Backward of the fused MLP heads (csrc/mlp.hip) as ONE persistent launch on fp32 MFMA.
//
// Per 32-sample tile a wavefront
//   1. recomputes the forward pass, keeping every hidden activation in registers (no activation
//      ever touches HBM, unlike autograd which saves each layer's input);
//   2. walks the layers backwards: the data gradient stays in the transposed register layout of
//      mlp_device.h (A operand = W^T read along torch weight rows, B operand = the previous
//      gradient registers), masked by ReLU' from the kept activations;
//   3. forms the weight gradient dW_l = G_l * a_l^T, whose reduction index is the SAMPLE, on MFMA
//      too: G_l and a_l are transposed through a small per-wave LDS scratch (feature-major rows),
//      the 32x32 product tile is accumulated into a workgroup-shared fp32 image of all weight
//      gradients in LDS with ds_add_f32, and that image is flushed to HBM once per workgroup.
// Weights are read from L2 (global) here because LDS holds the gradient image.
//
// The first layer's encodings are those of the forward (PE / [PE(d), d, x]); output activation
// derivatives: exp(y-1) uses the clamped derivative of truncated_exp (reference models.py:50-53),
// sigmoid s(1-s).
#include "mlp_device.h"
#include <algorithm>
#include <type_traits>

namespace {

using tn::f32x16;
using tn::f32x4;

// compile-time loop: f(integral_constant<0>) ... f(integral_constant<N-1>) -- register arrays indexed by the
// loop variable stay in registers (a runtime layer loop would send them to scratch)
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int SCR_STRIDE = 34;                       // floats per scratch row (32 samples + 2: conflict-free b64)
constexpr int SCR_FLOATS = 2 * 32 * SCR_STRIDE;      // G tile + A tile per wave

struct BwdArgs {
    int n_layers, in_dim, K0, K0_pad, enc, n_freqs, out_act, out_dim;
    const float *freqs;
    const float *W[TN_MLP_MAX_LAYERS];
    const float *B[TN_MLP_MAX_LAYERS];
    float *gW[TN_MLP_MAX_LAYERS];
    float *gB[TN_MLP_MAX_LAYERS];
    int K[TN_MLP_MAX_LAYERS], N[TN_MLP_MAX_LAYERS];
    int dw_off[TN_MLP_MAX_LAYERS], db_off[TN_MLP_MAX_LAYERS];
    int acc_floats;          // size of the gradient image
    int accum_gx;            // grad_x += (TN_MLP_ACCUM_GRAD_X)
};

__device__ __forceinline__ int col0(const BwdArgs &a, int q) {   // slot -> torch column of layer 0
    if (a.enc == TN_ENC_DIR_CAT) {
        const int pe = 6 * a.n_freqs + 3;
        return q < a.in_dim ? pe + q : q - a.in_dim;
    }
    return q;
}

__device__ __forceinline__ f32x4 fetch_in(const BwdArgs &a, const float *__restrict__ xrow, const float *aux3, bool valid,
                                          int g, int h)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!valid) return v;
    const int q0 = 8 * g + 4 * h;
    if (a.enc == TN_ENC_POSENC) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q0 + u < a.K0) v[u] = tn::posenc_value(aux3, q0 + u, a.n_freqs, a.freqs);
        return v;
    }
    if (q0 + 3 < a.in_dim && (a.in_dim & 3) == 0) return *reinterpret_cast<const f32x4 *>(xrow + q0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = q0 + u;
        if (q < a.in_dim) v[u] = xrow[q];
        else if (a.enc == TN_ENC_DIR_CAT) {
            const int p = q - a.in_dim;
            if (p < 6 * a.n_freqs) v[u] = tn::posenc_value(aux3, p, a.n_freqs, a.freqs);
            else if (p < 6 * a.n_freqs + 3) v[u] = aux3[p - 6 * a.n_freqs];
        }
    }
    return v;
}

// hidden layer from global weights, out-of-place: y = relu(W x + b)
template <int H>
__device__ __forceinline__ void hidden_fwd(const float *__restrict__ W, const float *__restrict__ bias,
                                           const f32x16 (&x)[H / 32], f32x16 (&y)[H / 32], int i, int h)
{
    constexpr int T = H / 32;
#pragma unroll
    for (int ob = 0; ob < T; ++ob) y[ob] = tn::bias_tile(bias, ob, h);
#pragma unroll
    for (int kb = 0; kb < T; ++kb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[T];
#pragma unroll
            for (int ob = 0; ob < T; ++ob)
                w[ob] = *reinterpret_cast<const f32x4 *>(W + (32 * ob + i) * H + 32 * kb + 8 * q + 4 * h);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ob = 0; ob < T; ++ob) y[ob] = tn::mfma32(w[ob][u], x[kb][4 * q + u], y[ob]);

        }
    }
#pragma unroll
    for (int ob = 0; ob < T; ++ob) y[ob] = tn::relu16(y[ob]);
}

__device__ __forceinline__ float act_grad(float pre, int act) {
    if (act == TN_ACT_EXP_M1) return expf(fminf(fmaxf(pre - 1.0f, -15.0f), 15.0f));     // models.py:50-53
    if (act == TN_ACT_SIGMOID) { const float s = 1.0f / (1.0f + expf(-pre)); return s * (1.0f - s); }
    return 1.0f;
}

// transpose one D-layout tile (32 features x 32 samples) into feature-major scratch rows
__device__ __forceinline__ void scratch_write(float *scr, const f32x16 &t, int j, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * h) * SCR_STRIDE + j] = t[r];
}

// the 16 operand values of lane (i = lane&31, h' = lane>>5): row i, samples 16h' .. 16h'+15
// The rows were written by OTHER lanes of this wave: DS operations of a wave execute in issue order, so a
// compiler-level barrier (no instruction) between the writes and these reads is all the ordering needed.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void scratch_read(const float *scr, int i, int h, float (&v)[16]) {
    asm volatile("" ::: "memory");
    const f32x2 *p = reinterpret_cast<const f32x2 *>(scr + i * SCR_STRIDE + 16 * h);
#pragma unroll
    for (int s = 0; s < 8; ++s) { const f32x2 t = p[s]; v[2 * s] = t[0]; v[2 * s + 1] = t[1]; }
    asm volatile("" ::: "memory");
}

// one 32x32 tile of dW_l = G_l a_l^T: reduce over the 32 samples on MFMA, add into the LDS image.
// Out-of-range rows / columns add 0.0f to element 0 instead of branching around each ds_add_f32.
__device__ __forceinline__ void wgrad_tile(const BwdArgs &a, const float *scrA, const float (&gop)[16], float *dW, int Kl,
                                           int Nl, int tn, int tk, int j, int h, bool first_layer)
{
    float aop[16];
    scratch_read(scrA, j, h, aop);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = tn::mfma32(gop[s], aop[s], acc);
    tn::pin16(acc);
    const int k = 32 * tk + j;
    const bool kok = k < Kl;
    const int kc = kok ? (first_layer ? col0(a, k) : k) : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nn = 32 * tn + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool ok = kok && nn < Nl;
        atomicAdd(&dW[ok ? nn * Kl + kc : 0], ok ? acc[r] : 0.0f);
    }

}

// NH = number of hidden activations kept (= n_layers - 1)
template <int H, int NH, int WPB>
__global__ __launch_bounds__(WPB * 64) void mlp_bwd_kernel(BwdArgs a, const float *__restrict__ x,
                                                           const float *__restrict__ aux, const float *__restrict__ gy,
                                                           int64_t n, float *__restrict__ gx)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = H / 32;
    constexpr int L = NH + 1;
    const int lane = tn::lane_id();
    const int wave = threadIdx.x >> 6;
    float *scrG = lds + a.acc_floats + wave * SCR_FLOATS;
    float *scrA = scrG + 32 * SCR_STRIDE;
    for (int e = threadIdx.x; e < a.acc_floats; e += WPB * 64) lds[e] = 0.0f;
    __syncthreads();

    const int64_t n_tiles = (n + 31) >> 5;
    const int G0 = a.K0_pad >> 3;
    const int out = a.out_dim;
    const int n_ot = (out + 31) >> 5;          // output tiles (<= T)

    const int j_ = lane & 31, h_ = lane >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        // Re-materialise the lane coordinates every tile: otherwise LICM hoists the hundreds of
        // loop-invariant per-lane weight addresses out of this persistent loop and spills them.
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
        // ================= forward recompute =================
        f32x16 act[NH][T];
        {
#pragma unroll
            for (int ob = 0; ob < T; ++ob) act[0][ob] = tn::bias_tile(a.B[0], ob, h);
#pragma clang loop unroll(disable)
            for (int g = 0; g < G0; ++g) {
                const f32x4 b = fetch_in(a, xrow, aux3, valid, g, h);
                f32x4 w[T];
#pragma unroll
                for (int ob = 0; ob < T; ++ob) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = 8 * g + 4 * h + u;
                        w[ob][u] = q < a.K0 ? a.W[0][(32 * ob + j) * a.K0 + col0(a, q)] : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int ob = 0; ob < T; ++ob) act[0][ob] = tn::mfma32(w[ob][u], b[u], act[0][ob]);
            }
#pragma unroll
            for (int ob = 0; ob < T; ++ob) { tn::pin16(act[0][ob]); act[0][ob] = tn::relu16(act[0][ob]); }
        }
        static_for<NH - 1>([&](auto lc) {
            constexpr int l = decltype(lc)::value + 1;
            hidden_fwd<H>(a.W[l], a.B[l], act[l - 1], act[l], j, h);
        });


        // ================= output gradient  G = gy * act'(pre) in D layout =================
        f32x16 G[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) G[t][r] = 0.0f;
        if (out <= 4) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < out) {
                    const float pre = tn::small_out<H>(a.W[L - 1] + o * H, a.B[L - 1][o], act[NH - 1], h);
                    const float g = valid ? gy[row * out + o] * act_grad(pre, a.out_act) : 0.0f;
                    if (h == 0) G[0][o] = g;              // feature o lives in half 0, reg o
                }
            }
        } else {
#pragma unroll
            for (int ob = 0; ob < T; ++ob) {
                if (ob < n_ot) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
                        acc[r] = f < out ? a.B[L - 1][f] : 0.f;
                    }
                    const int arow = 32 * ob + j;
#pragma unroll
                    for (int kb = 0; kb < T; ++kb)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 w = {0.f, 0.f, 0.f, 0.f};
                            if (arow < out) w = *reinterpret_cast<const f32x4 *>(a.W[L - 1] + arow * H + 32 * kb + 8 * q + 4 * h);
#pragma unroll
                            for (int u = 0; u < 4; ++u) acc = tn::mfma32(w[u], act[NH - 1][kb][4 * q + u], acc);
                        }
                    tn::pin16(acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
                        G[ob][r] = (valid && f < out) ? gy[row * out + f] * act_grad(acc[r], a.out_act) : 0.0f;
                    }
                }
            }
        }


        // ================= layers, last to first =================
        static_for<L>([&](auto lc) {
            constexpr int l = L - 1 - decltype(lc)::value;
            constexpr int lm1 = l > 0 ? l - 1 : 0;
            const int Nl = a.N[l], Kl = a.K[l];
            const int Tn = (l == L - 1) ? n_ot : T;
            float *dW = lds + a.dw_off[l];
            float *dB = lds + a.db_off[l];
            const int Tk = (a.K0_pad + 31) >> 5;           // k-tiles of the first layer's (encoded) input
            // ---- weight / bias gradient ----
            static_for<T>([&](auto tc) {
                constexpr int tn = decltype(tc)::value;
                if (tn < Tn) {
                    scratch_write(scrG, G[tn], j, h);
                    float gop[16];
                    scratch_read(scrG, j, h, gop);
                    {
                        float s = 0.f;
#pragma unroll
                        for (int k = 0; k < 16; ++k) s += gop[k];
                        if (32 * tn + j < Nl) atomicAdd(&dB[32 * tn + j], s);
                    }
                    if constexpr (l == 0) {
#pragma clang loop unroll(disable)
                        for (int tk = 0; tk < Tk; ++tk) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                                if (4 * tk + q < G0) v = fetch_in(a, xrow, aux3, valid, 4 * tk + q, h);
#pragma unroll
                                for (int u = 0; u < 4; ++u) scrA[(8 * q + 4 * h + u) * SCR_STRIDE + j] = v[u];
                            }
                            wgrad_tile(a, scrA, gop, dW, Kl, Nl, tn, tk, j, h, true);
                        }
                    } else {
#pragma unroll
                        for (int tk = 0; tk < T; ++tk) {
                            scratch_write(scrA, act[lm1][tk], j, h);
                            wgrad_tile(a, scrA, gop, dW, Kl, Nl, tn, tk, j, h, false);
                        }
                    }
                }
            });

            // ---- data gradient: G <- relu'(a_l) * (W_l^T G), or grad_x for l == 0 ----
            const int ng = (Nl + 7) >> 3;                     // groups of 8 out-features actually present
            if constexpr (l > 0) {
                f32x16 Gn[T];
#pragma unroll
                for (int kt = 0; kt < T; ++kt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Gn[kt][r] = 0.0f;
                }
#pragma unroll
                for (int tn = 0; tn < T; ++tn) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int g = 4 * tn + q;
                        if (g < ng) {
                            float w[T][4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int nn = 8 * g + 4 * h + u;
#pragma unroll
                                for (int kt = 0; kt < T; ++kt) w[kt][u] = nn < Nl ? a.W[l][nn * H + 32 * kt + j] : 0.0f;
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int kt = 0; kt < T; ++kt) Gn[kt] = tn::mfma32(w[kt][u], G[tn][4 * q + u], Gn[kt]);

                        }
                    }
                }
#pragma unroll
                for (int kt = 0; kt < T; ++kt) {
                    tn::pin16(Gn[kt]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) G[kt][r] = act[lm1][kt][r] > 0.0f ? Gn[kt][r] : 0.0f;
                }
            } else if (gx != nullptr && a.enc != TN_ENC_POSENC) {
                const int n_kt = (a.in_dim + 31) >> 5;
#pragma clang loop unroll(disable)
                for (int kt = 0; kt < n_kt; ++kt) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    const int k = 32 * kt + j;
                    const int kc = col0(a, k);
#pragma unroll
                    for (int tn = 0; tn < T; ++tn) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int g = 4 * tn + q;
                            if (g < ng) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const int nn = 8 * g + 4 * h + u;
                                    const float w = (nn < Nl && k < a.in_dim) ? a.W[0][nn * a.K0 + kc] : 0.0f;
                                    acc = tn::mfma32(w, G[tn][4 * q + u], acc);
                                }
                            }
                        }
                    }
                    tn::pin16(acc);
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int f0 = 32 * kt + 8 * q + 4 * h;
                            if (f0 + 3 < a.in_dim && (a.in_dim & 3) == 0) {
                                f32x4 v;
#pragma unroll
                                for (int u = 0; u < 4; ++u) v[u] = acc[4 * q + u];
                                f32x4 *dst = reinterpret_cast<f32x4 *>(gx + row * a.in_dim + f0);
                                if (a.accum_gx) v += *dst;
                                *dst = v;
                            } else {
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                    if (f0 + u < a.in_dim) {
                                        float *dst = gx + row * a.in_dim + f0 + u;
                                        *dst = a.accum_gx ? *dst + acc[4 * q + u] : acc[4 * q + u];
                                    }
                            }
                        }
                    }
                }
            }
        });
    }

    // ================= flush the gradient image =================
    __syncthreads();
    for (int l = 0; l < L; ++l) {
        const int nw = a.N[l] * a.K[l];
        const float *dW = lds + a.dw_off[l];
        for (int e = threadIdx.x; e < nw; e += WPB * 64) {
            const float v = dW[e];
            if (v != 0.0f) atomicAdd(&a.gW[l][e], v);
        }
        const float *dB = lds + a.db_off[l];
        for (int e = threadIdx.x; e < a.N[l]; e += WPB * 64) {
            const float v = dB[e];
            if (v != 0.0f) atomicAdd(&a.gB[l][e], v);
        }
    }
}

constexpr int LDS_LIMIT_FLOATS = 160 * 1024 / 4;

int plan(const tn_mlp_desc *d, float *const *gw, float *const *gb, BwdArgs &a, int &H)
{
    TN_REQUIRE(d, TN_E_NULL, "mlp_bwd: null descriptor");
    const int L = d->n_layers;
    TN_REQUIRE(L >= 2 && L <= 5, TN_E_CONFIG, "mlp_bwd: the register-resident backward supports 2..5 layers");
    H = d->dims[1];
    TN_REQUIRE(H == 32 || H == 64 || H == 128, TN_E_CONFIG, "mlp_bwd: hidden width must be 32, 64 or 128");
    TN_REQUIRE((L - 1) * H <= 256, TN_E_CONFIG, "mlp_bwd: hidden activations do not fit the register file");
    for (int l = 1; l < L; ++l) TN_REQUIRE(d->dims[l] == H, TN_E_CONFIG, "mlp_bwd: all hidden layers must share one width");
    TN_REQUIRE(gw && gb, TN_E_NULL, "mlp_bwd: null gradient pointer arrays");
    a.n_layers = L; a.in_dim = d->in_dim; a.K0 = d->dims[0]; a.K0_pad = (a.K0 + 7) & ~7;
    a.enc = d->encoding; a.n_freqs = d->n_freqs; a.out_act = d->out_activation; a.out_dim = d->dims[L]; a.accum_gx = d->flags & TN_MLP_ACCUM_GRAD_X;
    a.freqs = d->freqs;
    TN_REQUIRE(a.out_dim >= 1 && a.out_dim <= H, TN_E_SIZE, "mlp_bwd: out width must be in [1, hidden]");
    switch (a.enc) {
    case TN_ENC_NONE: TN_REQUIRE(a.K0 == a.in_dim, TN_E_CONFIG, "mlp_bwd: dims[0] must equal in_dim"); break;
    case TN_ENC_POSENC: TN_REQUIRE(a.in_dim == 3 && a.K0 == 6 * a.n_freqs, TN_E_CONFIG, "mlp_bwd: bad posenc widths"); break;
    case TN_ENC_DIR_CAT: TN_REQUIRE(a.K0 == a.in_dim + 6 * a.n_freqs + 3, TN_E_CONFIG, "mlp_bwd: bad dir_cat widths"); break;
    default: return tn::fail(TN_E_CONFIG, "mlp_bwd: unknown encoding");
    }
    int off = 0;
    for (int l = 0; l < L; ++l) {
        TN_REQUIRE(d->weights[l] && d->biases[l] && gw[l] && gb[l], TN_E_NULL, "mlp_bwd: null weight / gradient pointer");
        a.W[l] = d->weights[l]; a.B[l] = d->biases[l]; a.gW[l] = gw[l]; a.gB[l] = gb[l];
        a.K[l] = d->dims[l]; a.N[l] = d->dims[l + 1];
        a.dw_off[l] = off; off += (a.N[l] * a.K[l] + 3) & ~3;
        a.db_off[l] = off; off += (a.N[l] + 3) & ~3;
    }
    a.acc_floats = off;
    TN_REQUIRE(off + 8 * SCR_FLOATS <= LDS_LIMIT_FLOATS, TN_E_CONFIG, "mlp_bwd: gradient image does not fit LDS");
    return TN_OK;
}

template <int H, int NH>
int launch(const BwdArgs &a, const float *x, const float *aux, const float *gy, int64_t n, float *gx, hipStream_t s)
{
    const int64_t n_tiles = (n + 31) / 32;
    // two waves per SIMD when the kept activations are small enough to leave room in the register file
    constexpr int WPB = (NH * H / 2 <= 64) ? 8 : 4;
    int wpb_fit = WPB;
    if (a.acc_floats + WPB * SCR_FLOATS > LDS_LIMIT_FLOATS) wpb_fit = 4;
    if (wpb_fit != WPB) return tn::fail(TN_E_CONFIG, "mlp_bwd: gradient image + scratch exceed LDS");
    const size_t lds_bytes = (size_t)(a.acc_floats + WPB * SCR_FLOATS) * 4;
    auto kern = mlp_bwd_kernel<H, NH, WPB>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)LDS_LIMIT_FLOATS * 4 / lds_bytes, (size_t)(16 / WPB)));
    const int64_t blocks = std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * per_cu);
    kern<<<dim3((unsigned)blocks), dim3(WPB * 64), lds_bytes, s>>>(a, x, aux, gy, n, gx);
    return tn::check_launch("mlp_bwd_kernel");
}

template <int H>
int launch_h(const BwdArgs &a, const float *x, const float *aux, const float *gy, int64_t n, float *gx, hipStream_t s)
{
    switch (a.n_layers - 1) {
    case 1: return launch<H, 1>(a, x, aux, gy, n, gx, s);
    case 2: if constexpr (2 * H <= 256) return launch<H, 2>(a, x, aux, gy, n, gx, s); break;
    case 3: if constexpr (3 * H <= 256) return launch<H, 3>(a, x, aux, gy, n, gx, s); break;
    case 4: if constexpr (4 * H <= 256) return launch<H, 4>(a, x, aux, gy, n, gx, s); break;
    default: break;
    }
    return tn::fail(TN_E_CONFIG, "mlp_bwd: unsupported depth for this width");
}

}  // namespace

// single-kernel form; the public entry point tn_mlp_bwd (mlp_bwd2.hip) dispatches here when no workspace is given
extern "C" __attribute__((visibility("hidden"))) int tn_mlp_bwd_fused1(const tn_mlp_desc *desc, const float *x, const float *aux,
                                                                      const float *grad_y, int64_t n, float *const *grad_weights,
                                                                      float *const *grad_biases, float *grad_x, void *stream)
{
    BwdArgs a;
    int H = 0;
    if (int rc = plan(desc, grad_weights, grad_biases, a, H)) return rc;
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_mlp_bwd: negative n");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && grad_y, TN_E_NULL, "tn_mlp_bwd: null pointer");
    TN_REQUIRE(a.enc != TN_ENC_DIR_CAT || aux, TN_E_NULL, "tn_mlp_bwd: dir_cat needs aux (ray directions)");
    hipStream_t s = (hipStream_t)stream;
    switch (H) {
    case 32: return launch_h<32>(a, x, aux, grad_y, n, grad_x, s);
    case 64: return launch_h<64>(a, x, aux, grad_y, n, grad_x, s);
    default: return launch_h<128>(a, x, aux, grad_y, n, grad_x, s);
    }
}
