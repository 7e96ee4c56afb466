// Backward of the fused MLP heads, two-pass form (the width-64 decoders; wide / deep stacks: mlp_bwd_layers.hip).
//
// A single kernel that keeps every hidden activation AND the gradient image on chip is left with one wave per
// SIMD and weights in L2 (round 1 measured ~8 % of the fp32 MFMA peak; that form is gone).  Splitting along the
// one place where the data layout has to change anyway -- the weight gradient reduces over SAMPLES,
// everything else over FEATURES -- gives two clean kernels:
//
//   chain kernel  (samples on lanes, weights resident in LDS, 2+ waves per SIMD)
//       recompute forward -> output gradient -> data-gradient chain -> grad_x.  Hidden activations are
//       streamed to a workspace as they are produced (only their ReLU bit masks stay in registers), and so
//       is every layer's pre-activation gradient.  Workspace rows are [feature][32 samples] (128 B): the
//       transposition the weight gradient needs is done by the store / load address pattern, for free.
//   wgrad kernel  (features on lanes)
//       dW_l += G_l * H_l^T as 32x32 MFMA tiles whose reduction index is the sample; each wave owns a few
//       (layer, row-tile, column-tile) accumulators for its workgroup's share of the samples and flushes
//       them once with full-line atomics.  The first layer's input operand is rebuilt from x / aux (PE
//       recomputed), so the encoded input is never stored.
//
// Workspace: (2*NH*H + 4) rows of 128 B per 32 samples (2.06 KB/sample for the K-Planes colour head).
#include "mlp_stage.h"
#include "kplanes_scatter.h"
#include <algorithm>
#include <type_traits>

extern "C" int64_t tn_mlp_bwd_layers_workspace_bytes(const tn_mlp_desc *desc, int64_t n);
extern "C" int tn_mlp_bwd_layers(const tn_mlp_desc *desc, const float *x, const float *aux, const float *grad_y, int64_t n,
                                 float *const *grad_weights, float *const *grad_biases, float *grad_x, float *workspace, void *stream);
extern "C" int tn_mlp_wgrad_rows2(const float *g_rows, int64_t g_stride, const float *g_rows2, int64_t g_stride2, const float *a_rows,
                                  int64_t a_stride, int na, float *gW, int ldw, int col0, float *gB, float *gW2, int ldw2, int col02,
                                  float *gB2, int64_t n, void *stream);
extern "C" int tn_heads_dx_rows(const float *w_a, int ld_a, int col0_a, const float *w_b, int ld_b, int col0_b, const float *g_a, int64_t gs_a,
                                const float *g_b, int64_t gs_b, int F, float *out, int64_t out_stride, const void *mask, int64_t mask_stride,
                                int64_t n, void *stream);
extern "C" int tn_mlp_wgrad_lean_pair(const tn_mlp_desc *desc, const tn_mlp_desc *partner, const float *x, const float *aux, int64_t n,
                                      float *const *gw, float *const *gb, float *const *gws, float *const *gbs, const float *ws_a,
                                      const float *ws_b, void *stream);
extern "C" int tn_mlp_wgrad_rows(const float *g_rows, int64_t g_stride, int ng, const float *a_rows, int64_t a_stride, int na, float *gW,
                                 int ldw, int col0, int kmax, float *gB, int64_t n, void *stream);

#ifdef TN_PHASE_TIMERS
__device__ unsigned long long tn_phase_cycles_b[16];
#define TN_PTB_BEGIN unsigned long long ptb_ = __builtin_amdgcn_s_memtime();
#define TN_PTB(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (tn::lane_id() == 0) atomicAdd(&tn_phase_cycles_b[k], n_ - ptb_); ptb_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
extern "C" int tn_debug_phase_cycles_b(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(tn_phase_cycles_b), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(tn_phase_cycles_b), z, sizeof(z)); }
    return 0;
}
#else
#define TN_PTB_BEGIN
#define TN_PTB(k)
#endif
namespace {

using tn::f32x16;
using tn::f32x4;
using namespace tn::mlp;

template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float act_grad(float pre, int act) {
    if (act == TN_ACT_EXP_M1) return expf(fminf(fmaxf(pre - 1.0f, -15.0f), 15.0f));     // models.py:50-53
    if (act == TN_ACT_SIGMOID) { const float s = 1.0f / (1.0f + expf(-pre)); return s * (1.0f - s); }
    return 1.0f;
}

// ------------------------------------------------------------------------------------------------
// chain kernel
// ------------------------------------------------------------------------------------------------
// STASHED: the training forward (tn_mlp_fwd_stash) already wrote H_l, the ReLU masks and the pre-activation of the
// last layer; the kernel starts at the output gradient.
// ACCUM (with STASHED, in_dim % 32 == 0): grad_x += ...; the old values of the next 32-column block are requested before the
// stores of the current one, for the same in-order-retirement reason as the tile prefetch below.
// PAIR (with STASHED): a second head `b` with ONE hidden layer of the same width that reads the same x (the K-Planes
// sigma head next to the colour head): its data gradient runs in the same pass, so x's gradient is written once as the
// sum of both heads instead of written by one launch and read-modified-written by the next.
struct PairArgs { MlpArgs b; const float *gy; float *stash; int64_t g0_stride = 0; };

// KP (with STASHED, PAIR; north star: the K-Planes lookup in the same launch as the MLP, backward half): x is the K-Planes
// feature row, so d(loss)/d(x) is not an output but an intermediate: every wave keeps the three 32-column blocks of its
// tile's data gradient in registers and scatters them into the nine plane gradients itself (kplanes_scatter.h) -- the
// atomics of one wave's scatter travel while the other waves of the SIMD run their MFMA chains, and grad_x is written only
// if the caller asks for it.
struct KpBwd {
    int H[3], W[3];
    const float *planes[3][3];
    float *grads[3][3];
    const float *coords;
    int64_t coord_stride;
};

template <int H>
__device__ __forceinline__ void first_dgrad(const float *__restrict__ Wf, int sf, int out, const float (&gp)[4],
                                            const unsigned (&mask)[H / 32], int h, f32x16 (&G)[H / 32])
{
#pragma unroll
    for (int kb = 0; kb < H / 32; ++kb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < out) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(Wf + o * sf + 32 * kb + 8 * q + 4 * h);
                    acc4 += w * gp[o];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) G[kb][4 * q + u] = mask_keep(acc4[u], mask[kb], 4 * q + u);
        }
    }
}

// G0B (with STASHED, without PAIR; round 5, heads behind a wide stack): grad_x additionally takes W_0b^T G_0b of ANOTHER head whose data
// gradient already ran WITHOUT its grad_x part -- G_0b are rows of that head's workspace (pr.stash + tile * pr.g0_stride), pr.b holds
// only its first layer's x columns.  The 5-layer colour head's chain then needs no first layer in LDS (16 waves per CU instead of 8),
// and the 2-layer sigma head's launch -- which otherwise read, added to and re-wrote the colour head's grad_x rows -- writes the sum once.
template <int H, int NH, int WPB, bool STASHED, bool ACCUM = false, bool PAIR = false, bool KP = false, bool G0B = false>
__global__ __launch_bounds__(WPB * 64) void mlp_chain_kernel(MlpArgs a, const float *__restrict__ x, const float *__restrict__ aux,
                                                             const float *__restrict__ gy, int64_t n, float *__restrict__ gx,
                                                             float *__restrict__ stash, PairArgs pr, KpBwd kp)
{
    static_assert(!KP || (STASHED && PAIR && !ACCUM && H == 64), "the fused scatter belongs to the paired, stashed chain of the width-64 heads");
    static_assert(!G0B || (STASHED && !PAIR && !KP && !ACCUM), "G0B: another head's G_0 rows join a stashed, unpaired chain's grad_x");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = H / 32;
    constexpr int L = NH + 1;
    stage_weights(a, lds);
    const float *ldsb = lds + a.lds_floats;
    if constexpr (PAIR || G0B) stage_weights(pr.b, lds + a.lds_floats);
    __syncthreads();
    if constexpr (STASHED && NH > 1) {
        // nothing is recomputed: the hidden layers are only ever read transposed (W_l^T G_l), so transpose them in place once
        // and let every dgrad operand be one ds_read_b128 (4 consecutive n of one k) instead of four strided ds_read_b32
        constexpr int sl = H + 4;
        for (int l = 1; l < NH; ++l) {
            float *W = lds + a.w_off[l];
            for (int e = threadIdx.x; e < H * H; e += blockDim.x) {
                const int r = e / H, c = e - r * H;
                if (r < c) { const float t = W[r * sl + c]; W[r * sl + c] = W[c * sl + r]; W[c * sl + r] = t; }
            }
        }
        __syncthreads();
    }
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: tile indices and workspace bases are scalars
    const int64_t n_tiles = (n + 31) >> 5;
    const int G0 = a.K0_pad >> 3;
    const int out = a.out_dim;

    // STASHED: the per-tile inputs (ReLU masks, last pre-activation, output gradient) of the NEXT tile are requested at the
    // top of the current one.  VMEM operations of a wave retire in order, so a load issued after a tile's ~150 row stores
    // could only be consumed once all of them had drained: one full store round trip per tile, on every wave.
    unsigned nmask[NH][T];
    float npre[4], ngy[4];
    unsigned bmask[T];
    float bpre[4], bgy[4];
    const int Rb = stash_rows(H, 1, 0);                    // PAIR: workspace rows per tile of head b (NH = 1, no E rows)
    const int extra_ = extra_rows(a.enc, a.in_dim, a.K0_pad);
    auto fetch_tile = [&](int64_t t) {
        if constexpr (PAIR) {
            const float *qb = pr.stash + t * (int64_t)(Rb * 32) + stash_rows_w(H, 1, 0) * 32;
            const unsigned *mb = reinterpret_cast<const unsigned *>(qb + 4 * 32);
#pragma unroll
            for (int ob = 0; ob < T; ++ob) bmask[ob] = mb[ob * 64 + lane];
            int64_t rb = t * 32 + j_;
            rb = rb < n ? rb : n - 1;
            const int outb = pr.b.out_dim;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int oc = o < outb ? o : outb - 1;
                bpre[o] = qb[oc * 32 + j_];
                bgy[o] = pr.gy[rb * outb + oc];
            }
        }
        const float *q = stash + t * (int64_t)(stash_rows(H, NH, extra_) * 32) + stash_rows_w(H, NH, extra_) * 32;
        const unsigned *mm = reinterpret_cast<const unsigned *>(q + 4 * 32);
#pragma unroll
        for (int l = 0; l < NH; ++l)
#pragma unroll
            for (int ob = 0; ob < T; ++ob) nmask[l][ob] = mm[(l * T + ob) * 64 + lane];
        int64_t r = t * 32 + j_;
        r = r < n ? r : n - 1;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int oc = o < out ? o : out - 1;          // unconditional loads (see mlp.hip): duplicates are ignored
            npre[o] = q[oc * 32 + j_];
            ngy[o] = gy[r * out + oc];
        }
    };
    if constexpr (STASHED) {
        const int64_t t0 = (int64_t)blockIdx.x * WPB + wave;
        fetch_tile(t0 < n_tiles ? t0 : n_tiles - 1);
    }

    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));           // keep per-lane LDS addresses out of LICM's reach
        TN_PTB_BEGIN
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        const int extra = extra_rows(a.enc, a.in_dim, a.K0_pad);
        float *st = stash + tile * (int64_t)(stash_rows(H, NH, extra) * 32);
        float *stH = st;                                // H_1 .. H_NH
        float *stG = st + NH * H * 32;                  // G_0 .. G_{NH-1}
        float *stP = st + 2 * NH * H * 32;              // g_pre (4 rows)
        float *stE = stP + 4 * 32;                      // encoded extras
        float *stQ = st + stash_rows_w(H, NH, extra) * 32;             // pre-activation of the last layer (4 rows)
        unsigned *stM = reinterpret_cast<unsigned *>(stQ + 4 * 32);    // ReLU masks
        const float *Wf = lds + a.w_off[L - 1];
        const float *Bf = lds + a.b_off[L - 1];
        const int sf = a.stride[L - 1];
        unsigned mask[NH][T];
        float gp[4];
        unsigned pmask[T];
        float gpb[4];
        if constexpr (STASHED) {
#pragma unroll
            for (int l = 0; l < NH; ++l)
#pragma unroll
                for (int ob = 0; ob < T; ++ob) mask[l][ob] = nmask[l][ob];
#pragma unroll
            for (int o = 0; o < 4; ++o) gp[o] = (valid && o < out) ? ngy[o] * act_grad(npre[o], a.out_act) : 0.0f;
            if constexpr (PAIR) {
#pragma unroll
                for (int ob = 0; ob < T; ++ob) pmask[ob] = bmask[ob];
#pragma unroll
                for (int o = 0; o < 4; ++o) gpb[o] = (valid && o < pr.b.out_dim) ? bgy[o] * act_grad(bpre[o], pr.b.out_act) : 0.0f;
            }
            {
                const int64_t tn_ = tile + (int64_t)gridDim.x * WPB;
                fetch_tile(tn_ < n_tiles ? tn_ : n_tiles - 1);       // before this tile's stores
            }
            if constexpr (PAIR) {                                    // head b: g_pre rows, G_0 = relu'(H_1) * (W_1^T g_pre), G_0 rows
                float *stb = pr.stash + tile * (int64_t)(Rb * 32);
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (h == 0) stb[(2 * H + o) * 32 + j] = gpb[o];
                f32x16 Gb[T];
                first_dgrad<H>(ldsb + pr.b.w_off[1], pr.b.stride[1], pr.b.out_dim, gpb, pmask, h, Gb);
#pragma unroll
                for (int ob = 0; ob < T; ++ob) store_rows(stb + H * 32, Gb[ob], ob, j, h);
                // (G_0b is rebuilt from g_pre and the mask where grad_x needs it: 6 live registers instead of 32)
            }
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (h == 0) stP[o * 32 + j] = gp[o];
            (void)stQ; (void)stM;
        } else {
        const float *xrow = x + (valid ? row : 0) * a.in_dim;
        float aux3[3] = {0.f, 0.f, 0.f};
        const float *auxrow = nullptr;
        if (valid) {
            if (a.enc == TN_ENC_POSENC) { aux3[0] = xrow[0]; aux3[1] = xrow[1]; aux3[2] = xrow[2]; }
            else if (a.enc == TN_ENC_DIR_CAT) { aux3[0] = aux[3 * row]; aux3[1] = aux[3 * row + 1]; aux3[2] = aux[3 * row + 2]; }
            else if (a.enc == TN_ENC_AUX_CAT) auxrow = aux + (int64_t)(a.aux_index ? a.aux_index[row] : row) * a.aux_stride;
        }
        const int xs = x_slots(a.enc, a.in_dim);

        // ---------------- forward ----------------
        f32x16 act[T];
        {
            const float *W0 = lds + a.w_off[0];
#pragma unroll
            for (int ob = 0; ob < T; ++ob) act[ob] = tn::bias_tile(lds + a.b_off[0], ob, h);
            f32x4 b = fetch_input(a, xrow, aux3, valid, 0, h, auxrow);
            for (int g = 0; g < G0; ++g) {
                f32x4 bn = {0.f, 0.f, 0.f, 0.f};
                if (g + 1 < G0) bn = fetch_input(a, xrow, aux3, valid, g + 1, h, auxrow);
                f32x4 w[T];
#pragma unroll
                for (int ob = 0; ob < T; ++ob) w[ob] = load_a4<true>(W0, 32 * ob + j, 8 * g + 4 * h, a.K0, a.stride[0]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int ob = 0; ob < T; ++ob) act[ob] = tn::mfma32(w[ob][u], b[u], act[ob]);
                if (extra > 0 && 8 * g + 4 * h + 3 >= xs) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = 8 * g + 4 * h + u;
                        if (q >= xs) stE[(q - xs) * 32 + j] = b[u];
                    }
                }
                b = bn;
            }
#pragma unroll
            for (int ob = 0; ob < T; ++ob) {
                tn::pin16(act[ob]);
                act[ob] = tn::relu16(act[ob]);
                mask[0][ob] = relu_bits(act[ob]);
                store_rows(stH, act[ob], ob, j, h);
            }
        }
        static_for<NH - 1>([&](auto lc) {
            constexpr int l = decltype(lc)::value + 1;          // layer l: H_l -> H_{l+1}
            tn::hidden_layer<H>(lds + a.w_off[l], lds + a.b_off[l], a.stride[l], act, j, h);
#pragma unroll
            for (int ob = 0; ob < T; ++ob) {
                mask[l][ob] = relu_bits(act[ob]);
                store_rows(stH + l * H * 32, act[ob], ob, j, h);
            }
        });

        // ---------------- output gradient (out <= 4) and first data-gradient step on the VALU ----------------
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            gp[o] = 0.0f;
            if (o < out) {
                const float pre = tn::small_out<H>(Wf + o * sf, Bf[o], act, h);
                gp[o] = valid ? gy[row * out + o] * act_grad(pre, a.out_act) : 0.0f;
            }
            if (h == 0) stP[o * 32 + j] = gp[o];
        }
        }
        TN_PTB(0)
        f32x16 G[T];
#pragma unroll
        for (int kb = 0; kb < T; ++kb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (o < out) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(Wf + o * sf + 32 * kb + 8 * q + 4 * h);
                        acc4 += w * gp[o];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) G[kb][4 * q + u] = mask_keep(acc4[u], mask[NH - 1][kb], 4 * q + u);
            }
        }

        // ---------------- hidden layers, last to first ----------------
        static_for<NH - 1>([&](auto lc) {
            constexpr int l = NH - 1 - decltype(lc)::value;     // l = NH-1 .. 1 : G holds G_l
#pragma unroll
            for (int ob = 0; ob < T; ++ob) store_rows(stG + l * H * 32, G[ob], ob, j, h);
            const float *Wl = lds + a.w_off[l];
            constexpr int sl = H + 4;                       // stride of every hidden layer (mlp_stage.h plan()): a constant lets the
                                                            // row offsets below become ds_read immediates instead of VALU adds
            f32x16 Gn[T];
#pragma unroll
            for (int kt = 0; kt < T; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) Gn[kt][r] = 0.0f;
            // software pipeline over the 4T groups (tn_, q): operands of group g+1 are requested before group g's MFMAs
            {
                constexpr int NG = 4 * T;
                f32x4 cur[T], nxt[T];
                auto operands = [&](int g, f32x4 (&w)[T]) {
#pragma unroll
                    for (int kt = 0; kt < T; ++kt) {
                        if constexpr (STASHED) w[kt] = *reinterpret_cast<const f32x4 *>(Wl + (32 * kt + j) * sl + 8 * g + 4 * h);   // W^T rows
                        else {
#pragma unroll
                            for (int u = 0; u < 4; ++u) w[kt][u] = Wl[(8 * g + 4 * h + u) * sl + 32 * kt + j];
                        }
                    }
                };
                operands(0, cur);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int tn_ = g >> 2, q = g & 3;
                    if (g + 1 < NG) operands(g + 1, nxt);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int kt = 0; kt < T; ++kt) Gn[kt] = tn::mfma32(cur[kt][u], G[tn_][4 * q + u], Gn[kt]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kt = 0; kt < T; ++kt) cur[kt] = nxt[kt];
                }
            }
#pragma unroll
            for (int kt = 0; kt < T; ++kt) {
                tn::pin16(Gn[kt]);
#pragma unroll
                for (int r = 0; r < 16; ++r) G[kt][r] = mask_keep(Gn[kt][r], mask[l - 1][kt], r);
            }
        });
#pragma unroll
        for (int ob = 0; ob < T; ++ob) store_rows(stG, G[ob], ob, j, h);          // G_0

        TN_PTB(1)
        // ---------------- grad_x = W_0^T G_0 over the x slots ----------------
        if ((KP || gx != nullptr || (!KP && a.gx_rows != nullptr)) && a.enc != TN_ENC_POSENC) {
            f32x16 gacc[KP ? 3 : 1];                // KP: d loss / d features of the three scales, kept for the scatter
            const float *W0 = lds + a.w_off[0];
            const int s0 = a.stride[0];
            const int n_kt = (a.in_dim + 31) >> 5;
            f32x16 Gb[T];
            if constexpr (PAIR) first_dgrad<H>(ldsb + pr.b.w_off[1], pr.b.stride[1], pr.b.out_dim, gpb, pmask, h, Gb);
            if constexpr (G0B) {
                const float *g0b = pr.stash + tile * pr.g0_stride;
#pragma unroll
                for (int ob = 0; ob < T; ++ob) tn::mlp::load_rows(g0b, Gb[ob], ob, j, h);
            }
            const int64_t rowc = row < n ? row : n - 1;
            f32x4 old[4], oldn[4];
            // d loss / d x as [feature][32-sample] rows (tn_mlp_desc::grad_x_rows: the consumer is a wide stack's layer kernel):
            // the D tile of a 32-column block IS 32 such rows, written like every workspace row
            float *const gxr = (!KP && a.gx_rows != nullptr) ? a.gx_rows + tile * a.gx_rows_stride : nullptr;      // (wave-uniform)
            if constexpr (ACCUM) {
                if (gxr == nullptr) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) old[q] = *reinterpret_cast<const f32x4 *>(gx + rowc * a.in_dim + 8 * q + 4 * h);
                }
            }
#pragma clang loop unroll(disable)
            for (int kt = 0; kt < n_kt; ++kt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                if constexpr (ACCUM) {
                    if (gxr != nullptr) tn::mlp::load_rows(gxr + kt * 32 * 32, acc, 0, j, h);     // the MFMAs below accumulate on top
                }
                if constexpr (ACCUM) {
                    if (gxr == nullptr) {
                    const int ktn = kt + 1 < n_kt ? kt + 1 : kt;
#pragma unroll
                    for (int q = 0; q < 4; ++q) oldn[q] = *reinterpret_cast<const f32x4 *>(gx + rowc * a.in_dim + 32 * ktn + 8 * q + 4 * h);
                    }
                }
                {
                    constexpr int NG = 4 * T;
                    float cur[4], nxt[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) cur[u] = W0[(4 * h + u) * s0 + 32 * kt + j];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const int tn_ = g >> 2, q = g & 3;
                        if (g + 1 < NG) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) nxt[u] = W0[(8 * (g + 1) + 4 * h + u) * s0 + 32 * kt + j];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc = tn::mfma32(cur[u], G[tn_][4 * q + u], acc);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
                    }
                }
                if constexpr (PAIR || G0B) {                          // + W_0b^T G_0b (x slots of head b are columns 0 .. in_dim-1)
                    const float *W0b = ldsb + pr.b.w_off[0];
                    const int s0b = pr.b.stride[0];
#pragma unroll
                    for (int tn_ = 0; tn_ < T; ++tn_)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                acc = tn::mfma32(W0b[(32 * tn_ + 8 * q + 4 * h + u) * s0b + 32 * kt + j], Gb[tn_][4 * q + u], acc);
                }
                tn::pin16(acc);
                if constexpr (KP) {                 // (uniform branches: kt is a loop counter)
                    if (kt == 0) gacc[0] = acc; else if (kt == 1) gacc[1] = acc; else gacc[2] = acc;
                    if (gx == nullptr) continue;
                }
                if (gxr != nullptr) {               // rows of samples >= n carry zeros (their output gradient was zeroed)
                    if (!KP && a.gx_mask_rows != nullptr) {      // x is a hidden activation of the producer: d / d (its pre-activation)
                        const unsigned mb = a.gx_mask_rows[tile * a.gx_mask_stride + kt * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = mask_keep(acc[r], mb, r);
                    }
                    store_rows(gxr + kt * 32 * 32, acc, 0, j, h);
                    continue;
                }
                if constexpr (ACCUM) {
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = old[q];
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] += acc[4 * q + u];
                            *reinterpret_cast<f32x4 *>(gx + row * a.in_dim + 32 * kt + 8 * q + 4 * h) = v;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) old[q] = oldn[q];
                } else if (valid) {
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
            TN_PTB(2)
            if constexpr (KP) {
                // ---------------- plane gradients: scatter of this tile's three 32-channel blocks ----------------
                const float *cr = kp.coords + (valid ? row : 0) * kp.coord_stride;
                const float xs[3] = {cr[0], cr[1], cr[2]};
                float *wave_lds = lds + a.lds_floats + pr.b.lds_floats + wave * tn::KP_WAVE_LDS;
#pragma unroll
                for (int sc = 0; sc < 3; ++sc) {
                    tn::f32x4k g4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) g4[q] = tn::f32x4k{gacc[sc][4 * q], gacc[sc][4 * q + 1], gacc[sc][4 * q + 2], gacc[sc][4 * q + 3]};
                    const float *const pl[3] = {kp.planes[sc][0], kp.planes[sc][1], kp.planes[sc][2]};
                    float *const gr[3] = {kp.grads[sc][0], kp.grads[sc][1], kp.grads[sc][2]};
                    tn::kp_scatter_scale<4, 8>(pl, gr, kp.H[sc], kp.W[sc], 32, xs, valid, g4, 4 * h, wave_lds, j, h);
                    TN_PTB(3 + sc)
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad kernel
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    int n_layers, in_dim, K0, K0_pad, enc, n_freqs, out_dim, Tk0, total_tiles;
    int tk_skip;                // first-layer k tiles [0, tk_skip) are somebody else's (x columns taken from row views, mlp_wgrad_rows.hip)
    const int *aux_index;       // TN_ENC_AUX_CAT
    int aux_stride;
    float *gW[TN_MLP_MAX_LAYERS];
    float *gB[TN_MLP_MAX_LAYERS];
    int K[TN_MLP_MAX_LAYERS], N[TN_MLP_MAX_LAYERS];
};

__device__ __forceinline__ int wg_col0(const WgradArgs &a, int q) {
    if (a.enc == TN_ENC_DIR_CAT || a.enc == TN_ENC_AUX_CAT) {
        const int pe = a.K0 - a.in_dim;
        return q < a.in_dim ? pe + q : q - a.in_dim;
    }
    return q;
}

// Tile id -> (layer, tn, tk).  Order: layer 0 tiles (tn-major), hidden layers, last layer.
template <int H, int NH>
__device__ __forceinline__ void decode_tile(const WgradArgs &a, int id, int &l, int &tn_, int &tk) {
    constexpr int T = H / 32;
    const int tk0n = a.Tk0 - a.tk_skip;
    const int n0 = T * tk0n;
    if (id < n0) { l = 0; tn_ = id / tk0n; tk = a.tk_skip + id - tn_ * tk0n; return; }
    id -= n0;
    if (id < (NH - 1) * T * T) { l = 1 + id / (T * T); id -= (l - 1) * T * T; tn_ = id / T; tk = id - tn_ * T; return; }
    id -= (NH - 1) * T * T;
    l = NH; tn_ = 0; tk = id;
}

// The workgroup (NW waves) stages one 32-sample tile of the workspace (all rows, plus the x rows of the tile) in
// LDS, so every byte is read from HBM exactly once (PMC: reading operands straight from global re-fetched each
// row ~2x), then each wave runs the MFMA tiles it owns (tile id = wave + NW*m) from LDS.  The next tile is
// prefetched into registers while the current one is being multiplied (issue early / write late).
constexpr int RS = 36;      // LDS row stride in floats: 16-B aligned rows, conflict-free ds_read_b128 across 16 lanes

template <int H, int NH, int MAXS, int NW, int NCH, bool AUX>
__global__ __launch_bounds__(NW * 64) void mlp_wgrad_kernel(WgradArgs a, const float *__restrict__ x, const float *__restrict__ aux,
                                                            int64_t n, const float *__restrict__ stash)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = tn::lane_id(), i_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: tile ownership lives in SGPRs
    const int64_t n_tiles = (n + 31) >> 5;
    const int xs = x_slots(a.enc, a.in_dim);
    const int xs_st = 32 * a.tk_skip >= xs ? 0 : xs;      // x columns that are this kernel's: staged sample-major (none when skipped)
    const int extra = extra_rows(a.enc, a.in_dim, a.K0_pad);
    const int R = stash_rows_w(H, NH, extra);             // rows staged per tile
    const int Rt = stash_rows(H, NH, extra);              // rows per tile in the workspace
    const int row_chunks = R * 8;                         // float4 chunks of the workspace tile
    const int x_chunks = xs_st > 0 ? (32 * a.in_dim) / 4 : 0;   // float4 chunks of the x rows of the tile (contiguous)
    const int aw = AUX ? a.K0_pad - a.in_dim : 0;         // aux-table columns (multiple of 8, <= 64)
    float *ldsR = lds;
    float *ldsX = lds + R * RS;
    float *ldsA = ldsX + (xs_st > 0 ? 32 * a.in_dim : 0);
    f32x16 acc[MAXS];
    float dbacc[MAXS];
    int tl[MAXS], ttn[MAXS], ttk[MAXS];
#pragma unroll
    for (int m = 0; m < MAXS; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
        dbacc[m] = 0.0f;
        tl[m] = -1; ttn[m] = 0; ttk[m] = 0;
        const int id = wave + NW * m;
        if (id < a.total_tiles) decode_tile<H, NH>(a, id, tl[m], ttn[m], ttk[m]);
    }
    // staging registers: NCH float4 of workspace rows / x rows per thread, plus (AUX) one float4 of the aux-table row of
    // one sample: 32 samples x 16 float4 slots = 512 chunks, one per thread of the first 8 waves
    f32x4 pre[NCH];
    f32x4 preA = {0.f, 0.f, 0.f, 0.f};
    // Every load below is unconditional (addresses clamped into valid memory, results of idle slots never committed):
    // a load whose result is merged with a constant at a control-flow join makes the compiler wait for it right there
    // (s_waitcnt vmcnt(0) after every load), which serialises the prefetch with the MFMA phase it is meant to overlap.
    auto prefetch = [&](int64_t tile) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(stash + tile * (int64_t)Rt * 32);
        const int64_t x0 = tile * 32 * (int64_t)a.in_dim;          // first float of the tile's x rows
        const int64_t xlast = n * (int64_t)a.in_dim - 4;           // in_dim % 4 == 0: chunks never straddle the end
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = threadIdx.x + k * NW * 64;
            int64_t e = x0 + 4 * (int64_t)(c - row_chunks);
            e = e < 0 ? 0 : (e > xlast ? xlast : e);
            const f32x4 *px = reinterpret_cast<const f32x4 *>(x + e);
            const f32x4 *pr = src + (c < row_chunks ? c : 0);
            pre[k] = *((c < row_chunks || xs_st == 0) ? pr : px);
        }
        if constexpr (AUX) {
            const int s_ = (threadIdx.x >> 4) & 31, part = threadIdx.x & 15;
            int64_t r_ = tile * 32 + s_;
            r_ = r_ < n ? r_ : n - 1;
            if (a.aux_index) r_ = a.aux_index[r_];
            preA = *reinterpret_cast<const f32x4 *>(aux + r_ * a.aux_stride + (4 * part < aw ? 4 * part : 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = threadIdx.x + k * NW * 64;
            if (c < row_chunks) *reinterpret_cast<f32x4 *>(ldsR + (c >> 3) * RS + (c & 7) * 4) = pre[k];
            else if (c < row_chunks + x_chunks) *reinterpret_cast<f32x4 *>(ldsX + 4 * (c - row_chunks)) = pre[k];
        }
        if constexpr (AUX) {
            const int s_ = threadIdx.x >> 4, part = threadIdx.x & 15;
            if (threadIdx.x < 512 && 4 * part < aw) *reinterpret_cast<f32x4 *>(ldsA + s_ * aw + 4 * part) = preA;
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) prefetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                   // everyone finished reading the previous tile
        commit();
        __syncthreads();
        if (tile + gridDim.x < n_tiles) prefetch(tile + gridDim.x);      // in flight during the MFMAs below
        int i = i_, h = h_;
        asm volatile("" : "+v"(i), "+v"(h));     // per-lane LDS addresses are rebuilt per tile, not kept live (spills)
        const float *rowsE = ldsR + (2 * NH * H + 4) * RS;
#pragma unroll
        for (int m = 0; m < MAXS; ++m) {
            const int l = tl[m];
            if (l < 0) continue;
            // G operand: row segment [16 samples of this half]; A-side operand: a row segment (hidden layers, E rows) or
            // a column of the sample-major x / aux rows.  Four samples (one e) at a time keeps the live set small.
            const int grow = l < NH ? (NH + l) * H + 32 * ttn[m] + i : 2 * NH * H + (i < 4 ? i : 0);
            const f32x4 *gp = reinterpret_cast<const f32x4 *>(ldsR + grow * RS + 16 * h);
            const bool gok = l < NH || i < 4;
            const f32x4 *rp = nullptr;           // row-type source
            const float *cp = nullptr;           // column-type source
            int cstride = 0;
            if (l == 0) {
                const int q = 32 * ttk[m] + i;
                if (q < xs) { cp = ldsX + 16 * h * a.in_dim + q; cstride = a.in_dim; }
                else if (AUX && q < a.K0_pad) { cp = ldsA + 16 * h * aw + (q - xs); cstride = aw; }
                else if (!AUX && extra > 0 && q < a.K0_pad) rp = reinterpret_cast<const f32x4 *>(rowsE + (q - xs) * RS + 16 * h);
            } else {
                rp = reinterpret_cast<const f32x4 *>(ldsR + ((l - 1) * H + 32 * ttk[m] + i) * RS + 16 * h);
            }
            float gsum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 gv = gp[e];
                if (!gok) gv = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 av = {0.f, 0.f, 0.f, 0.f};
                if (l == 0) {
                    if (cp != nullptr) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) av[u] = cp[(4 * e + u) * cstride];
                    } else if (rp != nullptr) av = rp[e];
                } else av = rp[e];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m] = tn::mfma32(gv[u], av[u], acc[m]);
                gsum += (gv[0] + gv[1]) + (gv[2] + gv[3]);
            }
            if (ttk[m] == 0) dbacc[m] += gsum;
        }
    }
    // ---- flush: full-line atomics (lanes = consecutive columns of one weight row) ----
    const int i = i_, h = h_;
#pragma unroll
    for (int m = 0; m < MAXS; ++m) {
        const int l = tl[m];
        if (l < 0) continue;
        tn::pin16(acc[m]);
        const int Nl = a.N[l], Kl = a.K[l];
        const int k = 32 * ttk[m] + i;
        const bool kok = k < Kl;
        const int kc = kok ? (l == 0 ? wg_col0(a, k) : k) : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = 32 * ttn[m] + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (kok && nn < Nl) atomicAdd(&a.gW[l][(int64_t)nn * Kl + kc], acc[m][r]);
        }
        if (ttk[m] == 0) {
            float sum = dbacc[m];
            sum += __shfl_xor(sum, 32, 64);
            const int nn = 32 * ttn[m] + i;
            if (h == 0 && nn < Nl) atomicAdd(&a.gB[l][nn], sum);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad kernel, paired form (hidden width 64 = two 32-row tiles per layer)
//
// The single-buffer kernel above spends every iteration as  commit | barrier | MFMA phase | barrier , with the next
// tile's loads in flight during the MFMA phase only; the tile needs longer from HBM (~25 MB chip-wide per iteration)
// than the MFMAs take, so the difference is exposed on every iteration.  Here only the rows every wave shares (H_NH,
// G_l, g_pre, x, aux) are staged, in TWO LDS buffers, and each wave reads the H_{l-1} rows of its own tiles straight
// from the workspace in MFMA operand layout (64 contiguous bytes per lane).  Iteration t: write tile t+1 into the other
// buffer, request tile t+2 (staging registers), multiply tile t, ONE barrier: every load has a full iteration to arrive.
// All loads are unconditional (clamped tile indices) so that the compiler keeps exact vmcnt counts instead of draining
// the queue at control-flow joins.  Ownership: a wave owns the TWO tiles that share one A-side operand --
// (layer l, tn = 0 and 1, one tk) -- so every H_{l-1} operand is read from the workspace by exactly one wave (the
// 16-byte-per-lane operand loads are what the texture-address unit handles worst: 64 requests per instruction), and
// every LDS operand column is read once for both tiles.  The last layer's two tiles (one G = g_pre, two H blocks) go to
// one more wave, with H_NH staged in LDS next to the G rows (it directly precedes them in the workspace).
// Waves: Tk0 (first layer) + (NH-1)*2 (hidden) + 1 (last) <= NW.
// ------------------------------------------------------------------------------------------------
template <int NH, int NW, int NCH, bool AUX>
__global__ __launch_bounds__(NW * 64) void mlp_wgrad4_kernel(WgradArgs a, const float *__restrict__ x, const float *__restrict__ aux,
                                                             int64_t n, const float *__restrict__ stash)
{
    constexpr int H = 64, T = 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = tn::lane_id(), i_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int xs = a.in_dim;
    constexpr int RG = H + NH * H + 4;                    // staged workspace rows: H_NH, G_0 .. G_{NH-1}, g_pre
    constexpr int R0 = (NH - 1) * H;                      // first staged row of a tile
    const int Rt = stash_rows(H, NH, 0);
    constexpr int g_chunks = RG * 8;
    // tk_skip > 0 (round 4): the first layer's x tiles -- k tiles below in_dim / 32 -- are computed by the row-operand kernel
    // from the feature stack's workspace (mlp_wgrad_rows.hip); here only its table columns are left and no x rows are staged
    const int xw = a.tk_skip > 0 ? 0 : a.in_dim;          // staged x columns per sample
    const int x_chunks = (32 * xw) / 4;
    const int aw = AUX ? a.K0_pad - a.in_dim : 0;
    const int buf_floats = RG * RS + 32 * xw + 32 * aw;
    // ownership (wave-uniform): kind 0 = first layer (tk), 1 = hidden layer l (tk), 2 = last layer, -1 = idle
    const int T0 = a.Tk0 - a.tk_skip;
    int kind = -1, l = 0, tk = 0;
    if (wave < T0) { kind = 0; tk = wave + a.tk_skip; }
    else if (wave < T0 + (NH - 1) * T) { const int q = wave - T0; kind = 1; l = 1 + q / T; tk = q - (l - 1) * T; }
    else if (wave == T0 + (NH - 1) * T) { kind = 2; l = NH; }
    f32x16 acc[2];
    float dbacc[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    f32x4 pre[NCH];
    f32x4 preA = {0.f, 0.f, 0.f, 0.f};
    f32x4 hn[4];
    int aidx = 0;
    auto prefetch = [&](int64_t tile, int64_t tile_after) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(stash + tile * (int64_t)Rt * 32 + R0 * 32);
        const int64_t x0 = tile * 32 * (int64_t)a.in_dim;
        const int64_t xlast = n * (int64_t)a.in_dim - 4;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = threadIdx.x + k * NW * 64;
            int64_t e = x0 + 4 * (int64_t)(c - g_chunks);
            e = e < 0 ? 0 : (e > xlast ? xlast : e);
            const f32x4 *px = reinterpret_cast<const f32x4 *>(x + e);
            const f32x4 *pr = src + (c < g_chunks ? c : 0);
            // (xw == 0: x lives in a producer's row view -- the idle chunk slots re-read workspace chunk 0 instead of streaming 20 KB of an x
            // nobody stages per tile; x may then be a placeholder of any size)
            pre[k] = *((c < g_chunks || xw == 0) ? pr : px);
        }
        if constexpr (AUX) {
            const int s_ = (threadIdx.x >> 4) & 31, part = threadIdx.x & 15;
            preA = *reinterpret_cast<const f32x4 *>(aux + (int64_t)aidx * a.aux_stride + (4 * part < aw ? 4 * part : 0));
            int64_t r_ = tile_after * 32 + s_;
            r_ = r_ < n ? r_ : n - 1;
            aidx = a.aux_index ? a.aux_index[r_] : (int)r_;
        }
    };
    float *dummy = lds + 2 * buf_floats;
    auto commit = [&](float *buf) {
        float *ldsX = buf + RG * RS, *ldsA = ldsX + 32 * xw;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = threadIdx.x + k * NW * 64;
            float *dst = c < g_chunks ? buf + (c >> 3) * RS + (c & 7) * 4 : (c < g_chunks + x_chunks ? ldsX + 4 * (c - g_chunks) : dummy);
            *reinterpret_cast<f32x4 *>(dst) = pre[k];
        }
        if constexpr (AUX) {
            const int s_ = threadIdx.x >> 4, part = threadIdx.x & 15;
            float *dst = (threadIdx.x < 512 && 4 * part < aw) ? ldsA + s_ * aw + 4 * part : dummy;
            *reinterpret_cast<f32x4 *>(dst) = preA;
        }
    };
    // the one H operand of a hidden-layer wave; other waves read one (broadcast) line so that the load stays unconditional
    const int hoff = kind == 1 ? ((l - 1) * H + 32 * tk + i_) * 32 + 16 * h_ : 0;
    auto fetch_h = [&](int64_t tile) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(stash + tile * (int64_t)Rt * 32 + hoff);
#pragma unroll
        for (int e = 0; e < 4; ++e) hn[e] = p[e];
    };
    const int64_t last = n_tiles - 1;
    int64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if constexpr (AUX) {
        int64_t r_ = tile * 32 + ((threadIdx.x >> 4) & 31);
        r_ = r_ < n ? r_ : n - 1;
        aidx = a.aux_index ? a.aux_index[r_] : (int)r_;
    }
    {
        const int64_t t1 = tile + gridDim.x, t2 = t1 + gridDim.x;
        prefetch(tile, t1 < last ? t1 : last);
        fetch_h(tile);
        commit(lds);
        __syncthreads();
        prefetch(t1 < last ? t1 : last, t2 < last ? t2 : last);
    }
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        int i = i_, h = h_;
        asm volatile("" : "+v"(i), "+v"(h));
        float *buf = lds + cur * buf_floats;
        float *nbuf = lds + (cur ^ 1) * buf_floats;
        const int64_t t1 = tile + gridDim.x, t2 = t1 + gridDim.x, t3 = t2 + gridDim.x;
        commit(nbuf);
        prefetch(t2 < last ? t2 : last, t3 < last ? t3 : last);
        f32x4 hc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hc[e] = hn[e];
        fetch_h(t1 < last ? t1 : last);
        const float *ldsX = buf + RG * RS, *ldsA = ldsX + 32 * xw;
        if (kind >= 0) {
            // staged rows: [0, H): H_NH   [H + l*H, ...): G_l   [H + NH*H, +4): g_pre
            const f32x4 *g0, *g1;          // G-side operands of the two tiles
            bool gok = true;
            if (kind == 2) {
                g0 = g1 = reinterpret_cast<const f32x4 *>(buf + (H + NH * H + (i < 4 ? i : 0)) * RS + 16 * h);
                gok = i < 4;
            } else {
                g0 = reinterpret_cast<const f32x4 *>(buf + (H + l * H + i) * RS + 16 * h);
                g1 = reinterpret_cast<const f32x4 *>(buf + (H + l * H + 32 + i) * RS + 16 * h);
            }
            const float *cp = nullptr;      // first layer: a column of the sample-major x / aux rows
            int cstride = 0;
            if (kind == 0) {
                const int q = 32 * tk + i;
                if (q < xs) { cp = ldsX + 16 * h * a.in_dim + q; cstride = a.in_dim; }
                else if (AUX && q < a.K0_pad) { cp = ldsA + 16 * h * aw + (q - xs); cstride = aw; }
            }
            const f32x4 *r0 = reinterpret_cast<const f32x4 *>(buf + i * RS + 16 * h);            // last layer: H_NH rows tk = 0
            const f32x4 *r1 = reinterpret_cast<const f32x4 *>(buf + (32 + i) * RS + 16 * h);     //                       tk = 1
            float gs0 = 0.f, gs1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 gv0 = g0[e], gv1 = g1[e];
                if (!gok) { gv0 = f32x4{0.f, 0.f, 0.f, 0.f}; gv1 = gv0; }
                f32x4 av0 = {0.f, 0.f, 0.f, 0.f}, av1;
                if (kind == 0) {
                    if (cp != nullptr) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) av0[u] = cp[(4 * e + u) * cstride];
                    }
                    av1 = av0;
                } else if (kind == 1) { av0 = hc[e]; av1 = av0; }
                else { av0 = r0[e]; av1 = r1[e]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0] = tn::mfma32(gv0[u], av0[u], acc[0]);
                    acc[1] = tn::mfma32(gv1[u], av1[u], acc[1]);
                }
                gs0 += (gv0[0] + gv0[1]) + (gv0[2] + gv0[3]);
                gs1 += (gv1[0] + gv1[1]) + (gv1[2] + gv1[3]);
            }
            if (kind == 2 || tk == 0) { dbacc[0] += gs0; dbacc[1] += gs1; }
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- flush ----
    if (kind < 0) return;
    const int i = i_, h = h_;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        tn::pin16(acc[m]);
        const int tn_ = kind == 2 ? 0 : m, tkm = kind == 2 ? m : tk;
        const int Nl = a.N[l], Kl = a.K[l];
        const int k = 32 * tkm + i;
        const bool kok = k < Kl;
        const int kc = kok ? (l == 0 ? wg_col0(a, k) : k) : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = 32 * tn_ + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (kok && nn < Nl) atomicAdd(&a.gW[l][(int64_t)nn * Kl + kc], acc[m][r]);
        }
        if (tkm == 0) {                                   // bias gradient: row sums of G (once per row block)
            float sum = dbacc[m];
            sum += __shfl_xor(sum, 32, 64);
            const int nn = 32 * tn_ + i;
            if (h == 0 && nn < Nl) atomicAdd(&a.gB[l][nn], sum);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// chunk / LDS budget of the wgrad kernel for a descriptor (host)
struct WgradPlan { int R, Rt, xs, aw, chunks; size_t lds; };
WgradPlan wgrad_plan(int enc, int in_dim, int K0_pad, int H, int NH, bool x_elsewhere = false) {
    WgradPlan p;
    const int extra = extra_rows(enc, in_dim, K0_pad);
    p.R = stash_rows_w(H, NH, extra); p.Rt = stash_rows(H, NH, extra);
    p.xs = x_elsewhere ? 0 : x_slots(enc, in_dim);         // x columns taken from row views: nothing of x is staged here
    p.aw = enc == TN_ENC_AUX_CAT ? K0_pad - in_dim : 0;
    p.chunks = p.R * 8 + (p.xs > 0 ? 8 * in_dim : 0);           // float4 chunks of workspace rows + x rows per tile
    p.lds = ((size_t)p.R * 36 + (p.xs > 0 ? 32 * (size_t)in_dim : 0) + 32 * (size_t)p.aw) * 4;
    return p;
}

bool v2_supported(const tn_mlp_desc *d) { return two_pass_supported(d); }

// phase bit 0: data-gradient chain, bit 1: weight gradient.  pair != nullptr: the chain also runs head `pair->b`.
// first layer of a head without its aux columns: the stashed chain only ever reads W_0's x columns (W_0^T G_0 over the x
// slots), and the 14 KB this saves in LDS are what the fused scatter's per-wave tiles need
MlpArgs compact_first_layer(const MlpArgs &a)
{
    MlpArgs c = a;
    const int H = a.N[0];
    c.K0_pad = (a.in_dim + 7) & ~7;
    int off = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        const int Kp = l == 0 ? c.K0_pad : H;
        const int rows = (l == a.n_layers - 1) ? (a.out_dim <= 4 ? a.out_dim : ((a.out_dim + 31) & ~31)) : H;
        c.stride[l] = Kp + 4;
        c.w_off[l] = off; off += rows * c.stride[l];
        c.b_off[l] = off; off += (rows + 3) & ~3;
    }
    c.lds_floats = off;
    return c;
}

// ... and nothing but that layer (G0B partner)
MlpArgs first_layer_only(const MlpArgs &a)
{
    MlpArgs c = compact_first_layer(a);
    c.n_layers = 1;
    c.lds_floats = c.b_off[0] + ((a.N[0] + 3) & ~3);
    return c;
}

// data gradient of the 2-layer head `b` with the first-layer part of head `a` folded into its grad_x (mlp_chain_kernel, G0B)
int launch_chain_g0b(const MlpArgs &b, const MlpArgs &a, const float *x, const float *gy_b, int64_t n, float *gx, float *stash_b,
                     const float *g0_rows_a, int64_t g0_stride_a, hipStream_t s)
{
    constexpr int WPS = 8;                     // (133 KB of LDS at 256 columns: one workgroup per CU anyway; 10 waves capped at 168 VGPRs spill 29)
    PairArgs pr;
    pr.b = first_layer_only(a); pr.gy = nullptr; pr.stash = const_cast<float *>(g0_rows_a); pr.g0_stride = g0_stride_a;
    const size_t lds_bytes = ((size_t)b.lds_floats + (size_t)pr.b.lds_floats) * 4;
    if (lds_bytes > (size_t)LDS_LIMIT_BYTES) return tn::fail(TN_E_CONFIG, "mlp_bwd: both first layers do not fit LDS");
    auto kern = mlp_chain_kernel<64, 1, WPS, true, false, false, false, true>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(LDS_LIMIT_BYTES / lds_bytes, 2048 / (WPS * 64)));
    const int64_t blocks = std::min<int64_t>((n_tiles + WPS - 1) / WPS, 256 * per_cu);
    kern<<<dim3((unsigned)blocks), dim3(WPS * 64), lds_bytes, s>>>(b, x, nullptr, gy_b, n, gx, stash_b, pr, KpBwd());
    return tn::check_launch("mlp_chain_kernel(g0b)");
}

template <int H, int NH>
int launch_v2(const MlpArgs &a, const tn_mlp_desc *d, const float *x, const float *aux, const float *gy, int64_t n,
              float *const *gw, float *const *gb, float *gx, float *stash, bool stashed, hipStream_t s,
              const PairArgs *pair = nullptr, int phase = 3, const KpBwd *kpb = nullptr)
{
    const int64_t n_tiles = (n + 31) / 32;
    if ((phase & 1) && kpb) {                 // paired, stashed chain with the plane scatter inside
        if constexpr (H == 64 && NH == 4) {
            if (!(pair && stashed && a.enc == TN_ENC_AUX_CAT && a.in_dim == 96)) return tn::fail(TN_E_CONFIG, "mlp_bwd: fused scatter needs the paired K-Planes heads");
            constexpr int WPK = 8;
            const MlpArgs ac = compact_first_layer(a);
            PairArgs pr = *pair;
            pr.b = compact_first_layer(pair->b);
            const size_t lds_bytes = ((size_t)ac.lds_floats + (size_t)pr.b.lds_floats + (size_t)WPK * tn::KP_WAVE_LDS) * 4;
            if (lds_bytes > (size_t)LDS_LIMIT_BYTES) return tn::fail(TN_E_CONFIG, "mlp_bwd: weights + scatter tiles do not fit LDS");
            auto kern = mlp_chain_kernel<H, NH, WPK, true, false, true, true>;
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
            const int64_t blocks = std::min<int64_t>((n_tiles + WPK - 1) / WPK, 256);
            kern<<<dim3((unsigned)blocks), dim3(WPK * 64), lds_bytes, s>>>(ac, x, aux, gy, n, gx, stash, pr, *kpb);
            if (int rc = tn::check_launch("mlp_chain_kernel(kplanes)")) return rc;
        } else return tn::fail(TN_E_CONFIG, "mlp_bwd: fused scatter is built for the 5-layer colour head");
    } else if (phase & 1) {
    size_t lds_bytes = (size_t)a.lds_floats * 4;
    constexpr int WPB = 8;
    constexpr int WPS = NH == 1 ? 10 : 16;   // stashed chain: no forward -> ~100 live registers: 4 waves per SIMD, or 2 x 10
                                             // waves per CU for the single-hidden-layer head (25 KB of LDS)
    constexpr int WPP = 8;                   // pair: both heads' G_0 live at grad_x: 207 VGPRs (9-12 waves are capped at 168 and spill 38)
    int wpb = stashed ? WPS : WPB;
    auto kern = stashed ? mlp_chain_kernel<H, NH, WPS, true> : mlp_chain_kernel<H, NH, WPB, false>;
    if (stashed && a.accum_gx && (gx != nullptr || a.gx_rows != nullptr) && a.enc != TN_ENC_POSENC && (a.in_dim & 31) == 0) kern = mlp_chain_kernel<H, NH, WPS, true, true>;
    PairArgs pr;
    pr.gy = nullptr; pr.stash = nullptr;
    if (pair) {
        pr = *pair;
        if constexpr (NH == 4) kern = mlp_chain_kernel<H, NH, WPP, true, false, true>;
        else return tn::fail(TN_E_CONFIG, "mlp_bwd: the paired chain is built for the 5-layer colour head");
        wpb = WPP;
        lds_bytes += (size_t)pr.b.lds_floats * 4;
    } else pr.b = a;
    if (lds_bytes > (size_t)LDS_LIMIT_BYTES) return tn::fail(TN_E_CONFIG, "mlp_bwd: weights do not fit LDS");
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(LDS_LIMIT_BYTES / lds_bytes, 2048 / (wpb * 64)));
    const int64_t blocks = std::min<int64_t>((n_tiles + wpb - 1) / wpb, 256 * per_cu);
    kern<<<dim3((unsigned)blocks), dim3(wpb * 64), lds_bytes, s>>>(a, x, aux, gy, n, gx, stash, pr, KpBwd());
    if (int rc = tn::check_launch("mlp_chain_kernel")) return rc;
    }
    if (!(phase & 2)) return TN_OK;

    WgradArgs w;
    constexpr int T = H / 32;
    w.n_layers = a.n_layers; w.in_dim = a.in_dim; w.K0 = a.K0; w.K0_pad = a.K0_pad; w.enc = a.enc; w.n_freqs = a.n_freqs;
    w.out_dim = a.out_dim;
    w.Tk0 = (a.K0_pad + 31) / 32;
    w.tk_skip = 0;
    for (int l = 0; l < a.n_layers; ++l) { w.gW[l] = gw[l]; w.gB[l] = gb[l]; w.K[l] = a.K[l]; w.N[l] = a.N[l]; }
    w.aux_index = a.aux_index; w.aux_stride = a.aux_stride;
    // x as [feature][32-sample] rows (the workspace of the wide stack that produced it): the first layer's x columns -- all of
    // its k tiles below in_dim / 32 and its bias gradient -- go to the layer-kernel form (mlp_wgrad_rows.hip); what is left
    // here are the encoding columns, the hidden layers and the output layer
    const bool x_rows = a.x_rows != nullptr && stashed && H == 64 && (a.in_dim & 31) == 0 && x_slots(a.enc, a.in_dim) == a.in_dim &&
                        (a.in_dim == 128 || a.in_dim == 256);
    if (x_rows) {
        const int extra_r = extra_rows(a.enc, a.in_dim, a.K0_pad);
        const int col0 = (a.enc == TN_ENC_DIR_CAT || a.enc == TN_ENC_AUX_CAT) ? a.K0 - a.in_dim : 0;      // torch order [PE(d), d, x]
        if (a.x_wgrad_done) {}              // (tn_mlp_bwd_pair: both heads' x columns went out in one launch, tn_mlp_wgrad_rows2)
        else if (int rc = tn_mlp_wgrad_rows(stash + (int64_t)NH * H * 32, (int64_t)stash_rows(H, NH, extra_r) * 32, H, a.x_rows, a.x_rows_stride,
                                       a.in_dim, gw[0], a.K0, col0, a.in_dim, gb[0], n, s)) return rc;
        w.tk_skip = a.in_dim / 32;
    }
    w.total_tiles = T * (w.Tk0 - w.tk_skip) + (NH - 1) * T * T + T;
    const WgradPlan wp = wgrad_plan(a.enc, a.in_dim, a.K0_pad, H, NH, x_rows);
    const size_t wlds = wp.lds;
    if (wlds > (size_t)LDS_LIMIT_BYTES) return tn::fail(TN_E_CONFIG, "mlp_bwd: workspace tile does not fit LDS");
    if (wp.xs > 0 && (a.in_dim & 3)) return tn::fail(TN_E_CONFIG, "mlp_bwd: in_dim must be a multiple of 4");
    const int chunks = wp.chunks;
    const int64_t wblocks = std::min<int64_t>(n_tiles, 256 * (wlds * 2 <= (size_t)LDS_LIMIT_BYTES ? 2 : 1));
    const int extra_rows_ = extra_rows(a.enc, a.in_dim, a.K0_pad);
    if (extra_rows_ == 0 && (wp.xs > 0 || (x_rows && a.enc == TN_ENC_AUX_CAT)) && NH >= 2) {   // double-buffered, paired form (measured slower for the 2-layer sigma head)
        if constexpr (H == 64 && NH >= 2) {                // paired ownership: one wave per shared operand
            constexpr int RG4 = H + NH * H + 4;
            const int xw4 = x_rows ? 0 : a.in_dim;         // (x tiles from row views: nothing of x is staged)
            const size_t lds4 = 2 * ((size_t)RG4 * RS + 32 * (size_t)xw4 + 32 * (size_t)wp.aw) * 4 + 32 * 16;
            const int chunks4 = RG4 * 8 + 8 * xw4;
            if (lds4 <= (size_t)LDS_LIMIT_BYTES && w.Tk0 - w.tk_skip + (NH - 1) * 2 + 1 <= 12 && chunks4 <= 5 * 768) {
                auto wk = a.enc == TN_ENC_AUX_CAT ? mlp_wgrad4_kernel<NH, 12, 5, true> : mlp_wgrad4_kernel<NH, 12, 5, false>;
                hipError_t we = hipFuncSetAttribute((const void *)wk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                if (we != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds4, hipGetErrorString(we)); return (int)we; }
                wk<<<dim3((unsigned)std::min<int64_t>(n_tiles, 256)), dim3(12 * 64), lds4, s>>>(w, x, aux, n, stash);
                return tn::check_launch("mlp_wgrad4_kernel");
            }
        }
    }
#define TN_WGRAD_X(MAXS_, NW_, NCH_, AUX_)                                                                                 \
    do {                                                                                                                    \
        auto wk = mlp_wgrad_kernel<H, NH, MAXS_, NW_, NCH_, AUX_>;                                                         \
        hipError_t we = hipFuncSetAttribute((const void *)wk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds);      \
        if (we != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", wlds, hipGetErrorString(we)); return (int)we; } \
        wk<<<dim3((unsigned)wblocks), dim3(NW_ * 64), wlds, s>>>(w, x, aux, n, stash);                                      \
    } while (0)
#define TN_WGRAD(MAXS_, NW_, NCH_) TN_WGRAD_X(MAXS_, NW_, NCH_, false)
    if (a.enc == TN_ENC_AUX_CAT) {
        // per-ray table columns + hidden + output layers of a head whose x columns went to the row-operand kernel above (the
        // colour head behind a 256- / 128-wide stack): the general tiling with the table rows staged per tile
        if constexpr (NH == 4) {
            if (x_rows && w.total_tiles <= 24 && chunks <= 7 * 768 && wp.aw <= 64) {
                TN_WGRAD_X(2, 12, 7, true);
                return tn::check_launch("mlp_wgrad_kernel(aux)");
            }
        }
        return tn::fail(TN_E_CONFIG, "mlp_bwd: TN_ENC_AUX_CAT outside the paired weight-gradient tiling");
    }
    if (w.total_tiles <= 8 && chunks <= 4 * 512) TN_WGRAD(1, 8, 4);
    else if (w.total_tiles <= 16 && chunks <= 4 * 1024) TN_WGRAD(1, 16, 4);
    else if (w.total_tiles <= 24 && chunks <= 7 * 768) TN_WGRAD(2, 12, 7);     // 12 waves x 2 tiles: 170-VGPR budget, no spills
    else if (w.total_tiles <= 32 && chunks <= 6 * 1024) TN_WGRAD(2, 16, 6);
    else if (w.total_tiles <= 48 && chunks <= 10 * 1024) TN_WGRAD(3, 16, 10);
    else return tn::fail(TN_E_CONFIG, "mlp_bwd: configuration outside the wgrad tiling");
#undef TN_WGRAD
#undef TN_WGRAD_X
    return tn::check_launch("mlp_wgrad_kernel");
}

template <int H>
int launch_v2_h(const MlpArgs &a, const tn_mlp_desc *d, const float *x, const float *aux, const float *gy, int64_t n,
                float *const *gw, float *const *gb, float *gx, float *stash, bool stashed, hipStream_t s,
                const PairArgs *pair = nullptr, int phase = 3, const KpBwd *kpb = nullptr)
{
    // the reference's two decoder shapes (two_pass_supported): one hidden layer (sigma) or four (colour)
    if (a.n_layers == 2) return launch_v2<H, 1>(a, d, x, aux, gy, n, gw, gb, gx, stash, stashed, s, pair, phase, kpb);
    return launch_v2<H, 4>(a, d, x, aux, gy, n, gw, gb, gx, stash, stashed, s, pair, phase, kpb);
}

}  // namespace

extern "C" int64_t tn_mlp_bwd_workspace_bytes(const tn_mlp_desc *desc, int64_t n)
{
    if (n <= 0 || !desc) return 0;
    if (!v2_supported(desc)) {
        if (desc->encoding == TN_ENC_AUX_CAT) return 0;
        return tn_mlp_bwd_layers_workspace_bytes(desc, n);
    }
    const int H = desc->dims[1], NH = desc->n_layers - 1;
    const int extra = extra_rows(desc->encoding, desc->in_dim, (desc->dims[0] + 7) & ~7);
    return ((n + 31) / 32) * (int64_t)stash_rows(H, NH, extra) * 32 * (int64_t)sizeof(float);
}

extern "C" int tn_mlp_bwd(const tn_mlp_desc *desc, const float *x, const float *aux, const float *grad_y, int64_t n,
                          float *const *grad_weights, float *const *grad_biases, float *grad_x, void *workspace,
                          int64_t workspace_bytes, void *stream)
{
    TN_REQUIRE(desc, TN_E_NULL, "tn_mlp_bwd: null descriptor");
    const int64_t need = tn_mlp_bwd_workspace_bytes(desc, n);
    const bool stashed = (desc->flags & TN_MLP_STASHED) != 0;
    const bool v2 = v2_supported(desc);
    if (stashed || desc->encoding == TN_ENC_AUX_CAT) {
        TN_REQUIRE(v2 || (stashed && desc->encoding != TN_ENC_AUX_CAT && need > 0), TN_E_CONFIG,
                   "tn_mlp_bwd: TN_ENC_AUX_CAT needs a configuration of the two-pass form, TN_MLP_STASHED one with a workspace");
        if (n == 0) return TN_OK;
        TN_REQUIRE(workspace && workspace_bytes >= need, TN_E_NULL, "tn_mlp_bwd: TN_MLP_STASHED / TN_ENC_AUX_CAT need the workspace");
    }
    if (!v2) {                                                 // wide / deep / odd-shaped stack (or its stash): layer-by-layer form
        if (n == 0) return TN_OK;
        TN_REQUIRE(need > 0, TN_E_CONFIG, "tn_mlp_bwd: unsupported layer configuration");
        TN_REQUIRE(workspace && workspace_bytes >= need, TN_E_NULL, "tn_mlp_bwd: this configuration needs the workspace");
        TN_REQUIRE(((uintptr_t)workspace & 15) == 0, TN_E_ALIGN, "tn_mlp_bwd: workspace must be 16-byte aligned");
        return tn_mlp_bwd_layers(desc, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, (float *)workspace, stream);
    }
    if (n == 0) return TN_OK;
    TN_REQUIRE(workspace != nullptr && workspace_bytes >= need, TN_E_NULL, "tn_mlp_bwd: workspace missing or too small (tn_mlp_bwd_workspace_bytes)");
    MlpArgs a;
    int H = 0;
    if (int rc = plan(desc, a, H)) return rc;
    TN_REQUIRE(x && grad_y && grad_weights && grad_biases, TN_E_NULL, "tn_mlp_bwd: null pointer");
    TN_REQUIRE((a.enc != TN_ENC_DIR_CAT && a.enc != TN_ENC_AUX_CAT) || aux, TN_E_NULL, "tn_mlp_bwd: dir_cat / aux_cat need aux");
    TN_REQUIRE(((uintptr_t)workspace & 15) == 0, TN_E_ALIGN, "tn_mlp_bwd: workspace must be 16-byte aligned");
    TN_REQUIRE(a.enc != TN_ENC_AUX_CAT || ((uintptr_t)aux & 15) == 0, TN_E_ALIGN, "tn_mlp_bwd: aux table must be 16-byte aligned");
    for (int l = 0; l < a.n_layers; ++l)
        TN_REQUIRE(grad_weights[l] && grad_biases[l], TN_E_NULL, "tn_mlp_bwd: null gradient pointer");
    hipStream_t s = (hipStream_t)stream;
    return launch_v2_h<64>(a, desc, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, (float *)workspace, stashed, s);
}

static int bwd_pair_common(const tn_mlp_desc *desc, const tn_mlp_desc *partner, const float *x, const float *aux,
                           const float *grad_y, const float *partner_grad_y, int64_t n, float *const *grad_weights,
                           float *const *grad_biases, float *const *partner_grad_weights, float *const *partner_grad_biases,
                           float *grad_x, void *workspace, int64_t workspace_bytes, void *partner_workspace,
                           int64_t partner_workspace_bytes, void *stream, const KpBwd *kpb)
{
    TN_REQUIRE(desc && partner, TN_E_NULL, "tn_mlp_bwd_pair: null descriptor");
    TN_REQUIRE((desc->flags & TN_MLP_STASHED) && (partner->flags & TN_MLP_STASHED), TN_E_CONFIG,
               "tn_mlp_bwd_pair: both heads need TN_MLP_STASHED workspaces (tn_mlp_fwd_stash)");
    TN_REQUIRE(v2_supported(desc) && v2_supported(partner), TN_E_CONFIG, "tn_mlp_bwd_pair: outside the two-pass form's configurations");
    TN_REQUIRE(desc->n_layers == 5 && partner->n_layers == 2 && partner->encoding == TN_ENC_NONE && partner->in_dim == desc->in_dim &&
                   partner->dims[1] == desc->dims[1] && desc->dims[1] == 64 && (desc->in_dim & 31) == 0 && desc->encoding != TN_ENC_POSENC,
               TN_E_CONFIG, "tn_mlp_bwd_pair: a 5-layer head and a 2-layer partner on the same x (width 64, in_dim % 32 == 0)");
    if (n == 0) return TN_OK;
    const int64_t need_a = tn_mlp_bwd_workspace_bytes(desc, n), need_b = tn_mlp_bwd_workspace_bytes(partner, n);
    TN_REQUIRE(workspace && workspace_bytes >= need_a && partner_workspace && partner_workspace_bytes >= need_b, TN_E_NULL,
               "tn_mlp_bwd_pair: workspace missing or too small");
    TN_REQUIRE(x && grad_y && partner_grad_y && (grad_x || kpb || desc->grad_x_rows) && grad_weights && grad_biases && partner_grad_weights && partner_grad_biases,
               TN_E_NULL, "tn_mlp_bwd_pair: null pointer");
    MlpArgs a, b;
    int H = 0, Hb = 0;
    if (int rc = plan(desc, a, H)) return rc;
    if (int rc = plan(partner, b, Hb)) return rc;
    TN_REQUIRE((a.enc != TN_ENC_DIR_CAT && a.enc != TN_ENC_AUX_CAT) || aux, TN_E_NULL, "tn_mlp_bwd_pair: dir_cat / aux_cat need aux");
    for (int l = 0; l < a.n_layers; ++l) TN_REQUIRE(grad_weights[l] && grad_biases[l], TN_E_NULL, "tn_mlp_bwd_pair: null gradient pointer");
    for (int l = 0; l < b.n_layers; ++l)
        TN_REQUIRE(partner_grad_weights[l] && partner_grad_biases[l], TN_E_NULL, "tn_mlp_bwd_pair: null gradient pointer");
    hipStream_t s = (hipStream_t)stream;
    PairArgs pr;
    pr.b = b; pr.gy = partner_grad_y; pr.stash = (float *)partner_workspace;
    a.accum_gx = 0;
    const bool chain = !(desc->flags & TN_MLP_WGRAD_ONLY), wgrad = !(desc->flags & TN_MLP_CHAIN_ONLY);
    TN_REQUIRE(chain || wgrad, TN_E_CONFIG, "tn_mlp_bwd_pair: TN_MLP_CHAIN_ONLY and TN_MLP_WGRAD_ONLY exclude each other");
    const bool lean = (desc->flags & TN_MLP_LEAN) != 0;
    TN_REQUIRE(lean == ((partner->flags & TN_MLP_LEAN) != 0), TN_E_CONFIG, "tn_mlp_bwd_pair: TN_MLP_LEAN must be set on both heads or on neither");
    if (lean) {
        // the workspaces hold no H rows (the forward ran with TN_MLP_LEAN): the chain half needs none; the weight-gradient half
        // rebuilds them from the feature rows (mlp_wgrad_rc.hip)
        if (chain)
            if (int rc = launch_v2_h<64>(a, desc, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, (float *)workspace, true, s, &pr, 1, kpb)) return rc;
        if (!wgrad) return TN_OK;
        return tn_mlp_wgrad_lean_pair(desc, partner, x, aux, n, grad_weights, grad_biases, partner_grad_weights, partner_grad_biases,
                                      (const float *)workspace, (const float *)partner_workspace, stream);
    }
    // Heads behind a wide stack (x and grad_x as rows of its workspace): the colour head's chain runs WITHOUT its grad_x part (no first
    // layer in LDS), and the sigma head's launch adds W_0c^T G_0c -- G_0c read back as rows -- to its own W_0s^T G_0s and writes the sum
    // once (mlp_chain_kernel, G0B).  Before: the colour launch wrote grad_x, the sigma launch read it, added and wrote it again
    // (0.67 + 0.54 ms per 2^20 samples at 256 columns; one paired launch does not fit LDS there and, as two column windows, was slower).
    const bool rows_split = kpb == nullptr && a.gx_rows != nullptr && b.gx_rows == a.gx_rows && a.x_rows != nullptr && grad_x == nullptr &&
                            a.enc == TN_ENC_AUX_CAT && (a.in_dim & 31) == 0 &&
                            ((size_t)b.lds_floats + (size_t)first_layer_only(a).lds_floats) * 4 <= (size_t)LDS_LIMIT_BYTES;
    const bool fits = kpb != nullptr || ((size_t)a.lds_floats + (size_t)b.lds_floats) * 4 <= (size_t)LDS_LIMIT_BYTES;
    if (chain) {
        if (rows_split) {
            MlpArgs a0 = a;
            a0.gx_rows = nullptr;                 // (no grad_x output: the launch stops at G_0)
            if (int rc = launch_v2_h<64>(a0, desc, x, aux, grad_y, n, grad_weights, grad_biases, nullptr, (float *)workspace, true, s, nullptr, 1)) return rc;
            if (a.f2 && b.f2 && (a.in_dim == 128 || a.in_dim == 256)) {
                // f16x2: the sigma head's chain stops at G_0 as well, and grad_x = W_0c[:, x]^T G_0c + W_0s^T G_0s is a launch of its own on
                // the fp16 matrix cores (heads_dx.hip): 6 k instead of 33 k matrix-pipe cycles per tile, bound by the rows it writes
                MlpArgs b0 = b;
                b0.gx_rows = nullptr;
                if (int rc = launch_v2_h<64>(b0, partner, x, nullptr, partner_grad_y, n, partner_grad_weights, partner_grad_biases, nullptr,
                                             (float *)partner_workspace, true, s, nullptr, 1)) return rc;
                const int col0 = a.K0 - a.in_dim;                    // torch order [PE(d), d, x]
                if (int rc = tn_heads_dx_rows(desc->weights[0], a.K0, col0, partner->weights[0], b.K0, 0,
                                              (const float *)workspace + (int64_t)4 * 64 * 32, (int64_t)stash_rows(64, 4, 0) * 32,
                                              (const float *)partner_workspace + (int64_t)1 * 64 * 32, (int64_t)stash_rows(64, 1, 0) * 32, a.in_dim,
                                              a.gx_rows, a.gx_rows_stride, a.gx_mask_rows, a.gx_mask_stride, n, s)) return rc;
            } else if (int rc = launch_chain_g0b(b, a, x, partner_grad_y, n, nullptr, (float *)partner_workspace,
                                                 (const float *)workspace + (int64_t)4 * 64 * 32, (int64_t)stash_rows(64, 4, 0) * 32, s)) return rc;
        } else if (fits) {
            if (int rc = launch_v2_h<64>(a, desc, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, (float *)workspace, true, s, &pr, 1, kpb)) return rc;
        } else {
            if (int rc = launch_v2_h<64>(a, desc, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, (float *)workspace, true, s, nullptr, 1)) return rc;
            MlpArgs b1 = b;
            b1.accum_gx = 1;
            if (int rc = launch_v2_h<64>(b1, partner, x, nullptr, partner_grad_y, n, partner_grad_weights, partner_grad_biases, grad_x,
                                         (float *)partner_workspace, true, s, nullptr, 1)) return rc;
        }
    }
    if (!wgrad) return TN_OK;
    // first layers over the x columns: with x as rows of the producer's workspace both heads' tiles go out in ONE launch (x is 80 % of
    // what a single-head launch reads at 256 columns)
    MlpArgs aw = a, bw = b;
    if (a.x_rows != nullptr && b.x_rows == a.x_rows && a.x_rows_stride == b.x_rows_stride && (a.in_dim == 128 || a.in_dim == 256) &&
        x_slots(a.enc, a.in_dim) == a.in_dim && extra_rows(a.enc, a.in_dim, a.K0_pad) == 0) {
        const int col0 = (a.enc == TN_ENC_DIR_CAT || a.enc == TN_ENC_AUX_CAT) ? a.K0 - a.in_dim : 0;      // torch order [PE(d), d, x]
        if (int rc = tn_mlp_wgrad_rows2((const float *)workspace + (int64_t)4 * 64 * 32, (int64_t)stash_rows(64, 4, 0) * 32,
                                        (const float *)partner_workspace + (int64_t)1 * 64 * 32, (int64_t)stash_rows(64, 1, 0) * 32,
                                        a.x_rows, a.x_rows_stride, a.in_dim, grad_weights[0], a.K0, col0, grad_biases[0],
                                        partner_grad_weights[0], b.K0, 0, partner_grad_biases[0], n, s)) return rc;
        aw.x_wgrad_done = 1; bw.x_wgrad_done = 1;
    }
    if (int rc = launch_v2_h<64>(aw, desc, x, aux, grad_y, n, grad_weights, grad_biases, grad_x, (float *)workspace, true, s, &pr, 2, kpb)) return rc;
    return launch_v2_h<64>(bw, partner, x, nullptr, partner_grad_y, n, partner_grad_weights, partner_grad_biases, nullptr,
                           (float *)partner_workspace, true, s, nullptr, 2);
}

extern "C" int tn_mlp_bwd_pair(const tn_mlp_desc *desc, const tn_mlp_desc *partner, const float *x, const float *aux,
                               const float *grad_y, const float *partner_grad_y, int64_t n, float *const *grad_weights,
                               float *const *grad_biases, float *const *partner_grad_weights, float *const *partner_grad_biases,
                               float *grad_x, void *workspace, int64_t workspace_bytes, void *partner_workspace,
                               int64_t partner_workspace_bytes, void *stream)
{
    return bwd_pair_common(desc, partner, x, aux, grad_y, partner_grad_y, n, grad_weights, grad_biases, partner_grad_weights,
                           partner_grad_biases, grad_x, workspace, workspace_bytes, partner_workspace, partner_workspace_bytes, stream,
                           nullptr);
}

extern "C" int tn_kplanes_mlp_bwd_pair(const tn_kplanes_desc *kd, const float *coords, int64_t coord_stride,
                                       float *const (*grad_planes)[3], const tn_mlp_desc *desc, const tn_mlp_desc *partner,
                                       const float *feat, const float *aux, const float *grad_y, const float *partner_grad_y, int64_t n,
                                       float *const *grad_weights, float *const *grad_biases, float *const *partner_grad_weights,
                                       float *const *partner_grad_biases, float *grad_feat, void *workspace, int64_t workspace_bytes,
                                       void *partner_workspace, int64_t partner_workspace_bytes, void *stream)
{
    TN_REQUIRE(kd && desc && partner && grad_planes, TN_E_NULL, "tn_kplanes_mlp_bwd_pair: null descriptor");
    TN_REQUIRE(kd->n_scales == 3 && kd->channels == 32 && desc->in_dim == 96 && partner->in_dim == 96 && desc->encoding == TN_ENC_AUX_CAT,
               TN_E_CONFIG, "tn_kplanes_mlp_bwd_pair: 3 scales x 32 channels feeding the aux-table colour head and the sigma head (run.py:136-139)");
    TN_REQUIRE(coord_stride >= 3, TN_E_SIZE, "tn_kplanes_mlp_bwd_pair: bad coordinate stride");
    if (n == 0) return TN_OK;
    TN_REQUIRE(coords, TN_E_NULL, "tn_kplanes_mlp_bwd_pair: null coordinates");
    KpBwd kp;
    for (int s = 0; s < 3; ++s) {
        TN_REQUIRE(kd->height[s] > 0 && kd->width[s] > 0 && (int64_t)kd->height[s] * kd->width[s] * 32 < (1ll << 30), TN_E_SIZE,
                   "tn_kplanes_mlp_bwd_pair: bad plane resolution");
        kp.H[s] = kd->height[s]; kp.W[s] = kd->width[s];
        for (int p = 0; p < 3; ++p) {
            TN_REQUIRE(kd->planes[s][p], TN_E_NULL, "tn_kplanes_mlp_bwd_pair: null plane pointer");
            TN_REQUIRE(((uintptr_t)kd->planes[s][p] & 15) == 0, TN_E_ALIGN, "tn_kplanes_mlp_bwd_pair: planes must be 16-byte aligned");
            kp.planes[s][p] = kd->planes[s][p];
            kp.grads[s][p] = grad_planes[s][p];           // NULL: no gradient for that plane
        }
    }
    kp.coords = coords; kp.coord_stride = coord_stride;
    return bwd_pair_common(desc, partner, feat, aux, grad_y, partner_grad_y, n, grad_weights, grad_biases, partner_grad_weights,
                           partner_grad_biases, grad_feat, workspace, workspace_bytes, partner_workspace, partner_workspace_bytes, stream,
                           &kp);
}
