// Hidden layers of the wide stacks (Vanilla 256 x 10, reference models.py:59-68, run.py:131; Cobafa 128 x 6) -- forward, data
// gradient and weight gradient -- on the fp16 matrix cores with TWO-term operand splits and power-of-two scaling ("f16x2",
// TN_MLP_F16X2).
//
// bf16x3 (mlp_b3_layers.hip) represents every fp32 operand exactly as three bf16 terms and pays six MFMAs per 32 x 32 x 16 block.
// An fp16 term carries 11 significand bits instead of 8: x s = hi + lo + r with hi = fp16(x s), lo = fp16(x s - hi) (both
// round-to-nearest; x s - hi is exact in fp32) leaves |r| <= 2^-23 |x s| -- 22 of fp32's 24 bits -- and THREE products (lo hi,
// hi lo, hi hi; lo lo is below 2^-22 of the product) per block: half the matrix time, 8 instead of 11 VALU instructions per
// converted value pair, two LDS planes instead of three.  What fp16 lacks is range, hence the scales s (powers of two: exact):
//   * B operand (activations / incoming gradients [feature][32 samples]): one scale PER SAMPLE, 2^(9 - e) with e the exponent of the
//     largest |value| of that sample's column -- no overflow whatever the data, and a column's small entries keep an absolute
//     error of 2^-35 of its largest one.  The column maxima of tile t + 2 are taken by the wave that staged the rows (its 32
//     rows), published in LDS behind the tile barrier the kernel has anyway, and combined by the waves that convert it a tile
//     later -- no extra barrier; the second staging buffer this needs is what the third LDS plane used to occupy.
//   * A operand (weights, in registers for the whole launch): one scale per layer from the largest |weight| (prologue).
//   * the accumulators are unscaled in the epilogue (D x 2^-(e_W + e_j) [+ bias]): one multiply-add per value, exact.
// Measured against an fp64 evaluation (scratch experiment in DESIGN 4.2): the ten-layer forward is as far from it as the fp32
// MFMA's (5e-7 of the largest output; bf16x3: 2e-7).  The weight gradient's reduction runs over samples, so no per-tile scale
// can leave its accumulators: it takes one scale per operand and launch (see wgrad_f2_kernel).
//
// Structure, geometry and instruction scheduling follow mlp_b3_layers.hip (a wave owns one 32-row block of the output with its
// weights in registers, eight waves per stream at H = 256 / four at H = 128, LDS-direct staging, conversion micro-steps pinned
// behind the MFMAs of the first half of the k loop).
#include "mlp_layers.h"
#include "b3_device.h"
#include <algorithm>
#include <type_traits>

namespace {

using namespace tn::layers;
using namespace tn::mlp;
using tn::b3::u32x4;
using tn::f32x16;
using tn::f32x4;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct Op2 { u32x4 hi, lo; };            // 8 values as two packed-fp16 operands

// TN_F2_NT: non-temporal hints on the row streams of these kernels (bit 0: LDS-direct loads of the forward / data gradient,
// bit 1: row stores).  In the slab layout a layer launch reads one dense 1 GB array and writes another; nothing is re-read from a
// cache, and the same copy runs at 5.9 instead of 5.5 TB/s with the hints (scripts/microbench/row_copy.hip).
// Ablation switches of DESIGN 4.2 (round 5; scripts/build_dev_lib.sh builds a second library with one of them set): TN_ABL_NOMFMA /
// TN_ABL_NOSTORE / TN_ABL_NOMAX / TN_ABL_NOEPI take the matrix products / row stores / column maxima / epilogue arithmetic out of the
// forward kernel, TN_F2_SB=(void)0 the scheduling barriers out of the k loop.  None is set in the shipped build.
#ifndef TN_F2_SB
#define TN_F2_SB __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef TN_F2_NT
#define TN_F2_NT 3
#endif
__device__ __forceinline__ void glds16(const float *src, float *dst) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, (TN_F2_NT & 1) ? 2 : 0);      // aux bit 1 = nt
}
__device__ __forceinline__ f32x16 mfma_h(const u32x4 &a, const u32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// two (already scaled) fp32 values -> packed fp16 pairs of their two terms (low half = first value)
__device__ __forceinline__ void split2h(float a, float b, unsigned &hi, unsigned &lo) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h[0], rb = b - (float)h[1];            // exact
    const f16x2 l = {(_Float16)ra, (_Float16)rb};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// 2^(9 - e) and 2^(e - 9) for a column / layer maximum m >= 0 (e = its exponent); m = 0 or denormal: the largest scale (0 s = 0)
__device__ __forceinline__ void pow2_scales(float m, float &s, float &inv) {
    int e = (int)(__float_as_uint(m) >> 23);                    // biased
    int se = 263 - e;                                            // biased exponent of 2^(9 - (e - 127))
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);           // 2^-(se - 127); se = 254 -> 0: the true values are below 2^-118
}

template <int H> struct F2Geom {
    static constexpr int T = H / 32;
    static constexpr int WPS = T;                   // waves per tile stream: one 32-row block each
    static constexpr int STREAMS = H == 256 ? 1 : 2;
    static constexpr int THREADS = STREAMS * WPS * 64;
    static constexpr int KS = H / 16;               // k steps
    static constexpr int CONV = KS / 2;             // steps that carry the conversion of the next tile
    static constexpr int SB = H + 8;                // fp16 elements per LDS tile row
    static constexpr int PLANE = 32 * SB;
    static constexpr int TILE_B = 2 * PLANE * 2;    // bytes per tile buffer (two planes)
    static constexpr int STAGE_F = 32 * H;          // floats per staging buffer ([row][32 samples])
    static constexpr int PM_F = 32 * WPS;           // floats per column-maximum slot ([sample][wave])
    static constexpr int STREAM_B = 2 * TILE_B + 2 * STAGE_F * 4 + 2 * PM_F * 4;
    static constexpr size_t lds_bytes = (size_t)STREAMS * STREAM_B + H * 4 + 64;
};

template <int SB>
__device__ __forceinline__ Op2 read_b2(const unsigned short *tile, int j, int h, int s) {
    constexpr int PLANE = 32 * SB;
    const unsigned short *p = tile + j * SB + 16 * s + 8 * h;
    Op2 o;
    o.hi = *reinterpret_cast<const u32x4 *>(p);
    o.lo = *reinterpret_cast<const u32x4 *>(p + PLANE);
    return o;
}

template <int NROWS>
__device__ __forceinline__ void request_rows(const float *rows, int r0, float *stage, int lane) {
    const float *src = rows + r0 * 32 + 4 * lane;            // 16 B per lane: one instruction = 8 rows x 128 B
#pragma unroll
    for (int e = 0; e < NROWS / 8; ++e) glds16(src + e * 256, stage + e * 256);
}

// A operands of one 32-row block, all KS steps, scaled by `s` (a power of two)
template <int KS>
__device__ __forceinline__ void split_weights(const float (&v)[KS][8], float s, Op2 (&A)[KS]) {
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned a, b;
            split2h(v[st][2 * p] * s, v[st][2 * p + 1] * s, a, b);
            A[st].hi[p] = a; A[st].lo[p] = b;
        }
}

template <int H>
struct Stream2 {
    using G = F2Geom<H>;
    static constexpr int KS = G::KS, SB = G::SB, PLANE = G::PLANE;
    int lane, j, h, wave, stream, wib;
    unsigned short *tiles;
    float *stage0;                              // this stream's two staging buffers
    float *pm;                                  // this stream's two column-maximum slots
    float *red;                                 // workgroup scratch (prologue: layer maximum)
    float seen;                                 // largest |value| this lane has staged so far (-> the weight gradient's scale)
    int64_t n_tiles, stride, first, iters;

    __device__ __forceinline__ void init(unsigned char *lds_raw, int64_t n) {
        lane = tn::lane_id(); j = lane & 31; h = lane >> 5;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        stream = wave / G::WPS; wib = wave % G::WPS;
        seen = 0.0f;
        unsigned char *sbase = lds_raw + stream * G::STREAM_B;
        tiles = reinterpret_cast<unsigned short *>(sbase);
        stage0 = reinterpret_cast<float *>(sbase + 2 * G::TILE_B);
        pm = stage0 + 2 * G::STAGE_F;
        red = reinterpret_cast<float *>(lds_raw + G::STREAMS * G::STREAM_B + H * 4);
        n_tiles = (n + 31) >> 5;
        stride = (int64_t)gridDim.x * G::STREAMS;
        first = (int64_t)blockIdx.x * G::STREAMS;
        iters = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    }
    __device__ __forceinline__ int64_t tile_of(int64_t it) const { const int64_t t = first + stream + it * stride; return t < n_tiles ? t : n_tiles - 1; }
    __device__ __forceinline__ float *stage(int sbuf) const { return stage0 + sbuf * G::STAGE_F + (32 * wib) * 32; }      // this wave's 32 rows
    __device__ __forceinline__ const float *sp(int sbuf) const { return stage(sbuf) + (16 * h) * 32 + j; }
    __device__ __forceinline__ unsigned short *np(int buf) const { return tiles + buf * (2 * PLANE) + j * SB + 32 * wib + 16 * h; }
    __device__ __forceinline__ void request(const float *stash, int64_t tile, int rows_total, int off, int sbuf) const {
        request_rows<32>(urow(stash, tile * rows_total + off), 32 * wib, stage(sbuf), lane);
    }
    // largest |value| of this wave's 32 staged rows per sample -> pm[slot][sample][wave]
    __device__ __forceinline__ void publish_max(int sbuf) {
#ifdef TN_ABL_NOMAX
        if (h == 0) pm[sbuf * G::PM_F + j * G::WPS + wib] = 1.0f;
        return;
#endif
        const float *s_ = sp(sbuf);
        float m = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(s_[e * 32]));
        seen = fmaxf(seen, m);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (h == 0) pm[sbuf * G::PM_F + j * G::WPS + wib] = m;
    }
    // scale of sample j for the tile whose maxima sit in slot `sbuf` (all waves' partial maxima)
    __device__ __forceinline__ void sample_scale(int sbuf, float &s, float &inv) const {
        const float *p = pm + sbuf * G::PM_F + j * G::WPS;
        float m = 0.0f;
#pragma unroll
        for (int q = 0; q < G::WPS / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * q);
            m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        }
        pow2_scales(m, s, inv);
    }
    // value pair p (0 .. 7) of the wave's staged rows -> the two fp16 planes of a tile buffer (see mlp_b3_layers.hip convert_pair)
    __device__ __forceinline__ void convert_pair(const float *sp_, unsigned short *np_, int p, float s) const {
        const float v0 = sp_[(2 * p) * 32] * s, v1 = sp_[(2 * p + 1) * 32] * s;
        unsigned hi, lo;
        split2h(v0, v1, hi, lo);
        unsigned short *d = np_ + 2 * p;
        *reinterpret_cast<unsigned *>(d) = hi;
        *reinterpret_cast<unsigned *>(d + PLANE) = lo;
    }
    // tiles 0 and 1 staged, tile 0 converted, the maxima of tile 1 published, tile 2 requested; returns 1 / scale of tile 0
    __device__ __forceinline__ float prologue(const float *stash, int rows_total, int off) {
        request(stash, tile_of(0), rows_total, off, 0);
        request(stash, tile_of(1), rows_total, off, 1);
        __syncthreads();                                     // (vmcnt(0): the wave's own rows have landed)
        publish_max(0);
        publish_max(1);
        __syncthreads();
        float s, inv;
        sample_scale(0, s, inv);
#pragma unroll
        for (int p = 0; p < 8; ++p) convert_pair(sp(0), np(0), p, s);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads done before the next request overwrites the area
        request(stash, tile_of(2), rows_total, off, 0);
        __syncthreads();
        return inv;
    }
    // acc += (W s_W) (x s_j) for the wave's block from tile buffer `cur`; converts the staged tile it + 1 (staging buffer, maxima slot
    // `nb` = (it + 1) & 1) into buffer cur ^ 1 during the first CONV steps.  Returns through `inv_next` 1 / scale of tile it + 1.
    // `next_request` runs behind step CONV: the staged tile it + 1 is consumed then, and the wave asks for tile it + 3 into the same rows
    template <typename Req>
    __device__ __forceinline__ void k_loop(const Op2 (&A)[KS], f32x16 &acc, int cur, int nb, float &inv_next, Req next_request) const {
        const unsigned short *tc = tiles + cur * (2 * PLANE);
        const float *s_ = sp(nb);
        unsigned short *n_ = np(cur ^ 1);
        constexpr int CONV = G::CONV;
        constexpr int PPS = 8 / CONV;                        // pairs per conversion step (1 at H = 256, 2 at H = 128)
        float sc;
        sample_scale(nb, sc, inv_next);
        Op2 b = read_b2<SB>(tc, j, h, 0);
        float cv[2 * PPS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Op2 bn = b;
            if (s + 1 < KS) bn = read_b2<SB>(tc, j, h, s + 1);
            float cn[2 * PPS];
            if (s < CONV) {
#pragma unroll
                for (int u = 0; u < PPS; ++u) {
                    const int p = PPS * s + u;
                    cn[2 * u] = s_[(2 * p) * 32];
                    cn[2 * u + 1] = s_[(2 * p + 1) * 32];
                }
            }
            TN_F2_SB;
            const bool conv = s >= 1 && s <= CONV;
            constexpr int NMS = 3 * PPS;                     // micro-steps per step: scale + hi | residual + lo | two LDS writes
            unsigned cu[2];
            float cf[2];
            auto micro = [&](int m) {
                if (!conv || m >= NMS) return;
                const int u = m / 3, mm = m % 3;
                if (mm == 0) {
                    cf[0] = cv[2 * u] * sc; cf[1] = cv[2 * u + 1] * sc;
                    const f16x2 hh = {(_Float16)cf[0], (_Float16)cf[1]};
                    cu[0] = __builtin_bit_cast(unsigned, hh);
                } else if (mm == 1) {
                    const f16x2 hh = __builtin_bit_cast(f16x2, cu[0]);
                    const f16x2 ll = {(_Float16)(cf[0] - (float)hh[0]), (_Float16)(cf[1] - (float)hh[1])};
                    cu[1] = __builtin_bit_cast(unsigned, ll);
                } else {
                    const int p = PPS * (s - 1) + u;
                    unsigned short *d = n_ + 2 * p;
                    *reinterpret_cast<unsigned *>(d) = cu[0];
                    *reinterpret_cast<unsigned *>(d + PLANE) = cu[1];
                }
            };
            // three partial products, small terms first; one conversion micro-step (two at H = 128) behind every MFMA
#ifndef TN_ABL_NOMFMA
            acc = mfma_h(A[s].lo, b.hi, acc);
#else
            asm volatile("" :: "v"(A[s].lo), "v"(A[s].hi), "v"(b.hi), "v"(b.lo));
#endif
#pragma unroll
            for (int e = 0; e < PPS; ++e) micro(0 * PPS + e);
            TN_F2_SB;
#ifndef TN_ABL_NOMFMA
            acc = mfma_h(A[s].hi, b.lo, acc);
#endif
#pragma unroll
            for (int e = 0; e < PPS; ++e) micro(1 * PPS + e);
            TN_F2_SB;
#ifndef TN_ABL_NOMFMA
            acc = mfma_h(A[s].hi, b.hi, acc);
#endif
#pragma unroll
            for (int e = 0; e < PPS; ++e) micro(2 * PPS + e);
            if (s == CONV) next_request();
            TN_F2_SB;
            b = bn;
#pragma unroll
            for (int u = 0; u < 2 * PPS; ++u) cv[u] = cn[u];
        }
    }
    // behind the k loop of iteration `it`: the rows of tile it + 2 (requested in the middle of iteration it - 1 into staging buffer
    // it & 1) have landed -- the only younger requests are the four of tile it + 3 from the middle of this k loop (vector memory
    // operations retire in order; the stores of tile it - 1 are older) -- and their maxima go out
    __device__ __forceinline__ void finish_staging(int sbuf) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        publish_max(sbuf);
    }
    __device__ __forceinline__ void tile_barrier() const {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // conversion writes + maxima retired; the row stores may stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // end of the launch: the largest |value| of every row this workgroup staged -> *dst (non-negative floats order like their bits)
    __device__ __forceinline__ void report_max(float *dst) const {
        if (dst == nullptr) return;
        float m = seen;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(dst), __float_as_uint(m));
    }
    // the layer's weight scale from every wave's largest |weight| (prologue)
    __device__ __forceinline__ void layer_scale(float wmax, float &s, float &inv) const {
        float m = wmax;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        float g = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) g = fmaxf(g, red[w]);
        pow2_scales(g, s, inv);
    }
};

// weights of output rows 32 ob + i: step s, lane (i, h): W[row][16 s + 8 h + 0..7]   (forward)
template <int KS>
__device__ __forceinline__ float load_rows(const float *__restrict__ W, int ldw, int row, bool ok, int h, float (&v)[KS][8]) {
    const float *wr = W + (int64_t)(ok ? row : 0) * ldw + 8 * h;
    float m = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wr + 16 * s), w1 = *reinterpret_cast<const f32x4 *>(wr + 16 * s + 4);
        const float t[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[s][e] = ok ? t[e] : 0.0f; m = fmaxf(m, fabsf(v[s][e])); }
    }
    return m;
}
// ... of W^T: A[i = column 32 kb + i][k = row n]: step s, lane (i, h): W[16 s + 8 h + e][col]   (data gradient)
template <int KS>
__device__ __forceinline__ float load_cols(const float *__restrict__ W, int ldw, int col, int h, float (&v)[KS][8]) {
    float m = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[s][e] = W[(int64_t)(16 * s + 8 * h + e) * ldw + col]; m = fmaxf(m, fabsf(v[s][e])); }
    return m;
}

template <int H, bool LAST>
__global__ __launch_bounds__(F2Geom<H>::THREADS) void fwd_f2_kernel(FwdLayerArgs a, int64_t n, float *__restrict__ stash, float *__restrict__ y)
{
    using G = F2Geom<H>;
    constexpr int KS = G::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Stream2<H> st;
    st.init(lds_raw, n);
    float *bias_s = reinterpret_cast<float *>(lds_raw + G::STREAMS * G::STREAM_B);
    for (int e = threadIdx.x; e < H; e += blockDim.x) bias_s[e] = e < a.N ? a.B[e] : 0.0f;
    const int j = st.j, h = st.h, lane = st.lane;
    const int ob = st.wib;
    Op2 A[KS];
    float inv_w;
    {
        float v[KS][8];
        const int row = 32 * ob + j;
        const float wmax = load_rows<KS>(a.W, a.K, row, row < a.N, h, v);
        float s_w;
        st.layer_scale(wmax, s_w, inv_w);                    // (carries the barrier behind the bias staging)
        split_weights<KS>(v, s_w, A);
    }
    if (st.iters == 0) return;
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = bias_s[32 * ob + frow(r, h)];
    float inv_cur = st.prologue(stash, a.rows_total, a.off_in);
    int cur = 0;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it < st.iters; ++it) {
        const int par = (int)(it & 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        float inv_next;
        st.k_loop(A, acc, cur, par ^ 1, inv_next, [&]() { st.request(stash, st.tile_of(it + 3), a.rows_total, a.off_in, par ^ 1); });
        st.finish_staging(par);
        const int64_t tile = st.tile_of(it);
        tn::pin16(acc);
        const float c = inv_w * inv_cur;
#ifndef TN_ABL_NOEPI
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fmaf(acc[r], c, bias[r]);
#endif
        if constexpr (!LAST) {
#ifndef TN_ABL_NOEPI
            acc = tn::relu16(acc);
#endif
#ifndef TN_ABL_NOSTORE
            wreg_store_block<(TN_F2_NT & 2) != 0>(urow(stash, tile * a.rows_total + a.off_out), ob, j, h, acc);
#else
            tn::pin16(acc);
#endif
            if (a.off_bits >= 0) {
                unsigned *bits = reinterpret_cast<unsigned *>(urow(stash, tile * a.rows_total + a.off_bits + 2 * ob));
                bits[lane] = relu_bits(acc);
            }
        } else if (a.N == H) {
            const int64_t row = tile * 32 + j;
            const bool valid = row < n;
            f32x16 pre;
#pragma unroll
            for (int r = 0; r < 16; ++r) pre[r] = valid ? acc[r] : 0.0f;
            wreg_store_block(urow(stash, tile * a.rows_total + a.off_out), ob, j, h, pre);
            if (y != nullptr) {                      // (TN_MLP_ROWS_ONLY: the consumer reads the rows)
                float *yr = y + (valid ? row : 0) * H + 32 * ob + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = tn::apply_act(acc[4 * q + u], a.out_act);
                    if (valid) *reinterpret_cast<f32x4 *>(yr + 8 * q) = v;
                }
            }
        } else if (32 * ob < a.N) {
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
        st.tile_barrier();
        inv_cur = inv_next;
        cur ^= 1;
    }
    st.report_max(a.max_in);
}

template <int H>
__global__ __launch_bounds__(F2Geom<H>::THREADS) void dgrad_f2_kernel(DgradArgs a, int64_t n, float *__restrict__ stash)
{
    using G = F2Geom<H>;
    constexpr int KS = G::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Stream2<H> st;
    st.init(lds_raw, n);
    const int j = st.j, h = st.h, lane = st.lane;
    const int kb = st.wib;
    Op2 A[KS];
    float inv_w;
    {
        float v[KS][8];
        const float wmax = load_cols<KS>(a.W, a.K, 32 * kb + j, h, v);
        float s_w;
        st.layer_scale(wmax, s_w, inv_w);
        split_weights<KS>(v, s_w, A);
    }
    if (st.iters == 0) return;
    float inv_cur = st.prologue(stash, a.rows_total, a.off_gin);
    int cur = 0;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it < st.iters; ++it) {
        const int par = (int)(it & 1);
        const int64_t tile = st.tile_of(it);
        const unsigned mbits = reinterpret_cast<const unsigned *>(urow(stash, tile * a.rows_total + a.off_bits + 2 * kb))[lane];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        float inv_next;
        st.k_loop(A, acc, cur, par ^ 1, inv_next, [&]() { st.request(stash, st.tile_of(it + 3), a.rows_total, a.off_gin, par ^ 1); });
        st.finish_staging(par);
        tn::pin16(acc);
        const float c = inv_w * inv_cur;
        f32x16 res;
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = mask_keep(acc[r] * c, mbits, r);
        wreg_store_block<(TN_F2_NT & 2) != 0>(urow(stash, tile * a.rows_total + a.off_gout), kb, j, h, res);
        st.tile_barrier();
        inv_cur = inv_next;
        cur ^= 1;
    }
    st.report_max(a.max_in);
}

// ------------------------------------------------------------------------------------------------
// weight gradient  dW[n][k] += sum_s G[n][s] A[k][s]  as two-term fp16 splits.  The reduction runs over the SAMPLES: a scale
// can only leave the accumulators if it is the same for every tile of the launch, so each operand takes ONE power of two from
// the largest |value| of all its rows -- known before the launch because the f16x2 forward / data-gradient kernels of the
// same layer stage exactly these rows and report their maximum (FwdLayerArgs / DgradArgs::max_in); run_layers therefore runs a
// layer's data gradient in front of its weight gradient.  An element 2^-k below the maximum keeps 22 - max(0, k - 13) bits:
// what it contributes to a sum over all samples is that much smaller, too.  Same pipeline as wgrad_b3_kernel
// (mlp_b3_layers.hip): half tiles of 16 samples, two LDS buffers, conversion micro-steps behind the MFMAs, one barrier.
// ------------------------------------------------------------------------------------------------
template <int H, int BN, int BK>
__global__ __launch_bounds__(64 * (H / 32 / BN) * (H / 32 / BK)) void wgrad_f2_kernel(WgradArgs a, int64_t n, const float *__restrict__ stash)
{
    constexpr int TH = 64 * (H / 32 / BN) * (H / 32 / BK);
    constexpr int NR = 2 * H;                          // rows per half tile: G rows [0, H), A rows [H, 2 H)
    constexpr int RS = 24;                             // fp16 elements per LDS row: 16 samples + 8 pad
    constexpr int PLANE = NR * RS;
    constexpr int BUF = 2 * PLANE;
    constexpr int NCH = (NR * 4) / TH;                 // 16-byte chunks (4 samples of a row) per thread and half tile
    constexpr int WK = (H / 32) / BK;
    static_assert(NCH * TH == NR * 4 && NCH >= 2 && (NCH & 1) == 0, "the waves own all tiles, every thread holds G and A chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned short *lds = reinterpret_cast<unsigned short *>(lds_raw);
    const int lane = tn::lane_id(), i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int tn0 = (wave / WK) * BN, tk0 = (wave % WK) * BK;
    float s_g, inv_g, s_a, inv_a;
    pow2_scales(*a.g_max, s_g, inv_g);
    pow2_scales(*a.a_max, s_a, inv_a);
    f32x16 acc[BN][BK];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn)
#pragma unroll
        for (int bk = 0; bk < BK; ++bk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bn][bk][r] = 0.0f;
    const int qd = threadIdx.x & 3;
    int src_off[NCH], dst_off[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int row = (threadIdx.x + TH * c) >> 2;
        src_off[c] = (row < H ? row : row - H) * 32 + 4 * qd;          // relative to the tile's G rows (c < NCH / 2) / A rows
        dst_off[c] = row * RS + 4 * qd;
    }
    float dbacc[NCH / 2];
#pragma unroll
    for (int c = 0; c < NCH / 2; ++c) dbacc[c] = 0.0f;
    const int64_t stride = gridDim.x;
    typedef const __attribute__((address_space(1))) char gchar;
    struct Bases { gchar *g, *a; };                                // wave-uniform: the half tile's G rows and A rows
    auto uniform_global = [](const float *p) {
        const uint64_t v = (uint64_t)p;
        const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        return (gchar *)u;
    };
    auto half_src = [&](int64_t it) {
        int64_t tile = blockIdx.x + (it >> 1) * stride;
        tile = tile < n_tiles ? tile : n_tiles - 1;
        return Bases{uniform_global(stash + (tile * a.rows_total + a.off_g) * 32 + 16 * (it & 1)),
                     uniform_global(stash + (tile * a.rows_total + a.off_a) * 32 + 16 * (it & 1))};
    };
    const int64_t my_tiles = (int64_t)blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + stride - 1) / stride : 0;
    const int64_t iters = 2 * my_tiles;
    if (iters == 0) return;
    // two register sets of raw chunks: half tile it + 1 (being converted during half step it) and it + 2 (in flight); a chunk's
    // successor two half tiles on is requested into its own registers as soon as it is converted -- every request has two half
    // steps (~3 us) to land, 64 KB per CU in flight (one set: 32 KB and one half step, which is the loaded HBM latency -- the
    // kernel ran at exactly 32 KB x 256 CUs per half step = 4.9 TB/s)
    f32x4 st[2][NCH];
    unsigned src_boff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) src_boff[c] = (unsigned)src_off[c] * 4u;
    auto load_chunk = [&](const Bases &base, int c, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        unsigned off = src_boff[c];
        asm volatile("" : "+v"(off));
        // (no nt hint here: a half tile takes 64 of a row's 128 bytes and the other half comes out of L2 one half step later -- with
        // the hint the launch takes 0.56 instead of 0.39 ms)
        st[SET][c] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>((c < NCH / 2 ? base.g : base.a) + off);
    };
    // conversion of chunk c in six micro-steps (0 / 2: scale + hi of a value pair, 1 / 3: residual + lo, 4: the two 8-byte LDS
    // writes, 5: bias sum + request of the chunk's successor); G chunks come first (c < NCH / 2)
    unsigned cu[4];
    float cf[2];
    auto micro = [&](int c, int m, unsigned short *buf, const Bases &nb, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const float sc = c < NCH / 2 ? s_g : s_a;
        if (m < 4) {
            const int e = m >= 2 ? 2 : 0;
            if ((m & 1) == 0) {
                cf[0] = st[SET][c][e] * sc; cf[1] = st[SET][c][e + 1] * sc;
                const f16x2 hh = {(_Float16)cf[0], (_Float16)cf[1]};
                cu[e] = __builtin_bit_cast(unsigned, hh);
            } else {
                const f16x2 hh = __builtin_bit_cast(f16x2, cu[e]);
                const f16x2 ll = {(_Float16)(cf[0] - (float)hh[0]), (_Float16)(cf[1] - (float)hh[1])};
                cu[e + 1] = __builtin_bit_cast(unsigned, ll);
            }
        } else if (m == 4) {
            unsigned short *d = buf + dst_off[c];
            *reinterpret_cast<uint2 *>(d) = make_uint2(cu[0], cu[2]);
            *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(cu[1], cu[3]);
        } else {
            if (c < NCH / 2) dbacc[c] += (st[SET][c][0] + st[SET][c][1]) + (st[SET][c][2] + st[SET][c][3]);
            load_chunk(nb, c, set_tag);
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    {
        const Bases b0 = half_src(0), b1 = half_src(1), b2 = half_src(2 < iters ? 2 : 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_chunk(b0, c, Set0{});
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_chunk(b1, c, Set1{});
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int m = 0; m < 6; ++m) micro(c, m, lds, b2, Set0{});
        }
    }
    __syncthreads();
    int g_off[BN], a_off[BK];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) g_off[bn] = (32 * (tn0 + bn) + i) * RS + 8 * h;
#pragma unroll
    for (int bk = 0; bk < BK; ++bk) a_off[bk] = (H + 32 * (tk0 + bk) + i) * RS + 8 * h;
    auto read_op = [&](const unsigned short *buf, int off) {
        Op2 o;
        o.hi = *reinterpret_cast<const u32x4 *>(buf + off);
        o.lo = *reinterpret_cast<const u32x4 *>(buf + off + PLANE);
        return o;
    };
    // half step `it` (parity P = it & 1, a compile-time constant: the loop below is unrolled by two): products from LDS buffer P,
    // half tile it + 1 (register set P ^ 1) converted into buffer P ^ 1, half tile it + 3 requested into that set
    auto half_step = [&](int64_t it, auto par_tag, auto convert_tag) {
        constexpr int P = decltype(par_tag)::value;
        constexpr bool CONVERT = decltype(convert_tag)::value;
        using SetN = std::integral_constant<int, P ^ 1>;
        const unsigned short *bc = lds + P * BUF;
        unsigned short *bnx = lds + (P ^ 1) * BUF;
        const Bases nb = half_src(it + 3 < iters ? it + 3 : it + 1 < iters ? it + 1 : it);
        Op2 gop[BN];
#pragma unroll
        for (int bn = 0; bn < BN; ++bn) gop[bn] = read_op(bc, g_off[bn]);
        Op2 aop = read_op(bc, a_off[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            Op2 anx = aop;
            if (bk + 1 < BK) anx = read_op(bc, a_off[bk + 1]);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NM = BN * BK * 3, MS = NCH * 6, PER = MS / NM;
            static_assert(PER * NM == MS, "micro-steps divide evenly over the MFMAs");
            auto term = [&](int t, const u32x4 &(*ga)(const Op2 &), const u32x4 &(*ab)(const Op2 &)) {
#pragma unroll
                for (int bn = 0; bn < BN; ++bn) {
                    acc[bn][bk] = mfma_h(ga(gop[bn]), ab(aop), acc[bn][bk]);
                    if constexpr (CONVERT) {
#pragma unroll
                        for (int u = 0; u < PER; ++u) {
                            const int m = ((bk * 3 + t) * BN + bn) * PER + u;
                            micro(m / 6, m % 6, bnx, nb, SetN{});
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            term(0, [](const Op2 &o) -> const u32x4 & { return o.lo; }, [](const Op2 &o) -> const u32x4 & { return o.hi; });
            term(1, [](const Op2 &o) -> const u32x4 & { return o.hi; }, [](const Op2 &o) -> const u32x4 & { return o.lo; });
            term(2, [](const Op2 &o) -> const u32x4 & { return o.hi; }, [](const Op2 &o) -> const u32x4 & { return o.hi; });
            aop = anx;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it + 2 < iters; it += 2) {               // (iters is even: two half tiles per tile)
        half_step(it, Set0{}, std::true_type{});
        half_step(it + 1, Set1{}, std::true_type{});
    }
    half_step(iters - 2, Set0{}, std::true_type{});
    half_step(iters - 1, Set1{}, std::false_type{});
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const float c_out = inv_g * inv_a;
#pragma unroll
    for (int bn = 0; bn < BN; ++bn)
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            const int k = 32 * (tk0 + bk) + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * (tn0 + bn) + frow(r, h);
                atomicAdd(&a.gW[(int64_t)nn * a.K + k], acc[bn][bk][r] * c_out);
            }
        }
#pragma unroll
    for (int c = 0; c < NCH / 2; ++c) {
        float sgm = dbacc[c];
        sgm += __shfl_xor(sgm, 1, 64);
        sgm += __shfl_xor(sgm, 2, 64);
        const int row = (threadIdx.x + TH * c) >> 2;
        if (qd == 0) atomicAdd(&a.gB[row], sgm);
    }
}

template <int H, int BN, int BK>
int launch_wgrad(const WgradArgs &w, int64_t n, const float *stash, hipStream_t s)
{
    constexpr size_t lds_bytes = (size_t)2 * 2 * (2 * H) * 24 * 2;
    auto kern = wgrad_f2_kernel<H, BN, BK>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd(f16x2): cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int per_cu = lds_bytes * 2 <= (size_t)LDS_LIMIT_BYTES ? 2 : 1;
    kern<<<dim3((unsigned)std::min<int64_t>(n_tiles, 256 * per_cu)), dim3(64 * (H / 32 / BN) * (H / 32 / BK)), lds_bytes, s>>>(w, n, stash);
    return tn::check_launch("wgrad_f2_kernel");
}


// ------------------------------------------------------------------------------------------------
// First layer of a wide stack (<= 64 encoded or plain inputs as workspace rows, enc_rows_kernel) in the f16x2 arithmetic: the
// fwd_lds_kernel of mlp_bwd_layers.hip with 12 fp16 MFMAs per 32 x 32 output block instead of 32 fp32 ones (0.52 ms per Vanilla
// step at 0.50 matrix-pipe busy: the last matrix-bound launch of the stack's forward).  A workgroup stages all H weight rows once
// as hi / lo planes (row stride 72 halfs: conflict-free b128 operand reads), scaled by the layer's largest |weight|; a wave takes
// a tile's 64 input rows in the D layout -- registers 8 (b & 1) .. + 7 of block b >> 1 ARE the B operand of k block b (the
// heads' layout, mlp_f2_heads.h) --, converts them once with the sample's scale, and walks the H / 32 output blocks.
// ------------------------------------------------------------------------------------------------
template <int H, int WPB>
__global__ __launch_bounds__(WPB * 64) void fwd_first_f2_kernel(FwdLayerArgs a, int64_t n, float *__restrict__ stash)
{
    constexpr int NOT = H / 32, ST = 72, PLANE = H * ST;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    _Float16 *wh = reinterpret_cast<_Float16 *>(lds_raw), *wl = wh + PLANE;
    float *bias_s = reinterpret_cast<float *>(lds_raw + 2 * PLANE * 2);
    float *red = bias_s + H;
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float m = 0.0f;
    for (int e = threadIdx.x; e < a.N * a.K; e += blockDim.x) m = fmaxf(m, fabsf(a.W[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    for (int e = threadIdx.x; e < H; e += blockDim.x) bias_s[e] = e < a.N ? a.B[e] : 0.0f;
    __syncthreads();
    float g = 0.0f;
    for (int w = 0; w < WPB; ++w) g = fmaxf(g, red[w]);
    float s_w, inv_w;
    pow2_scales(g, s_w, inv_w);
    for (int e = threadIdx.x; e < H * 64; e += blockDim.x) {
        const int o = e >> 6, q = e & 63;
        const float v = (o < a.N && q < a.K) ? a.W[(int64_t)o * a.K + q] * s_w : 0.0f;
        const _Float16 vh = (_Float16)v;
        const int w = q & 15, pos = o * ST + (q & ~15) + ((w >> 2) & 1) * 8 + (w & 3) + ((w >> 3) << 2);      // [k block][half h][8]
        wh[pos] = vh;
        wl[pos] = (_Float16)(v - (float)vh);
    }
    __syncthreads();
    const int64_t n_tiles = (n + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        float *st = stash + tile * (int64_t)a.rows_total * 32;
        const float *in = st + a.off_in * 32;
        float x[2][16];
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * t + frow(r, h);
                const float v = in[(k < a.Kp ? k : 0) * 32 + j];
                x[t][r] = k < a.Kp ? v : 0.0f;                            // rows past Kp belong to another buffer
                mx = fmaxf(mx, fabsf(x[t][r]));
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float s_x, inv_x;
        pow2_scales(mx, s_x, inv_x);
        Op2 B[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned hi, lo;
                split2h(x[b >> 1][8 * (b & 1) + 2 * p] * s_x, x[b >> 1][8 * (b & 1) + 2 * p + 1] * s_x, hi, lo);
                B[b].hi[p] = hi; B[b].lo[p] = lo;
            }
        const float c = inv_w * inv_x;
#pragma clang loop unroll(disable)
        for (int ot = 0; ot < NOT; ++ot) {
            if (32 * ot >= a.N) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const _Float16 *wr = wh + (32 * ot + j) * ST + 8 * h;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const u32x4 ah = *reinterpret_cast<const u32x4 *>(wr + 16 * b), al = *reinterpret_cast<const u32x4 *>(wr + 16 * b + PLANE);
                acc = mfma_h(al, B[b].hi, acc);
                acc = mfma_h(ah, B[b].lo, acc);
                acc = mfma_h(ah, B[b].hi, acc);
            }
            tn::pin16(acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(acc[r], c, bias_s[32 * ot + frow(r, h)]);
            acc = tn::relu16(acc);
            wreg_store_block<(TN_F2_NT & 2) != 0>(urow(stash, tile * a.rows_total + a.off_out), ot, j, h, acc);
            if (a.off_bits >= 0) reinterpret_cast<unsigned *>(st + (a.off_bits + 2 * ot) * 32)[lane] = relu_bits(acc);
        }
    }
}

template <int H>
int launch_first(const FwdLayerArgs &f, int64_t n, float *stash, hipStream_t s)
{
    constexpr int WPB = 8;
    constexpr size_t lds_bytes = (size_t)2 * H * 72 * 2 + H * 4 + 64;
    auto kern = fwd_first_f2_kernel<H, WPB>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_fwd(f16x2, first layer): cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int per_cu = (int)std::min<size_t>(4, (size_t)LDS_LIMIT_BYTES / lds_bytes);
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * per_cu));
    kern<<<dim3((unsigned)bl), dim3(WPB * 64), lds_bytes, s>>>(f, n, stash);
    return tn::check_launch("fwd_first_f2_kernel");
}

template <int H, bool LAST>
int launch_fwd(const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s)
{
    using G = F2Geom<H>;
    auto kern = fwd_f2_kernel<H, LAST>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_fwd(f16x2): cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + G::STREAMS - 1) / G::STREAMS, 256));
    kern<<<dim3((unsigned)bl), dim3(G::THREADS), G::lds_bytes, s>>>(f, n, stash, y);
    return tn::check_launch("fwd_f2_kernel");
}

template <int H>
int launch_dgrad(const DgradArgs &d, int64_t n, float *stash, hipStream_t s)
{
    using G = F2Geom<H>;
    if (d.off_bits < 0) return tn::fail(TN_E_CONFIG, "mlp_bwd(f16x2): the data gradient takes its ReLU masks as bit rows");
    auto kern = dgrad_f2_kernel<H>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd(f16x2): cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + G::STREAMS - 1) / G::STREAMS, 256));
    kern<<<dim3((unsigned)bl), dim3(G::THREADS), G::lds_bytes, s>>>(d, n, stash);
    return tn::check_launch("dgrad_f2_kernel");
}

}  // namespace

namespace tn {
namespace layers {

__attribute__((visibility("hidden"))) int launch_fwd_f2(int H, bool last, const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s)
{
    if (H == 256) return last ? launch_fwd<256, true>(f, n, stash, y, s) : launch_fwd<256, false>(f, n, stash, y, s);
    if (H == 128) return last ? launch_fwd<128, true>(f, n, stash, y, s) : launch_fwd<128, false>(f, n, stash, y, s);
    return tn::fail(TN_E_CONFIG, "mlp_fwd(f16x2): width 128 or 256");
}

// first layer (K <= 64 input rows, hidden output: ReLU, rows + bit rows)
__attribute__((visibility("hidden"))) int launch_fwd_first_f2(int H, const FwdLayerArgs &f, int64_t n, float *stash, hipStream_t s)
{
    if (f.Kp > 64 || f.K > 64 || f.N > H) return tn::fail(TN_E_CONFIG, "mlp_fwd(f16x2, first layer): at most 64 input rows");
    if (H == 256) return launch_first<256>(f, n, stash, s);
    if (H == 128) return launch_first<128>(f, n, stash, s);
    return tn::fail(TN_E_CONFIG, "mlp_fwd(f16x2): width 128 or 256");
}

__attribute__((visibility("hidden"))) int launch_dgrad_f2(int H, const DgradArgs &d, int64_t n, float *stash, hipStream_t s)
{
    if (H == 256) return launch_dgrad<256>(d, n, stash, s);
    if (H == 128) return launch_dgrad<128>(d, n, stash, s);
    return tn::fail(TN_E_CONFIG, "mlp_bwd(f16x2): width 128 or 256");
}

__attribute__((visibility("hidden"))) int launch_wgrad_f2(int H, const WgradArgs &w, int64_t n, const float *stash, hipStream_t s)
{
    if (w.first || w.N != H || w.K != H || !w.g_max || !w.a_max) return tn::fail(TN_E_CONFIG, "mlp_bwd(f16x2): square hidden layers with both maxima");
    if (H == 256) return launch_wgrad<256, 4, 2>(w, n, stash, s);
    if (H == 128) return launch_wgrad<128, 2, 2>(w, n, stash, s);
    return tn::fail(TN_E_CONFIG, "mlp_bwd(f16x2): width 128 or 256");
}

}  // namespace layers
}  // namespace tn
