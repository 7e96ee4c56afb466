// Layers of the wide stacks (Vanilla feature MLP 256 x 10, reference models.py:59-68, run.py:131; Cobafa 128 x 6) on the bf16
// matrix cores with exact three-way operand splits (b3_device.h) -- same workspace rows, same arguments and the same results to
// fp32 rounding as the fp32-MFMA kernels of mlp_bwd_layers.hip, at 6 x 32 instead of 8 x 64 matrix-pipe cycles per 32 x 32 x 16
// block.
//
// Forward / data gradient (fwd_b3_kernel / dgrad_b3_kernel): as in the fp32 form a wave owns 32-row blocks of the layer's
// output and keeps their weights -- all K columns -- in registers for the whole launch, now as three packed-bf16 A operands per
// 16-wide k step (12 registers per step and block), split once when the kernel starts.  H = 256: EIGHT waves per workgroup, one
// block each (TN_B3_BPW256 = 1, round 4: 192 weight registers of the 256 a wave may use with two waves per SIMD -- 250 VGPRs, no
// spills; A / B on one box against four waves x two blocks, one wave per SIMD with 384 weight registers, the round-3 form
// (TN_B3_BPW256 = 2): forward 757 -> 696 us, data gradient 634 -> 594 us per layer, Vanilla step 21.5 -> 20.8 ms: the second wave
// of a SIMD issues MFMAs while the first one is in its epilogue, at the price of every B operand being read from LDS twice);
// H = 128: four waves per tile stream, one block each, two streams per workgroup.
// The 32-sample input tile reaches the workgroup through LDS as three bf16 planes [sample][feature] (B operand of a k step =
// one ds_read_b128 per term, shared by the wave's blocks):
//   * a wave's rows of the NEXT tile arrive by LDS-direct loads (global_load_lds_dwordx4) in an fp32 staging area -- no
//     staging registers -- requested in the middle of the current tile's k loop, as soon as the area is free;
//   * in the first half of the k loop the wave converts its rows of the next tile, two value pairs per step: ds_read_b32 from
//     the staging area, the exact split (b3::split2), three ds_write_b32 into the other tile buffer.  The conversion
//     instructions sit between the MFMAs of a step and overlap with them; only the wave that loaded a row touches it;
//   * one barrier per tile (two tile buffers): everybody has finished reading the current tile and writing the next.  It is a
//     raw s_barrier behind a COUNTED s_waitcnt: the LDS-direct loads must have landed, the tile's own row stores (issued
//     behind them, retired in order) need not -- __syncthreads() would drain them too (vmcnt(0)).
// LDS: 2 x 3 planes x 32 x (H + 8) bf16 + 32 x H fp32 staging per stream + bias = 135 KB for H = 256.
#include "mlp_layers.h"
#include "b3_device.h"
#include <algorithm>
#include <type_traits>

#ifndef TN_B3_ABLATE
#define TN_B3_ABLATE 0      // timing experiments only (wrong results): 1 = no conversion, 2 = no LDS-direct requests, 4 = no stores
#endif

namespace {

using namespace tn::layers;
using namespace tn::mlp;
using tn::b3::Op;
using tn::b3::u32x4;
using tn::f32x16;
using tn::f32x4;

__device__ __forceinline__ void glds16(const float *src, float *dst) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

template <int H> struct B3Geom {
    static constexpr int T = H / 32;                // 32-row blocks of the layer's input and of its output
#ifndef TN_B3_BPW256
#define TN_B3_BPW256 1
#endif
    static constexpr int BPW = H == 256 ? TN_B3_BPW256 : 1;   // blocks per wave (see the file header)
    static constexpr int WPS = T / BPW;             // waves per tile stream (4)
    static constexpr int STREAMS = H == 256 ? 1 : 2;
    static constexpr int THREADS = STREAMS * WPS * 64;
    static constexpr int KS = H / 16;               // k steps
#ifndef TN_B3_CONV_DIV
#define TN_B3_CONV_DIV 2
#endif
    static constexpr int CONV = H == 256 ? KS / TN_B3_CONV_DIV : KS / 2;   // steps that carry the conversion of the next tile: the sooner
                                                    // the staging area is free, the longer the request of the tile after next has to land
    static constexpr int SB = H + 8;                // bf16 elements per LDS tile row: (H + 8) * 2 B = odd multiple of 16 B
    static constexpr int PLANE = 32 * SB;           // bf16 elements per term plane
    static constexpr int TILE_B = 3 * PLANE * 2;    // bytes per tile buffer
    static constexpr int STAGE_B = 32 * H * 4;      // bytes of the fp32 staging area of a stream ([row][32 samples], lane-linear)
    static constexpr int STREAM_B = 2 * TILE_B + STAGE_B;
    static constexpr size_t lds_bytes = (size_t)STREAMS * STREAM_B + H * 4;
};

// weights of output rows 32 ob + i as A operands: step s, lane (i, h): W[row][16 s + 8 h + 0..7]
template <int KS>
__device__ __forceinline__ void load_weights_rows(const float *__restrict__ W, int ldw, int row, bool ok, int h, Op (&A)[KS]) {
    const float *wr = W + (int64_t)(ok ? row : 0) * ldw + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wr + 16 * s), w1 = *reinterpret_cast<const f32x4 *>(wr + 16 * s + 4);
        float v[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        if (!ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        }
        A[s] = tn::b3::split8(v);
    }
}
// ... of W^T: A[i = column 32 kb + i][k = row n]: step s, lane (i, h): W[16 s + 8 h + e][col]
template <int KS>
__device__ __forceinline__ void load_weights_cols(const float *__restrict__ W, int ldw, int col, int h, Op (&A)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W[(int64_t)(16 * s + 8 * h + e) * ldw + col];
        A[s] = tn::b3::split8(v);
    }
}

// B operand of step s for sample j: three ds_read_b128
template <int SB>
__device__ __forceinline__ Op read_b(const unsigned short *tile, int j, int h, int s) {
    constexpr int PLANE = 32 * SB;
    const unsigned short *p = tile + j * SB + 16 * s + 8 * h;
    Op o;
    o.hi = *reinterpret_cast<const u32x4 *>(p);
    o.mid = *reinterpret_cast<const u32x4 *>(p + PLANE);
    o.lo = *reinterpret_cast<const u32x4 *>(p + 2 * PLANE);
    return o;
}

// LDS-direct request of `nrows` (multiple of 8) rows of a tile, starting at row r0 of `rows` ([row][32 samples]) -> stage
// (lane-linear: one instruction = 8 rows x 128 B = 1 KB)
template <int NROWS>
__device__ __forceinline__ void request_rows(const float *rows, int r0, float *stage, int lane) {
    const float *src = rows + r0 * 32 + 4 * lane;            // 16 B per lane
#pragma unroll
    for (int e = 0; e < NROWS / 8; ++e) glds16(src + e * 256, stage + e * 256);
}

// value pair p (0 .. 8 BPW - 1) of the wave's staged rows -> the three bf16 planes of a tile buffer.  The wave's 32 BPW rows x 32
// samples are 16 BPW values per lane: lane (j, h) owns rows 32 b + 16 h + e (b < BPW, e < 16) of sample j; pair p = (b, e = 2 q,
// 2 q + 1) with b = p / 8, q = p % 8.  Feature index of row r (within the wave's rows) = f0 + r.
template <int SB>
__device__ __forceinline__ void convert_pair(const float *sp, unsigned short *np, int p) {
    constexpr int PLANE = 32 * SB;
    const int b = p >> 3, q = p & 7;
    const float v0 = sp[(32 * b + 2 * q) * 32], v1 = sp[(32 * b + 2 * q + 1) * 32];
    unsigned hi, mid, lo;
    tn::b3::split2(v0, v1, hi, mid, lo);
    unsigned short *d = np + 32 * b + 2 * q;
    *reinterpret_cast<unsigned *>(d) = hi;
    *reinterpret_cast<unsigned *>(d + PLANE) = mid;
    *reinterpret_cast<unsigned *>(d + 2 * PLANE) = lo;
}

// Everything of one tile stream that forward and data gradient share: geometry, prologue, the k loop with the conversion of the
// next tile and the request of the one after it in the middle, the counted wait + barrier.
template <int H>
struct Stream {
    using G = B3Geom<H>;
    static constexpr int KS = G::KS, SB = G::SB, PLANE = G::PLANE, BPW = G::BPW;
    int lane, j, h, wave, stream, wib;          // wib: wave in stream; it owns blocks BPW wib .. + BPW - 1
    unsigned short *tiles;
    float *stage;                               // this wave's 32 BPW rows of the staging area
    int64_t n_tiles, stride, first, iters;

    __device__ __forceinline__ void init(unsigned char *lds_raw, int64_t n) {
        lane = tn::lane_id(); j = lane & 31; h = lane >> 5;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        stream = wave / G::WPS; wib = wave % G::WPS;
        unsigned char *sbase = lds_raw + stream * G::STREAM_B;
        tiles = reinterpret_cast<unsigned short *>(sbase);
        stage = reinterpret_cast<float *>(sbase + 2 * G::TILE_B) + (32 * BPW * wib) * 32;
        n_tiles = (n + 31) >> 5;
        stride = (int64_t)gridDim.x * G::STREAMS;
        first = (int64_t)blockIdx.x * G::STREAMS;
        iters = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    }
    __device__ __forceinline__ int64_t tile_of(int64_t it) const { const int64_t t = first + stream + it * stride; return t < n_tiles ? t : n_tiles - 1; }
    __device__ __forceinline__ const float *sp() const { return stage + (16 * h) * 32 + j; }
    __device__ __forceinline__ unsigned short *np(int buf) const { return tiles + buf * (3 * PLANE) + j * SB + 32 * BPW * wib + 16 * h; }
    __device__ __forceinline__ void request(const float *stash, int64_t tile, int rows_total, int off) const {
        request_rows<32 * BPW>(urow(stash, tile * rows_total + off), 32 * BPW * wib, stage, lane);
    }
    // tiles 0 (converted in one go) and 1 (requested); ends with a full barrier
    __device__ __forceinline__ void prologue(const float *stash, int rows_total, int off) {
        request(stash, tile_of(0), rows_total, off);
        __syncthreads();                                     // (vmcnt(0): the wave's own rows have landed)
#pragma unroll
        for (int p = 0; p < 8 * BPW; ++p) convert_pair<SB>(sp(), np(0), p);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads done before the next request overwrites the area
        request(stash, tile_of(1), rows_total, off);
        __syncthreads();
    }
    // acc[b] += W_b x tile(cur) for the wave's blocks; converts the staged tile into buffer cur ^ 1 during the first CONV steps,
    // then requests tile `it + 2`.  Software-pipelined by hand and pinned with scheduling barriers (one wave per SIMD at H = 256:
    // nobody else hides an LDS round trip; left alone, hipcc sinks every ds_read to just in front of its first use):
    //   step s:  [ds_read: B operands of step s + 1, staged values of conversion slot s]  |  [MFMAs of step s, interleaved
    //            over the wave's blocks so that consecutive MFMAs never share an accumulator, + split / ds_write of the values
    //            read in step s - 1]
    __device__ __forceinline__ void k_loop(const Op (&A)[BPW][KS], f32x16 (&acc)[BPW], int cur, const float *stash, int64_t next2, int rows_total,
                                           int off) const {
        const unsigned short *tc = tiles + cur * (3 * PLANE);
        const float *s_ = sp();
        unsigned short *n_ = np(cur ^ 1);
        constexpr int CONV = G::CONV;
        constexpr int PPS = (8 * BPW) / CONV;                // pairs per conversion step
        Op b = read_b<SB>(tc, j, h, 0);
        float cv[2 * PPS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Op bn = b;
            if (s + 1 < KS) bn = read_b<SB>(tc, j, h, s + 1);
            float cn[2 * PPS];
            if (s < CONV && !(TN_B3_ABLATE & 1)) {
#pragma unroll
                for (int u = 0; u < PPS; ++u) {
                    const int p = PPS * s + u, bq = p >> 3, q = p & 7;
                    cn[2 * u] = s_[(32 * bq + 2 * q) * 32];
                    cn[2 * u + 1] = s_[(32 * bq + 2 * q + 1) * 32];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == CONV + 1) {
                // every staged value has been read AND consumed (split in step CONV): the area is free for the tile after next,
                // which then has the rest of the k loop to arrive
                if (!(TN_B3_ABLATE & 2)) request(stash, next2, rows_total, off);
            }
            // six partial products, small terms first, blocks interleaved (consecutive MFMAs never share an accumulator).  The
            // split / ds_write of the values read one step ago is cut into micro-steps of 2-3 instructions, one (two for
            // H = 128) pinned behind every MFMA: with one wave per SIMD an instruction overlaps with the matrix pipe only while
            // an MFMA is executing, i.e. in the ~28 cycles behind each MFMA's issue; clustered behind the MFMAs (hipcc's
            // choice, whatever sched_group_barrier asks for) the conversion runs while the pipe is idle
            const bool conv = s >= 1 && s <= CONV && !(TN_B3_ABLATE & 1);
            constexpr int NMS = 6 * PPS, NMF = 6 * BPW, PER = (NMS + NMF - 1) / NMF;      // micro-steps / MFMAs per step
            unsigned cu[4];
            float cf[4];
            auto micro = [&](int m) {
                if (!conv || m >= NMS) return;
                const int u = m / 6, mm = m % 6;
                const unsigned MSK = 0xffff0000u;
                if (mm == 0) { cu[0] = __float_as_uint(cv[2 * u]) & MSK; cu[1] = __float_as_uint(cv[2 * u + 1]) & MSK; }
                else if (mm == 1) { cf[0] = cv[2 * u] - __uint_as_float(cu[0]); cf[1] = cv[2 * u + 1] - __uint_as_float(cu[1]); }
                else if (mm == 2) { cu[2] = __float_as_uint(cf[0]) & MSK; cu[3] = __float_as_uint(cf[1]) & MSK; }
                else if (mm == 3) { cf[2] = cf[0] - __uint_as_float(cu[2]); cf[3] = cf[1] - __uint_as_float(cu[3]); }
                else if (mm == 4) {
                    cu[0] = __builtin_amdgcn_perm(cu[1], cu[0], 0x07060302u);
                    cu[2] = __builtin_amdgcn_perm(cu[3], cu[2], 0x07060302u);
                    cu[1] = __builtin_amdgcn_perm(__float_as_uint(cf[3]), __float_as_uint(cf[2]), 0x07060302u);
                } else {
                    const int p = PPS * (s - 1) + u, bq = p >> 3, q = p & 7;
                    unsigned short *d = n_ + 32 * bq + 2 * q;
                    *reinterpret_cast<unsigned *>(d) = cu[0];
                    *reinterpret_cast<unsigned *>(d + PLANE) = cu[2];
                    *reinterpret_cast<unsigned *>(d + 2 * PLANE) = cu[1];
                }
            };
            int g = 0;
            auto term = [&](const u32x4 &(*wa)(const Op &), const u32x4 &(*xb)(const Op &)) {
#pragma unroll
                for (int bq = 0; bq < BPW; ++bq) {
                    acc[bq] = tn::b3::mfma16(wa(A[bq][s]), xb(b), acc[bq]);
#pragma unroll
                    for (int e = 0; e < PER; ++e) micro(g * PER + e);
                    ++g;
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            term([](const Op &o) -> const u32x4 & { return o.lo; }, [](const Op &o) -> const u32x4 & { return o.hi; });
            term([](const Op &o) -> const u32x4 & { return o.hi; }, [](const Op &o) -> const u32x4 & { return o.lo; });
            term([](const Op &o) -> const u32x4 & { return o.mid; }, [](const Op &o) -> const u32x4 & { return o.mid; });
            term([](const Op &o) -> const u32x4 & { return o.mid; }, [](const Op &o) -> const u32x4 & { return o.hi; });
            term([](const Op &o) -> const u32x4 & { return o.hi; }, [](const Op &o) -> const u32x4 & { return o.mid; });
            term([](const Op &o) -> const u32x4 & { return o.hi; }, [](const Op &o) -> const u32x4 & { return o.hi; });
            __builtin_amdgcn_sched_barrier(0);
            b = bn;
#pragma unroll
            for (int u = 0; u < 2 * PPS; ++u) cv[u] = cn[u];
        }
    }
    // end of a tile: NSTORES vector-memory stores were issued behind the request of k_loop (retired in order: once at most
    // NSTORES operations are outstanding the request has landed), LDS writes of the conversion retired, then the barrier
    template <int NSTORES>
    __device__ __forceinline__ void tile_barrier() const {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTORES) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
};

template <int H, bool LAST>
__global__ __launch_bounds__(B3Geom<H>::THREADS) void fwd_b3_kernel(FwdLayerArgs a, int64_t n, float *__restrict__ stash, float *__restrict__ y)
{
    using G = B3Geom<H>;
    constexpr int KS = G::KS, BPW = G::BPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Stream<H> st;
    st.init(lds_raw, n);
    float *bias_s = reinterpret_cast<float *>(lds_raw + G::STREAMS * G::STREAM_B);
    for (int e = threadIdx.x; e < H; e += blockDim.x) bias_s[e] = e < a.N ? a.B[e] : 0.0f;
    if (st.iters == 0) return;
    const int j = st.j, h = st.h, lane = st.lane;
    Op A[BPW][KS];
#pragma unroll
    for (int bq = 0; bq < BPW; ++bq) {
        const int row = 32 * (BPW * st.wib + bq) + j;
        load_weights_rows<KS>(a.W, a.K, row, row < a.N, h, A[bq]);
    }
    st.prologue(stash, a.rows_total, a.off_in);
    int cur = 0;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it < st.iters; ++it) {
        f32x16 acc[BPW];
#pragma unroll
        for (int bq = 0; bq < BPW; ++bq)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_s + 32 * (BPW * st.wib + bq) + 8 * q + 4 * h);
                acc[bq][4 * q] = b4[0]; acc[bq][4 * q + 1] = b4[1]; acc[bq][4 * q + 2] = b4[2]; acc[bq][4 * q + 3] = b4[3];
            }
        st.k_loop(A, acc, cur, stash, st.tile_of(it + 2), a.rows_total, a.off_in);
        const int64_t tile = st.tile_of(it);
#pragma unroll
        for (int bq = 0; bq < BPW; ++bq) {
            const int ob = BPW * st.wib + bq;
            tn::pin16(acc[bq]);
            if constexpr (!LAST) {
                acc[bq] = tn::relu16(acc[bq]);
                if (!(TN_B3_ABLATE & 4) || acc[bq][0] == 123.f) wreg_store_block(urow(stash, tile * a.rows_total + a.off_out), ob, j, h, acc[bq]);
            } else if (a.N == H) {
                // full-width output (the feature stacks: y = 256 / 128 features): pre-activation rows through the SGPR-base stores,
                // y as four 16-byte stores per block -- a fixed number of vector-memory operations behind the k loop's request,
                // so that the tile ends with a counted wait instead of draining them
                const int64_t row = tile * 32 + j;
                const bool valid = row < n;
                f32x16 pre;
#pragma unroll
                for (int r = 0; r < 16; ++r) pre[r] = valid ? acc[bq][r] : 0.0f;
                wreg_store_block(urow(stash, tile * a.rows_total + a.off_out), ob, j, h, pre);
                float *yr = y + (valid ? row : 0) * H + 32 * ob + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = tn::apply_act(acc[bq][4 * q + u], a.out_act);
                    if (valid) *reinterpret_cast<f32x4 *>(yr + 8 * q) = v;
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
                        outp[(u + 8 * q) * 32] = ok ? acc[bq][4 * q + u] : 0.0f;
                        v[u] = tn::apply_act(acc[bq][4 * q + u], a.out_act);
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
        if constexpr (!LAST) {
            if (a.off_bits >= 0) {                  // (wave-uniform) ReLU bits of the wave's blocks for the data-gradient kernel
#pragma unroll
                for (int bq = 0; bq < BPW; ++bq) {
                    unsigned *bits = reinterpret_cast<unsigned *>(urow(stash, tile * a.rows_total + a.off_bits + 2 * (BPW * st.wib + bq)));
                    bits[lane] = relu_bits(acc[bq]);
                }
            }
            st.template tile_barrier<16 * BPW>();   // (with the bit rows 17 BPW stores follow the request: the bound still covers it)
        } else {
            if (a.N == H) st.template tile_barrier<20 * BPW>();   // (a tile always holds a valid sample: the y stores are issued)
            else st.template tile_barrier<0>();                   // (narrow outputs: conditional stores, count nothing)
        }
        cur ^= 1;
    }
}

// data gradient twin: the wave's blocks are input-feature blocks kb (A = W^T rows), B = the incoming gradient tile
template <int H>
__global__ __launch_bounds__(B3Geom<H>::THREADS) void dgrad_b3_kernel(DgradArgs a, int64_t n, float *__restrict__ stash)
{
    using G = B3Geom<H>;
    constexpr int KS = G::KS, BPW = G::BPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Stream<H> st;
    st.init(lds_raw, n);
    if (st.iters == 0) return;
    const int j = st.j, h = st.h, lane = st.lane;
    Op A[BPW][KS];
#pragma unroll
    for (int bq = 0; bq < BPW; ++bq) load_weights_cols<KS>(a.W, a.K, 32 * (BPW * st.wib + bq) + j, h, A[bq]);
    st.prologue(stash, a.rows_total, a.off_gin);
    int cur = 0;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it < st.iters; ++it) {
        const int64_t tile = st.tile_of(it);
        unsigned mbits[BPW];
#pragma unroll
        for (int bq = 0; bq < BPW; ++bq)
            mbits[bq] = reinterpret_cast<const unsigned *>(urow(stash, tile * a.rows_total + a.off_bits + 2 * (BPW * st.wib + bq)))[lane];
        f32x16 acc[BPW];
#pragma unroll
        for (int bq = 0; bq < BPW; ++bq)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bq][r] = 0.0f;
        st.k_loop(A, acc, cur, stash, st.tile_of(it + 2), a.rows_total, a.off_gin);
#pragma unroll
        for (int bq = 0; bq < BPW; ++bq) {
            tn::pin16(acc[bq]);
            f32x16 res;
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = mask_keep(acc[bq][r], mbits[bq], r);
            if (!(TN_B3_ABLATE & 4) || res[0] == 123.f) wreg_store_block(urow(stash, tile * a.rows_total + a.off_gout), BPW * st.wib + bq, j, h, res);
        }
        st.template tile_barrier<16 * BPW>();
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW[n][k] += sum_s G[n][s] A[k][s],  db[n] += sum_s G[n][s]   (N == K == H)
//
// The reduction index is the SAMPLE: a 32-sample tile is two 16-wide k steps.  Four waves per workgroup (H = 256: one per SIMD,
// 512 registers), a wave owns BN x BK output tiles of 32 x 32 for the whole launch (4 x 4 for H = 256: 256 accumulator
// registers -- the whole accumulator file -- and 24 ds_read_b128 for 96 MFMAs per k step).  Both operands are rows
// [feature][32 samples] of the workspace; a k step needs their 16-sample halves as bf16 triplets [row][16 samples] in LDS
// (operand of lane (i, h) = samples 8 h .. 8 h + 7 of row i: one ds_read_b128 per term).  Pipeline over HALF tiles, two LDS
// buffers, one barrier per half tile:
//     while the MFMAs of half tile t run from buffer t & 1, every thread splits the 2 H x 16 values of half tile t + 1 it holds
//     in registers (8 x 4 consecutive samples of a row, loaded while half tile t - 1 was being multiplied) and writes them to
//     the other buffer; then it requests half tile t + 2.
// The bias gradient is summed on the way by the threads that convert G rows (a thread meets the same rows in every tile).
// ------------------------------------------------------------------------------------------------
template <int H, int BN, int BK>
__global__ __launch_bounds__(64 * (H / 32 / BN) * (H / 32 / BK)) void wgrad_b3_kernel(WgradArgs a, int64_t n, const float *__restrict__ stash)
{
    constexpr int TH = 64 * (H / 32 / BN) * (H / 32 / BK);     // threads: 4 waves (one per SIMD, 4 x 4 tiles each) or 8 (two per SIMD, 4 x 2)
    constexpr int NR = 2 * H;                          // rows per half tile: G rows [0, H), A rows [H, 2 H)
    constexpr int RS = 24;                             // bf16 elements per LDS row: 16 samples + 8 pad = 48 B (3 x 16 B: odd)
    constexpr int PLANE = NR * RS;                     // bf16 elements per term plane
    constexpr int BUF = 3 * PLANE;                     // ... per buffer
    constexpr int NCH = (NR * 4) / TH;                 // 16-byte chunks (4 samples of a row) per thread and half tile
    constexpr int WK = (H / 32) / BK;                  // waves along k
    static_assert(NCH * TH == NR * 4 && NCH >= 2 && (NCH & 1) == 0, "the waves own all tiles, every thread holds G and A chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned short *lds = reinterpret_cast<unsigned short *>(lds_raw);
    const int lane = tn::lane_id(), i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int tn0 = (wave / WK) * BN, tk0 = (wave % WK) * BK;
    f32x16 acc[BN][BK];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn)
#pragma unroll
        for (int bk = 0; bk < BK; ++bk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bn][bk][r] = 0.0f;
    // this thread's chunks of a half tile: chunk id = threadIdx.x + TH c -> row id / 4, samples 4 (id % 4) .. + 3 of the half
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
    const int64_t n_half = 2 * n_tiles;                           // half tile t: tile t / 2, samples 16 (t & 1) .. + 15
    const int64_t stride = gridDim.x;                             // tiles are dealt round-robin to the workgroups
    struct Bases { const global_char *g, *a; };                   // wave-uniform: the half tile's G rows and A rows
    auto half_src = [&](int64_t it) {                             // `it`-th half tile of this workgroup
        int64_t tile = blockIdx.x + (it >> 1) * stride;
        tile = tile < n_tiles ? tile : n_tiles - 1;
        return Bases{wave_uniform_global(stash + (tile * a.rows_total + a.off_g) * 32 + 16 * (it & 1)),
                     wave_uniform_global(stash + (tile * a.rows_total + a.off_a) * 32 + 16 * (it & 1))};
    };
    const int64_t my_tiles = (int64_t)blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + stride - 1) / stride : 0;
    const int64_t iters = 2 * my_tiles;
    if (iters == 0) return;
    (void)n_half;
    // The half tile in flight lives in ONE register set: chunk c of half tile k + 1 is requested right behind the conversion of
    // chunk c of half tile k (its registers are free at that moment) and consumed one half tile later -- a whole iteration
    // (~3000 matrix-pipe cycles) for every load, and at each conversion exactly NCH - 1 younger loads are outstanding, so the
    // wait is a constant vmcnt(NCH - 1).  (Two sets swapped per iteration made hipcc's waitcnt pass drain the queue at the top
    // of every half tile; a 64-bit address per lane makes it build the address in the load's destination registers: the
    // loads take the SGPR-base form, wave-uniform base + the thread's 32-bit byte offset.)
    f32x4 st[NCH];
    unsigned src_boff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) src_boff[c] = (unsigned)src_off[c] * 4u;
    auto load_chunk = [&](const Bases &base, int c) {
        unsigned off = src_boff[c];
        asm volatile("" : "+v"(off));                          // (keeps the zero-extension at the access: SGPR-base form)
        st[c] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>((c < NCH / 2 ? base.g : base.a) + off);
    };
    auto convert_chunk = [&](int c, unsigned short *buf) {
        unsigned h0, m0, l0, h1, m1, l1;
        tn::b3::split2(st[c][0], st[c][1], h0, m0, l0);
        tn::b3::split2(st[c][2], st[c][3], h1, m1, l1);
        unsigned short *d = buf + dst_off[c];
        *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(d + 2 * PLANE) = make_uint2(l0, l1);
        if (c < NCH / 2) dbacc[c] += (st[c][0] + st[c][1]) + (st[c][2] + st[c][3]);
    };
    // The same conversion cut into 12 micro-steps of 2-4 instructions per chunk (split of the first value pair: 0-4, of the second:
    // 5-9, LDS writes + bias sum + request of the chunk's successor: 10-11).  One wave per SIMD: an instruction overlaps with the
    // matrix pipe only while an MFMA is EXECUTING, i.e. in the ~28 cycles behind each MFMA's issue -- a block of twenty VALU
    // instructions behind eight MFMAs (what hipcc schedules) runs while the pipe is idle.  The k loop below pins one micro-step
    // behind every MFMA.
    unsigned cu[8];          // conversion state carried between micro-steps
    float cf[4];
    auto micro = [&](int c, int m, unsigned short *buf, const Bases &nb) {
        const unsigned MSK = 0xffff0000u;
        const int e = m >= 5 ? 2 : 0;                        // value pair (e, e + 1) of the chunk
        const int mm = m >= 5 ? m - 5 : m;
        if (m < 10) {
            if (mm == 0) { cu[0] = __float_as_uint(st[c][e]) & MSK; cu[1] = __float_as_uint(st[c][e + 1]) & MSK; }
            else if (mm == 1) { cf[0] = st[c][e] - __uint_as_float(cu[0]); cf[1] = st[c][e + 1] - __uint_as_float(cu[1]); }
            else if (mm == 2) { cu[2] = __float_as_uint(cf[0]) & MSK; cu[3] = __float_as_uint(cf[1]) & MSK; }
            else if (mm == 3) { cf[2] = cf[0] - __uint_as_float(cu[2]); cf[3] = cf[1] - __uint_as_float(cu[3]); }
            else {              // packed pairs: hi -> cu[4 + e/2 .. ], mid, lo
                const int q = e >> 1;
                cu[0 + 0] = __builtin_amdgcn_perm(cu[1], cu[0], 0x07060302u);
                cu[2] = __builtin_amdgcn_perm(cu[3], cu[2], 0x07060302u);
                cu[1] = __builtin_amdgcn_perm(__float_as_uint(cf[3]), __float_as_uint(cf[2]), 0x07060302u);
                if (q == 0) { cu[4] = cu[0]; cu[5] = cu[2]; cu[6] = cu[1]; }      // keep the first pair's (hi, mid, lo) for the 8-byte writes
            }
        } else if (m == 10) {
            unsigned short *d = buf + dst_off[c];
            *reinterpret_cast<uint2 *>(d) = make_uint2(cu[4], cu[0]);
            *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(cu[5], cu[2]);
            *reinterpret_cast<uint2 *>(d + 2 * PLANE) = make_uint2(cu[6], cu[1]);
        } else {
            if (c < NCH / 2) dbacc[c] += (st[c][0] + st[c][1]) + (st[c][2] + st[c][3]);
            load_chunk(nb, c);
        }
    };
    // prologue: half tile 0 converted, half tile 1 in registers
    {
        const Bases b0 = half_src(0), b1 = half_src(1);
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_chunk(b0, c);
#pragma unroll
        for (int c = 0; c < NCH; ++c) { convert_chunk(c, lds); load_chunk(b1, c); }
    }
    __syncthreads();
    int g_off[BN], a_off[BK];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) g_off[bn] = (32 * (tn0 + bn) + i) * RS + 8 * h;
#pragma unroll
    for (int bk = 0; bk < BK; ++bk) a_off[bk] = (H + 32 * (tk0 + bk) + i) * RS + 8 * h;
    auto read_op = [&](const unsigned short *buf, int off) {
        Op o;
        o.hi = *reinterpret_cast<const u32x4 *>(buf + off);
        o.mid = *reinterpret_cast<const u32x4 *>(buf + off + PLANE);
        o.lo = *reinterpret_cast<const u32x4 *>(buf + off + 2 * PLANE);
        return o;
    };
    // one half tile: MFMAs from buffer `cur`; (CONVERT) chunk by chunk, the next half (in `st`) is converted into the other buffer
    // and the half after that requested into the registers just freed.  The last half tile of the workgroup has nothing to
    // convert: it runs as a second instance of the body behind the loop, so that inside the loop conversion and requests are
    // unconditional (a branch around them makes hipcc's counted vmcnt waits collapse to vmcnt(0)).
    auto half_step = [&](int64_t it, int cur, auto convert_tag) {
        constexpr bool CONVERT = decltype(convert_tag)::value;
        const unsigned short *bc = lds + cur * BUF;
        unsigned short *bnx = lds + (cur ^ 1) * BUF;
        const Bases nb = half_src(it + 2 < iters ? it + 2 : it);     // (clamped: loaded, never used)
        Op gop[BN];
#pragma unroll
        for (int bn = 0; bn < BN; ++bn) gop[bn] = read_op(bc, g_off[bn]);
        Op aop = read_op(bc, a_off[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            Op anx = aop;
            if (bk + 1 < BK) anx = read_op(bc, a_off[bk + 1]);
            __builtin_amdgcn_sched_barrier(0);
            // six partial products per output tile, small terms first, tiles interleaved; one conversion micro-step (two for the
            // narrow stack) pinned behind every MFMA
            constexpr int NM = BN * BK * 6, MS = NCH * 12, PER = MS / NM;
            static_assert(PER * NM == MS, "micro-steps divide evenly over the MFMAs");
            auto term = [&](int t, const u32x4 &(*ga)(const Op &), const u32x4 &(*ab)(const Op &)) {
#pragma unroll
                for (int bn = 0; bn < BN; ++bn) {
                    acc[bn][bk] = tn::b3::mfma16(ga(gop[bn]), ab(aop), acc[bn][bk]);
                    if constexpr (CONVERT && !(TN_B3_ABLATE & 8)) {
#pragma unroll
                        for (int u = 0; u < PER; ++u) {
                            const int m = ((bk * 6 + t) * BN + bn) * PER + u;
                            micro(m / 12, m % 12, bnx, nb);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            term(0, [](const Op &o) -> const u32x4 & { return o.lo; }, [](const Op &o) -> const u32x4 & { return o.hi; });
            term(1, [](const Op &o) -> const u32x4 & { return o.hi; }, [](const Op &o) -> const u32x4 & { return o.lo; });
            term(2, [](const Op &o) -> const u32x4 & { return o.mid; }, [](const Op &o) -> const u32x4 & { return o.mid; });
            term(3, [](const Op &o) -> const u32x4 & { return o.mid; }, [](const Op &o) -> const u32x4 & { return o.hi; });
            term(4, [](const Op &o) -> const u32x4 & { return o.hi; }, [](const Op &o) -> const u32x4 & { return o.mid; });
            term(5, [](const Op &o) -> const u32x4 & { return o.hi; }, [](const Op &o) -> const u32x4 & { return o.hi; });
            aop = anx;
        }
        // next half converted by everybody, this one read by everybody: LDS traffic retired, then a RAW barrier -- __syncthreads()
        // carries vmcnt(0) and would wait for the loads just requested
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int cur = 0;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it + 1 < iters; ++it) {
        half_step(it, cur, std::true_type{});
        cur ^= 1;
    }
    half_step(iters - 1, cur, std::false_type{});
    // ---- flush: full-line atomics (lanes = consecutive columns of one weight row) ----
    // (all MFMA results are complete before the first accumulator read whichever way the loop was left: see tn::pin16 for the
    // hipcc wait-state bug; pinning 256 accumulator registers in VGPRs at once is not an option here)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int bn = 0; bn < BN; ++bn)
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            const int k = 32 * (tk0 + bk) + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * (tn0 + bn) + frow(r, h);
                atomicAdd(&a.gW[(int64_t)nn * a.K + k], acc[bn][bk][r]);
            }
        }
    // bias gradient: the four threads of a row (quarters of the 16 samples) are neighbours
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
    constexpr size_t lds_bytes = (size_t)2 * 3 * (2 * H) * 24 * 2;
    auto kern = wgrad_b3_kernel<H, BN, BK>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd(bf16x3): cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int per_cu = lds_bytes * 2 <= (size_t)LDS_LIMIT_BYTES ? 2 : 1;
    kern<<<dim3((unsigned)std::min<int64_t>(n_tiles, 256 * per_cu)), dim3(64 * (H / 32 / BN) * (H / 32 / BK)), lds_bytes, s>>>(w, n, stash);
    return tn::check_launch("wgrad_b3_kernel");
}

template <int H, bool LAST>
int launch_fwd(const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s)
{
    using G = B3Geom<H>;
    auto kern = fwd_b3_kernel<H, LAST>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_fwd(bf16x3): cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + G::STREAMS - 1) / G::STREAMS, 256));
    kern<<<dim3((unsigned)bl), dim3(G::THREADS), G::lds_bytes, s>>>(f, n, stash, y);
    return tn::check_launch("fwd_b3_kernel");
}

template <int H>
int launch_dgrad(const DgradArgs &d, int64_t n, float *stash, hipStream_t s)
{
    using G = B3Geom<H>;
    if (d.off_bits < 0) return tn::fail(TN_E_CONFIG, "mlp_bwd(bf16x3): the data gradient takes its ReLU masks as bit rows");
    auto kern = dgrad_b3_kernel<H>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd(bf16x3): cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + G::STREAMS - 1) / G::STREAMS, 256));
    kern<<<dim3((unsigned)bl), dim3(G::THREADS), G::lds_bytes, s>>>(d, n, stash);
    return tn::check_launch("dgrad_b3_kernel");
}

}  // namespace

namespace tn {
namespace layers {

__attribute__((visibility("hidden"))) int launch_fwd_b3(int H, bool last, const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s)
{
    if (H == 256) return last ? launch_fwd<256, true>(f, n, stash, y, s) : launch_fwd<256, false>(f, n, stash, y, s);
    if (H == 128) return last ? launch_fwd<128, true>(f, n, stash, y, s) : launch_fwd<128, false>(f, n, stash, y, s);
    return tn::fail(TN_E_CONFIG, "mlp_fwd(bf16x3): width 128 or 256");
}

__attribute__((visibility("hidden"))) int launch_dgrad_b3(int H, const DgradArgs &d, int64_t n, float *stash, hipStream_t s)
{
    if (H == 256) return launch_dgrad<256>(d, n, stash, s);
    if (H == 128) return launch_dgrad<128>(d, n, stash, s);
    return tn::fail(TN_E_CONFIG, "mlp_bwd(bf16x3): width 128 or 256");
}

__attribute__((visibility("hidden"))) int launch_wgrad_b3(int H, const WgradArgs &w, int64_t n, const float *stash, hipStream_t s)
{
    if (w.first || w.N != H || w.K != H) return tn::fail(TN_E_CONFIG, "mlp_bwd(bf16x3): square hidden layers only");
#ifndef TN_B3_WGRAD_BK256
#define TN_B3_WGRAD_BK256 2
#endif
    if (H == 256) return launch_wgrad<256, 4, TN_B3_WGRAD_BK256>(w, n, stash, s);
    if (H == 128) return launch_wgrad<128, 2, 2>(w, n, stash, s);
    return tn::fail(TN_E_CONFIG, "mlp_bwd(bf16x3): width 128 or 256");
}

}  // namespace layers
}  // namespace tn
