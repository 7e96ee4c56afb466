// The wide stacks (Vanilla 256 x 10, reference models.py:7-28,59-68, run.py:131; Cobafa 128 x 6, models.py:239-247) as ONE persistent
// launch across all layers -- the north star's "persistent-threadblock kernel ... with weights and ray-packed samples staged on chip":
// no activation crosses HBM between layers.  Round 6.
//
// The layer-by-layer kernels (mlp_f2_layers.hip) keep one layer's weights in registers and stream the activations of all samples
// through them: 1 KB read + 1 KB written per sample and layer, 0.41 ms per 256 x 256 layer and 2^20 samples at the copy roofline of
// that access pattern, against 0.16 ms of fp16 matrix time.  Here the roles are swapped:
//
//   * ACTIVATIONS STAY IN REGISTERS.  A wave owns 32 samples for the whole stack.  With the transposed formulation of
//     mlp_device.h / mlp_f2_heads.h the D registers of a layer's 32 x 32 output block, after bias + ReLU + conversion, ARE the B
//     operands of two k steps of the next layer (k step 2 ob + half = registers 8 half .. + 7 of block ob): H / 2 registers of
//     packed fp16 pairs (hi plane + lo plane) per lane hold a sample column, and nothing is exchanged between waves.
//   * WEIGHTS STREAM THROUGH LDS.  fused_pack_kernel writes every layer once per call as the exact byte stream the kernel consumes:
//     per (output block ob, k step s) a 2 KB "pair" = the A operand of the 64 lanes as fp16 hi values (1 KB) and lo values (1 KB),
//     scaled by the layer's power-of-two 2^(9 - e_W); two output blocks at a time, k step by k step (below).  The stream of a whole
//     stack is 2.4 MB (Vanilla): resident in every XCD's 4 MB L2.  A workgroup (4 waves, one per SIMD) pulls it through a ring of
//     four 32 KB chunks with LDS-direct loads (buffer_load_dwordx4 ... lds: SGPR base + one constant lane offset, no address
//     arithmetic, no registers), two chunks ahead, one 1 KB piece per wave and k step; all four waves read the same operands
//     (ds_read_b128, the 64 lanes' 16 B contiguous: conflict-free, 256 B per clock).  Per 256 x 256 layer a workgroup moves 256 KB
//     from L2 for 128 samples -- 21 B per CU and clock of the 64 the L2 delivers -- and reads 1 MB out of LDS (4 k cycles) in the
//     ~12 k cycles its 1536 MFMAs take.
//   * ONE BARRIER PER CHUNK (every 8 k steps).  s_waitcnt vmcnt(8) (the wave's own share of chunk g + 1 has landed; only the 8 pieces
//     of chunk g + 2 may be outstanding -- vector memory operations retire in order, so any other load or store issued in between
//     only makes the wait stricter), s_barrier (everybody's share has landed, everybody is done with chunk g - 1), then the pieces
//     of chunk g + 3 go into the slot of g - 1.  Chunks g and g + 1 are readable after boundary g.
//   * A LONE WAVE PER SIMD MUST NOT PUT ANYTHING BETWEEN TWO MFMAS ON THE SAME ACCUMULATOR (one extra issue slot there costs ~43
//     cycles, MI355X_MICROARCH.md; the first draft of this kernel, 48 chained MFMAs per output block with the epilogue dealt out
//     between them, ran at 0.38 of the fp16 matrix rate).  Two output blocks are therefore in flight at once and their MFMAs
//     alternate: c0 += A0.lo B.hi, c1 += A1.lo B.hi, c0 += A0.hi B.lo, ... -- consecutive MFMAs never share an accumulator, and at
//     most five single-issue instructions go into a gap: the operand reads two k steps ahead, the LDS-direct piece, and the
//     epilogue of the PREVIOUS two blocks (bias, ReLU, conversion to the next layer's operands: ~12 instructions per value pair).
//     The last two blocks of a layer are finished in the shadow of the next layer's first k steps (which only need the operands of
//     the earlier blocks); the operand pipeline never drains, across layers and across rounds.
//   * ALL LDS READS OF THE MAIN LOOP ARE INLINE ASSEMBLY WITH HAND-COUNTED s_waitcnt lgkmcnt(N).  hipcc's waitcnt pass lumps the
//     waits of a software pipeline into lgkmcnt(0) every fourth operand pair, which drains the prefetched pairs as well.
//   * SCALES WITHOUT A COLUMN MAXIMUM.  The f16x2 arithmetic needs one power-of-two scale per sample and layer that keeps the
//     scaled activations inside fp16's range.  The layer kernels take the exact column maximum -- here that would serialise a layer's
//     epilogue behind all of its output blocks.  Instead the scale comes from a bound known BEFORE the layer's first epilogue:
//     |y_i| <= max_i ||W_i||_1 max_k |x_k| + max |b|   (Hoelder), with max |x| taken in the previous layer's epilogue (one v_max3 per
//     value pair).  The bound maps to [2^13, 2^14) (fp16 overflows at 2^16).  How loose may it be?  An entry a = x s is carried as
//     hi + lo with an absolute error of at most 2^-25 (fp16's subnormal quantum on the lo term) or 2^-22 |a|; the dot product sees
//     that error relative to the column's largest entry.  With the bound a factor R above the true column maximum the latter sits
//     at 2^13.5 / R, so the error relative to it is max(2^-22, R 2^-38.5): fp32's own rounding (2^-24) is reached only at
//     R = 2^14.  For torch-initialised 256-wide layers R is 2^3 - 2^5.  So every block's conversion is independent of the others.
//
// Arithmetic per product block is that of mlp_f2_layers.hip: x s = hi + lo (two fp16 terms, 22 significand bits), three MFMAs
// (lo hi, hi lo, hi hi) on v_mfma_f32_32x32x16_f16 with fp32 accumulation, scales taken out of the accumulator in the epilogue.
#include "mlp_layers.h"
#include "mlp_f2_heads.h"
#include <algorithm>
#include <utility>

namespace {

using namespace tn::layers;
using namespace tn::mlp;
using tn::f32x16;
using tn::f32x4;

struct Op2 { u32x4h hi, lo; };
typedef __attribute__((address_space(3))) unsigned char lds_u8;      // (explicit: an LDS pointer that went through a struct must not turn generic)
__device__ __forceinline__ u32x4h lds_read16(const lds_u8 *p) { return *reinterpret_cast<const __attribute__((address_space(3))) u32x4h *>(p); }
__device__ __forceinline__ f32x4 lds_read4f(const lds_u8 *p) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(p); }

// Ablation switches for timing experiments (scripts/build_dev_lib.sh ... -DTN_FUSED_ABL=bits; results are WRONG with any bit set; the
// shipped build has none): 1 no LDS-direct requests, 2 no chunk barrier, 4 no y stores, 8 no operand reads from LDS, 16 no MFMAs, 32 no waits for the operand reads
#ifndef TN_FUSED_ABL
#define TN_FUSED_ABL 0
#endif
constexpr int PAIR_B = 2048;                 // one A operand: 64 lanes x 16 B of hi values + the same of lo values
constexpr int CHUNK_PAIRS = 16;
constexpr int CHUNK_B = CHUNK_PAIRS * PAIR_B;
constexpr int NRING = 4;
// Waves per workgroup (one workgroup per CU).  H = 256: one per SIMD -- a sample column and its successor fill the 512-register budget.
// H = 128: two per SIMD (256 registers each) where the kernel fits them -- the inference forward (222 registers) and the gradient chain (188);
// the training forward would spill 75.  The epilogue of a 128-wide layer is twice as long relative to its MFMAs, and a second wave's MFMAs
// run behind it: measured 0.86 -> 0.76 ms (Cobafa forward, 2^20 samples), chain 0.89 -> 0.84 ms.  All waves share the ring; a wave
// fetches 32 / NW of a chunk's 32 pieces.  CHAIN_OR_INFER: the kernel is fused_chain_kernel or fused_fwd_kernel<H, false>.
template <int H, bool CHAIN_OR_INFER> constexpr int NWAVES = (H == 128 && CHAIN_OR_INFER) ? 8 : 4;
constexpr int KS0 = 4;                       // k steps of the first layer (<= 64 input rows)

struct FusedArgs {
    const unsigned char *stream;     // packed weights of all layers, consumption order (tn_fused_pack)
    const float *bias;               // [L][H], zero padded
    const float *consts;             // [L][4]: 1 / s_W, max_i ||W_i||_1, max |bias|, -
    int L;                           // layers in the stream: 0 .. L - 2 hidden (ReLU), L - 1 the output layer (no activation)
    int n_chunks;                    // chunks per pass over the stream
    int out_act;
    const float *e_rows;             // first-layer inputs as [slot][32 samples] rows (enc_rows_kernel): tile t at rows t e_rows_total + e_off
    int e_rows_total;
    int64_t e_off;
    float *y;                        // [n][H] row-major (nullptr: not wanted -- TN_MLP_ROWS_ONLY)
    // training forward (STASH): what the layer-wise backward reads (mlp_bwd_layers.hip RowMap, slab layout)
    float *rows;                     // workspace: tile t, row set at row offset `off` = rows [t rows_total + off, ...)
    int rows_total;
    int out_linear;                  // 1: layer L - 1 is the stack's output layer (no activation); 0: every layer of the stream is hidden (TN_MLP_SKIP_LAST)
    int64_t off_out[TN_MLP_MAX_LAYERS];      // row offset of layer l's output (activation H_l; the output layer: buffer A)
    int64_t off_bits[TN_MLP_MAX_LAYERS];     // ... of its ReLU bit rows (2 per 32-feature block), hidden layers only
    float *tail;                     // tail[tail_idx[l]] receives the largest |value| of layer l's INPUT rows (scale of a weight-gradient launch); idx < 0: not wanted
    int tail_idx[TN_MLP_MAX_LAYERS];
    int64_t off_in;                  // data-gradient chain: row offset of the gradient it starts from (H rows per tile)
};

struct PackArgs {
    const float *W[TN_MLP_MAX_LAYERS];
    const float *B[TN_MLP_MAX_LAYERS];
    int N[TN_MLP_MAX_LAYERS], K[TN_MLP_MAX_LAYERS];
    int64_t off[TN_MLP_MAX_LAYERS];  // byte offset of layer l in the stream
    int L, H;
    int transpose;                   // data-gradient chain: layer l's A operand is W_l^T (no bias)
    unsigned char *stream;
    float *bias, *consts;
};

// 2^(13 - e) and its inverse for an upper bound u >= 0 with exponent e: u s in [2^13, 2^14)
__device__ __forceinline__ void bound_scales(float u, float &s, float &inv) {
    int se = 267 - (int)(__float_as_uint(u) >> 23);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
}

// ------------------------------------------------------------------------------------------------------------------------------
// gridDim.y workgroups per layer: the layer's scale, row-norm / bias bounds (every workgroup computes them: 256 KB out of L2; the first one
// stores them), and a gridDim.y-th of its weights as the kernel's operand stream
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fused_pack_kernel(PackArgs a)
{
    __shared__ float red[3][16];
    const int l = blockIdx.x, H = a.H;
    const int N = a.transpose ? a.K[l] : a.N[l], K = a.transpose ? a.N[l] : a.K[l];         // rows / columns of the A operand
    const int ldw = a.K[l];
    const float *W = a.W[l];
    auto elem = [&](int r, int k) -> float { return a.transpose ? W[(int64_t)k * ldw + r] : W[(int64_t)r * ldw + k]; };
    const int lane = tn::lane_id(), wave = threadIdx.x >> 6, n_waves = (int)(blockDim.x >> 6);
    float wmax = 0.0f, n2 = 0.0f, bmax = 0.0f;
    // row sums of |A|, every load coalesced.  A = W: a wave per row, lanes along it.  A = W^T: a row of A is a column of W -- lanes along
    // the rows of A, the waves share the columns and meet in LDS.  (The first form walked a row per thread: 0.07 - 0.10 ms per call.)
    if (!a.transpose) {
        for (int r = wave; r < N; r += n_waves) {
            float s2 = 0.0f;
            for (int k = lane; k < K; k += 64) { const float w = fabsf(W[(int64_t)r * ldw + k]); wmax = fmaxf(wmax, w); s2 += w; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
            n2 = fmaxf(n2, s2);
        }
        for (int r = threadIdx.x; r < N; r += blockDim.x) bmax = fmaxf(bmax, fabsf(a.B[l][r]));
    } else {
        __shared__ float colsum[4][256 + 1];
        const int rb = threadIdx.x & 255, part = threadIdx.x >> 8, parts = (int)(blockDim.x >> 8);      // (N <= H <= 256)
        float s2 = 0.0f;
        if (rb < N) for (int k = part; k < K; k += parts) { const float w = fabsf(W[(int64_t)k * ldw + rb]); wmax = fmaxf(wmax, w); s2 += w; }
        colsum[part][rb] = s2;
        __syncthreads();
        if (part == 0 && rb < N) { float t = 0.0f; for (int q = 0; q < parts; ++q) t += colsum[q][rb]; n2 = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64)); n2 = fmaxf(n2, __shfl_xor(n2, o, 64)); bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
    }
    if (lane == 0) { red[0][wave] = wmax; red[1][wave] = n2; red[2][wave] = bmax; }
    __syncthreads();
    wmax = n2 = bmax = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { wmax = fmaxf(wmax, red[0][w]); n2 = fmaxf(n2, red[1][w]); bmax = fmaxf(bmax, red[2][w]); }
    float s_w, inv_w;
    f2_scales(wmax, s_w, inv_w);
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        a.consts[4 * l + 0] = inv_w;
        a.consts[4 * l + 1] = n2 * 1.0005f;             // largest row sum of |w| (fp32 rounding of the sum)
        a.consts[4 * l + 2] = bmax;
        a.consts[4 * l + 3] = 0.0f;
    }
    if (blockIdx.y == 0) for (int e = threadIdx.x; e < H; e += blockDim.x) a.bias[l * H + e] = (e < N && !a.transpose) ? a.B[l][e] : 0.0f;
    const int KS = (l == 0 && !a.transpose) ? KS0 : H / 16;       // (the forward's first layer has <= 64 inputs; the chain has no such layer)
    unsigned char *dst = a.stream + a.off[l];
    for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < (H / 32) * KS * 64; idx += blockDim.x * gridDim.y) {
        // consumption order: group gi = output blocks 2 gi, 2 gi + 1; inside a group k step by k step, the even block first
        const int pair = idx >> 6, ln = idx & 63, gi = pair / (2 * KS), s = (pair - gi * 2 * KS) >> 1, ob = 2 * gi + (pair & 1), i = ln & 31, h = ln >> 5;
        const int row = 32 * ob + i;
        u32x4h hi, lo;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = 2 * p + u, col = 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);       // (mlp_f2_heads.h f2_perm: k block s, half h)
                v[u] = (row < N && col < K) ? elem(row, col) * s_w : 0.0f;
            }
            const f16x2h hh = {(_Float16)v[0], (_Float16)v[1]};
            const f16x2h ll = {(_Float16)(v[0] - (float)hh[0]), (_Float16)(v[1] - (float)hh[1])};
            hi[p] = __builtin_bit_cast(unsigned, hh);
            lo[p] = __builtin_bit_cast(unsigned, ll);
        }
        *reinterpret_cast<u32x4h *>(dst + (int64_t)pair * PAIR_B + ln * 16) = hi;
        *reinterpret_cast<u32x4h *>(dst + (int64_t)pair * PAIR_B + 1024 + ln * 16) = lo;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// ---- LDS reads and waits of the main loop: inline assembly with compile-time offsets / counts (see the file header) ----
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ u32x4h lds16(unsigned addr) {
    u32x4h v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ f32x4 lds16f(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// "at most N LDS operations outstanding"; the operands it guards are listed so that nothing that uses them can be moved above it
template <int N> __device__ __forceinline__ void wait_ops(Op2 &a, Op2 &b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.hi), "+v"(a.lo), "+v"(b.hi), "+v"(b.lo) : "n"(N));
}
template <int N> __device__ __forceinline__ void wait_bias(f32x4 (&b)[2][4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]) : "n"(N));
}
__device__ __forceinline__ float other_half_max(float v) {     // max(v, v of lane ^ 32): no LDS (ds_bpermute would drain the operand queue)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

struct Ring {
    __amdgpu_buffer_rsrc_t rsrc;     // the packed stream
    lds_u8 *lds_wave;                // ring base + the wave's 8 KB share of a chunk (wave-uniform)
    unsigned voff;                   // the lane's byte offset inside a chunk (share + 16 lane)
    unsigned rd0;                    // LDS byte address of the ring + 16 lane
    unsigned cur, nxt;               // ... of the chunk being consumed / of the next one (+ 16 lane)
    unsigned g;                      // chunks consumed so far (wave-uniform)
    unsigned next, n_chunks;         // position in the stream (in chunks) of the chunk whose pieces are being requested
    unsigned iss_slot, iss_off;      // where they go: ring slot, byte offset in the stream

    template <int E> __device__ __forceinline__ void piece() {     // 1 KB of the wave's share
        if (TN_FUSED_ABL & 1) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds_wave + iss_slot * CHUNK_B + E * 1024), 16,
                                                 (int)voff, (int)(iss_off + E * 1024), 0, 0);
    }
    __device__ __forceinline__ void target_next() {                // the chunk the next 8 pieces belong to: g + 3
        iss_slot = (g + 3) & (NRING - 1);
        iss_off = next * CHUNK_B;
        next = next + 1 == n_chunks ? 0 : next + 1;
    }
    // between the last MFMA on chunk g - 1 and the first on chunk g
    template <int NW> __device__ __forceinline__ void boundary() {   // (vmcnt: the wave's pieces of one chunk -- the newest -- may still be in flight)
        if (!(TN_FUSED_ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(32 / NW) : "memory");
        }
        target_next();
        ++g;
        cur = nxt;
        nxt = rd0 + (g & (NRING - 1)) * CHUNK_B;
    }
};

// The two output blocks whose epilogue is pending (the last group of a layer crosses into the next layer)
struct Pending {
    f32x16 c[2];
    f32x4 bias[2][4];                // D-layout bias of both blocks
    float cmul, s_out;
    float *rows; unsigned *bits;     // STASH: where that layer's activation rows / bit rows of this tile go (wave-uniform)
};

template <int H> struct State {
    Op2 P[H / 16], Q[H / 16];        // a sample column as operands: layer input / output, in turn
    Op2 aw[4][2];                    // operand window: k steps t, t + 1, t + 2 (slot t & 3), two blocks each
    Pending pend;
    float xmax;                      // largest scaled output of the layer being finished (this lane's half of the features)
};

// value pair (D registers r0, r0 + 1) of a finished hidden block -> one packed register of the next layer's hi plane and lo plane, in five
// phases of 2 - 4 plain VALU instructions that the caller deals out over the gaps behind consecutive MFMAs.  (The first form used
// v_fma_mixlo/hi_f16 for the conversion and v_dot2_f32_f16 for a sum of squares -- five instructions instead of thirteen -- and was
// SLOWER: beside MFMAs a VOP3P instruction costs about two plain ones.)  The column's largest scaled value goes into `xmax`: the
// next layer's a-priori bound is max_i ||W_i||_1 max |x| + max |b|.
struct PairTmp { float v0, v1, a0, a1; unsigned hi; };
// GRAD (the data-gradient chain): the pair is a gradient -- no bias, no ReLU; it is multiplied by relu'(h) of the activation it flows into,
// taken from that activation's bit rows (`mask`: bit r of the lane's dword of the block, r = R0, R0 + 1), and it is signed
// SPLIT (GRAD, one pair per step): the scale multiply moves from phase 1 to phase 2, so that no gap of the step holds more than five instructions
template <int PH, bool GRAD = false, int R0 = 0, bool SPLIT = false>
__device__ __forceinline__ void pair_phase(PairTmp &t, float c0, float c1, float b0, float b1, float cmul, float s_out, unsigned &hi_out, unsigned &lo_out, float &xmax,
                                           unsigned mask = 0) {
    // (inline assembly: left to itself hipcc rebuilds the v_fma_mix forms out of the plain ones and batches the packing)
    if constexpr (PH == 0) {
        if constexpr (GRAD) { t.v0 = c0 * cmul; t.v1 = c1 * cmul; }
        else { t.v0 = fmaf(c0, cmul, b0); t.v1 = fmaf(c1, cmul, b1); }
    }
    // (one asm statement per phase: between two statements hipcc's hazard recogniser, which cannot see into them, pads dependent ones with s_nop)
    if constexpr (PH == 1) {
        if constexpr (GRAD) {
            // bit R0 of the mask sign-extended to a word, ANDed in: two instructions per value (hipcc turns the C form into
            // and + compare + select through vcc, with wait states); then the scale (exact: s is a power of two)
            if constexpr (SPLIT)
                asm("v_bfe_i32 %2, %4, %5, 1\n\tv_bfe_i32 %3, %4, %6, 1\n\tv_and_b32_e32 %0, %0, %2\n\tv_and_b32_e32 %1, %1, %3"
                    : "+v"(t.v0), "+v"(t.v1), "=&v"(t.a0), "=&v"(t.a1) : "v"(mask), "n"(R0), "n"(R0 + 1));
            else
                asm("v_bfe_i32 %2, %4, %5, 1\n\tv_bfe_i32 %3, %4, %6, 1\n\tv_and_b32_e32 %0, %0, %2\n\tv_and_b32_e32 %1, %1, %3\n\t"
                    "v_mul_f32_e32 %2, %0, %7\n\tv_mul_f32_e32 %3, %1, %7"
                    : "+v"(t.v0), "+v"(t.v1), "=&v"(t.a0), "=&v"(t.a1) : "v"(mask), "n"(R0), "n"(R0 + 1), "v"(s_out));
        } else {
            asm("v_max_f32_e32 %0, 0, %0\n\tv_max_f32_e32 %1, 0, %1\n\tv_mul_f32_e32 %2, %0, %4\n\tv_mul_f32_e32 %3, %1, %4"
                : "+v"(t.v0), "+v"(t.v1), "=&v"(t.a0), "=&v"(t.a1) : "v"(s_out));
        }
    }
    if constexpr (PH == 2 && GRAD && SPLIT) {
        asm("v_mul_f32_e32 %0, %4, %6\n\tv_mul_f32_e32 %1, %5, %6\n\tv_cvt_pk_f16_f32 %2, %0, %1\n\tv_max3_f32 %3, %3, |%0|, |%1|"
            : "=&v"(t.a0), "=&v"(t.a1), "=&v"(t.hi), "+v"(xmax) : "v"(t.v0), "v"(t.v1), "v"(s_out));
        hi_out = t.hi;
    } else if constexpr (PH == 2) {
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(t.hi) : "v"(t.a0), "v"(t.a1));          // round to nearest even
        hi_out = t.hi;
        if constexpr (GRAD) asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(xmax) : "v"(xmax), "v"(t.a0), "v"(t.a1));
        else asm("v_max3_f32 %0, %1, %2, %3" : "=v"(xmax) : "v"(xmax), "v"(t.a0), "v"(t.a1));  // (a >= 0 behind the ReLU)
    }
    if constexpr (PH == 3) {
        float h0, h1;
        asm("v_cvt_f32_f16_e32 %2, %4\n\tv_cvt_f32_f16_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_sub_f32_e32 %0, %0, %2\n\tv_sub_f32_e32 %1, %1, %3"                     // exact
            : "+v"(t.a0), "+v"(t.a1), "=&v"(h0), "=&v"(h1) : "v"(t.hi));
    }
    if constexpr (PH == 4) { asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo_out) : "v"(t.a0), "v"(t.a1)); }
}

// One layer.  MODE 0: hidden (ReLU, operands of the next layer into `out`), MODE 1: the output layer (rows of y), MODE 2: a layer of the
// data-gradient chain (A = W^T blocks, no bias, relu' of the activation below from its bit rows, gradient rows out).  HAS_PREV: the last
// two blocks of the previous layer are still pending (st.pend) and are finished behind the first k steps of this one, into `in`.
template <int H, int KSL, int MODE, bool HAS_PREV, bool STASH>
struct Layer {
    static constexpr int NB = H / 32, NG = NB / 2, KS = H / 16;
    static constexpr int T = NG * KSL;                         // k steps of the layer (two blocks each)
    static constexpr int PPS = 16 / KSL > 0 ? 16 / KSL : 1;    // pending value pairs finished per step (groups >= 1)
    static constexpr int PPS_PREV = 32 / KSL;                  // ... of the previous layer's last group (first half of group 0)
    Ring &ring;
    State<H> &st;
    Op2 (&in)[KS];
    Op2 (&out)[KS];
    unsigned bias_addr;            // LDS byte address of this layer's bias + 16 h
    float inv_w, wn, bmax;         // the layer's constants (fused_pack_kernel)
    float inv_in;                  // 1 / scale of `in`
    float cmul, s_out, inv_out;    // this layer's epilogue factor and output scale (s_out: known once the previous layer is finished)
    float *yrow; bool store; int h;
    bool own_tile;                 // false: a wave without a tile of its own (it recomputes the last tile; what it would write differs for the masked samples)
    float *rows; unsigned *bits;   // STASH: this layer's output rows / bit rows of the wave's tile (wave-uniform); lane offset below
    unsigned lane_off;             // (4 h 32 + j) 4: byte offset of D register 0 of a block inside its 32 rows; an opaque 32-bit value (Tile), so that
                                   // base + zero-extended offset + immediate selects the SGPR-base form of the stores without a copy per store
    int lane;
    unsigned bitacc[2];            // ReLU bits of the two blocks being finished
    unsigned max_addr;             // STASH: LDS address of this lane's running maximum of the layer's input (0: not wanted)
    f32x16 acc[2][2];              // [group parity][block]: the group in flight and the finished one whose epilogue runs behind it
    f32x4 bias_cur[2][4];          // of the finished group (requested at its last step)

    __device__ __forceinline__ void scales_from_norm() {       // the a-priori bound of the file header
        const float xm = other_half_max(st.xmax) * inv_in;     // max |x| over the sample's input column
        bound_scales(fmaf(wn, xm, bmax) * 1.002f, s_out, inv_out);
        st.xmax = 0.0f;
        if constexpr (STASH) {                                 // largest input value of this layer so far, per lane (reduced at the end of the kernel)
            if (max_addr != 0) asm volatile("ds_max_f32 %0, %1" :: "v"(max_addr), "v"(xm) : "memory");
        }
    }
    // pending pair Q (0 .. 15: block Q >> 3, registers 2 (Q & 7), + 1) of hidden group `grp` (blocks 2 grp, 2 grp + 1) -> dst; phase PH
    PairTmp tmp[2];
    unsigned bit0[2], bit1[2];     // STASH: v > 0 of the pair's two values, on their way into bitacc
    // SPREAD (one pair per step, phase = gap): the row stores and the bit arithmetic are dealt over phases 2 .. 4 so that no gap holds more than
    // five instructions; otherwise (two or four pairs per step) everything of a pair sits in its five phases as densely as it comes
    template <int Q, int PH, bool SPREAD = false>
    __device__ __forceinline__ void hidden_phase(const f32x16 (&cc)[2], const f32x4 (&bb)[2][4], float cm, float so, Op2 (&dst)[KS], int grp,
                                                 float *rows_, unsigned *bits_) {
        constexpr int blk = Q >> 3, r0 = 2 * (Q & 7);
        Op2 &d = dst[2 * (2 * grp + blk) + (r0 >> 3)];
        unsigned hi_ = 0, lo_ = 0;
        if constexpr (MODE == 2)          // (the mask dword of the block travels in the first bias slot)
            pair_phase<PH, true, r0, SPREAD>(tmp[Q & 1], cc[blk][r0], cc[blk][r0 + 1], 0.0f, 0.0f, cm, so, hi_, lo_, st.xmax, __float_as_uint(bb[blk][0][0]));
        else
            pair_phase<PH>(tmp[Q & 1], cc[blk][r0], cc[blk][r0 + 1], bb[blk][r0 >> 2][r0 & 3], bb[blk][r0 >> 2][(r0 & 3) + 1], cm, so, hi_, lo_, st.xmax);
        if constexpr (PH == 2) d.hi[(r0 >> 1) & 3] = hi_;
        if constexpr (PH == 4) d.lo[(r0 >> 1) & 3] = lo_;
        if constexpr (MODE == 2 || STASH) {
            // MODE 2: the gradient, STASH: the activation (after the ReLU, before the scale), as two [feature][32-sample] rows (the weight-gradient
            // launches read them)
            char *p = reinterpret_cast<char *>(rows_ + 32 * (2 * grp + blk) * 32);
            constexpr int rr0 = ((r0 & 3) + 8 * (r0 >> 2)) * 128;
            if constexpr (PH == 2) __builtin_nontemporal_store(tmp[Q & 1].v0, reinterpret_cast<float *>(p + lane_off + rr0));
            if constexpr (PH == (SPREAD ? 3 : 2)) __builtin_nontemporal_store(tmp[Q & 1].v1, reinterpret_cast<float *>(p + lane_off + rr0 + 128));
        }
        if constexpr (MODE != 2 && STASH) {
            // v > 0 of a value behind the ReLU = min(its bit pattern, 1); shifted into place by v_lshl_or: two instructions per value
            // (hipcc turns the C form into class compare + select through vcc + or)
            if constexpr (PH == (SPREAD ? 2 : 3)) {
                asm("v_min_u32_e32 %0, 1, %1" : "=v"(bit0[Q & 1]) : "v"(tmp[Q & 1].v0));
                asm("v_min_u32_e32 %0, 1, %1" : "=v"(bit1[Q & 1]) : "v"(tmp[Q & 1].v1));
            }
            if constexpr (PH == (SPREAD ? 4 : 3)) {
                if constexpr ((Q & 7) == 0) asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(bitacc[blk]) : "v"(bit1[Q & 1]), "v"(bit0[Q & 1]));
                else
                    asm("v_lshl_or_b32 %0, %1, %3, %0\n\tv_lshl_or_b32 %0, %2, %4, %0"
                        : "+v"(bitacc[blk]) : "v"(bit0[Q & 1]), "v"(bit1[Q & 1]), "n"(r0), "n"(r0 + 1));
            }
            if constexpr (PH == 4 && (Q & 7) == 7) bits_[(2 * (2 * grp + blk)) * 32 + lane] = bitacc[blk];
        }
    }
    template <int Q> __device__ __forceinline__ void flush_pending() {     // a pending pair with nothing to hide behind (end of a TN_MLP_SKIP_LAST stack)
        hidden_phase<Q, 0>(st.pend.c, st.pend.bias, st.pend.cmul, st.pend.s_out, in, NG - 1, st.pend.rows, st.pend.bits);
        hidden_phase<Q, 1>(st.pend.c, st.pend.bias, st.pend.cmul, st.pend.s_out, in, NG - 1, st.pend.rows, st.pend.bits);
        hidden_phase<Q, 2>(st.pend.c, st.pend.bias, st.pend.cmul, st.pend.s_out, in, NG - 1, st.pend.rows, st.pend.bits);
        hidden_phase<Q, 3>(st.pend.c, st.pend.bias, st.pend.cmul, st.pend.s_out, in, NG - 1, st.pend.rows, st.pend.bits);
        hidden_phase<Q, 4>(st.pend.c, st.pend.bias, st.pend.cmul, st.pend.s_out, in, NG - 1, st.pend.rows, st.pend.bits);
    }
    // NPAIR pairs Q0 .. Q0 + NPAIR - 1 dealt over the six gaps of a step
    template <int Q0, int NPAIR>
    __device__ __forceinline__ void hidden_gap(int gap, const f32x16 (&cc)[2], const f32x4 (&bb)[2][4], float cm, float so, Op2 (&dst)[KS], int grp,
                                               float *rows_, unsigned *bits_) {
        if constexpr (NPAIR == 1) {
            if (gap == 0) hidden_phase<Q0, 0, true>(cc, bb, cm, so, dst, grp, rows_, bits_);
            if (gap == 1) hidden_phase<Q0, 1, true>(cc, bb, cm, so, dst, grp, rows_, bits_);
            if (gap == 2) hidden_phase<Q0, 2, true>(cc, bb, cm, so, dst, grp, rows_, bits_);
            if (gap == 3) hidden_phase<Q0, 3, true>(cc, bb, cm, so, dst, grp, rows_, bits_);
            if (gap == 4) hidden_phase<Q0, 4, true>(cc, bb, cm, so, dst, grp, rows_, bits_);
        } else if constexpr (NPAIR == 2) {
            if (gap == 0) { hidden_phase<Q0, 0>(cc, bb, cm, so, dst, grp, rows_, bits_); hidden_phase<Q0, 1>(cc, bb, cm, so, dst, grp, rows_, bits_); }
            if (gap == 1) { hidden_phase<Q0, 2>(cc, bb, cm, so, dst, grp, rows_, bits_); hidden_phase<Q0, 3>(cc, bb, cm, so, dst, grp, rows_, bits_); }
            if (gap == 2) { hidden_phase<Q0, 4>(cc, bb, cm, so, dst, grp, rows_, bits_); hidden_phase<Q0 + 1, 0>(cc, bb, cm, so, dst, grp, rows_, bits_); }
            if (gap == 3) { hidden_phase<Q0 + 1, 1>(cc, bb, cm, so, dst, grp, rows_, bits_); hidden_phase<Q0 + 1, 2>(cc, bb, cm, so, dst, grp, rows_, bits_); }
            if (gap == 4) { hidden_phase<Q0 + 1, 3>(cc, bb, cm, so, dst, grp, rows_, bits_); hidden_phase<Q0 + 1, 4>(cc, bb, cm, so, dst, grp, rows_, bits_); }
        } else {
            hidden_gap<Q0, 2>(gap < 3 ? 2 * gap : 9, cc, bb, cm, so, dst, grp, rows_, bits_);              // pairs 0, 1 in gaps 0 .. 2 (two of the five gap slots each)
            hidden_gap<Q0, 2>(gap < 3 ? 2 * gap + 1 : 9, cc, bb, cm, so, dst, grp, rows_, bits_);
            hidden_gap<Q0 + 2, 2>(gap >= 3 ? 2 * (gap - 3) : 9, cc, bb, cm, so, dst, grp, rows_, bits_);   // pairs 2, 3 in gaps 3 .. 5
            hidden_gap<Q0 + 2, 2>(gap >= 3 ? 2 * (gap - 3) + 1 : 9, cc, bb, cm, so, dst, grp, rows_, bits_);
        }
    }
    template <int Q4>                                           // output layer: four values (registers 4 (Q4 & 3) ..) of block Q4 >> 2 of group grp
    __device__ __forceinline__ void out_quad(const f32x16 (&cc)[2], const f32x4 (&bb)[2][4], int grp) {
        constexpr int blk = Q4 >> 2, q = Q4 & 3;
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fmaf(cc[blk][4 * q + u], cmul, bb[blk][q][u]);
        if (TN_FUSED_ABL & 4) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); return; }
        if (store && yrow != nullptr) *reinterpret_cast<f32x4 *>(yrow + 32 * (2 * grp + blk) + 8 * q + 4 * h) = v;
        if constexpr (STASH) {               // y as rows (tn_mlp_rows_view): samples past n hold 0 as in the layer-wise form
            char *p = reinterpret_cast<char *>(rows + 32 * (2 * grp + blk) * 32);
            unsigned off = lane_off;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int u = 0; u < 4; ++u) if (own_tile) *reinterpret_cast<float *>(p + off + (u + 8 * q) * 128) = store ? v[u] : 0.0f;
        }
    }
    template <int BLK> __device__ __forceinline__ void load_mask(f32x4 (&bb)[2][4], int grp) {       // MODE 2: relu' bits of the block the gradient flows into
        bb[BLK][0][0] = __uint_as_float(bits[(2 * (2 * grp + BLK)) * 32 + lane]);
    }
    template <int BLK> __device__ __forceinline__ void request_bias(f32x4 (&bb)[2][4], int grp) {
        // (grp is a compile-time constant at every call site; the offset is folded into the address register once per group)
        const unsigned a = bias_addr + (unsigned)(32 * (2 * grp + BLK)) * 4u;
        bb[BLK][0] = lds16f<0>(a); bb[BLK][1] = lds16f<32>(a); bb[BLK][2] = lds16f<64>(a); bb[BLK][3] = lds16f<96>(a);
    }

    template <int GI, int S>
    __device__ __forceinline__ void step() {
        constexpr int t = GI * KSL + S;
        // ---- front matter: chunk boundary, one LDS-direct piece, the operands of step t + 2, wait for those of step t ----
        constexpr int NW = NWAVES<H, MODE == 2 || !STASH>;
        if constexpr (t % 8 == 0) ring.template boundary<NW>();
        // operands of step t + 2 of the STREAM (the next layer's, the next round's: the ring does not care); pair offset inside its chunk
        constexpr int pt = (2 * (t + 2)) % CHUNK_PAIRS;
        const unsigned base = ((t % 8) + 2 >= 8) ? ring.nxt : ring.cur;
        Op2 (&w)[2] = st.aw[(t + 2) & 3];
        auto req = [&](int k) {
            if (TN_FUSED_ABL & 8) return;
            if (k == 0) w[0].hi = lds16<pt * PAIR_B>(base);
            if (k == 1) w[0].lo = lds16<pt * PAIR_B + 1024>(base);
            if (k == 2) w[1].hi = lds16<(pt + 1) * PAIR_B>(base);
            if (k == 3) w[1].lo = lds16<(pt + 1) * PAIR_B + 1024>(base);
        };
        // (requested behind the last MFMA of the step -- gap 5 -- so that the front of the step is the wait alone)
        Op2 (&A)[2] = st.aw[t & 3];
        // the first step of a group follows the bias requests of the previous step: they must be complete as well
        constexpr bool bias_due = MODE != 2 && S == 0 && (GI > 0 || HAS_PREV);      // (MODE 2: masks come by vector loads, hipcc counts those)
        if (!(TN_FUSED_ABL & (8 | 32))) {
            constexpr int own = 0;                                 // (this step's own requests come behind the wait)
            if constexpr (bias_due) {
                wait_ops<own>(A[0], A[1]);
                if constexpr (GI > 0) wait_bias<own>(bias_cur); else wait_bias<own>(st.pend.bias);
            } else wait_ops<4 + own>(A[0], A[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const Op2 &b = in[S];
        auto mm = [&](const u32x4h &a_, const u32x4h &b_, const f32x16 &c_) -> f32x16 {
            if (TN_FUSED_ABL & 16) { f32x16 r_ = c_; asm volatile("" : "+v"(r_[0]) : "v"(a_), "v"(b_)); return r_; }
            return mfma_f16(a_, b_, c_);
        };
        // ---- six MFMAs, the two blocks' accumulators in turn; the pending pairs of this step behind them ----
        auto fill = [&](int gap) {
            if constexpr (GI > 0) {
                if constexpr (MODE == 0 || MODE == 2) hidden_gap<S * PPS, PPS>(gap, acc[(GI - 1) & 1], bias_cur, cmul, s_out, out, GI - 1, rows, bits);     // pairs S PPS .. of group GI - 1
                else {
                    // eight quads over the group's KSL steps
                    if constexpr (KSL >= 16) { if ((S & 1) == 0 && gap == 1) out_quad<S / 2>(acc[(GI - 1) & 1], bias_cur, GI - 1); }
                    else if constexpr (KSL == 8) { if (gap == 1) out_quad<S>(acc[(GI - 1) & 1], bias_cur, GI - 1); }
                    else { if (gap == 1) out_quad<2 * S>(acc[(GI - 1) & 1], bias_cur, GI - 1); if (gap == 4) out_quad<2 * S + 1>(acc[(GI - 1) & 1], bias_cur, GI - 1); }
                }
            } else if constexpr (HAS_PREV) {
                // the previous layer's last group, PPS_PREV pairs per step, into THIS layer's input (k steps KS - 4 .. KS - 1: not read before step KSL - 4 >= KSL / 2)
                if constexpr (S < KSL / 2) hidden_gap<S * PPS_PREV, PPS_PREV>(gap, st.pend.c, st.pend.bias, st.pend.cmul, st.pend.s_out, in, NG - 1, st.pend.rows, st.pend.bits);
                else if constexpr (S == KSL / 2) { if (gap == 0) scales_from_norm(); }      // the previous layer is complete: this layer's output scale
            }
        };
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        f32x16 (&c)[2] = acc[GI & 1];
        // (a scheduling barrier on both sides of every MFMA: hipcc otherwise moves a gap's fillers in front of it)
#define TN_SB __builtin_amdgcn_sched_barrier(0)
        c[0] = mm(A[0].lo, b.hi, S == 0 ? z : c[0]); TN_SB; fill(0); TN_SB;
        c[1] = mm(A[1].lo, b.hi, S == 0 ? z : c[1]); TN_SB; fill(1); TN_SB;
        c[0] = mm(A[0].hi, b.lo, c[0]); TN_SB; fill(2); TN_SB;
        c[1] = mm(A[1].hi, b.lo, c[1]); TN_SB; fill(3); TN_SB;
        c[0] = mm(A[0].hi, b.hi, c[0]); TN_SB; fill(4); if constexpr ((t % 8) % (NW / 4) == 0) ring.template piece<(t % 8) / (NW / 4)>(); TN_SB;
        c[1] = mm(A[1].hi, b.hi, c[1]); TN_SB; fill(5); req(0); req(1); req(2); req(3); TN_SB;
#undef TN_SB
        // ---- last step of a group: its accumulators become the pending ones, their bias is requested (8 reads, complete by the next step) ----
        if constexpr (S == KSL - 1) {
            if constexpr (GI + 1 < NG) {
                if constexpr (MODE == 2) { load_mask<0>(bias_cur, GI); load_mask<1>(bias_cur, GI); }
                else { request_bias<0>(bias_cur, GI); request_bias<1>(bias_cur, GI); }
            } else {
                st.pend.c[0] = c[0]; st.pend.c[1] = c[1];
                if constexpr (MODE == 2) { load_mask<0>(st.pend.bias, GI); load_mask<1>(st.pend.bias, GI); }
                else { request_bias<0>(st.pend.bias, GI); request_bias<1>(st.pend.bias, GI); }
                st.pend.cmul = cmul; st.pend.s_out = s_out;
                st.pend.rows = rows; st.pend.bits = bits;
            }
        }
    }

    template <int GI, int... S> __device__ __forceinline__ void group(std::integer_sequence<int, S...>) { (step<GI, S>(), ...); }
    template <int... GI> __device__ __forceinline__ void groups(std::integer_sequence<int, GI...>) { (group<GI>(std::make_integer_sequence<int, KSL>{}), ...); }
    __device__ __forceinline__ void run() {
        cmul = inv_w * inv_in;
        if constexpr (!HAS_PREV) scales_from_norm();           // (layer 0: the norm of the encoded inputs is complete)
        groups(std::make_integer_sequence<int, NG>{});
    }
};

// per-layer context of a wave's tile
struct Tile {
    float *yrow; bool store, own_tile; int h, lane;
    unsigned lane_off;
    float *rows_base;              // STASH: workspace + tile * rows_total * 32 floats (wave-uniform)
    unsigned max0;                 // STASH: LDS address of this lane's slot in the maxima table of layer 0 (layer l: + l * 1024)
};

template <int H, int KSL, int MODE, bool HAS_PREV, bool STASH>
__device__ __forceinline__ float run_layer(Ring &ring, State<H> &st, Op2 (&in)[H / 16], Op2 (&out)[H / 16], const FusedArgs &a, int l, unsigned bias0,
                                           float inv_in, const Tile &tl)
{
    const float *cs = a.consts + 4 * l;
    Layer<H, KSL, MODE, HAS_PREV, STASH> L{ring, st, in, out, bias0 + (unsigned)(l * H * 4), cs[0], cs[1], cs[2], inv_in, 0.0f, 0.0f, 0.0f, tl.yrow, tl.store, tl.h};
    if constexpr (STASH) {
        L.rows = urow(tl.rows_base, a.off_out[l]);
        L.bits = reinterpret_cast<unsigned *>(urow(tl.rows_base, MODE != 1 ? a.off_bits[l] : 0));
        L.lane_off = tl.lane_off; L.lane = tl.lane;
        L.max_addr = (a.tail != nullptr && a.tail_idx[l] >= 0) ? tl.max0 + (unsigned)l * (256u * NWAVES<H, MODE == 2 || !STASH>) : 0u;
    }
    L.own_tile = tl.own_tile;
    L.run();
    return L.inv_out;
}

template <int H, bool STASH>
__global__ __launch_bounds__((64 * NWAVES<H, !STASH>)) void fused_fwd_kernel(FusedArgs a, int64_t n)
{
    constexpr int KS = H / 16, NB = H / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float *bias_s = reinterpret_cast<float *>(lds_raw + NRING * CHUNK_B);
    float *max_s = bias_s + a.L * H;                           // STASH: [L][256] running maxima of the layers' inputs
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < a.L * H; e += blockDim.x) bias_s[e] = a.bias[e];
    if constexpr (STASH) for (int e = threadIdx.x; e < a.L * 64 * NWAVES<H, !STASH>; e += blockDim.x) max_s[e] = 0.0f;
    lds_u8 *lds3 = (lds_u8 *)lds_raw;
    const unsigned lds0 = (unsigned)(uintptr_t)lds3;
    Ring ring;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.stream, 0, 0x7fffffff, 0x00020000);
    constexpr int NW = NWAVES<H, !STASH>;
    ring.lds_wave = lds3 + wave * (CHUNK_B / NW);
    ring.voff = (unsigned)(wave * (CHUNK_B / NW) + lane * 16);
    ring.rd0 = lds0 + lane * 16;
    ring.n_chunks = (unsigned)a.n_chunks;
    ring.next = 0;
    // chunks 0, 1, 2 of the stream into slots 0, 1, 2.  Ring state = "no boundary passed": the first boundary (step 0 of round 0) makes
    // chunk 0 (slot 0) current, chunk 1 next, and targets the pieces of chunk 3
    ring.g = (unsigned)-3;
    for (int k = 0; k < 3; ++k) {
        ring.target_next();
        ring.template piece<0>(); ring.template piece<1>(); ring.template piece<2>(); ring.template piece<3>();
        if constexpr (NW == 4) { ring.template piece<4>(); ring.template piece<5>(); ring.template piece<6>(); ring.template piece<7>(); }
        ++ring.g;
    }
    ring.cur = ring.nxt = ring.rd0;
    __syncthreads();                         // (vmcnt(0): all three chunks have landed; bias staged)
    const unsigned bias0 = lds0 + NRING * CHUNK_B + 16 * h;
    const int64_t n_tiles = (n + 31) >> 5;
    const int64_t per_round = (int64_t)gridDim.x * NW;
    const int64_t rounds = (n_tiles + per_round - 1) / per_round;
    State<H> st;
    // (operands of stream steps 0 and 1 -- chunk 0 -- for the window)
    {
        const unsigned base = ring.rd0;
        st.aw[0][0].hi = lds16<0>(base); st.aw[0][0].lo = lds16<1024>(base);
        st.aw[0][1].hi = lds16<PAIR_B>(base); st.aw[0][1].lo = lds16<PAIR_B + 1024>(base);
        st.aw[1][0].hi = lds16<2 * PAIR_B>(base); st.aw[1][0].lo = lds16<2 * PAIR_B + 1024>(base);
        st.aw[1][1].hi = lds16<3 * PAIR_B>(base); st.aw[1][1].lo = lds16<3 * PAIR_B + 1024>(base);
    }
    Tile tl;
    tl.h = h; tl.lane = lane;
    tl.lane_off = (unsigned)(4 * h * 32 + j) * 4u;
    asm volatile("" : "+v"(tl.lane_off));
    tl.max0 = lds0 + NRING * CHUNK_B + (unsigned)(a.L * H) * 4u + (unsigned)threadIdx.x * 4u;
    const int n_hidden = a.L - 1 - (a.out_linear ? 1 : 0);     // hidden layers behind layer 0
#pragma clang loop unroll(disable)
    for (int64_t round = 0; round < rounds; ++round) {
        const int64_t tile_raw = (round * gridDim.x + blockIdx.x) * NW + wave;
        const bool tile_ok = tile_raw < n_tiles;
        const int64_t tile = tile_ok ? tile_raw : n_tiles - 1;
        float inv_in;
        {   // first-layer inputs: 64 rows of the tile in the D layout (mlp_f2_layers.hip fwd_first_f2_kernel), exact column maximum
            const float *in = urow(a.e_rows, tile * a.e_rows_total + a.e_off);
            float x[2][16];
            float mx = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { x[t][r] = in[(32 * t + frow(r, h)) * 32 + j]; mx = fmaxf(mx, fabsf(x[t][r])); }
            mx = f2_xmax(mx);
            float s_in;
            f2_scales(mx, s_in, inv_in);
            st.xmax = mx * s_in;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x[b >> 1][8 * (b & 1) + e] * s_in;
                f2_split8(v, 1.0f, st.Q[b].hi, st.Q[b].lo);
            }
        }
        tl.yrow = a.y != nullptr ? a.y + (tile * 32 + j) * H : nullptr;
        tl.store = tile_ok && tile * 32 + j < n;
        tl.own_tile = tile_ok;
        // (a wave without a tile of its own recomputes the last tile and writes the same values to the same places)
        if constexpr (STASH) tl.rows_base = urow(a.rows, tile * a.rows_total);
        // layer 0: Q (4 operand steps) -> P; hidden layers in pairs P -> Q -> P; output layer (or the flush of the last hidden layer) from P
        inv_in = run_layer<H, KS0, 0, false, STASH>(ring, st, st.Q, st.P, a, 0, bias0, inv_in, tl);
        int l = 1;
#pragma clang loop unroll(disable)
        for (; l + 1 <= n_hidden; l += 2) {
            inv_in = run_layer<H, KS, 0, true, STASH>(ring, st, st.P, st.Q, a, l, bias0, inv_in, tl);
            inv_in = run_layer<H, KS, 0, true, STASH>(ring, st, st.Q, st.P, a, l + 1, bias0, inv_in, tl);
        }
        if (l <= n_hidden) {                 // an odd number of hidden layers (Cobafa: 5): a third copy of the loop body would cost 12 KB of code;
            // instead the last one runs P -> Q like the others and the column moves back once per round (H / 2 register moves)
            inv_in = run_layer<H, KS, 0, true, STASH>(ring, st, st.P, st.Q, a, l, bias0, inv_in, tl);
#pragma unroll
            for (int b = 0; b < KS - 4; ++b) st.P[b] = st.Q[b];          // (the last four operand steps are still pending: st.pend writes them into P)
            ++l;
        }
        if (a.out_linear) {
            const float *cs = a.consts + 4 * l;
            Layer<H, KS, 1, true, STASH> L{ring, st, st.P, st.Q, bias0 + (unsigned)(l * H * 4), cs[0], cs[1], cs[2], inv_in, 0.0f, 0.0f, 0.0f, tl.yrow, tl.store, h};
            if constexpr (STASH) {
                L.rows = urow(tl.rows_base, a.off_out[l]); L.bits = nullptr; L.lane_off = tl.lane_off; L.lane = lane;
                L.max_addr = (a.tail != nullptr && a.tail_idx[l] >= 0) ? tl.max0 + (unsigned)l * (256u * NWAVES<H, !STASH>) : 0u;
            }
            L.own_tile = tile_ok;
            L.run();
            // its last two blocks have nothing to hide behind
            wait_bias<0>(st.pend.bias);
            L.template out_quad<0>(st.pend.c, st.pend.bias, NB / 2 - 1); L.template out_quad<1>(st.pend.c, st.pend.bias, NB / 2 - 1);
            L.template out_quad<2>(st.pend.c, st.pend.bias, NB / 2 - 1); L.template out_quad<3>(st.pend.c, st.pend.bias, NB / 2 - 1);
            L.template out_quad<4>(st.pend.c, st.pend.bias, NB / 2 - 1); L.template out_quad<5>(st.pend.c, st.pend.bias, NB / 2 - 1);
            L.template out_quad<6>(st.pend.c, st.pend.bias, NB / 2 - 1); L.template out_quad<7>(st.pend.c, st.pend.bias, NB / 2 - 1);
        } else {
            // TN_MLP_SKIP_LAST: the stack ends in a hidden activation -- its last two blocks are finished here (rows and bit rows; the
            // operands they would become have no consumer)
            if constexpr (STASH) {
                Layer<H, KS, 0, true, STASH> L{ring, st, st.P, st.Q, bias0, 0.0f, 0.0f, 0.0f, inv_in, 0.0f, 0.0f, 0.0f, tl.yrow, tl.store, h};
                L.lane_off = tl.lane_off; L.lane = lane; L.max_addr = 0; L.own_tile = tile_ok;
                wait_bias<0>(st.pend.bias);
                L.template flush_pending<0>(); L.template flush_pending<1>(); L.template flush_pending<2>(); L.template flush_pending<3>();
                L.template flush_pending<4>(); L.template flush_pending<5>(); L.template flush_pending<6>(); L.template flush_pending<7>();
                L.template flush_pending<8>(); L.template flush_pending<9>(); L.template flush_pending<10>(); L.template flush_pending<11>();
                L.template flush_pending<12>(); L.template flush_pending<13>(); L.template flush_pending<14>(); L.template flush_pending<15>();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the ring's last requests: nothing may land in LDS after the workgroup has gone)
    if constexpr (STASH) {
        if (a.tail != nullptr) {
            __syncthreads();
            for (int l = 0; l < a.L; ++l) {
                if (a.tail_idx[l] < 0) continue;
                float m = max_s[l * 64 * NW + threadIdx.x];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(a.tail + a.tail_idx[l]), __float_as_uint(m));      // (non-negative floats order like their bits)
            }
        }
    }
}

// The data-gradient chain of a wide stack in the same form (round 6): from the gradient w.r.t. the top layer's pre-activation (rows at
// off_in) down to the first hidden layer's --  G_{l-1} = relu'(h_{l-1}) * (W_l^T G_l)  -- with the gradients in registers between
// the layers.  Every G_{l-1} is written once as rows (the weight-gradient launch of layer l - 1 reads it) and never read back here;
// relu' comes from the bit rows the training forward left.  Stream position i = layer top - i.
template <int H>
__global__ __launch_bounds__((64 * NWAVES<H, true>)) void fused_chain_kernel(FusedArgs a, int64_t n)
{
    constexpr int KS = H / 16, NB = H / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float *max_s = reinterpret_cast<float *>(lds_raw + NRING * CHUNK_B);           // [L][256] running maxima of the layers' input gradients
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < a.L * 64 * NWAVES<H, true>; e += blockDim.x) max_s[e] = 0.0f;
    lds_u8 *lds3 = (lds_u8 *)lds_raw;
    const unsigned lds0 = (unsigned)(uintptr_t)lds3;
    Ring ring;
    ring.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.stream, 0, 0x7fffffff, 0x00020000);
    constexpr int NW = NWAVES<H, true>;
    ring.lds_wave = lds3 + wave * (CHUNK_B / NW);
    ring.voff = (unsigned)(wave * (CHUNK_B / NW) + lane * 16);
    ring.rd0 = lds0 + lane * 16;
    ring.n_chunks = (unsigned)a.n_chunks;
    ring.next = 0;
    ring.g = (unsigned)-3;
    for (int k = 0; k < 3; ++k) {
        ring.target_next();
        ring.template piece<0>(); ring.template piece<1>(); ring.template piece<2>(); ring.template piece<3>();
        if constexpr (NW == 4) { ring.template piece<4>(); ring.template piece<5>(); ring.template piece<6>(); ring.template piece<7>(); }
        ++ring.g;
    }
    ring.cur = ring.nxt = ring.rd0;
    __syncthreads();
    const int64_t n_tiles = (n + 31) >> 5;
    const int64_t per_round = (int64_t)gridDim.x * NW;
    const int64_t rounds = (n_tiles + per_round - 1) / per_round;
    State<H> st;
    {
        const unsigned base = ring.rd0;
        st.aw[0][0].hi = lds16<0>(base); st.aw[0][0].lo = lds16<1024>(base);
        st.aw[0][1].hi = lds16<PAIR_B>(base); st.aw[0][1].lo = lds16<PAIR_B + 1024>(base);
        st.aw[1][0].hi = lds16<2 * PAIR_B>(base); st.aw[1][0].lo = lds16<2 * PAIR_B + 1024>(base);
        st.aw[1][1].hi = lds16<3 * PAIR_B>(base); st.aw[1][1].lo = lds16<3 * PAIR_B + 1024>(base);
    }
    Tile tl;
    tl.h = h; tl.lane = lane; tl.yrow = nullptr; tl.store = false;
    tl.lane_off = (unsigned)(4 * h * 32 + j) * 4u;
    asm volatile("" : "+v"(tl.lane_off));
    tl.max0 = lds0 + NRING * CHUNK_B + (unsigned)threadIdx.x * 4u;
    const unsigned bias0 = 0;               // (no bias in the chain)
#pragma clang loop unroll(disable)
    for (int64_t round = 0; round < rounds; ++round) {
        const int64_t tile_raw = (round * gridDim.x + blockIdx.x) * NW + wave;
        const bool tile_ok = tile_raw < n_tiles;
        const int64_t tile = tile_ok ? tile_raw : n_tiles - 1;
        tl.own_tile = tile_ok;
        tl.rows_base = urow(a.rows, tile * a.rows_total);
        float inv_in;
        {   // the gradient the chain starts from: H rows of the tile in the D layout, exact column maximum
            const float *in = urow(tl.rows_base, a.off_in);
            float mx = 0.0f;
            float x[NB][16];
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { x[t][r] = in[(32 * t + frow(r, h)) * 32 + j]; mx = fmaxf(mx, fabsf(x[t][r])); }
            mx = f2_xmax(mx);
            float s_in;
            f2_scales(mx, s_in, inv_in);
            st.xmax = mx * s_in;
#pragma unroll
            for (int b = 0; b < KS; ++b) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x[b >> 1][8 * (b & 1) + e] * s_in;
                f2_split8(v, 1.0f, st.Q[b].hi, st.Q[b].lo);
            }
        }
        inv_in = run_layer<H, KS, 2, false, true>(ring, st, st.Q, st.P, a, 0, bias0, inv_in, tl);
        int l = 1;
#pragma clang loop unroll(disable)
        for (; l + 1 < a.L; l += 2) {
            inv_in = run_layer<H, KS, 2, true, true>(ring, st, st.P, st.Q, a, l, bias0, inv_in, tl);
            inv_in = run_layer<H, KS, 2, true, true>(ring, st, st.Q, st.P, a, l + 1, bias0, inv_in, tl);
        }
        if (l < a.L) {
            inv_in = run_layer<H, KS, 2, true, true>(ring, st, st.P, st.Q, a, l, bias0, inv_in, tl);
#pragma unroll
            for (int b = 0; b < KS - 4; ++b) st.P[b] = st.Q[b];
        }
        // the last layer's last two blocks: their rows (the operands they would become have no consumer)
        {
            Layer<H, KS, 2, true, true> L{ring, st, st.P, st.Q, 0u, 0.0f, 0.0f, 0.0f, inv_in, 0.0f, 0.0f, 0.0f, nullptr, false, h};
            L.lane_off = tl.lane_off; L.lane = lane; L.max_addr = 0; L.own_tile = tile_ok;
            L.template flush_pending<0>(); L.template flush_pending<1>(); L.template flush_pending<2>(); L.template flush_pending<3>();
            L.template flush_pending<4>(); L.template flush_pending<5>(); L.template flush_pending<6>(); L.template flush_pending<7>();
            L.template flush_pending<8>(); L.template flush_pending<9>(); L.template flush_pending<10>(); L.template flush_pending<11>();
            L.template flush_pending<12>(); L.template flush_pending<13>(); L.template flush_pending<14>(); L.template flush_pending<15>();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (a.tail != nullptr) {
        __syncthreads();
        for (int l = 0; l < a.L; ++l) {
            if (a.tail_idx[l] < 0) continue;
            float m = max_s[l * 64 * NW + threadIdx.x];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(a.tail + a.tail_idx[l]), __float_as_uint(m));
        }
    }
}

template <int H>
int launch_chain(const FusedArgs &f, int64_t n, hipStream_t s)
{
    constexpr int NW = NWAVES<H, true>;
    const size_t lds_bytes = (size_t)NRING * CHUNK_B + (size_t)f.L * 256 * NW + 64;
    if (lds_bytes > (size_t)LDS_LIMIT_BYTES) return tn::fail(TN_E_CONFIG, "mlp_bwd(fused chain): maxima table does not fit LDS");
    auto kern = fused_chain_kernel<H>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd(fused chain): cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + NW - 1) / NW, 256));
    kern<<<dim3((unsigned)bl), dim3(64 * NW), lds_bytes, s>>>(f, n);
    return tn::check_launch("fused_chain_kernel");
}

template <int H, bool STASH>
int launch(const FusedArgs &f, int64_t n, hipStream_t s)
{
    constexpr int NW = NWAVES<H, !STASH>;
    const size_t lds_bytes = (size_t)NRING * CHUNK_B + (size_t)f.L * H * 4 + (STASH ? (size_t)f.L * 256 * NW : 0) + 64;
    if (lds_bytes > (size_t)LDS_LIMIT_BYTES) return tn::fail(TN_E_CONFIG, "mlp_fwd(fused): bias table does not fit LDS");
    auto kern = fused_fwd_kernel<H, STASH>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_fwd(fused): cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t bl = std::max<int64_t>(1, std::min<int64_t>((n_tiles + NW - 1) / NW, 256));
    kern<<<dim3((unsigned)bl), dim3(64 * NW), lds_bytes, s>>>(f, n);
    return tn::check_launch("fused_fwd_kernel");
}

}  // namespace

namespace tn {
namespace layers {

// bytes of the packed stream + bias + constants of a stack (first layer <= 64 input rows, L - 1 layers H -> H)
__attribute__((visibility("hidden"))) int64_t fused_pack_bytes(int H, int L)
{
    const int64_t pairs = (int64_t)(H / 32) * KS0 + (int64_t)(L - 1) * (H / 32) * (H / 16);
    const int64_t chunks = (pairs + CHUNK_PAIRS - 1) / CHUNK_PAIRS;
    return chunks * CHUNK_B + (int64_t)L * H * 4 + (int64_t)L * 16 + 256;
}

__attribute__((visibility("hidden"))) bool fused_fwd_ok(int H, const MlpArgs &a)
{
    if (!(H == 128 || H == 256) || !a.f2 || a.n_layers < 3 || a.n_layers > TN_MLP_MAX_LAYERS || a.out_dim != H || a.out_act != TN_ACT_NONE) return false;
    if (a.K[0] > 64 || a.N[0] != H) return false;
    for (int l = 1; l < a.n_layers; ++l) if (a.K[l] != H || a.N[l] != H) return false;
    return true;
}

// Inference (spec == nullptr): e_rows = 64 rows x 32 samples per tile, contiguous over the tiles; y [n][H].  Training forward: spec says
// where the workspace rows go.  pack_area: fused_pack_bytes(H, L), 256-byte aligned.  n_run: layers to evaluate (L, or L - 1 under
// TN_MLP_SKIP_LAST: every evaluated layer is then a hidden one).
__attribute__((visibility("hidden"))) int launch_fused_fwd_f2(int H, const MlpArgs &a, int64_t n, const float *e_rows, float *y, void *pack_area,
                                                              hipStream_t s, const FusedStash *spec)
{
    const int L = spec ? spec->n_run : a.n_layers;
    PackArgs p;
    p.L = L; p.H = H; p.transpose = 0;
    int64_t off = 0;
    for (int l = 0; l < L; ++l) {
        p.W[l] = a.W[l]; p.B[l] = a.B[l]; p.N[l] = a.N[l]; p.K[l] = l == 0 ? a.K0 : a.K[l];
        p.off[l] = off;
        off += (int64_t)(H / 32) * (l == 0 ? KS0 : H / 16) * PAIR_B;
    }
    const int64_t chunks = (off + CHUNK_B - 1) / CHUNK_B;
    p.stream = reinterpret_cast<unsigned char *>(pack_area);
    p.bias = reinterpret_cast<float *>(p.stream + chunks * CHUNK_B);
    p.consts = p.bias + (int64_t)L * H;
    if (off != chunks * CHUNK_B) {          // (H = 128: the first layer is 16 pairs, every hidden layer 32: always whole chunks; kept for other shapes)
        hipError_t me = hipMemsetAsync(p.stream + off, 0, (size_t)(chunks * CHUNK_B - off), s);
        if (me != hipSuccess) { tn::set_error("mlp_fwd(fused): memset: %s", hipGetErrorString(me)); return (int)me; }
    }
    fused_pack_kernel<<<dim3((unsigned)L, 4), dim3(1024), 0, s>>>(p);
    if (int rc = tn::check_launch("fused_pack_kernel")) return rc;
    FusedArgs f;
    f.stream = p.stream; f.bias = p.bias; f.consts = p.consts; f.L = L; f.n_chunks = (int)chunks; f.out_act = a.out_act;
    f.e_rows = e_rows; f.e_rows_total = 64; f.e_off = 0; f.y = y;
    f.rows = nullptr; f.rows_total = 0; f.out_linear = 1; f.tail = nullptr;
    f.off_in = 0;
    for (int l = 0; l < TN_MLP_MAX_LAYERS; ++l) { f.off_out[l] = 0; f.off_bits[l] = 0; f.tail_idx[l] = l >= 1 ? l : -1; }
    if (spec == nullptr) return H == 256 ? launch<256, false>(f, n, s) : launch<128, false>(f, n, s);
    f.e_rows_total = spec->rows_total; f.e_off = spec->off_e;
    f.rows = spec->rows; f.rows_total = spec->rows_total; f.out_linear = spec->n_run == a.n_layers ? 1 : 0; f.tail = spec->tail;
    for (int l = 0; l < L; ++l) { f.off_out[l] = spec->off_out[l]; f.off_bits[l] = spec->off_bits[l]; }
    return H == 256 ? launch<256, true>(f, n, s) : launch<128, true>(f, n, s);
}

// The data-gradient chain of layers `top` .. 1 (all H x H) in one launch: starts from the gradient rows at spec->off_in (w.r.t. layer
// top's pre-activation), writes the gradient w.r.t. layer l - 1's pre-activation to spec->off_out[top - l] using the bit rows
// spec->off_bits[top - l] of activation l - 1, and reports max |gradient w.r.t. layer l| to tail[spec->tail_idx[top - l]].
__attribute__((visibility("hidden"))) int launch_fused_chain_f2(int H, const MlpArgs &a, int top, int64_t n, void *pack_area, hipStream_t s, const FusedChain *spec)
{
    const int L = top;                      // chain positions 0 .. top - 1 = layers top .. 1
    if (L < 1 || L > TN_MLP_MAX_LAYERS) return tn::fail(TN_E_CONFIG, "mlp_bwd(fused chain): nothing to do");
    PackArgs p;
    p.L = L; p.H = H; p.transpose = 1;
    int64_t off = 0;
    for (int i = 0; i < L; ++i) {
        const int l = top - i;
        p.W[i] = a.W[l]; p.B[i] = a.B[l]; p.N[i] = a.N[l]; p.K[i] = a.K[l];
        p.off[i] = off;
        off += (int64_t)(H / 32) * (H / 16) * PAIR_B;
    }
    const int64_t chunks = off / CHUNK_B;
    p.stream = reinterpret_cast<unsigned char *>(pack_area);
    p.bias = reinterpret_cast<float *>(p.stream + chunks * CHUNK_B);
    p.consts = p.bias + (int64_t)L * H;
    fused_pack_kernel<<<dim3((unsigned)L, 4), dim3(1024), 0, s>>>(p);
    if (int rc = tn::check_launch("fused_pack_kernel(chain)")) return rc;
    FusedArgs f;
    f.stream = p.stream; f.bias = p.bias; f.consts = p.consts; f.L = L; f.n_chunks = (int)chunks; f.out_act = TN_ACT_NONE;
    f.e_rows = nullptr; f.e_rows_total = 0; f.e_off = 0; f.y = nullptr;
    f.rows = spec->rows; f.rows_total = spec->rows_total; f.out_linear = 0; f.tail = spec->tail; f.off_in = spec->off_in;
    for (int i = 0; i < TN_MLP_MAX_LAYERS; ++i) { f.off_out[i] = 0; f.off_bits[i] = 0; f.tail_idx[i] = -1; }
    for (int i = 0; i < L; ++i) { f.off_out[i] = spec->off_out[i]; f.off_bits[i] = spec->off_bits[i]; f.tail_idx[i] = spec->tail_idx[i]; }
    return H == 256 ? launch_chain<256>(f, n, s) : launch_chain<128>(f, n, s);
}

}  // namespace layers
}  // namespace tn
