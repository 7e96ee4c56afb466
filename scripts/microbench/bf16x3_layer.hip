// One 64 -> 64 hidden layer on register-resident activations (the building block of the fused MLP heads, mlp_device.h):
//   (a) exact fp32:  v_mfma_f32_32x32x2_f32, 64 MFMAs x 64 cycles per 32-sample tile;
//   (b) "bf16x3":    every fp32 operand split EXACTLY into three bf16 terms by truncation (x = hi + mid + lo, 8 + 8 + 8
//       significant bits), six of the nine partial products on v_mfma_f32_32x32x16_bf16 (hh, hm, mh, mm, hl, lh: the three
//       dropped ones are <= 2^-23 of |x w|), fp32 accumulate: 48 MFMAs x 32 cycles per tile + the activation split on the VALU.
// Reports time per layer and the error of both against an fp64 evaluation of the same layer.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x3_layer bf16x3_layer.hip && ./bf16x3_layer
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64, T = 2;
constexpr int SF = H + 4;            // fp32 LDS row stride (floats)
constexpr int SB = 72;               // bf16 LDS row stride (bf16 elements): 144 B = 9 x 16 B (odd)

__device__ __forceinline__ void pin16(f32x16 &v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[r]));
}

// ---------------------------------------------------------------- (a) fp32, as tn::hidden_layer
__device__ __forceinline__ void layer_f32(const float *W, const float *bias, f32x16 (&x)[T], int i, int h) {
    f32x16 y[T];
#pragma unroll
    for (int ob = 0; ob < T; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[ob][r] = bias[32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
    for (int g = 0; g < 4 * T; ++g) {
        const int kb = g >> 2, q = g & 3;
        f32x4 w[T];
#pragma unroll
        for (int ob = 0; ob < T; ++ob) w[ob] = *reinterpret_cast<const f32x4 *>(W + (32 * ob + i) * SF + 8 * g + 4 * h);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ob = 0; ob < T; ++ob) y[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob][u], x[kb][4 * q + u], y[ob], 0, 0, 0);
    }
#pragma unroll
    for (int ob = 0; ob < T; ++ob) {
        pin16(y[ob]);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[ob][r] = fmaxf(y[ob][r], 0.0f);
    }
}

// ---------------------------------------------------------------- (b) bf16 x 3
// truncation split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const unsigned ha = ua & 0xffff0000u, hb = ub & 0xffff0000u;
    const float ra = a - __uint_as_float(ha), rb = b - __uint_as_float(hb);            // exact
    const unsigned ma = __float_as_uint(ra) & 0xffff0000u, mb = __float_as_uint(rb) & 0xffff0000u;
    const float la = ra - __uint_as_float(ma), lb = rb - __uint_as_float(mb);          // exact, <= 8 significant bits
    hi = (ha >> 16) | hb;
    mid = (ma >> 16) | mb;
    lo = (__float_as_uint(la) >> 16) | (__float_as_uint(lb) & 0xffff0000u);
}

// K step s = 2 kb + s' consumes the lane's registers 8 s' .. 8 s' + 7 of block kb, i.e. features 32 kb + 16 s' + 4 h + {0..3}
// and 32 kb + 16 s' + 8 + 4 h + {0..3}: the LDS copies of the weights hold exactly those 8 columns contiguously
// (Wp[row][s][h][e]), so the A operand of a step is one ds_read_b128 per bf16 term.
__device__ __forceinline__ void layer_b3(const __bf16 *Whi, const __bf16 *Wmid, const __bf16 *Wlo, const float *bias, f32x16 (&x)[T], int i, int h) {
    f32x16 y[T];
#pragma unroll
    for (int ob = 0; ob < T; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[ob][r] = bias[32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int kb = s >> 1, sp = s & 1;
        u32x4 bh, bm, bl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned a_, b_, c_;
            split2(x[kb][8 * sp + 2 * p], x[kb][8 * sp + 2 * p + 1], a_, b_, c_);
            bh[p] = a_; bm[p] = b_; bl[p] = c_;
        }
        const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
#pragma unroll
        for (int ob = 0; ob < T; ++ob) {
            const int off = (32 * ob + i) * SB + 16 * s + 8 * h;
            const bf16x8 Ah = *reinterpret_cast<const bf16x8 *>(Whi + off), Am = *reinterpret_cast<const bf16x8 *>(Wmid + off),
                         Al = *reinterpret_cast<const bf16x8 *>(Wlo + off);
            // small terms first
            y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, y[ob], 0, 0, 0);
            y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, y[ob], 0, 0, 0);
            y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, y[ob], 0, 0, 0);
            y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, y[ob], 0, 0, 0);
            y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, y[ob], 0, 0, 0);
            y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, y[ob], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ob = 0; ob < T; ++ob) {
        pin16(y[ob]);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[ob][r] = fmaxf(y[ob][r], 0.0f);
    }
}

// x: [n][64] row-major in, y: [n][64] out after `layers` applications of relu(W x + b) (same W every layer)
template <bool B3>
__global__ __launch_bounds__(512) void k(const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ xin,
                                         float *__restrict__ yout, int n_tiles, int layers)
{
    __shared__ __attribute__((aligned(16))) float wf[H * SF];
    __shared__ __attribute__((aligned(16))) __bf16 wb[3][H * SB];
    __shared__ float bs[H];
    for (int e = threadIdx.x; e < H * H; e += blockDim.x) {
        const int r = e / H, c = e % H;
        const float w = W[e];
        wf[r * SF + c] = w;
        // column c = 32 kb + 16 s' + 8 g + 4 h + u  ->  slot [s = 2 kb + s'][h][e = 4 g + u]
        const int kb = c >> 5, sp = (c >> 4) & 1, g = (c >> 3) & 1, hh = (c >> 2) & 1, u = c & 3;
        const int slot = 16 * (2 * kb + sp) + 8 * hh + 4 * g + u;
        const unsigned uw = __float_as_uint(w), h1 = uw & 0xffff0000u;
        const float r1 = w - __uint_as_float(h1);
        const unsigned m1 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(m1);
        reinterpret_cast<unsigned short *>(wb[0])[r * SB + slot] = (unsigned short)(h1 >> 16);
        reinterpret_cast<unsigned short *>(wb[1])[r * SB + slot] = (unsigned short)(m1 >> 16);
        reinterpret_cast<unsigned short *>(wb[2])[r * SB + slot] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
    if (threadIdx.x < H) bs[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    for (int tile = blockIdx.x * 8 + wave; tile < n_tiles; tile += gridDim.x * 8) {
        f32x16 x[T];
        const float *xr = xin + (size_t)(tile * 32 + j) * H;
#pragma unroll
        for (int kb = 0; kb < T; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[kb][r] = xr[32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h];
        for (int l = 0; l < layers; ++l) {
            if constexpr (B3) layer_b3(wb[0], wb[1], wb[2], bs, x, j, h);
            else layer_f32(wf, bs, x, j, h);
        }
        float *yr = yout + (size_t)(tile * 32 + j) * H;
#pragma unroll
        for (int kb = 0; kb < T; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) yr[32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h] = x[kb][r];
    }
}

int main() {
    const int n = 1 << 20, n_tiles = n / 32;
    std::vector<float> W(H * H), b(H), x((size_t)n * H);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (auto &v : W) v = rnd() * 0.25f;
    for (auto &v : b) v = rnd() * 0.1f;
    for (auto &v : x) v = rnd();
    float *dW, *db, *dx, *dy;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dx, x.size() * 4); hipMalloc(&dy, x.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> y(x.size());
    for (int layers : {1, 4}) {
        // fp64 reference on the first 4096 samples
        const int nref = 4096;
        std::vector<double> ref((size_t)nref * H);
        for (int s = 0; s < nref; ++s) {
            double cur[H], nxt[H];
            for (int c = 0; c < H; ++c) cur[c] = x[(size_t)s * H + c];
            for (int l = 0; l < layers; ++l) {
                for (int r = 0; r < H; ++r) { double a = b[r]; for (int c = 0; c < H; ++c) a += (double)W[r * H + c] * cur[c]; nxt[r] = a > 0 ? a : 0; }
                for (int r = 0; r < H; ++r) cur[r] = (float)nxt[r] == 0 ? 0.0 : nxt[r];
            }
            for (int c = 0; c < H; ++c) ref[(size_t)s * H + c] = cur[c];
        }
        for (int v = 0; v < 2; ++v) {
            auto launch = [&](int L) { if (v) k<true><<<256, 512>>>(dW, db, dx, dy, n_tiles, L); else k<false><<<256, 512>>>(dW, db, dx, dy, n_tiles, L); };
            launch(layers); hipDeviceSynchronize();
            hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
            double emax = 0, rmax = 0;
            for (size_t e = 0; e < ref.size(); ++e) { emax = fmax(emax, fabs(y[e] - ref[e])); rmax = fmax(rmax, fabs(ref[e])); }
            hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
            const int LT = 64;      // long chain for the timing (load / store amortised)
            launch(LT); hipDeviceSynchronize();
            hipEventRecord(a); launch(LT); hipEventRecord(c); hipEventSynchronize(c);
            float ms; hipEventElapsedTime(&ms, a, c);
            printf("%s %d layer(s): max |err| %.3e (max |ref| %.3f, rel %.2e);  %.3f us per layer and 2^20 samples = %.1f TFLOP/s fp32-equivalent\n",
                   v ? "bf16x3" : "fp32  ", layers, emax, rmax, emax / rmax, ms * 1e3 / LT, 2.0 * H * H * n / (ms * 1e-3 / LT) / 1e12);
        }
    }
    return 0;
}
