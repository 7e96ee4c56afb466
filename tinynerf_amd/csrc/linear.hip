// One plain Linear: y = x W^T + b, its input gradient and its weight / bias gradients.
//
// The reference's path holds exactly one torch.nn.Linear that is not part of an MLP stack: KPlanesExplicitOpacityDecoder.net
// (reference src/models.py:183-191, 96 x 96 on the K-Planes features; sigma = exp(<f, W f + b> - 1), the dot product and the
// truncated exponential are tn_basis_dot_*).  Small dense layers of any shape up to 128 x 128 on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32, the exact fp32 fma chain of the other MLP kernels) in the transposed formulation of
// mlp_device.h: D[feature][sample], A = weights out of LDS, B = the wave's 32-sample tile out of LDS.
//
//   lin_apply_kernel<false>   y  = x W^T + b        A[i][k] = W[i][k]   (N = out features, K = in features)
//   lin_apply_kernel<true>    gx = gy W             A[i][k] = W[k][i]   (N = in features,  K = out features)
//   lin_wgrad_kernel          gW[o][i] += sum_s gy[s][o] x[s][i],  gb[o] += sum_s gy[s][o]     (reduction index = the sample)
//
// HBM-bound by construction (2 * K * N FLOP against 4 (K + N) bytes per sample: 48 FLOP/B at 96 x 96, the ridge of the fp32
// MFMA roofline is 25): the kernels stage whole 32-sample tiles with coalesced loads and write 16-byte runs.
#include "mlp_device.h"
#include <algorithm>

namespace {

using tn::f32x16;
using tn::f32x4;

constexpr int LIN_WAVES = 4;
constexpr int LIN_MAX = 128;       // weights + four 32-sample tiles in LDS: (128 + 128) x 129 floats = 132 KB

__device__ __forceinline__ int frow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// y [n][N] = x [n][K] A^T (+ bias [N]);  A[i][k] = TRANS ? W[k * ldw + i] : W[i * ldw + k]
template <bool TRANS>
__global__ __launch_bounds__(LIN_WAVES * 64) void lin_apply_kernel(const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ bias,
                                                                   float *__restrict__ y, int K, int N, int ldw, int64_t n)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Kp = (K + 1) & ~1, Np = (N + 31) & ~31;       // k pairs, 32-row blocks (padding is zero)
    const int SA = Kp + 1;                                    // odd row stride: lane i reads A[i][k] conflict-free
    float *As = lds;                                          // [Np][SA]
    float *xs = lds + (size_t)Np * SA + (threadIdx.x >> 6) * 32 * SA;      // this wave's tile [32][SA]
    for (int e = threadIdx.x; e < Np * Kp; e += blockDim.x) {
        const int i = e / Kp, k = e % Kp;
        As[i * SA + k] = (i < N && k < K) ? (TRANS ? W[(int64_t)k * ldw + i] : W[(int64_t)i * ldw + k]) : 0.0f;
    }
    __syncthreads();
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * LIN_WAVES + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * LIN_WAVES) {
        const int64_t r0 = tile * 32;
        const int rows = (int)(n - r0 < 32 ? n - r0 : 32);
        // the tile's rows are one contiguous block of rows * K floats
        const float *src = x + r0 * K;
        for (int e = lane; e < 32 * Kp; e += 64) {
            const int r = e / Kp, k = e % Kp;
            xs[r * SA + k] = (r < rows && k < K) ? src[(int64_t)r * K + k] : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        for (int ob = 0; ob < Np / 32; ++ob) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * ob + frow(r, h);
                acc[r] = (bias != nullptr && f < N) ? bias[f] : 0.0f;
            }
            const float *ar = As + (32 * ob + j) * SA + h, *br = xs + j * SA + h;
            for (int s = 0; s < Kp / 2; ++s) acc = tn::mfma32(ar[2 * s], br[2 * s], acc);
            if (j < rows) {
                float *dst = y + (r0 + j) * N + 32 * ob + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = 32 * ob + 8 * q + 4 * h;
                    if (f + 3 < N && (N & 3) == 0) *reinterpret_cast<f32x4 *>(dst + 8 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (f + u < N) dst[8 * q + u] = acc[4 * q + u];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// gW[o][i] += sum_s gy[s][o] x[s][i] for the 32 x 32 block (blockIdx.y = ob * KB + kb); gb[o] += sum_s gy[s][o] (kb == 0)
__global__ __launch_bounds__(LIN_WAVES * 64) void lin_wgrad_kernel(const float *__restrict__ gy, const float *__restrict__ x, float *__restrict__ gW,
                                                                   float *__restrict__ gb, int K, int N, int ldw, int64_t n)
{
    __shared__ float tiles[LIN_WAVES][2][32 * 33];           // [sample][column] with an odd stride
    const int KB = gW != nullptr ? (K + 31) / 32 : 1;          // bias gradient alone: one k block, no x operand
    const int ob = blockIdx.y / KB, kb = blockIdx.y % KB;
    const int lane = tn::lane_id(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    float *gs = tiles[wave][0], *xs = tiles[wave][1];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bsum = 0.0f;
    const int64_t n_tiles = (n + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * LIN_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * LIN_WAVES) {
        const int64_t r0 = tile * 32;
        const int rows = (int)(n - r0 < 32 ? n - r0 : 32);
        for (int e = lane; e < 32 * 32; e += 64) {
            const int r = e >> 5, c = e & 31;
            gs[r * 33 + c] = (r < rows && 32 * ob + c < N) ? gy[(r0 + r) * N + 32 * ob + c] : 0.0f;
            xs[r * 33 + c] = (gW != nullptr && r < rows && 32 * kb + c < K) ? x[(r0 + r) * K + 32 * kb + c] : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = gs[(2 * s + h) * 33 + i];
            acc = tn::mfma32(a, xs[(2 * s + h) * 33 + i], acc);
            bsum += a;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // D[m = out feature frow(r, h)][n = in feature i]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 32 * ob + frow(r, h), c = 32 * kb + i;
        if (gW != nullptr && o < N && c < K) atomicAdd(&gW[(int64_t)o * ldw + c], acc[r]);
    }
    if (kb == 0 && gb != nullptr) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (h == 0 && 32 * ob + i < N) atomicAdd(&gb[32 * ob + i], bsum);
    }
}

size_t apply_lds(int K, int N) {
    const int Kp = (K + 1) & ~1, Np = (N + 31) & ~31;
    return ((size_t)Np + 32 * LIN_WAVES) * (Kp + 1) * sizeof(float);
}

template <bool TRANS>
int launch_apply(const float *x, const float *W, const float *b, float *y, int K, int N, int ldw, int64_t n, hipStream_t s)
{
    const size_t lds = apply_lds(K, N);
    if (lds > 160 * 1024) return tn::fail(TN_E_CONFIG, "tn_linear: the layer does not fit LDS");
    auto kern = lin_apply_kernel<TRANS>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { tn::set_error("tn_linear: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + LIN_WAVES - 1) / LIN_WAVES, 512));
    kern<<<dim3(blocks), dim3(LIN_WAVES * 64), lds, s>>>(x, W, b, y, K, N, ldw, n);
    return tn::check_launch("lin_apply_kernel");
}

}  // namespace

extern "C" int tn_linear_fwd(const float *x, const float *weight, const float *bias, int64_t n, int32_t in_features, int32_t out_features,
                             float *y, void *stream)
{
    TN_REQUIRE(n >= 0 && in_features >= 1 && out_features >= 1 && in_features <= LIN_MAX && out_features <= LIN_MAX, TN_E_SIZE,
               "tn_linear_fwd: 1 <= in_features, out_features <= 128");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && weight && y, TN_E_NULL, "tn_linear_fwd: null pointer");
    return launch_apply<false>(x, weight, bias, y, in_features, out_features, in_features, n, (hipStream_t)stream);
}

extern "C" int tn_linear_bwd(const float *x, const float *weight, const float *grad_y, int64_t n, int32_t in_features, int32_t out_features,
                             float *grad_x, float *grad_weight, float *grad_bias, void *stream)
{
    TN_REQUIRE(n >= 0 && in_features >= 1 && out_features >= 1 && in_features <= LIN_MAX && out_features <= LIN_MAX, TN_E_SIZE,
               "tn_linear_bwd: 1 <= in_features, out_features <= 128");
    if (n == 0) return TN_OK;
    TN_REQUIRE(weight && grad_y && (x || !grad_weight), TN_E_NULL, "tn_linear_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (grad_x)
        if (int rc = launch_apply<true>(grad_y, weight, nullptr, grad_x, out_features, in_features, in_features, n, s)) return rc;
    if (grad_weight || grad_bias) {        // either gradient alone is served: the bias sums ride in the k-block-0 workgroups
        const int64_t n_tiles = (n + 31) / 32;
        const int NB = (out_features + 31) / 32, KB = grad_weight ? (in_features + 31) / 32 : 1;
        const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + LIN_WAVES - 1) / LIN_WAVES, 1024 / (NB * KB) + 1));
        lin_wgrad_kernel<<<dim3(bx, (unsigned)(NB * KB)), dim3(LIN_WAVES * 64), 0, s>>>(grad_y, x, grad_weight, grad_bias, in_features, out_features,
                                                                                       in_features, n);
        if (int rc = tn::check_launch("lin_wgrad_kernel")) return rc;
    }
    return TN_OK;
}
