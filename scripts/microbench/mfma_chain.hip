// fp32 MFMA rate as a function of waves per SIMD and independent accumulator chains per wave (v_mfma_f32_32x32x2_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float x = threadIdx.x * 1e-3f, y = 0.5f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    float s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(float *out, int wpc) {
    const int iters = 4000 / NACC;
    k<NACC><<<256, wpc * 64>>>(out, iters); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k<NACC><<<256, wpc * 64>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mf = 256.0 * wpc * iters * 8.0 * NACC;
    printf("%d waves/CU, %d chain(s): %.1f TFLOP/s, %.1f cycles per MFMA and SIMD at 2.4 GHz\n", wpc, NACC, mf * 4096 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (mf / 1024.0 * (wpc >= 4 ? 1.0 : 4.0 / wpc)));
}
int main() {
    float *out; hipMalloc(&out, 1 << 22);
    for (int wpc : {4, 8, 16}) { run<1>(out, wpc); run<2>(out, wpc); run<4>(out, wpc); }
    return 0;
}
