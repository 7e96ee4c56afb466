// VALU issue rate of one wave while another wave on the same SIMD streams fp32 MFMAs, with and without s_setprio.
// mode 0: VALU waves only; 1: MFMA + VALU waves, default priority; 2: VALU waves at s_setprio 3; 3: MFMA waves at s_setprio 3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void mix(float *out, int iters, int mode, unsigned long long *cycles) {
    const int wave = threadIdx.x >> 6, slot = wave >> 2;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[8]; for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3f + k;
    const float y = 1.0001f, z = 0.5f;
    if (mode == 2 && slot == 1) __builtin_amdgcn_s_setprio(3);
    if (mode == 3 && slot == 0) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (slot == 0) {
        if (mode != 0)
            for (int i = 0; i < iters; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y, z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z, y, acc, 0, 0, 0);
            }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], y, z);        // 32 independent-ish fmas (8 chains)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int k = 0; k < 8; ++k) s += v[k]; for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cycles[wave] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    const int iters = 20000;
    for (int mode : {0, 1, 2, 3}) {
        mix<<<256, 512>>>(out, iters, mode, cyc); hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); mix<<<256, 512>>>(out, iters, mode, cyc); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        printf("mode %d: kernel %.3f ms (MFMA alone: %.3f ms at 2.4 GHz); memtime ticks mfma-wave %llu valu-wave %llu\n",
               mode, ms, iters * 2 * 64 / 2.4e6, c[0], c[4]);
    }
    return 0;
}
