// LDS read-modify-write rates on gfx950: ds_add_f32 vs ds_add_u32 vs plain ds_read / v_add / ds_write, 64 lanes on 64
// consecutive dwords (conflict-free), 16 waves per CU on every CU.   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
    __shared__ float t[16 * 64 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 64 * 8; i += 1024) t[i] = 0.f;
    __syncthreads();
    float *p = t + w * 512 + lane;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float *q = p + 64 * ((r + it) & 7);
            if (MODE == 0) atomicAdd(q, v);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(q), (unsigned)lane);
            else if (MODE == 2) { *q = *q + v; }
            else { asm volatile("ds_add_f32 %0, %1" :: "v"((unsigned)(uintptr_t)(q) ), "v"(v) : "memory"); }
        }
    }
    __syncthreads();
    out[blockIdx.x * 1024 + threadIdx.x] = t[threadIdx.x];
}
int main()
{
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 2000;
    const char *names[3] = {"ds_add_f32 (atomicAdd float)", "ds_add_u32", "ds_read + v_add + ds_write"};
    for (int m = 0; m < 3; ++m) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (m == 0) k<0><<<256, 1024>>>(out, iters); else if (m == 1) k<1><<<256, 1024>>>(out, iters); else k<2><<<256, 1024>>>(out, iters);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ops = 256.0 * 16 * iters * 8;            // wave-level instructions
        printf("%-32s %.3f ms  %.1f cycles per wave-instruction per CU (at 2.4 GHz)  %.1f G lane-ops/s\n", names[m], ms,
               ms * 1e-3 * 2.4e9 / (16.0 * iters * 8), ops * 64 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
