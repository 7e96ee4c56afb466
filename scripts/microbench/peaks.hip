// calibration of the box: fp32 MFMA rate, HBM read / write / copy bandwidth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void rd(const f4 *p, size_t n4, float *out) {
    f4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void wr(f4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = f4{1, 2, 3, 4};
}
__global__ __launch_bounds__(256) void cp(const f4 *p, f4 *q, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) q[i] = p[i];
}
template <class F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    float *out; hipMalloc(&out, 1 << 24);
    for (int wpc : {4, 8, 16}) {
        const int blocks = 256 * wpc / 4, iters = 20000;
        float ms = timeit([&] { mfma_loop<4><<<blocks, 256>>>(out, iters); });
        double fl = (double)blocks * 4 * iters * 4 * 4096.0;
        printf("mfma f32 32x32x2: %d waves/CU, 4 acc: %.1f TFLOP/s\n", wpc, fl / ms / 1e9);
    }
    { const int blocks = 256 * 2, iters = 20000;
      float ms = timeit([&] { mfma_loop<1><<<blocks, 256>>>(out, iters); });
      printf("mfma f32 32x32x2: 8 waves/CU, 1 acc (dependent chain): %.1f TFLOP/s\n", (double)blocks * 4 * iters * 4096.0 / ms / 1e9); }
    { const int blocks = 256 * 2, iters = 20000;
      float ms = timeit([&] { mfma_loop<2><<<blocks, 256>>>(out, iters); });
      printf("mfma f32 32x32x2: 8 waves/CU, 2 acc: %.1f TFLOP/s\n", (double)blocks * 4 * iters * 2 * 4096.0 / ms / 1e9); }
    size_t bytes = (size_t)4 << 30; f4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    for (int bl : {1024, 4096, 16384}) {
        float r = timeit([&] { rd<<<bl, 256>>>(a, bytes / 16, out); });
        float w = timeit([&] { wr<<<bl, 256>>>(b, bytes / 16); });
        float c = timeit([&] { cp<<<bl, 256>>>(a, b, bytes / 16); });
        printf("blocks %5d: read %.2f TB/s  write %.2f TB/s  copy %.2f TB/s (r+w)\n", bl, bytes / r / 1e9, bytes / w / 1e9, 2.0 * bytes / c / 1e9);
    }
    return 0;
}
