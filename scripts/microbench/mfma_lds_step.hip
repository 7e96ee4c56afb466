// One wave per SIMD (4 waves per workgroup, one workgroup per CU): what a "k step" of the cross-layer MLP kernel costs -- six
// v_mfma_f32_32x32x16_f16 on two accumulators in turn, with / without four ds_read_b128 of the operands two steps ahead, with / without
// VALU fillers, with the reads in front of the MFMAs or one per gap.  Round 6 (csrc/mlp_fused_f2.hip).
//   bit 0: MFMAs   bit 1: ds_read_b128 x 4 per step   bit 2: 12 VALU per step   bit 3: reads spread over the gaps   bit 4: B operand = registers the reads do not touch
//   bit 5: MFMA A operands independent of the reads (reads only land in registers)
//   bit 6: operands are random fp16 values in +-[2^-3, 2^2) instead of the constants 1.0 / 0.5 (data toggling: the chip's clock under matrix load
//          depends on it -- GRBM_GUI_ACTIVE / duration of the real kernels says 1.5 - 1.9 GHz, this loop with constants runs at 2.26)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int OFF> __device__ __forceinline__ u32x4 lds16(unsigned addr) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v; }
__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
template <int MODE>
__global__ __launch_bounds__(256) void step_kernel(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    auto rnd16 = [](unsigned x) {            // a random fp16 pattern: sign, exponent 12 .. 16, 10 random significand bits
        x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        return ((x & 1u) << 15) | ((12u + (x >> 1) % 5u) << 10) | ((x >> 8) & 0x3ffu);
    };
    auto rnd32 = [&](unsigned x) { return rnd16(2 * x) | (rnd16(2 * x + 1) << 16); };
    for (int e = threadIdx.x; e < 32768; e += 256) reinterpret_cast<unsigned *>(lds)[e] = (MODE & 64) ? rnd32(e + 32768 * blockIdx.x) : 0x3c003c00u;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + (threadIdx.x & 63) * 16;
    f32x16 c0, c1; for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    u32x4 w[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) w[a][b] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 bop = u32x4{0x38003800u, 0x38003800u, 0x38003800u, 0x38003800u}, kop = bop;
    if (MODE & 64) for (int q = 0; q < 4; ++q) { bop[q] = rnd32(threadIdx.x * 8 + q + 77777 * blockIdx.x); kop[q] = rnd32(threadIdx.x * 8 + 4 + q); }
    asm volatile("" : "+v"(bop), "+v"(kop));
    float f[12]; for (int k = 0; k < 12; ++k) f[k] = threadIdx.x * 1e-3f + k;
    const float y = 1.0001f, z = 0.5f;
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {                     // four steps per iteration: window slot (t + 2) & 3 is requested, slot t consumed
            u32x4 (&n)[4] = w[(t + 2) & 3];
            u32x4 (&a)[4] = w[t];
            auto req = [&](int k) {
                if (!(MODE & 2)) return;
                if (k == 0) n[0] = lds16<0>(addr + t * 4096); if (k == 1) n[1] = lds16<1024>(addr + t * 4096);
                if (k == 2) n[2] = lds16<2048>(addr + t * 4096); if (k == 3) n[3] = lds16<3072>(addr + t * 4096);
            };
            auto valu = [&](int g) { if (MODE & 4) { f[2 * g] = fmaf(f[2 * g], y, z); f[2 * g + 1] = fmaf(f[2 * g + 1], y, z); } };
            if (!(MODE & 8)) { req(0); req(1); req(2); req(3); }
            if (MODE & 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"((MODE & 8) ? 4 : 8));
            __builtin_amdgcn_sched_barrier(0);
            const bool ind = (MODE & 32) != 0;
            if (MODE & 1) {
                c0 = mm(ind ? kop : a[1], bop, c0); valu(0); __builtin_amdgcn_sched_barrier(0);
                c1 = mm(ind ? kop : a[3], bop, c1); valu(1); __builtin_amdgcn_sched_barrier(0);
                c0 = mm(ind ? kop : a[0], bop, c0); valu(2); if (MODE & 8) req(0); __builtin_amdgcn_sched_barrier(0);
                c1 = mm(ind ? kop : a[2], bop, c1); valu(3); if (MODE & 8) req(1); __builtin_amdgcn_sched_barrier(0);
                c0 = mm(ind ? kop : a[0], bop, c0); valu(4); if (MODE & 8) req(2); __builtin_amdgcn_sched_barrier(0);
                c1 = mm(ind ? kop : a[2], bop, c1); valu(5); if (MODE & 8) req(3); __builtin_amdgcn_sched_barrier(0);
            } else {
                for (int g = 0; g < 6; ++g) valu(g);
                if (MODE & 8) { req(0); req(1); req(2); req(3); }
                asm volatile("" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0; for (int k = 0; k < 12; ++k) s += f[k]; for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += __uint_as_float(w[a][b][0]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float *out, int iters) {
    auto k = step_kernel<MODE>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    k<<<256, 256, 131072>>>(out, iters); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k<<<256, 256, 131072>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %2d (%s%s%s%s%s): %.3f ms = %.1f ns per step (6 MFMAs = 192 matrix-pipe cycles = 80 ns at 2.4 GHz)\n", MODE, MODE & 1 ? "mfma " : "", MODE & 2 ? "reads " : "",
           MODE & 4 ? "valu " : "", MODE & 8 ? "spread " : "", MODE & 32 ? (MODE & 64 ? "independent random " : "independent ") : (MODE & 64 ? "random " : ""), ms, ms * 1e6 / (iters * 4.0));
}
int main() {
    float *out; hipMalloc(&out, 1 << 20);
    const int iters = 20000;
    run<1>(out, iters); run<2>(out, iters); run<3>(out, iters); run<4>(out, iters); run<5>(out, iters); run<7>(out, iters); run<11>(out, iters); run<15>(out, iters);
    run<35>(out, iters); run<39>(out, iters); run<6>(out, iters);
    run<1 + 64>(out, iters); run<3 + 64>(out, iters); run<15 + 64>(out, iters); run<1>(out, iters); run<15>(out, iters); run<15 + 64>(out, 4 * iters);
    return 0;
}
