// follow-up of scripts/microbench/atomic_patterns.hip: does a wave64 atomic that covers 256 CONTIGUOUS bytes (two adjacent lines) retire
// faster than one covering two unrelated lines?  And how does the rate scale with the number of workgroups (per-CU or per-chip limit)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// MODE 1: lanes 0..31 -> line A, lanes 32..63 -> unrelated line B        MODE 3: lanes 0..63 -> lines A, A + 1 (256 contiguous bytes)
// MODE 4: like 1 but B = A + W (a texel one row below: what the scatter issues today, W = 512 lines)
template <int MODE>
__global__ void k(float* buf, const int* idx, int n_iters, int n_idx) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int it = 0; it < n_iters; ++it) {
        long base = (wave * n_iters + it) * 2;
        int line;
        if (MODE == 1) line = idx[(base + (lane >> 5)) % n_idx];
        else if (MODE == 3) line = (idx[base % n_idx] & ~1) + (lane >> 5);
        else line = idx[base % n_idx] + 512 * (lane >> 5);
        atomicAdd(buf + (long)line * 32 + (lane & 31), 1.0f);
    }
}
int main() {
    const long lines = 1 << 18;  // 32 MB region (+ slack for mode 4)
    float* buf; hipMalloc(&buf, (lines + 1024) * 128); hipMemset(buf, 0, (lines + 1024) * 128);
    const int n_idx = 1 << 22;
    int* h = (int*)malloc(n_idx * 4);
    srand(1);
    for (int i = 0; i < n_idx; ++i) h[i] = rand() % lines;
    int* idx; hipMalloc(&idx, n_idx * 4); hipMemcpy(idx, h, n_idx * 4, hipMemcpyHostToDevice);
    for (int blocks : {2048, 1024, 512, 256, 128}) {
        for (int m : {1, 3, 4}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 256 * (2048 / blocks);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (m == 1) k<1><<<blocks, 256>>>(buf, idx, iters, n_idx);
                if (m == 3) k<3><<<blocks, 256>>>(buf, idx, iters, n_idx);
                if (m == 4) k<4><<<blocks, 256>>>(buf, idx, iters, n_idx);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double atoms = (double)blocks * 4 * iters * 64;
            printf("blocks %4d (x 4 waves) mode %d: %.3f ms  %.1f G lane-atomics/s\n", blocks, m, ms, atoms / ms / 1e6);
        }
    }
    return 0;
}
