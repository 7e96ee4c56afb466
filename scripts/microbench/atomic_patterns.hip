#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// MODE 0: lane l adds to line idx[wave_iter*64 + l], dword (it&15)    -> 64 lines / instruction (1 dword each)
// MODE 1: lanes 0..31 -> line A dwords 0..31, lanes 32..63 -> line B  -> 2 lines / instruction (32 dwords each)
// MODE 2: like 0 but lanes l, l+32 share a line (2 dwords per line)   -> 32 lines / instruction
template <int MODE>
__global__ void k(float* buf, const int* idx, int n_iters, int n_idx) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int it = 0; it < n_iters; ++it) {
        long base = (wave * n_iters + it) * 64;
        if (MODE == 0) {
            int line = idx[(base + lane) % n_idx];
            atomicAdd(buf + (long)line * 32 + (it & 31), 1.0f);
        } else if (MODE == 1) {
            int line = idx[(base + (lane >> 5)) % n_idx];
            atomicAdd(buf + (long)line * 32 + (lane & 31), 1.0f);
        } else {
            int line = idx[(base + (lane & 31)) % n_idx];
            atomicAdd(buf + (long)line * 32 + (it & 15) + 16 * (lane >> 5), 1.0f);
        }
    }
}
int main() {
    const long lines = 1 << 18;  // 32 MB region
    float* buf; hipMalloc(&buf, lines * 128); hipMemset(buf, 0, lines * 128);
    const int n_idx = 1 << 22;
    int* h = (int*)malloc(n_idx * 4);
    for (int mode_idx = 0; mode_idx < 2; ++mode_idx) {   // 0: random lines, 1: locally coherent (runs of nearby lines)
        srand(1);
        int cur = 0;
        for (int i = 0; i < n_idx; ++i) { if (mode_idx == 0) h[i] = rand() % lines; else { if (i % 32 == 0) cur = rand() % lines; cur = (cur + (rand() % 3)) % lines; h[i] = cur; } }
        int* idx; hipMalloc(&idx, n_idx * 4); hipMemcpy(idx, h, n_idx * 4, hipMemcpyHostToDevice);
        for (int m = 0; m < 3; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int blocks = 2048, iters = 256;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (m == 0) k<0><<<blocks, 256>>>(buf, idx, iters, n_idx);
                if (m == 1) k<1><<<blocks, 256>>>(buf, idx, iters, n_idx);
                if (m == 2) k<2><<<blocks, 256>>>(buf, idx, iters, n_idx);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double atoms = (double)blocks * 4 * iters * 64;
            printf("pattern %s mode %d: %.3f ms  %.1f G lane-atomics/s  %.1f G instr/s\n", mode_idx ? "coherent" : "random", m, ms, atoms / ms / 1e6, atoms / 64 / ms / 1e6);
        }
        hipFree(idx);
    }
    return 0;
}
