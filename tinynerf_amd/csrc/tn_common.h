// Shared host/device helpers for libtinynerf_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tinynerf_hip.h"

#define TN_WAVE 64

namespace tn {

void set_error(const char *fmt, ...);
void warn_once(int id, const char *fmt, ...);      // (common.hip) a valid call on a slow general-shape fallback: said once per id

inline int fail(int code, const char *what) {
    set_error("%s", what);
    return code;
}

inline int check_launch(const char *kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", kernel, hipGetErrorString(e));
        return (int)e;
    }
    return TN_OK;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// number of set bits of `m` strictly below this lane
__device__ __forceinline__ int rank_below(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// counter-based U[0,1): two rounds of a 64-bit mix (splitmix64 finaliser) -> 24 random bits
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}

}  // namespace tn

#define TN_REQUIRE(cond, code, msg) do { if (!(cond)) return tn::fail((code), (msg)); } while (0)
