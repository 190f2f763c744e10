// Shared internals of libflid_tg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "flid_tg.h"

namespace tg {

void set_error(const std::string& s);

#define TG_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::tg::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return TG_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

#define TG_REQUIRE(cond, msg)                                                                      \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::tg::set_error(std::string("invalid argument: ") + msg);                              \
            return TG_EINVAL;                                                                      \
        }                                                                                          \
    } while (0)

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        return TG_EHIP;
    }
    return TG_OK;
}

// One incidence of the time-sorted adjacency (16 B): a lookup of the newest k is one contiguous read.
struct __attribute__((aligned(16))) Incidence {
    int32_t nbr;
    int32_t eid;
    double t;
};

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kMaxGridBlocks = 2048; // 256 CUs x 8 resident blocks: grid-stride beyond this

// ---- wave64 helpers -------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// counter-based RNG for attention dropout: the same (seed, row, head, slot) gives the same bit in fwd and bwd
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)(x >> 11);
}
__device__ __forceinline__ float dropout_keep_scale(uint64_t seed, int64_t row, int head, int slot, float p) {
    if (p <= 0.f) return 1.f;
    uint64_t key = seed ^ ((uint64_t)row * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)(head * 1315423911u + slot) << 20);
    float u = (float)(mix32(key) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f / (1.0f - p) : 0.f;
}

}  // namespace tg
