// Split-bf16 helpers shared by the product kernels (tg_gemm_rows.hip, tg_wgrad.hip, tg_chain.hip): x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi); a product keeps hi*hi + hi*lo + lo*hi (see tg_gemm_bf16x3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tgs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);     // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xFFFF0000u); }
// 4 floats -> 4 bf16 hi + 4 bf16 lo (two 8-byte words)
__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
    hi.x = pack_bf16(v.x, v.y);
    hi.y = pack_bf16(v.z, v.w);
    lo.x = pack_bf16(v.x - bf16_lo_to_f32(hi.x), v.y - bf16_hi_to_f32(hi.x));
    lo.y = pack_bf16(v.z - bf16_lo_to_f32(hi.y), v.w - bf16_hi_to_f32(hi.y));
}

// LDS image of a row block for the 16x16x32 MFMA: 32-k chunks of [row][64 B]; the four 16-byte slots of a row are permuted so that
// the fragment read (lane l: row l & 15, slot l >> 4) is conflict-free for ds_read_b128's lane groups
__device__ __forceinline__ int slot_swz(int row) { return ((row >> 3) & 1) << 1; }
// byte offset, inside one plane of one chunk, of the 4 floats [cw, cw + 4) (cw = column inside the chunk, a multiple of 4) of `row`
__device__ __forceinline__ int chunk_off(int row, int cw) { return row * 64 + ((((cw >> 3) ^ slot_swz(row))) << 4) + ((cw >> 2) & 1) * 8; }
// fragment read offset of lane l inside one plane of one chunk (add 1024 per 16-row block)
__device__ __forceinline__ int frag_off(int lane) { return (lane & 15) * 64 + ((((lane >> 4) ^ slot_swz(lane & 15))) << 4); }

}  // namespace tgs
