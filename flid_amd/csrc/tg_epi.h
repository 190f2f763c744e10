// Element-wise pieces of the pre-LN transformer block (models/DyGFormer.py:448-461) as device functions, shared by the stand-alone
// passes (tg_seq.hip) and the product epilogue that fuses them (tg_gemm_bf16x3.hip): erf GELU, its derivative, and the counter-based
// dropout mask of tg_dropout -- keep / rescale factor of element i of a tensor under `seed`.
#pragma once
#include "tg_common.h"

namespace tg {

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    return cdf + v * pdf;
}
__device__ __forceinline__ float drop_keep(uint64_t seed, int64_t i, float p, float scale) {
    const float u = (float)(mix32(seed ^ ((uint64_t)i * 0x9E3779B97F4A7C15ULL)) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return u >= p ? scale : 0.f;
}

// what a product's epilogue does with its (row, col) element v = A B^T + bias:
//   EPI_MASK      C = aux > 0 ? v : 0                       (ReLU backward)
//   EPI_GELU_DROP C = v, out2 = dropout(gelu(v))           (feed-forward layer 1: the pre-activation is kept for the backward)
//   EPI_GELU_BWD  C = v * dropmask * gelu'(aux)             (gradient w.r.t. that pre-activation; aux = the pre-activation)
//   EPI_RES_DROP  C = aux + dropout(v)                      (a residual branch's end)
// dropout index = row * idx_ld + col (the flat index of the contiguous (rows, idx_ld) tensor the stand-alone pass would see)
enum { EPI_NONE = 0, EPI_MASK = 1, EPI_GELU_DROP = 2, EPI_GELU_BWD = 3, EPI_RES_DROP = 4 };
struct Epi {
    int op;
    const float* aux; int64_t ld_aux;
    float* out2; int64_t ld_out2;
    float p; uint64_t seed; int64_t idx_ld;
};

}  // namespace tg
