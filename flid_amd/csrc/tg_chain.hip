// Row-block chain kernels: everything of a temporal-attention layer that sits between the fused attention kernel and the layer's
// output (forward), and between the layer's output gradient and the attention backward (backward), as ONE launch each.
//
// replaces per layer: the value projection, residual_fc, dropout + residual + LayerNorm (models/modules.py:228-238) and the MergeLayer
//                     (models/modules.py:58-69) -- five product launches + one LayerNorm launch forward, five + one backward.
//
// Why: these are products of a tall activation (R rows) with weights of at most 272 x 444.  As separate launches each one pays
// its own prologue / epilogue and a round trip of its intermediate through HBM; for the 1 200-row root layer they are pure
// launch latency (13 launches of 5-16 us for 0.6 GFLOP).  Here a workgroup owns 32 rows and walks the whole chain with the
// intermediates in LDS; the weights stream from L2 straight into MFMA operand registers (k-contiguous rows: lane (c, h) of a
// 32-column tile loads W[col c][16 s + 8 h .. + 7] as two float4 -- no LDS staging for the weights at all).
//
// Arithmetic: split-bf16 (x = hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate), the same as
// tg_gemm_bf16x3.hip; the activation fragment of a k-step is split once and reused by all of a wave's column tiles.
#include <math.h>

#include "tg_common.h"

namespace {

using tg::kWave;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BR = 32;            // rows per workgroup
constexpr int NW = 4;             // waves per workgroup
constexpr int MAXT = 3;           // column tiles per wave and stage: N <= 32 * NW * MAXT = 384

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo16(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi16(uint32_t p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

// 8 consecutive floats -> bf16x8 hi and lo fragments
__device__ __forceinline__ void split8(const float4& a, const float4& b, bf16x8& hi, bf16x8& lo) {
    uint4 h, l;
    h.x = pack_bf16(a.x, a.y); h.y = pack_bf16(a.z, a.w); h.z = pack_bf16(b.x, b.y); h.w = pack_bf16(b.z, b.w);
    l.x = pack_bf16(a.x - lo16(h.x), a.y - hi16(h.x));
    l.y = pack_bf16(a.z - lo16(h.y), a.w - hi16(h.y));
    l.z = pack_bf16(b.x - lo16(h.z), b.y - hi16(h.z));
    l.w = pack_bf16(b.z - lo16(h.w), b.w - hi16(h.w));
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t idx, float p) {          // == tg_layer.hip (dropout after residual_fc)
    if (p <= 0.f) return 1.f;
    const float u = (float)(tg::mix32(seed ^ ((uint64_t)idx * 0x9E3779B97F4A7C15ULL)) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return u >= p ? 1.f / (1.f - p) : 0.f;
}

// One product stage of a workgroup:  C (BR x N) = A (BR x K) W^T, W: (N x K) row-major in global memory (ldw), both k-contiguous.
// A comes from `a_row` = pointer to the lane's row (lane & 31) at k = 0 -- LDS or global alike -- with `a_ok` false for rows past
// the end (read as zeros).  K % 4 == 0.  Wave w owns the column tiles w, w + NW, ...; `epi(col0, acc)` receives each finished tile
// in the MFMA C/D layout: acc[r] = C[row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col0 + (lane & 31)].
template <class Epi>
__device__ __forceinline__ void stage(const float* __restrict__ a_row, bool a_ok, const float* __restrict__ W, int64_t ldw, int N, int K, Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int tiles = (N + 31) >> 5;
    f32x16 acc[MAXT];
    const float* wrow[MAXT];
    bool live[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int tile = wave + NW * t;
        live[t] = tile < tiles;
        int col = tile * 32 + c;
        col = col < N ? col : N - 1;                              // clamped columns are computed but never stored
        wrow[t] = W + (int64_t)col * ldw + 8 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int steps = (K + 15) >> 4;
    for (int s = 0; s < steps; ++s) {
        const int k = 16 * s + 8 * h;
        const bool k0 = k < K, k1 = k + 4 < K;                    // K % 4 == 0: each half of the 8-float fragment is wholly in or out
        float4 a0 = z4, a1 = z4;
        if (a_ok && k0) a0 = *reinterpret_cast<const float4*>(a_row + k);
        if (a_ok && k1) a1 = *reinterpret_cast<const float4*>(a_row + k + 4);
        float4 b0[MAXT], b1[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            b0[t] = z4; b1[t] = z4;
            if (live[t] && k0) b0[t] = *reinterpret_cast<const float4*>(wrow[t] + 16 * s);
            if (live[t] && k1) b1[t] = *reinterpret_cast<const float4*>(wrow[t] + 16 * s + 4);
        }
        bf16x8 ah, al;
        split8(a0, a1, ah, al);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (!live[t]) continue;                               // wave-uniform
            bf16x8 bh, bl;
            split8(b0[t], b1[t], bh, bl);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
        if (live[t]) epi((wave + NW * t) * 32, acc[t]);
}

// visit the 16 elements of a tile held by this lane: f(row_in_block, col, value)
template <class F>
__device__ __forceinline__ void for_tile(int col0, const f32x16& acc, F f) {
    const int lane = threadIdx.x & 63;
    const int col = col0 + (lane & 31), rb = 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) f((r & 3) + 8 * (r >> 2) + rb, col, acc[r]);
}

struct FwdArgs {
    const float *agg, *own, *raw, *cosb, *Wv, *Wr, *br, *ln_g, *ln_b, *W1, *b1, *W2, *b2;
    int64_t own_ld, raw_ld, R;
    float *ctx, *res, *y, *mean, *rstd, *f1, *out;
    int H, dn, T, dq, dk, hd;
    float p;
    uint64_t seed;
};

// LDS (floats): ctx [BR][dq + 4] | xs [BR][dq + 4] | yraw [BR][dq + dn + 4] ; f1 [BR][dn + 4] aliases ctx
__global__ void __launch_bounds__(NW* kWave) chain_fwd_kernel(FwdArgs a) {
    extern __shared__ __align__(16) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dq = a.dq, dn = a.dn, dk = a.dk, hd = a.hd, H = a.H;
    const int ldc = dq + 4, ldy = dq + dn + 4, ldf = dn + 4;
    float* s_ctx = sm;
    float* s_xs = s_ctx + BR * ldc;
    float* s_yr = s_xs + BR * ldc;
    float* s_f1 = s_ctx;
    const int64_t row0 = (int64_t)blockIdx.x * BR;
    const int64_t myrow = row0 + (lane & 31);
    const bool row_ok = myrow < a.R;
    const int64_t hk = (int64_t)H * dk;

    // ---- S1: ctx_h = agg_h Wv_h^T  (per head; A straight from global memory) ------------------------------------------------------
    for (int h = 0; h < H; ++h) {
        stage(a.agg + (row_ok ? myrow : 0) * hk + (int64_t)h * dk, row_ok, a.Wv + (int64_t)h * hd * dk, dk, hd, dk,
              [&](int col0, const f32x16& acc) {
                  for_tile(col0, acc, [&](int r, int c, float v) {
                      if (c < hd) {
                          s_ctx[r * ldc + h * hd + c] = v;
                          if (row0 + r < a.R) a.ctx[(row0 + r) * dq + h * hd + c] = v;
                      }
                  });
              });
    }
    __syncthreads();
    // ---- S2: res = ctx Wr^T + br ; xs = dropout(res) + [own | cos b] --------------------------------------------------------------
    stage(s_ctx + (lane & 31) * ldc, true, a.Wr, dq, dq, dq, [&](int col0, const f32x16& acc) {
        for_tile(col0, acc, [&](int r, int c, float v) {
            if (c < dq) {
                const int64_t gr = row0 + r;
                const float rv = v + a.br[c];
                float x = 0.f;
                if (gr < a.R) {
                    a.res[gr * dq + c] = rv;
                    x = rv * keep_scale(a.seed, gr * dq + c, a.p) + (c < dn ? a.own[gr * a.own_ld + c] : a.cosb[c - dn]);
                }
                s_xs[r * ldc + c] = x;
            }
        });
    });
    __syncthreads();
    // ---- LayerNorm (one wave per 8 rows), y -> global and [y | raw] -> LDS --------------------------------------------------------
    for (int r = wave; r < BR; r += NW) {
        const int64_t gr = row0 + r;
        float s = 0.f;
        for (int c = lane; c < dq; c += kWave) s += s_xs[r * ldc + c];
        const float mu = tg::wave_sum(s) / dq;
        float q = 0.f;
        for (int c = lane; c < dq; c += kWave) { const float d = s_xs[r * ldc + c] - mu; q = fmaf(d, d, q); }
        const float rs = rsqrtf(tg::wave_sum(q) / dq + 1e-5f);
        for (int c = lane; c < dq; c += kWave) {
            const float yv = (s_xs[r * ldc + c] - mu) * rs * a.ln_g[c] + a.ln_b[c];
            s_yr[r * ldy + c] = yv;
            if (gr < a.R) a.y[gr * dq + c] = yv;
        }
        for (int c = lane; c < dn; c += kWave) s_yr[r * ldy + dq + c] = gr < a.R ? a.raw[gr * a.raw_ld + c] : 0.f;
        if (lane == 0 && gr < a.R) { a.mean[gr] = mu; a.rstd[gr] = rs; }
    }
    __syncthreads();
    // ---- S3: f1 = relu([y | raw] W1^T + b1) ------------------------------------------------------------------------------------------
    stage(s_yr + (lane & 31) * ldy, true, a.W1, dq + dn, dn, dq + dn, [&](int col0, const f32x16& acc) {
        for_tile(col0, acc, [&](int r, int c, float v) {
            if (c < dn) {
                const float f = fmaxf(v + a.b1[c], 0.f);
                s_f1[r * ldf + c] = f;
                if (row0 + r < a.R) a.f1[(row0 + r) * dn + c] = f;
            }
        });
    });
    __syncthreads();
    // ---- S4: out = f1 W2^T + b2 -------------------------------------------------------------------------------------------------------
    stage(s_f1 + (lane & 31) * ldf, true, a.W2, dn, dn, dn, [&](int col0, const f32x16& acc) {
        for_tile(col0, acc, [&](int r, int c, float v) {
            if (c < dn && row0 + r < a.R) a.out[(row0 + r) * dn + c] = v + a.b2[c];
        });
    });
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace tg {

// post-attention forward chain of one layer; false = dimensions not covered (the caller runs the separate launches)
bool chain_fwd(const tg_layer_desc* L, hipStream_t s) {
    const tg_attn_desc& at = L->attn;
    const int H = at.heads, dn = at.dn, T = at.dt_dim, dq = dn + T, dk = dn + at.de + T, hd = dq / H;
    if (dn % 4 || dq % 4 || dk % 4 || hd % 4 || dq > 32 * NW * MAXT || hd > 32 * NW * MAXT || L->own_ld % 4 || L->raw_ld % 4) return false;
    if (!(al16(L->agg) && al16(L->own) && al16(L->raw) && al16(L->params.Wv) && al16(L->params.Wr) && al16(L->params.W1) && al16(L->params.W2))) return false;
    const size_t lds = sizeof(float) * BR * ((size_t)2 * (dq + 4) + (dq + dn + 4));
    if (lds > 160 * 1024) return false;
    FwdArgs a{L->agg, L->own, L->raw, L->cosb, L->params.Wv, L->params.Wr, L->params.br, L->params.ln_g, L->params.ln_b, L->params.W1, L->params.b1,
              L->params.W2, L->params.b2, L->own_ld, L->raw_ld, at.m, L->ctx, L->res, L->y, L->mean, L->rstd, L->f1, L->out, H, dn, T, dq, dk, hd,
              L->res_dropout_p, L->res_seed};
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const double flops = 2.0 * at.m * ((double)dq * dk + (double)dq * dq + (double)dn * (dq + dn) + (double)dn * dn);
    ProfScope prof("gemm", flops, s);
    chain_fwd_kernel<<<(unsigned)((at.m + BR - 1) / BR), NW * kWave, lds, s>>>(a);
    return true;
}

}  // namespace tg
