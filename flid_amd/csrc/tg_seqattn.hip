// Self-attention core of DyGFormer's transformer block as ONE launch per direction: for every (sequence, head)
//     P = softmax(Q K^T / sqrt(hd));  O = dropout(P) V                                                  (forward)
//     dV = Pd^T dO;  dP = dropout'(dO V^T);  dS = P (dP - rowsum(dP P));  dQ = dS K / sqrt(hd);  dK = dS^T Q / sqrt(hd)   (backward)
// on the packed (B, S, 3 d) in-projection, the way nn.MultiheadAttention computes it (no masks).
//
// replaces: the attention core of models/DyGFormer.py:442-461 (TransformerEncoder.forward: self.multi_head_attention(...)) and its
//           autograd -- before: two (forward) / four (backward) batched products of 64 x 64 x 100 per (sequence, head), a softmax and a
//           dropout pass, each a launch over 1 200 tiny problems (85 / 130 us per block and direction for 1-2 GFLOP).
//
// One workgroup (4 waves) per (sequence, head); S <= 64 tokens, head_dim <= 104.  Q, K, V (and dO) sit in LDS as fp32 for the whole
// launch; every product is v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: the arithmetic of the reference's fp32 bmm, exact fp32 like the
// batched products this replaces); the probabilities pass through LDS between the products.  The MFMA's two k slots take the two HALVES
// of a contraction (k and k + K/2), so that a lane reads its operand row as consecutive 16-byte chunks.
// Dropout: the mask of tg_dropout on the (B, heads, S, S) probability tensor (hash of seed and flat index), regenerated in the backward.
#include <math.h>

#include "tg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SMAX = 64;            // tokens per sequence at most
constexpr int HDP = 104;            // head_dim padded (two halves of 52)
constexpr int LDQ = 104;            // row stride of the Q / K / V / dO images (no padding: three of them are 79.9 KB, two workgroups of the
                                    // forward share a CU; the 16-byte fragment reads of 16 rows then collide two-way, 26 reads per wave)
constexpr int LDP = 65;             // row stride of the probability images: odd, they are read along rows AND along columns

__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t i, float p, float scale) {      // = tg_dropout's mask (tg_seq.hip)
    (void)scale;
    return tg::res_keep_scale(seed, i, p);
}

// rows [0, S) x [0, hd) of a (S, ld) global block -> image (SMAX x LDQ), zeros elsewhere (columns up to HDP, rows up to SMAX).  Two
// phases: EVERY 16-byte load of the image is issued before the first LDS store (a rolled load-then-store loop waited out the full memory
// latency once per iteration: 20 round trips per workgroup, 2/3 of the first version's time).
constexpr int IMG_CH = (SMAX * (HDP / 4) + 255) / 256;       // float4 per thread and image
struct ImageRegs { float4 v[IMG_CH]; };
__device__ __forceinline__ void fetch_image(ImageRegs& g, const float* __restrict__ src, int64_t ld, int S, int hd) {
#pragma unroll
    for (int j = 0; j < IMG_CH; ++j) {
        const int f = threadIdx.x + 256 * j;
        const int r = f / (HDP / 4), c = (f % (HDP / 4)) * 4;
        g.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < S && c < hd) g.v[j] = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + c);
    }
}
__device__ __forceinline__ void store_image(float* __restrict__ img, const ImageRegs& g) {
#pragma unroll
    for (int j = 0; j < IMG_CH; ++j) {
        const int f = threadIdx.x + 256 * j;
        const int r = f / (HDP / 4), c = (f % (HDP / 4)) * 4;
        if (r < SMAX) *reinterpret_cast<float4*>(img + r * LDQ + c) = g.v[j];
    }
}

// C (32 x 32 tile at rows r0, columns c0) += sum_k A[r0 + i][k] B[c0 + j][k], k over [0, 2 KH): both operands k-contiguous images of
// row stride lda / ldb.  Lane (i = l & 31, h = l >> 5) walks k = h KH .. h KH + KH in 16-byte chunks.
template <int KH>
__device__ __forceinline__ void mma_nt(f32x16& acc, const float* __restrict__ A, int lda, int r0, const float* __restrict__ B, int ldb, int c0) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const float* ap = A + (r0 + i) * lda + h * KH;
    const float* bp = B + (c0 + i) * ldb + h * KH;
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(ap + 4 * q), b = *reinterpret_cast<const float4*>(bp + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
}
// C (32 x 32 at r0, c0) += sum_k A(r0 + i, k) B[k][c0 + j], k over [0, 64): A either k-contiguous (A[i][k], TA = false) or given
// transposed (A[k][i], TA = true); B is (k, column) with row stride ldb, read as scalars (consecutive lanes, consecutive banks).
template <bool TA>
__device__ __forceinline__ void mma_nn(f32x16& acc, const float* __restrict__ A, int lda, int r0, const float* __restrict__ B, int ldb, int c0,
                                       int cmax) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int col = c0 + i < cmax ? c0 + i : cmax;            // (columns past the image read its last one: never stored)
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
        const int k = 32 * h + kk;
        const float a = TA ? A[k * lda + r0 + i] : A[(r0 + i) * lda + k];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B[k * ldb + col], acc, 0, 0, 0);
    }
}
// wave maximum through DPP (as tg::wave_sum: no LDS crossbar; tg::wave_max's shuffles are ds_bpermute round trips, six per row)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_max(float v) {
    const int moved = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
    return fmaxf(v, __builtin_bit_cast(float, moved));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = dpp_max<0xB1>(v);
    v = dpp_max<0x4E>(v);
    v = dpp_max<0x141>(v);
    v = dpp_max<0x140>(v);
    v = dpp_max<0x142, 0xA>(v);
    v = dpp_max<0x143, 0xC>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// accumulator element r of a lane: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 (inside the 32 x 32 tile)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ void __launch_bounds__(256, 2) seq_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ prob,
                                                           int S, int d, int heads, float alpha, float p, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Qs = lds, *Ks = Qs + SMAX * LDQ, *Vs = Ks + SMAX * LDQ, *Ps = Qs;      // the probabilities take Q's place once the scores exist
    const int hd = d / heads, b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qkv + (int64_t)b * S * 3 * d + h * hd;
    {
        ImageRegs gq, gk, gv;
        fetch_image(gq, base, 3 * d, S, hd);
        fetch_image(gk, base + d, 3 * d, S, hd);
        fetch_image(gv, base + 2 * d, 3 * d, S, hd);
        store_image(Qs, gq);
        store_image(Ks, gk);
        store_image(Vs, gv);
    }
    __syncthreads();
    {   // scores: wave w owns the 32 x 32 tile (w >> 1, w & 1)
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
        mma_nt<HDP / 2>(acc, Qs, LDQ, r0, Ks, LDQ, c0);
        __syncthreads();                                   // every wave is done reading Q: its image becomes the score matrix
#pragma unroll
        for (int r = 0; r < 16; ++r) Ps[(r0 + acc_row(r, lane)) * LDP + c0 + (lane & 31)] = acc[r] * alpha;
    }
    __syncthreads();
    // softmax + dropout: one wave per row, lane = key
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int r = wave; r < SMAX; r += 4) {
        float v = (r < S && lane < S) ? Ps[r * LDP + lane] : -INFINITY;
        const float m = wave_max_dpp(v);
        const float e = (r < S && lane < S) ? expf(v - m) : 0.f;
        const float s = tg::wave_sum(e);
        const float pr = r < S ? e / s : 0.f;
        float pd = 0.f;
        if (r < S && lane < S) {
            const int64_t idx = (((int64_t)b * heads + h) * S + r) * S + lane;
            prob[idx] = pr;
            pd = pr * keep_scale(seed, idx, p, scale);
        }
        Ps[r * LDP + lane] = pd;
    }
    __syncthreads();
    // O = Pd V: wave w owns columns [32 w, 32 w + 32) of both row tiles
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        mma_nn<false>(acc, Ps, LDP, 32 * tm, Vs, LDQ, 32 * wave, LDQ - 1);
        const int col = 32 * wave + (lane & 31);
        if (col < hd) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * tm + acc_row(r, lane);
                if (row < S) out[((int64_t)b * S + row) * d + h * hd + col] = acc[r];
            }
        }
    }
}

__global__ void __launch_bounds__(256) seq_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ prob,
                                                           const float* __restrict__ dout, float* __restrict__ dqkv, int S, int d, int heads,
                                                           float alpha, float p, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Qs = lds, *Ks = Qs + SMAX * LDQ, *Vs = Ks + SMAX * LDQ, *Os = Vs + SMAX * LDQ, *Ps = Os + SMAX * LDQ, *Ds = Ps + SMAX * LDP;
    const int hd = d / heads, b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qkv + (int64_t)b * S * 3 * d + h * hd;
    const int64_t pb = ((int64_t)b * heads + h) * S * S;
    {
        ImageRegs gq, gk;
        float pr[SMAX * SMAX / 256];
        fetch_image(gq, base, 3 * d, S, hd);
        fetch_image(gk, base + d, 3 * d, S, hd);
#pragma unroll
        for (int j = 0; j < SMAX * SMAX / 256; ++j) {
            const int f = threadIdx.x + 256 * j, r = f >> 6, c = f & 63;
            pr[j] = (r < S && c < S) ? prob[pb + (int64_t)r * S + c] : 0.f;
        }
        store_image(Qs, gq);
        store_image(Ks, gk);
        fetch_image(gq, base + 2 * d, 3 * d, S, hd);
        fetch_image(gk, dout + (int64_t)b * S * d + h * hd, d, S, hd);
#pragma unroll
        for (int j = 0; j < SMAX * SMAX / 256; ++j) {
            const int f = threadIdx.x + 256 * j;
            Ps[(f >> 6) * LDP + (f & 63)] = pr[j];
        }
        store_image(Vs, gq);
        store_image(Os, gk);
    }
    __syncthreads();
    {   // dPd = dO V^T (tile per wave), times the dropout mask -> Ds
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
        mma_nt<HDP / 2>(acc, Os, LDQ, r0, Vs, LDQ, c0);
        const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + acc_row(r, lane), col = c0 + (lane & 31);
            const float ks = (row < S && col < S) ? keep_scale(seed, pb + (int64_t)row * S + col, p, scale) : 0.f;
            Ds[row * LDP + col] = acc[r] * ks;
        }
    }
    __syncthreads();
    {   // softmax backward per row; Ps becomes Pd (dropped probabilities), Ds becomes dS * alpha
        const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
        for (int r = wave; r < SMAX; r += 4) {
            const float pr = Ps[r * LDP + lane], g = Ds[r * LDP + lane];
            const float t = tg::wave_sum(pr * g);
            Ds[r * LDP + lane] = pr * (g - t) * alpha;
            const float ks = (r < S && lane < S) ? keep_scale(seed, pb + (int64_t)r * S + lane, p, scale) : 0.f;
            Ps[r * LDP + lane] = pr * ks;
        }
    }
    __syncthreads();
    // dV = Pd^T dO, dQ = dS K, dK = dS^T Q: wave w owns columns [32 w, 32 w + 32) of both row tiles of each
    const int col = 32 * wave + (lane & 31);
#pragma unroll
    for (int which = 0; which < 3; ++which) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (which == 0) mma_nn<true>(acc, Ps, LDP, 32 * tm, Os, LDQ, 32 * wave, LDQ - 1);          // dV[j][c] = sum_i Pd[i][j] dO[i][c]
            else if (which == 1) mma_nn<false>(acc, Ds, LDP, 32 * tm, Ks, LDQ, 32 * wave, LDQ - 1);    // dQ[i][c] = sum_j dS[i][j] K[j][c]
            else mma_nn<true>(acc, Ds, LDP, 32 * tm, Qs, LDQ, 32 * wave, LDQ - 1);                     // dK[j][c] = sum_i dS[i][j] Q[i][c]
            if (col < hd) {
                const int off = which == 0 ? 2 * d : (which == 1 ? 0 : d);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * tm + acc_row(r, lane);
                    if (row < S) dqkv[((int64_t)b * S + row) * 3 * d + off + h * hd + col] = acc[r];
                }
            }
        }
    }
}

constexpr size_t kFwdLds = sizeof(float) * (3 * SMAX * LDQ);
constexpr size_t kBwdLds = sizeof(float) * (4 * SMAX * LDQ + 2 * SMAX * LDP);

bool shape_ok(int64_t B, int S, int d, int heads) {
    return B >= 1 && heads >= 1 && d % heads == 0 && S >= 1 && S <= SMAX && (d / heads) % 4 == 0 && d / heads <= HDP - 4 && d % 4 == 0 &&
           B * heads < ((int64_t)1 << 31);
}
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// TG_ESHAPE (nothing launched) for sequences longer than 64 tokens or heads wider than 100: the caller keeps the batched products.
extern "C" int tg_seq_attn_fwd(const float* d_qkv, int64_t B, int S, int d, int heads, float dropout_p, uint64_t seed, float* d_out,
                               float* d_prob, void* stream) {
    TG_REQUIRE(d_qkv && d_out && d_prob && dropout_p >= 0.f && dropout_p < 1.f, "tg_seq_attn_fwd: arguments");
    if (!shape_ok(B, S, d, heads) || !al16(d_qkv)) { tg::set_error("tg_seq_attn_fwd: shape not covered (S <= 64, head_dim <= 100, multiples of 4)"); return TG_ESHAPE; }
    static bool attr = false;
    if (!attr) {
        TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds));
        attr = true;
    }
    tg::ProfScope prof("gemm", 4.0 * B * heads * S * S * (d / heads), (hipStream_t)stream);
    seq_attn_fwd_kernel<<<(unsigned)(B * heads), 256, kFwdLds, (hipStream_t)stream>>>(d_qkv, d_out, d_prob, S, d, heads,
                                                                                      (float)pow((double)(d / heads), -0.5), dropout_p, seed);
    return tg::launch_status("seq_attn_fwd_kernel");
}

extern "C" int tg_seq_attn_bwd(const float* d_qkv, const float* d_prob, const float* d_dout, int64_t B, int S, int d, int heads, float dropout_p,
                               uint64_t seed, float* d_dqkv, void* stream) {
    TG_REQUIRE(d_qkv && d_prob && d_dout && d_dqkv && dropout_p >= 0.f && dropout_p < 1.f, "tg_seq_attn_bwd: arguments");
    if (!shape_ok(B, S, d, heads) || !al16(d_qkv) || !al16(d_dout)) { tg::set_error("tg_seq_attn_bwd: shape not covered"); return TG_ESHAPE; }
    static bool attr = false;
    if (!attr) {
        TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds));
        attr = true;
    }
    tg::ProfScope prof("gemm", 8.0 * B * heads * S * S * (d / heads), (hipStream_t)stream);
    seq_attn_bwd_kernel<<<(unsigned)(B * heads), 256, kBwdLds, (hipStream_t)stream>>>(d_qkv, d_prob, d_dout, d_dqkv, S, d, heads,
                                                                                      (float)pow((double)(d / heads), -0.5), dropout_p, seed);
    return tg::launch_status("seq_attn_bwd_kernel");
}
