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
#include <stdlib.h>

#include <algorithm>

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
// Both 32-row tiles of a 64-row product at once: they share B's values (read once) and give the matrix pipe two independent
// accumulation chains -- one chain of dependent 64-cycle MFMAs with its reads and address arithmetic in between left the pipe 24 % busy
// (PMC, round 5: 244 MFMAs and 3 100 VALU instructions per wave and problem in 65 k cycles).
template <bool TA>
__device__ __forceinline__ void mma_nn2(f32x16& acc0, f32x16& acc1, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                        int c0, int cmax) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int col = c0 + i < cmax ? c0 + i : cmax;
    const float* bp = B + (32 * h) * ldb + col;
    const float* ap = TA ? A + (32 * h) * lda + i : A + i * lda + 32 * h;
    const int astep = TA ? lda : 1, atile = TA ? 32 : 32 * lda;
#pragma unroll 16
    for (int kk = 0; kk < 32; ++kk) {
        const float b = bp[kk * ldb];
        const float a0 = ap[kk * astep], a1 = ap[kk * astep + atile];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
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
    {
        f32x16 acc[2] = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
        mma_nn2<false>(acc[0], acc[1], Ps, LDP, Vs, LDQ, 32 * wave, LDQ - 1);
        const int col = 32 * wave + (lane & 31);
        if (col < hd) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * tm + acc_row(r, lane);
                    if (row < S) out[((int64_t)b * S + row) * d + h * hd + col] = acc[tm][r];
                }
        }
    }
}

// LDS-only barrier: every hand-over between the phases below goes through LDS, so a phase boundary waits for the wave's LDS traffic
// alone -- __syncthreads() also drains vmcnt, i.e. it would wait for the NEXT problem's images, which are on their way into registers
// while this one computes.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Persistent: one workgroup per CU (140 KB of LDS) walks problems blockIdx.x, + gridDim.x, ...; the four images and the probabilities
// of the next problem are fetched into registers as soon as this problem's have been stored to LDS.  (One problem per workgroup: 131.8 us
// for 1 200 problems of 64 tokens -- load, five products, store, strictly one after the other on every CU.)
__global__ void __launch_bounds__(256) seq_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ prob,
                                                           const float* __restrict__ dout, float* __restrict__ dqkv, int S, int d, int heads,
                                                           float alpha, float p, uint64_t seed, int nprob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Qs = lds, *Ks = Qs + SMAX * LDQ, *Vs = Ks + SMAX * LDQ, *Os = Vs + SMAX * LDQ, *Ps = Os + SMAX * LDQ, *Ds = Ps + SMAX * LDP;
    const int hd = d / heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    ImageRegs gq, gk, gv, go;
    float pr[SMAX * SMAX / 256];
    auto fetch = [&](int pid) {
        const int b = pid / heads, h = pid % heads;
        const float* base = qkv + (int64_t)b * S * 3 * d + h * hd;
        const int64_t pb = ((int64_t)b * heads + h) * S * S;
        fetch_image(gq, base, 3 * d, S, hd);
        fetch_image(gk, base + d, 3 * d, S, hd);
        fetch_image(gv, base + 2 * d, 3 * d, S, hd);
        fetch_image(go, dout + (int64_t)b * S * d + h * hd, d, S, hd);
#pragma unroll
        for (int j = 0; j < SMAX * SMAX / 256; ++j) {
            const int f = threadIdx.x + 256 * j, r = f >> 6, c = f & 63;
            pr[j] = (r < S && c < S) ? prob[pb + (int64_t)r * S + c] : 0.f;
        }
    };
    int pid = blockIdx.x;
    if (pid < nprob) fetch(pid);
    for (; pid < nprob; pid += gridDim.x) {
        const int b = pid / heads, h = pid % heads;
        const int64_t pb = ((int64_t)b * heads + h) * S * S;
        store_image(Qs, gq);
        store_image(Ks, gk);
        store_image(Vs, gv);
        store_image(Os, go);
#pragma unroll
        for (int j = 0; j < SMAX * SMAX / 256; ++j) {
            const int f = threadIdx.x + 256 * j;
            Ps[(f >> 6) * LDP + (f & 63)] = pr[j];
        }
        lds_barrier();
        if (pid + (int)gridDim.x < nprob) fetch(pid + (int)gridDim.x);      // (uniform) in flight under everything below
        {   // dPd = dO V^T (tile per wave), times the dropout mask -> Ds
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const int r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
            mma_nt<HDP / 2>(acc, Os, LDQ, r0, Vs, LDQ, c0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + acc_row(r, lane), col = c0 + (lane & 31);
                const float ks = (row < S && col < S) ? keep_scale(seed, pb + (int64_t)row * S + col, p, scale) : 0.f;
                Ds[row * LDP + col] = acc[r] * ks;
            }
        }
        lds_barrier();
        for (int r = wave; r < SMAX; r += 4) {      // softmax backward per row; Ps becomes Pd (dropped probabilities), Ds becomes dS * alpha
            const float q = Ps[r * LDP + lane], g = Ds[r * LDP + lane];
            const float t = tg::wave_sum(q * g);
            Ds[r * LDP + lane] = q * (g - t) * alpha;
            const float ks = (r < S && lane < S) ? keep_scale(seed, pb + (int64_t)r * S + lane, p, scale) : 0.f;
            Ps[r * LDP + lane] = q * ks;
        }
        lds_barrier();
        // dV = Pd^T dO, dQ = dS K, dK = dS^T Q: wave w owns columns [32 w, 32 w + 32) of both row tiles of each
        const int col = 32 * wave + (lane & 31);
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            f32x16 acc[2] = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
            if (which == 0) mma_nn2<true>(acc[0], acc[1], Ps, LDP, Os, LDQ, 32 * wave, LDQ - 1);          // dV[j][c] = sum_i Pd[i][j] dO[i][c]
            else if (which == 1) mma_nn2<false>(acc[0], acc[1], Ds, LDP, Ks, LDQ, 32 * wave, LDQ - 1);    // dQ[i][c] = sum_j dS[i][j] K[j][c]
            else mma_nn2<true>(acc[0], acc[1], Ds, LDP, Qs, LDQ, 32 * wave, LDQ - 1);                     // dK[j][c] = sum_i dS[i][j] Q[i][c]
            if (col < hd) {
                const int off = which == 0 ? 2 * d : (which == 1 ? 0 : d);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * tm + acc_row(r, lane);
                        if (row < S) dqkv[((int64_t)b * S + row) * 3 * d + off + h * hd + col] = acc[tm][r];
                    }
            }
        }
        lds_barrier();                                     // every wave is done with this problem's images
    }
}

// ---- sequences of at most 32 tokens (config 4: 32 patches): the 64-row form above spends three of its four score tiles and half of
// every other product on padding and holds twice the LDS (one workgroup per CU backward).  Here the images are 32 rows, the 32 x 32
// score matrix is four 16 x 16 tiles (v_mfma_f32_16x16x4_f32, one per wave, full contraction), the other products one 32-row tile with
// a 32-deep contraction; 40 / 62 KB of LDS forward / backward.
constexpr int SM32 = 32;
constexpr int LDP32 = 33;
constexpr int IMG32_CH = (SM32 * (HDP / 4) + 255) / 256;
struct Image32Regs { float4 v[IMG32_CH]; };
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fetch_image32(Image32Regs& g, const float* __restrict__ src, int64_t ld, int S, int hd) {
#pragma unroll
    for (int j = 0; j < IMG32_CH; ++j) {
        const int f = threadIdx.x + 256 * j;
        const int r = f / (HDP / 4), c = (f % (HDP / 4)) * 4;
        g.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < S && c < hd) g.v[j] = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + c);
    }
}
__device__ __forceinline__ void store_image32(float* __restrict__ img, const Image32Regs& g) {
#pragma unroll
    for (int j = 0; j < IMG32_CH; ++j) {
        const int f = threadIdx.x + 256 * j;
        const int r = f / (HDP / 4), c = (f % (HDP / 4)) * 4;
        if (r < SM32) *reinterpret_cast<float4*>(img + r * LDQ + c) = g.v[j];
    }
}
// C (16 x 16 at r0, c0) = sum_k A[r0 + i][k] B[c0 + j][k], k over [0, HDP): lane (i = l & 15, q = l >> 4) supplies k = 16 c + 4 q + {0..3}
// of chunk c to four MFMAs (A and B agree on the map, so any map is a valid order of the sum); the last 8 columns come from q < 2.
// acc[r] = C[r0 + 4 (l >> 4) + r][c0 + (l & 15)]
__device__ __forceinline__ f32x4v mma16_nt(const float* __restrict__ A, int r0, const float* __restrict__ B, int c0) {
    static_assert(HDP == 104, "six 16-column chunks and one of 8");
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const float* ap = A + (r0 + i) * LDQ + 4 * q;
    const float* bp = B + (c0 + i) * LDQ + 4 * q;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (c < 6 || q < 2) { a = *reinterpret_cast<const float4*>(ap + 16 * c); b = *reinterpret_cast<const float4*>(bp + 16 * c); }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    return acc;
}
// C (32 x 32 at rows 0.., columns c0..) = sum_{k < 32} A(i, k) B[k][c0 + j]: A k-contiguous (TA false) or transposed; lane (i, h) walks
// k = 16 h .. 16 h + 15
template <bool TA>
__device__ __forceinline__ void mma_nn32(f32x16& acc, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int c0, int cmax) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int col = c0 + i < cmax ? c0 + i : cmax;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = 16 * h + kk;
        const float a = TA ? A[k * lda + i] : A[i * lda + k];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B[k * ldb + col], acc, 0, 0, 0);
    }
}

__global__ void __launch_bounds__(256, 4) seq_attn_fwd32_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ prob,
                                                             int S, int d, int heads, float alpha, float p, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Qs = lds, *Ks = Qs + SM32 * LDQ, *Vs = Ks + SM32 * LDQ, *Ps = Qs;
    const int hd = d / heads, b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qkv + (int64_t)b * S * 3 * d + h * hd;
    {
        Image32Regs gq, gk, gv;
        fetch_image32(gq, base, 3 * d, S, hd);
        fetch_image32(gk, base + d, 3 * d, S, hd);
        fetch_image32(gv, base + 2 * d, 3 * d, S, hd);
        store_image32(Qs, gq);
        store_image32(Ks, gk);
        store_image32(Vs, gv);
    }
    __syncthreads();
    {
        const int r0 = 16 * (wave >> 1), c0 = 16 * (wave & 1);
        const f32x4v acc = mma16_nt(Qs, r0, Ks, c0);
        __syncthreads();                                   // every wave is done reading Q: its image becomes the score matrix
#pragma unroll
        for (int r = 0; r < 4; ++r) Ps[(r0 + 4 * (lane >> 4) + r) * LDP32 + c0 + (lane & 15)] = acc[r] * alpha;
    }
    __syncthreads();
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int r = wave; r < SM32; r += 4) {
        const bool in = r < S && lane < S;
        float v = in ? Ps[r * LDP32 + lane] : -INFINITY;
        const float m = wave_max_dpp(v);
        const float e = in ? expf(v - m) : 0.f;
        const float s = tg::wave_sum(e);
        const float pr = r < S ? e / s : 0.f;
        float pd = 0.f;
        if (in) {
            const int64_t idx = (((int64_t)b * heads + h) * S + r) * S + lane;
            prob[idx] = pr;
            pd = pr * keep_scale(seed, idx, p, scale);
        }
        if (lane < SM32) Ps[r * LDP32 + lane] = pd;
    }
    __syncthreads();
    {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        mma_nn32<false>(acc, Ps, LDP32, Vs, LDQ, 32 * wave, LDQ - 1);
        const int col = 32 * wave + (lane & 31);
        if (col < hd) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, lane);
                if (row < S) out[((int64_t)b * S + row) * d + h * hd + col] = acc[r];
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2) seq_attn_bwd32_kernel(const float* __restrict__ qkv, const float* __restrict__ prob,
                                                             const float* __restrict__ dout, float* __restrict__ dqkv, int S, int d, int heads,
                                                             float alpha, float p, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Qs = lds, *Ks = Qs + SM32 * LDQ, *Vs = Ks + SM32 * LDQ, *Os = Vs + SM32 * LDQ, *Ps = Os + SM32 * LDQ, *Ds = Ps + SM32 * LDP32;
    const int hd = d / heads, b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qkv + (int64_t)b * S * 3 * d + h * hd;
    const int64_t pb = ((int64_t)b * heads + h) * S * S;
    {
        Image32Regs gq, gk, gv, go;
        float pr[SM32 * SM32 / 256];
        fetch_image32(gq, base, 3 * d, S, hd);
        fetch_image32(gk, base + d, 3 * d, S, hd);
        fetch_image32(gv, base + 2 * d, 3 * d, S, hd);
        fetch_image32(go, dout + (int64_t)b * S * d + h * hd, d, S, hd);
#pragma unroll
        for (int j = 0; j < SM32 * SM32 / 256; ++j) {
            const int f = threadIdx.x + 256 * j, r = f >> 5, c = f & 31;
            pr[j] = (r < S && c < S) ? prob[pb + (int64_t)r * S + c] : 0.f;
        }
        store_image32(Qs, gq);
        store_image32(Ks, gk);
        store_image32(Vs, gv);
        store_image32(Os, go);
#pragma unroll
        for (int j = 0; j < SM32 * SM32 / 256; ++j) {
            const int f = threadIdx.x + 256 * j;
            Ps[(f >> 5) * LDP32 + (f & 31)] = pr[j];
        }
    }
    __syncthreads();
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    {   // dPd = dO V^T (16 x 16 tile per wave), times the dropout mask -> Ds
        const int r0 = 16 * (wave >> 1), c0 = 16 * (wave & 1);
        const f32x4v acc = mma16_nt(Os, r0, Vs, c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 4 * (lane >> 4) + r, col = c0 + (lane & 15);
            const float ks = (row < S && col < S) ? keep_scale(seed, pb + (int64_t)row * S + col, p, scale) : 0.f;
            Ds[row * LDP32 + col] = acc[r] * ks;
        }
    }
    __syncthreads();
    for (int r = wave; r < SM32; r += 4) {      // softmax backward per row; Ps becomes Pd, Ds becomes dS * alpha
        const bool li = lane < SM32;
        const float pr = li ? Ps[r * LDP32 + lane] : 0.f, g = li ? Ds[r * LDP32 + lane] : 0.f;
        const float t = tg::wave_sum(pr * g);
        const float ks = (r < S && lane < S) ? keep_scale(seed, pb + (int64_t)r * S + lane, p, scale) : 0.f;
        if (li) {
            Ds[r * LDP32 + lane] = pr * (g - t) * alpha;
            Ps[r * LDP32 + lane] = pr * ks;
        }
    }
    __syncthreads();
    const int col = 32 * wave + (lane & 31);
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (which == 0) mma_nn32<true>(acc, Ps, LDP32, Os, LDQ, 32 * wave, LDQ - 1);          // dV[j][c] = sum_i Pd[i][j] dO[i][c]
        else if (which == 1) mma_nn32<false>(acc, Ds, LDP32, Ks, LDQ, 32 * wave, LDQ - 1);    // dQ[i][c] = sum_j dS[i][j] K[j][c]
        else mma_nn32<true>(acc, Ds, LDP32, Qs, LDQ, 32 * wave, LDQ - 1);                     // dK[j][c] = sum_i dS[i][j] Q[i][c]
        if (col < hd) {
            const int off = which == 0 ? 2 * d : (which == 1 ? 0 : d);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, lane);
                if (row < S) dqkv[((int64_t)b * S + row) * 3 * d + off + h * hd + col] = acc[r];
            }
        }
    }
}

constexpr size_t kFwdLds32 = sizeof(float) * (3 * SM32 * LDQ);
constexpr size_t kBwdLds32 = sizeof(float) * (4 * SM32 * LDQ + 2 * SM32 * LDP32);
constexpr size_t kFwdLds = sizeof(float) * (3 * SMAX * LDQ);
constexpr size_t kBwdLds = sizeof(float) * (4 * SMAX * LDQ + 2 * SMAX * LDP);

bool shape_ok(int64_t B, int S, int d, int heads) {
    return B >= 1 && heads >= 1 && d % heads == 0 && S >= 1 && S <= SMAX && (d / heads) % 4 == 0 && d / heads <= HDP - 4 && d % 4 == 0 &&
           B * heads < ((int64_t)1 << 31);
}
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int num_cus() {                       // workgroups of the persistent backward: one per CU of the current device
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) { (void)hipGetLastError(); v = 256; }
        n = v;
    }
    return n;
}
// FLID_SEQATTN_LONG=1 (with FLID_GEMM_TUNE): the 64-row form for every length (A/B runs, tests of that form at short lengths)
const bool g_short = !(getenv("FLID_GEMM_TUNE") && getenv("FLID_SEQATTN_LONG") && atoi(getenv("FLID_SEQATTN_LONG")) != 0);

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
    if (S <= SM32 && g_short) {
        static bool attr32 = false;
        if (!attr32) {
            TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_fwd32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds32));
            attr32 = true;
        }
        seq_attn_fwd32_kernel<<<(unsigned)(B * heads), 256, kFwdLds32, (hipStream_t)stream>>>(d_qkv, d_out, d_prob, S, d, heads,
                                                                                            (float)pow((double)(d / heads), -0.5), dropout_p, seed);
        return tg::launch_status("seq_attn_fwd32_kernel");
    }
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
    if (S <= SM32 && g_short) {
        static bool attr32 = false;
        if (!attr32) {
            TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_bwd32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds32));
            attr32 = true;
        }
        seq_attn_bwd32_kernel<<<(unsigned)(B * heads), 256, kBwdLds32, (hipStream_t)stream>>>(d_qkv, d_prob, d_dout, d_dqkv, S, d, heads,
                                                                                            (float)pow((double)(d / heads), -0.5), dropout_p, seed);
        return tg::launch_status("seq_attn_bwd32_kernel");
    }
    const int nprob = (int)(B * heads);
    seq_attn_bwd_kernel<<<(unsigned)std::min(nprob, num_cus()), 256, kBwdLds, (hipStream_t)stream>>>(d_qkv, d_prob, d_dout, d_dqkv, S, d, heads,
                                                                                      (float)pow((double)(d / heads), -0.5), dropout_p, seed, nprob);
    return tg::launch_status("seq_attn_bwd_kernel");
}
