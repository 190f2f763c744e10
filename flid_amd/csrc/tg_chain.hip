// Row-block CHAINS of a temporal-attention layer: the products that follow one another on the same rows run in ONE launch, a
// 64-row block per workgroup, the intermediate rows handed from product to product through LDS (bf16 hi / lo images) instead of HBM.
//
// replaces, per layer (models/modules.py:229-238 + :58-69 as called from models/TGAT.py:132-142, and their autograd):
//   forward  (tg_chain_fwd):  ctx_h = agg_h Wv_h^T  ->  res = ctx Wr^T + br  ->  y = LayerNorm(dropout(res) + [own | cos b])
//                             ->  f1 = relu([y | raw] W1^T + b1)  ->  out = f1 W2^T + b2
//   backward (tg_chain_bwd):  df1 = (dout W2) * (f1 > 0)  ->  dy = df1 W1[:, :dq]  ->  LayerNorm backward (dsum, dres, column sums)
//                             ->  dctx = dres Wr  ->  dagg_h = dctx_h Wv_h
// Every intermediate the backward / the weight gradients need is still written to HBM once (ctx, res, y, f1; df1, dres, dctx,
// dagg) -- what disappears is reading them back, and above all the launch + first-load + drain latency of five dependent launches:
// at 13.6 k rows a product is 213 workgroups that each wait ~2 us for their first rows, compute for 1-3 us and drain
// (tools/rows_prof.sh: 11-17 us per launch whatever is ablated), so a layer's ten products + two LayerNorm passes cost ~200 us for
// ~25 us of MFMA work.
//
// Machine.  4 waves; wave w owns a block of 16-column tiles of the current product's output, all 64 rows (four 16-row blocks):
//   * weights come PACKED (tg_pack_weights: bf16 hi / lo in MFMA fragment order) straight from L2 into registers, two register sets
//     of two 32-deep steps each, the set for steps s+2, s+3 in flight while s, s+1 multiply;
//   * the MFMA is issued with the WEIGHT fragment as operand A: the accumulator then holds, per lane, 4 consecutive output columns
//     of one row -- a float4 for the HBM store and an 8-byte hi / lo pair for the LDS image of the next product;
//   * the first product of a chain streams its rows from HBM through a two-buffer ring (as gemm_rows_kernel); later products read
//     the whole K from the panel the previous epilogue wrote.  LDS: 15 chunks x (64 rows x 32 k x hi|lo) panel + ring + reductions
//     = 160 128 B -- one workgroup per CU.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "tg_common.h"
#include "tg_split.h"

namespace {

using namespace tgs;

constexpr int NTH = 256, ROWS = 64, RB = 4;
constexpr int CHS = ROWS * 64 + 64;          // one plane of one 32-k chunk (+64: the chunk stores of one row spread over banks)
constexpr int CHUNK = 2 * CHS;               // hi plane | lo plane
constexpr int NCH = 15;                      // panel chunks: [y (9) | raw (6)] of the merge layer is the widest operand
constexpr int PANEL = NCH * CHUNK;
constexpr int RING_BUF = 2 * CHUNK;          // one ring buffer = one group of two steps
constexpr int RED_OFF = PANEL + 2 * RING_BUF;
constexpr int LDS_BYTES = RED_OFF + 2 * 4 * ROWS * 4;
static_assert(LDS_BYTES <= 163840, "one workgroup must fit the CU's LDS");

__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t idx, float p) {      // as ln_res_fwd/bwd_kernel (tg_layer.hip)
    if (p <= 0.f) return 1.f;
    const float u = (float)(tg::mix32(seed ^ ((uint64_t)idx * 0x9E3779B97F4A7C15ULL)) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return u >= p ? 1.f / (1.f - p) : 0.f;
}

// state of one wave inside a chain
template <int NTW>
struct Wave {
    char* lds;
    int lane, wave;
    int64_t row0, R;
    f32x4 acc[RB][NTW];
    bf16x8 b0h[2][NTW], b0l[2][NTW], b1h[2][NTW], b1l[2][NTW];
    const uint4* bptr[NTW];
    int S, t0, tcnt;          // current product: steps, first tile of this wave, tiles it owns

    __device__ __forceinline__ void begin(const void* packed, int nt, int steps) {
        S = steps;
        const int cpw = (nt + 3) >> 2;
        t0 = wave * cpw;
        tcnt = nt - t0 < cpw ? nt - t0 : cpw;
        if (tcnt < 0) tcnt = 0;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            int t = t0 + j;
            if (t > nt - 1) t = nt - 1;
            bptr[j] = reinterpret_cast<const uint4*>(packed) + (int64_t)t * S * 128 + lane;
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void loadB(bf16x8 (&bh)[2][NTW], bf16x8 (&bl)[2][NTW], int g) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int s = 2 * g + sl;
            const int sc = s < S ? s : S - 1;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                bh[sl][j] = __builtin_bit_cast(bf16x8, bptr[j][sc * 128]);
                bl[sl][j] = __builtin_bit_cast(bf16x8, bptr[j][sc * 128 + 64]);
            }
        }
    }
    // one 32-deep step: A fragments of the four row blocks from the chunk at `chunk`, against step sl of a B set
    __device__ __forceinline__ void step(const char* chunk, const bf16x8 (&bh)[NTW], const bf16x8 (&bl)[NTW]) {
        const char* p = chunk + frag_off(lane);
        bf16x8 ah[RB], al[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            ah[rb] = *reinterpret_cast<const bf16x8*>(p + rb * 1024);
            al[rb] = *reinterpret_cast<const bf16x8*>(p + CHS + rb * 1024);
        }
        // weight fragment as operand A: acc[rb][j][r] = C[row 16 rb + (lane & 15)][column 16 (t0 + j) + 4 (lane >> 4) + r]
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[rb], acc[rb][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[rb], acc[rb][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[rb], acc[rb][j], 0, 0, 0);
    }

    // ---- product whose rows come from the LDS panel: chunks [chunk0, chunk0 + S).  No barrier inside.
    __device__ __forceinline__ void run_panel(int chunk0) {
        const int ngroups = (S + 1) >> 1;
        const char* base = lds + chunk0 * CHUNK;
        loadB(b0h, b0l, 0);
        for (int g = 0; g < ngroups; g += 2) {
            loadB(b1h, b1l, g + 1);
            step(base + (2 * g) * CHUNK, b0h[0], b0l[0]);
            if (2 * g + 1 < S) step(base + (2 * g + 1) * CHUNK, b0h[1], b0l[1]);
            loadB(b0h, b0l, g + 2);
            if (2 * g + 2 < S) step(base + (2 * g + 2) * CHUNK, b1h[0], b1l[0]);
            if (2 * g + 3 < S) step(base + (2 * g + 3) * CHUNK, b1h[1], b1l[1]);
        }
    }

    // ---- product whose rows stream from HBM (fp32, split here) through the two ring buffers behind the panel.  Ends with a barrier.
    __device__ __forceinline__ void run_stream(const float* __restrict__ A, int64_t lda, int K) {
        const int tid = wave * 64 + lane;
        const int c4 = tid & 15, r16 = tid >> 4;           // 16 float4 per row and group, 16 rows per pass
        const float* aptr[4];
        bool rok[4];
        int aoff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r16 + 16 * i;
            int64_t rg = row0 + r;
            rok[i] = rg < R;
            if (rg > R - 1) rg = R - 1;
            aptr[i] = A + rg * lda;
            aoff[i] = (c4 >> 3) * CHUNK + chunk_off(r, (c4 & 7) * 4);
        }
        auto loadA = [&](float4 (&ra)[4], int g) {
            const int k = 64 * g + 4 * c4;
            const int ko = k < K ? k : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(aptr[i] + ko);
        };
        auto writeA = [&](int buf, const float4 (&ra)[4], int g) {
            char* base = lds + PANEL + buf * RING_BUF;
            const bool kok = 64 * g + 4 * c4 < K;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = kok && rok[i];
                const float4 v = make_float4(ok ? ra[i].x : 0.f, ok ? ra[i].y : 0.f, ok ? ra[i].z : 0.f, ok ? ra[i].w : 0.f);
                uint2 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<uint2*>(base + aoff[i]) = hi;
                *reinterpret_cast<uint2*>(base + CHS + aoff[i]) = lo;
            }
        };
        const int ngroups = (S + 1) >> 1;
        float4 ra0[4], ra1[4];
        loadA(ra0, 0);
        loadB(b0h, b0l, 0);
        loadA(ra1, 1);
        writeA(0, ra0, 0);
        loadA(ra0, 2);
        __syncthreads();
        const char* r0 = lds + PANEL, *r1 = lds + PANEL + RING_BUF;
        for (int g = 0; g < ngroups; g += 2) {
            writeA(1, ra1, g + 1);
            loadA(ra1, g + 3);
            loadB(b1h, b1l, g + 1);
            step(r0, b0h[0], b0l[0]);
            step(r0 + CHUNK, b0h[1], b0l[1]);               // (a step past the end of K multiplies zero rows)
            __syncthreads();
            writeA(0, ra0, g + 2);
            loadA(ra0, g + 4);
            loadB(b0h, b0l, g + 2);
            if (g + 1 < ngroups) { step(r1, b1h[0], b1l[0]); step(r1 + CHUNK, b1h[1], b1l[1]); }
            __syncthreads();
        }
    }
    // coordinates of acc[rb][j]: row (inside the block) and first of its 4 columns
    __device__ __forceinline__ int out_row(int rb) const { return rb * 16 + (lane & 15); }
    __device__ __forceinline__ int out_col(int j) const { return (t0 + j) * 16 + 4 * (lane >> 4); }
    // 4 columns [col, col + 4) of `row` into the panel image whose column 0 sits at chunk `chunk0`
    __device__ __forceinline__ void panel_store(int chunk0, int row, int col, const float4& v) {
        uint2 hi, lo;
        split4(v, hi, lo);
        char* p = lds + (chunk0 + (col >> 5)) * CHUNK + chunk_off(row, col & 31);
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + CHS) = lo;
    }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(const f32x4& a) { return make_float4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (they hold the other column groups of the same row)
__device__ __forceinline__ float quad_rows_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// sum over the 16 lanes that share l >> 4 (they hold the same columns of the block's 16 rows)
__device__ __forceinline__ float rows16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

struct ChainFwdArgs {
    int64_t R;
    int H, dn, T, de;                 // dq = dn + T, hd = dq / H, dk = dn + de + T
    int hp;                           // per-head block of ctx inside the panel (hd rounded up to 16)
    const float* agg;                 // (R, H dk)
    const void *pWv, *pWr, *pW1, *pW2;   // packed: Wv heads back to back (tg_packed_floats(hd, dk) apart), Wr with K = H hp, W1 with K = [y | raw] padded
    const float *br, *b1, *b2, *ln_g, *ln_b, *cosb;
    const float* own; int64_t own_ld;
    float p_res; uint64_t seed;
    float* ctx;                       // (R, dq)
    float* res;                       // (R, dq)
    float* y; int64_t y_ld;           // (R, dq) inside the [y | raw] buffer
    const float* raw; int64_t raw_ld; // (R, dn)
    float *mean, *rstd, *f1, *out;
};

// tiles a wave can own: dq <= 320 -> 5
__global__ void __launch_bounds__(NTH, 1) chain_fwd_kernel(ChainFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    Wave<5> w;
    w.lds = lds;
    w.lane = threadIdx.x & 63;
    w.wave = threadIdx.x >> 6;
    w.row0 = (int64_t)blockIdx.x * ROWS;
    w.R = a.R;
    const int lane = w.lane, tid = threadIdx.x;
    const int dq = a.dn + a.T, hd = dq / a.H, dk = a.dn + a.de + a.T;
    const int ychunks = (dq + 31) >> 5, rchunks = (a.dn + 31) >> 5;
    float* red = reinterpret_cast<float*>(lds + RED_OFF);

    // ---- the merge layer's raw rows: in flight now, into the panel (behind y) after the first product
    constexpr int RAWN = 3;                                   // float4 per thread: 64 rows x 192 columns / 256 threads / 4
    float4 rawv[RAWN];
    const int rcols4 = a.dn >> 2;
#pragma unroll
    for (int i = 0; i < RAWN; ++i) {
        const int f = tid + NTH * i, r = f / 48, c = f % 48;
        int64_t rg = w.row0 + r;
        if (rg > a.R - 1) rg = a.R - 1;
        rawv[i] = ld4(a.raw + rg * a.raw_ld + (c < rcols4 ? 4 * c : 0));
    }

    // ---- ctx_h = agg_h Wv_h^T, head by head (all four waves on one head: they share its rows of agg)
    const int ht = (hd + 15) >> 4;                            // tiles per head
    const int64_t wv_stride = tg::packed_floats(1, 1) * 0 + (int64_t)((hd + 15) / 16) * ((dk + 31) / 32) * 512;   // floats per packed head
    for (int h = 0; h < a.H; ++h) {
        w.begin(reinterpret_cast<const float*>(a.pWv) + h * wv_stride, ht, (dk + 31) >> 5);
        w.run_stream(a.agg + (int64_t)h * dk, (int64_t)a.H * dk, dk);
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < RAWN; ++i) {
                const int f = tid + NTH * i, r = f / 48, c = f % 48;
                const bool ok = c < rcols4 && w.row0 + r < a.R;
                const float4 v = ok ? rawv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (4 * c < 32 * rchunks) w.panel_store(ychunks, r, 4 * c, v);
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);                     // inside the head's block
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = w.out_row(rb);
                const float4 v = f4(w.acc[rb][j]);
                w.panel_store(0, r, h * a.hp + col, v);       // columns >= hd of the block are zero (zero rows of the packed weight)
                if (col < hd && w.row0 + r < a.R) st4(a.ctx + (w.row0 + r) * dq + h * hd + col, v);
            }
        }
    }
    __syncthreads();

    // ---- res = ctx Wr^T + br ;  y = LayerNorm(dropout(res) + [own | cos b]) * g + b
    {
        const int kc = (a.H * a.hp + 31) >> 5;
        w.begin(a.pWr, (dq + 15) >> 4, kc);
        w.run_panel(0);
        f32x4 x[RB][5];
        float s1[RB];
        bool rok[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) { s1[rb] = 0.f; rok[rb] = w.row0 + w.out_row(rb) < a.R; }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);
            const float4 b4 = ld4(a.br + col);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                int64_t rg = w.row0 + w.out_row(rb);
                if (rg > a.R - 1) rg = a.R - 1;
                float4 v = add4(f4(w.acc[rb][j]), b4);
                if (rok[rb]) st4(a.res + rg * dq + col, v);
                const float4 o = col < a.dn ? ld4(a.own + rg * a.own_ld + col) : ld4(a.cosb + (col - a.dn));
                const int64_t e = rg * dq + col;
                v.x = v.x * keep_scale(a.seed, e, a.p_res) + o.x;
                v.y = v.y * keep_scale(a.seed, e + 1, a.p_res) + o.y;
                v.z = v.z * keep_scale(a.seed, e + 2, a.p_res) + o.z;
                v.w = v.w * keep_scale(a.seed, e + 3, a.p_res) + o.w;
                x[rb][j] = f32x4{v.x, v.y, v.z, v.w};
                s1[rb] += (v.x + v.y) + (v.z + v.w);
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            s1[rb] = quad_rows_sum(s1[rb]);
            if (lane < 16) red[w.wave * ROWS + rb * 16 + lane] = s1[rb];
        }
        __syncthreads();                                       // (also: every wave is done reading the ctx panel)
        float mu[RB], s2[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            mu[rb] = (red[r] + red[ROWS + r] + red[2 * ROWS + r] + red[3 * ROWS + r]) / dq;
            s2[rb] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j >= w.tcnt) break;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = x[rb][j][e] - mu[rb]; s2[rb] = fmaf(d, d, s2[rb]); }
        }
        float* red2 = red + 4 * ROWS;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            s2[rb] = quad_rows_sum(s2[rb]);
            if (lane < 16) red2[w.wave * ROWS + rb * 16 + lane] = s2[rb];
        }
        __syncthreads();
        float rs[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            rs[rb] = rsqrtf((red2[r] + red2[ROWS + r] + red2[2 * ROWS + r] + red2[3 * ROWS + r]) / dq + 1e-5f);
            if (w.wave == 0 && lane < 16 && rok[rb]) { a.mean[w.row0 + r] = mu[rb]; a.rstd[w.row0 + r] = rs[rb]; }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);
            const float4 g4 = ld4(a.ln_g + col), be4 = ld4(a.ln_b + col);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = w.out_row(rb);
                float4 v;
                v.x = (x[rb][j][0] - mu[rb]) * rs[rb] * g4.x + be4.x;
                v.y = (x[rb][j][1] - mu[rb]) * rs[rb] * g4.y + be4.y;
                v.z = (x[rb][j][2] - mu[rb]) * rs[rb] * g4.z + be4.z;
                v.w = (x[rb][j][3] - mu[rb]) * rs[rb] * g4.w + be4.w;
                if (!rok[rb]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                w.panel_store(0, r, col, v);
                if (rok[rb]) st4(a.y + (w.row0 + r) * a.y_ld + col, v);
            }
        }
        // the tail of y's last chunk (columns dq .. 32 ychunks) multiplies zero columns of the packed W1 but must be finite
        if (32 * ychunks > dq) {
            for (int f = tid; f < ROWS * ((32 * ychunks - dq) >> 2); f += NTH) {
                const int per = (32 * ychunks - dq) >> 2, r = f / per, c = dq + 4 * (f % per);
                w.panel_store(0, r, c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
    }
    __syncthreads();

    // ---- f1 = relu([y | raw] W1^T + b1)
    w.begin(a.pW1, (a.dn + 15) >> 4, ychunks + rchunks);
    w.run_panel(0);
    __syncthreads();                                           // every wave is done with [y | raw]: f1 takes its place
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        if (j >= w.tcnt) break;
        const int col = w.out_col(j);
        if (col >= a.dn) {                                     // padding columns of the last tile: zeros for the next product
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) w.panel_store(0, w.out_row(rb), col, make_float4(0.f, 0.f, 0.f, 0.f));
            continue;
        }
        const float4 b4 = ld4(a.b1 + col);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            float4 v = add4(f4(w.acc[rb][j]), b4);
            v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            w.panel_store(0, r, col, v);
            if (w.row0 + r < a.R) st4(a.f1 + (w.row0 + r) * a.dn + col, v);
        }
    }
    __syncthreads();

    // ---- out = f1 W2^T + b2
    w.begin(a.pW2, (a.dn + 15) >> 4, rchunks);
    w.run_panel(0);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        if (j >= w.tcnt) break;
        const int col = w.out_col(j);
        if (col >= a.dn) continue;
        const float4 b4 = ld4(a.b2 + col);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            if (w.row0 + r < a.R) st4(a.out + (w.row0 + r) * a.dn + col, add4(f4(w.acc[rb][j]), b4));
        }
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace tg {

// geometry the chain kernels cover (everything else takes the launch-per-product path of tg_layer.hip)
bool chain_shape_ok(int H, int dn, int T, int de) {
    if (H < 1 || H > 2 || dn % 4 || T % 4 || de % 4) return false;
    const int dq = dn + T, dk = dn + de + T;
    if (dq % H || (dq / H) % 4) return false;
    const int hd = dq / H, hp = (hd + 15) / 16 * 16;
    if (dq > 320 || dn > 192 || H * hp > 32 * 10) return false;               // tiles per wave <= 5; raw = 6 chunks; ctx panel
    if ((dq + 31) / 32 + (dn + 31) / 32 > NCH || dk > 32 * 64) return false;
    return true;
}
int chain_hp(int H, int dn, int T) { const int hd = (dn + T) / H; return (hd + 15) / 16 * 16; }

int chain_fwd(const tg_layer_desc* L, const void* pWv, const void* pWr, const void* pW1, const void* pW2, hipStream_t s) {
    const tg_attn_desc& at = L->attn;
    ChainFwdArgs a;
    a.R = at.m; a.H = at.heads; a.dn = at.dn; a.T = at.dt_dim; a.de = at.de;
    a.hp = chain_hp(at.heads, at.dn, at.dt_dim);
    a.agg = L->agg;
    a.pWv = pWv; a.pWr = pWr; a.pW1 = pW1; a.pW2 = pW2;
    const tg_layer_params& P = L->params;
    a.br = P.br; a.b1 = P.b1; a.b2 = P.b2; a.ln_g = P.ln_g; a.ln_b = P.ln_b; a.cosb = L->cosb;
    a.own = L->own; a.own_ld = L->own_ld;
    a.p_res = L->res_dropout_p; a.seed = L->res_seed;
    a.ctx = L->ctx; a.res = L->res; a.y = L->y; a.y_ld = L->y_ld ? L->y_ld : at.dn + at.dt_dim;
    a.raw = L->raw; a.raw_ld = L->raw_ld;
    a.mean = L->mean; a.rstd = L->rstd; a.f1 = L->f1; a.out = L->out;
    static bool attr_set = false;
    if (!attr_set) {
        TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int dq = at.dn + at.dt_dim, dk = at.dn + at.de + at.dt_dim;
    const double macs = (double)dk * dq + (double)dq * dq + (double)(dq + at.dn) * at.dn + (double)at.dn * at.dn;
    ProfScope prof("gemm", 2.0 * at.m * macs, s);
    chain_fwd_kernel<<<(unsigned)((at.m + ROWS - 1) / ROWS), NTH, LDS_BYTES, s>>>(a);
    return launch_status("chain_fwd_kernel");
}

}  // namespace tg
