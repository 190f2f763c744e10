// One temporal-attention layer (attention block + merge MLP), forward and backward, as ONE native call each:
// the launch sequence that flid_amd/engine.py otherwise issues op by op from Python, plus the fusions that only make sense
// here (dropout + split residual inside LayerNorm, ReLU mask + bias gradient, LayerNorm backward emitting every column sum the
// layer needs).
//
// replaces per layer: models/modules.py:167-245 (MultiHeadAttention.forward) + :58-69 (MergeLayer.forward) as called from
//                     models/TGAT.py:132-142 / models/MemoryModel.py:703-713, and their autograd.
#include <math.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <mutex>
#include <unordered_map>

#include "tg_common.h"
#include "tg_pack.h"
#include "tg_colsum.h"
#include "tg_tail.h"

namespace {

using tg::kWave;
constexpr int ROW_WAVES = 4;

__host__ __device__ inline int64_t row_grid(int64_t n) {
    int64_t b = (n + ROW_WAVES - 1) / ROW_WAVES;
    return b < 1 ? 1 : (b > tg::kMaxGridBlocks ? tg::kMaxGridBlocks : b);
}

__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t idx, float p) { return tg::res_keep_scale(seed, idx, p); }

// y = LayerNorm(dropout(res) + [own | cosb]) * gamma + beta        (modules.py:235-238 with residual = cat[node, time(0)])
template <int MAXC>
__global__ void __launch_bounds__(256) ln_res_fwd_kernel(const float* __restrict__ res, const float* __restrict__ own, int64_t own_ld,
        const float* __restrict__ cosb, int64_t n, int dn, int cols, float p, uint64_t seed, const float* __restrict__ gamma,
        const float* __restrict__ beta, float* __restrict__ y, int64_t ldy, float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * ROW_WAVES + wave; r < n; r += (int64_t)gridDim.x * ROW_WAVES) {
        float x[MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            float v = 0.f;
            if (c < cols) {
                v = res[r * cols + c] * keep_scale(seed, r * cols + c, p);
                v += c < dn ? own[r * own_ld + c] : cosb[c - dn];
            }
            x[i] = v;
            s += v;
        }
        const float mu = tg::wave_sum(s) / cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const float d = (lane + 64 * i) < cols ? x[i] - mu : 0.f;
            q = fmaf(d, d, q);
        }
        const float rs = rsqrtf(tg::wave_sum(q) / cols + 1e-5f);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) y[r * ldy + c] = (x[i] - mu) * rs * gamma[c] + beta[c];
        }
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}

// dsum = dLN(dy); dres = dsum * dropout mask.  Workgroup slab (4 * cols): [dgamma | dbeta | colsum(dsum) | colsum(dres)].
// d_own (optional): the residual's share of the gradient w.r.t. the layer's own rows, d_own[r, :dn] (+)= dsum[r, :dn] -- the
// product that delivers the query path's share later accumulates into it.
template <int MAXC>
__global__ void __launch_bounds__(256) ln_res_bwd_kernel(const float* __restrict__ res, const float* __restrict__ own, int64_t own_ld,
        const float* __restrict__ cosb, const float* __restrict__ dy, int64_t n, int dn, int cols, float p, uint64_t seed,
        const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dsum,
        float* __restrict__ dres, float* __restrict__ part, float* __restrict__ d_own, int64_t d_own_ld, int d_own_acc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ float red[];   // ROW_WAVES * 4 * cols
    float a0[MAXC], a1[MAXC], a2[MAXC], a3[MAXC], gm[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        a0[i] = a1[i] = a2[i] = a3[i] = 0.f;
        const int c = lane + 64 * i;
        gm[i] = c < cols ? gamma[c] : 0.f;
    }
    for (int64_t r = (int64_t)blockIdx.x * ROW_WAVES + wave; r < n; r += (int64_t)gridDim.x * ROW_WAVES) {
        const float mu = mean[r], rs = rstd[r];
        float xh[MAXC], g[MAXC], ks[MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < cols;
            ks[i] = ok ? keep_scale(seed, r * cols + c, p) : 0.f;
            float v = 0.f;
            if (ok) {
                v = res[r * cols + c] * ks[i];
                v += c < dn ? own[r * own_ld + c] : cosb[c - dn];
            }
            const float d = ok ? dy[r * cols + c] : 0.f;
            xh[i] = ok ? (v - mu) * rs : 0.f;
            g[i] = d * gm[i];
            s1 += g[i];
            s2 = fmaf(g[i], xh[i], s2);
            a0[i] = fmaf(d, xh[i], a0[i]);
            a1[i] += d;
        }
        const float m1 = tg::wave_sum(s1) / cols, m2 = tg::wave_sum(s2) / cols;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) {
                const float dx = rs * (g[i] - m1 - xh[i] * m2);
                const float dr = dx * ks[i];
                dsum[r * cols + c] = dx;
                if (dres != dsum) dres[r * cols + c] = dr;
                if (d_own && c < dn) {
                    float* o = d_own + r * d_own_ld + c;
                    *o = d_own_acc ? *o + dx : dx;
                }
                a2[i] += dx;
                a3[i] += dr;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < cols) {
            float* w = red + wave * 4 * cols;
            w[c] = a0[i]; w[cols + c] = a1[i]; w[2 * cols + c] = a2[i]; w[3 * cols + c] = a3[i];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 4 * cols; j += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ROW_WAVES; ++w) s += red[w * 4 * cols + j];
        part[(int64_t)blockIdx.x * 4 * cols + j] = s;
    }
}

// dy *= (y > 0) in place; slab (cols) of column sums of the masked dy per workgroup.  Thread t owns columns t, t+256, ...
template <int MAXC>
__global__ void __launch_bounds__(256) relu_bwd_colsum_kernel(float* __restrict__ dy, const float* __restrict__ y, int64_t n, int cols,
                                                              int64_t rows_per_block, float* __restrict__ part) {
    float acc[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) acc[i] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (int64_t r = r0; r < r1; ++r) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + 256 * i;
            if (c < cols) {
                const int64_t o = r * cols + c;
                const float v = y[o] > 0.f ? dy[o] : 0.f;
                dy[o] = v;
                acc[i] += v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < cols) part[(int64_t)blockIdx.x * cols + c] = acc[i];
    }
}


// Column sums of a tall matrix added (float atomics) into up to 6 destination vectors: column c belongs to the first segment
// with c < end[i] and lands at p[i][c - begin_i]; a null p[i] drops the segment.  One launch finishes a bias / LayerNorm /
// time-encoder gradient that used to take two reduction passes plus copies.
using tg::SegDst;
__global__ void __launch_bounds__(256) colsum_seg_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int cols, SegDst d) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < cols)
        for (int64_t r = (int64_t)blockIdx.y * 4 + wave; r < n; r += (int64_t)gridDim.y * 4) s += x[r * ld + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || c >= cols) return;
    const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    int beg = 0;
    for (int i = 0; i < d.n; ++i) {
        if (c < d.end[i]) {
            if (d.p[i]) atomicAdd(d.p[i] + (c - beg), t);
            return;
        }
        beg = d.end[i];
    }
}
int colsum_seg(const float* x, int64_t ld, int64_t n, int cols, const SegDst& d, hipStream_t s) {
    if (n == 0) return TG_OK;
    const int col_groups = (cols + 63) / 64;
    const int64_t max_slices = std::max<int64_t>(64, std::min<int64_t>(512, 2048 / col_groups));
    const int slices = (int)std::min<int64_t>(max_slices, std::max<int64_t>(1, n / 32));
    colsum_seg_kernel<<<dim3((cols + 63) / 64, slices), 256, 0, s>>>(x, ld, n, cols, d);
    return tg::launch_status("colsum_seg_kernel");
}
SegDst seg1(float* p, int cols) { SegDst d{}; d.p[0] = p; d.end[0] = cols; d.n = 1; return d; }

// two slabs in one launch (a layer's LayerNorm slabs and its attention backward's time-encoder slabs)
using tg::ColJob;
using tg::colsum_seg2_body;
__global__ void __launch_bounds__(256) colsum_seg2_kernel(ColJob a, ColJob b, int groups_a) {
    __shared__ float red[4][64];
    colsum_seg2_body(a, b, groups_a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, red);
}
int colsum_seg2(const ColJob& a, const ColJob& b, hipStream_t s) {
    if (a.n == 0) return colsum_seg(b.x, b.ld, b.n, b.cols, b.d, s);
    if (b.n == 0) return colsum_seg(a.x, a.ld, a.n, a.cols, a.d, s);
    const int ga = (a.cols + 63) / 64, gb = (b.cols + 63) / 64;
    const int64_t max_slices = std::max<int64_t>(64, std::min<int64_t>(512, 2048 / (ga + gb)));
    const int slices = (int)std::min<int64_t>(max_slices, std::max<int64_t>(1, std::max(a.n, b.n) / 32));
    colsum_seg2_kernel<<<dim3(ga + gb, slices), 256, 0, s>>>(a, b, ga);
    return tg::launch_status("colsum_seg2_kernel");
}

using tg::WQT_ROWS;
using tg::wq_time_body;
// The layer's PRELUDE in one launch: up to 10 small matrix transposes (weights, once per step): dst[c * ldd + r] = src[r * lds + c];
// grid row n: cos(b) (the time encoding of a zero interval, models/TGAT.py:84-85; written out when cosb_out is set, else read from
// mv_x) and the matrix-vector product mv_y[i] = sum_t mv_W[i * mv_ld + t] cos(b_t) (the constant half of the query, qb = Wq[:, dn:] cos b),
// one wave per output; grid rows n + 1 ..: the gather of the layer's raw rows g_out[r] = g_table[g_idx[r]] (utils of TGAT.py:77-79).
struct TrJob { const float* src; float* dst; int rows, cols; int64_t lds, ldd; };
struct TrJobs {
    TrJob j[10];
    int n;
    const float *mv_W, *mv_x;
    float* mv_y;
    int mv_rows, mv_cols;
    int64_t mv_ld;
    const float* te_b;         // != nullptr: cos(b) is computed here (and stored to cosb_out); else mv_x holds it
    float* cosb_out;
    const float* g_table;      // != nullptr: gather job
    const int32_t* g_idx;
    float* g_out;
    int64_t g_tld, g_old, g_n;
    int g_cols, g_rows_y;      // grid rows that walk the gather
};
// one wave: qb_i = sum_t Wq[i, dn + t] cos(b_t) (every lane returns the sum)
__device__ __forceinline__ float query_bias_row(const TrJobs& jobs, int i, int lane) {
    float acc = 0.f;
    for (int t = lane; t < jobs.mv_cols; t += 64) {
        const float cb = jobs.te_b ? tg::cos_phase(jobs.te_b[t]) : jobs.mv_x[t];
        acc = fmaf(jobs.mv_W[(int64_t)i * jobs.mv_ld + t], cb, acc);
    }
    return tg::wave_sum(acc);
}
// (bx, by) of a (gx, n + 1 + g_rows_y) grid of 256-thread workgroups
__device__ __forceinline__ void transpose_many_body(const TrJobs& jobs, int bx, int by, int gx, float (*tile)[33]) {
    if (by > jobs.n) {                                    // ---- row gather: one wave per row, 16-byte chunks
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int64_t first = ((int64_t)(by - jobs.n - 1) * gx + bx) * 4 + wave;
        const int64_t stride = (int64_t)jobs.g_rows_y * gx * 4;
        const bool vec = (jobs.g_cols & 3) == 0 && (jobs.g_tld & 3) == 0 && (jobs.g_old & 3) == 0 &&
                         ((reinterpret_cast<uintptr_t>(jobs.g_table) | reinterpret_cast<uintptr_t>(jobs.g_out)) & 15) == 0;
        for (int64_t r = first; r < jobs.g_n; r += stride) {
            const float* src = jobs.g_table + (int64_t)jobs.g_idx[r] * jobs.g_tld;
            float* dst = jobs.g_out + r * jobs.g_old;
            if (vec) {
                for (int c = lane * 4; c < jobs.g_cols; c += 256) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
            } else {
                for (int c = lane; c < jobs.g_cols; c += 64) dst[c] = src[c];
            }
        }
        return;
    }
    if (by == jobs.n) {                                   // ---- cos(b) and the query bias
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (jobs.cosb_out && bx == 0)
            for (int t = threadIdx.x; t < jobs.mv_cols; t += blockDim.x) jobs.cosb_out[t] = tg::cos_phase(jobs.te_b[t]);
        for (int i = bx * 4 + wave; i < jobs.mv_rows; i += gx * 4) {
            const float acc = query_bias_row(jobs, i, lane);
            if (lane == 0) jobs.mv_y[i] = acc;
        }
        return;
    }
    const TrJob jb = jobs.j[by];
    const int tiles_c = (jb.cols + 31) / 32, tiles_r = (jb.rows + 31) / 32;
    for (int t = bx; t < tiles_c * tiles_r; t += gx) {
        const int tr = t / tiles_c, tc = t % tiles_c;
        const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
        __syncthreads();
        for (int i = ly; i < 32; i += 8) {
            const int r = tr * 32 + i, c = tc * 32 + lx;
            tile[i][lx] = (r < jb.rows && c < jb.cols) ? jb.src[(int64_t)r * jb.lds + c] : 0.f;
        }
        __syncthreads();
        for (int i = ly; i < 32; i += 8) {
            const int c = tc * 32 + i, r = tr * 32 + lx;
            if (r < jb.rows && c < jb.cols) jb.dst[(int64_t)c * jb.ldd + r] = tile[lx][i];
        }
    }
}

// Merged projections of one attention layer, weights only (once per step and layer, ~0.1 GFLOP):
//   P_h = Wk_h^T Wq_h[:, :dn]   (dk x dn)    u_h = own P_h^T + ub_h      replaces  q = [own | cos b] Wq^T ; u_h = Wk_h^T q_h
//   V_h = Wr[:, h] Wv_h         (dq x dk)    res = agg V^T + br          replaces  ctx_h = Wv_h agg_h ; res = Wr ctx + br
// Two products per direction leave the main chain; each output element is one hd-deep dot product, written in both layouts.
struct MergeJob {            // C[m, n] (+)= sum_k A(m, k) B(k, n); optionally also written transposed to CT (ldct)
    const float *A, *B;
    float *C, *CT;
    int M, N, K, sAm, sAk, sBk, sBn, ldc, ldct, tiles_n, tile0, accumulate;
};
struct MergeJobs { MergeJob j[4]; int n, total_tiles; };

// Small weight-space products, up to 4 jobs per launch (the merged projections forward, their gradient chains backward):
// 32 x 32 output tile per workgroup, K in LDS chunks of 32; thread (ty, tx) owns rows ty, ty+8, ty+16, ty+24 of column tx.
// Operand tiles are loaded along whichever index is contiguous in memory.  Blocks past the tiles compute ub (forward only).
// bx: workgroup index inside the role; qb: the query bias in LDS (computed by this workgroup: the launch that writes it to memory is this one)
constexpr int MERGE_ALL = 5;            // chunks of 32 a merge tile fetches at once (contractions up to 160 deep)
__device__ __forceinline__ void merge_weights_body(const MergeJobs& jobs, int bx, const float* __restrict__ Wk, const float* qb, int H, int hd,
                                                   int dk, float* __restrict__ ub, float (*As)[33], float (*Bs)[33], float (*Cs)[33]) {
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    if (bx >= jobs.total_tiles) {                        // ub[h dk + j] = sum_k Wk[h hd + k, j] qb[h hd + k]
        // 32 outputs per workgroup, the contraction split over 8 thread groups whose loads are all in flight at once (one thread per
        // output walking all hd rows was a chain of ~17 exposed load round trips: 20 us for 0.2 MFLOP)
        const int64_t hj = (int64_t)(bx - jobs.total_tiles) * 32 + tx;
        const int per = (hd + 7) / 8, k0 = ty * per, k1 = min(hd, k0 + per);
        float acc = 0.f;
        if (hj < (int64_t)H * dk) {
            const int h = (int)(hj / dk), jj = (int)(hj % dk);
            if (per <= 20) {                                 // (uniform) head_dim <= 160: the group's whole share in flight at once
                float wv[20];
#pragma unroll
                for (int q = 0; q < 20; ++q) wv[q] = k0 + q < k1 ? Wk[((int64_t)h * hd + k0 + q) * dk + jj] : 0.f;
#pragma unroll
                for (int q = 0; q < 20; ++q) if (k0 + q < k1) acc = fmaf(wv[q], qb[h * hd + k0 + q], acc);
            } else {
#pragma unroll 4
                for (int k = k0; k < k1; ++k) acc = fmaf(Wk[((int64_t)h * hd + k) * dk + jj], qb[h * hd + k], acc);
            }
        }
        As[ty][tx] = acc;
        __syncthreads();
        if (ty == 0 && hj < (int64_t)H * dk) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += As[g][tx];
            ub[hj] = s;
        }
        return;
    }
    int ji = 0;
    while (ji + 1 < jobs.n && bx >= jobs.j[ji + 1].tile0) ++ji;
    const MergeJob J = jobs.j[ji];
    const int tile = bx - J.tile0, m0 = (tile / J.tiles_n) * 32, n0 = (tile % J.tiles_n) * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // operand tiles of chunk k0 + 32 are loaded into registers while chunk k0 is multiplied (K = head_dim = 136 is five chunks: without
    // the look-ahead the kernel was five exposed global round trips, 20 us for 0.1 GFLOP)
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t + 256 * q;                                   // 1024 elements of each operand tile
            const int am = J.sAm == 1 ? (e & 31) : (e >> 5), ak = J.sAm == 1 ? (e >> 5) : (e & 31);
            ra[q] = (m0 + am < J.M && k0 + ak < J.K) ? J.A[(int64_t)(m0 + am) * J.sAm + (int64_t)(k0 + ak) * J.sAk] : 0.f;
            const int bn = J.sBn == 1 ? (e & 31) : (e >> 5), bk = J.sBn == 1 ? (e >> 5) : (e & 31);
            rb[q] = (n0 + bn < J.N && k0 + bk < J.K) ? J.B[(int64_t)(k0 + bk) * J.sBk + (int64_t)(n0 + bn) * J.sBn] : 0.f;
        }
    };
    if (J.K <= 32 * MERGE_ALL) {
        // a short contraction (head_dim = 136: five chunks) is fetched WHOLE before the first chunk is multiplied: one global round trip
        // instead of one per chunk (the merge tiles were the prelude launch's long pole: 20.4 us with them, 11.0 without -- ablation,
        // round 5); same chunk order, same sums
        float fa[MERGE_ALL][4], fb[MERGE_ALL][4];
#pragma unroll
        for (int c = 0; c < MERGE_ALL; ++c) {
            const int k0 = 32 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = t + 256 * q;
                const int am = J.sAm == 1 ? (e & 31) : (e >> 5), ak = J.sAm == 1 ? (e >> 5) : (e & 31);
                fa[c][q] = (m0 + am < J.M && k0 + ak < J.K) ? J.A[(int64_t)(m0 + am) * J.sAm + (int64_t)(k0 + ak) * J.sAk] : 0.f;
                const int bn = J.sBn == 1 ? (e & 31) : (e >> 5), bk = J.sBn == 1 ? (e >> 5) : (e & 31);
                fb[c][q] = (n0 + bn < J.N && k0 + bk < J.K) ? J.B[(int64_t)(k0 + bk) * J.sBk + (int64_t)(n0 + bn) * J.sBn] : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < MERGE_ALL; ++c) {
            if (32 * c < J.K) {                                              // (uniform)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = t + 256 * q;
                    const int am = J.sAm == 1 ? (e & 31) : (e >> 5), ak = J.sAm == 1 ? (e >> 5) : (e & 31);
                    As[ak][am] = fa[c][q];
                    const int bn = J.sBn == 1 ? (e & 31) : (e >> 5), bk = J.sBn == 1 ? (e >> 5) : (e & 31);
                    Bs[bk][bn] = fb[c][q];
                }
                __syncthreads();
#pragma unroll 8
                for (int k = 0; k < 32; ++k) {
                    const float b = Bs[k][tx];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(As[k][ty + 8 * q], b, acc[q]);
                }
                __syncthreads();
            }
        }
    } else {
    fetch(0);
    for (int k0 = 0; k0 < J.K; k0 += 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t + 256 * q;
            const int am = J.sAm == 1 ? (e & 31) : (e >> 5), ak = J.sAm == 1 ? (e >> 5) : (e & 31);
            As[ak][am] = ra[q];
            const int bn = J.sBn == 1 ? (e & 31) : (e >> 5), bk = J.sBn == 1 ? (e >> 5) : (e & 31);
            Bs[bk][bn] = rb[q];
        }
        __syncthreads();
        if (k0 + 32 < J.K) fetch(k0 + 32);
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float b = Bs[k][tx];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(As[k][ty + 8 * q], b, acc[q]);
        }
        __syncthreads();
    }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + ty + 8 * q, n = n0 + tx;
        Cs[ty + 8 * q][tx] = acc[q];
        if (m < J.M && n < J.N) {
            float* c = J.C + (int64_t)m * J.ldc + n;
            *c = J.accumulate ? *c + acc[q] : acc[q];
        }
    }
    if (!J.CT) return;                                                   // (uniform per workgroup)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                        // transposed copy: consecutive threads walk m
        const int n = n0 + ty + 8 * q, m = m0 + tx;
        if (m < J.M && n < J.N) J.CT[(int64_t)n * J.ldct + m] = Cs[tx][ty + 8 * q];
    }
}

// job table helper: appends a job and advances the running tile count
inline void add_job(MergeJobs& mj, const float* A, const float* B, float* C, float* CT, int M, int N, int K, int sAm, int sAk, int sBk,
                    int sBn, int ldc, int ldct, int accumulate) {
    MergeJob& J = mj.j[mj.n++];
    J = MergeJob{A, B, C, CT, M, N, K, sAm, sAk, sBk, sBn, ldc, ldct, (N + 31) / 32, mj.total_tiles, accumulate};
    mj.total_tiles += ((M + 31) / 32) * J.tiles_n;
}

// Everything of a layer that depends on the WEIGHTS alone (and the raw-row gather), in one launch of three kinds of workgroups:
//   [0, nb_merge)         the merged query projection P_h = Wk_h^T Wq_h[:, :dn] and its constant part ub (merge_weights_body)
//   [.., + nb_pack)       the packed split-bf16 operands of the chain kernels (tg_pack.h)
//   the rest (64 tr_gy)   transposed copies, cos(b), the query bias, the raw-row gather (transpose_many_body)
// (three launches before: 6.5 + 8.4 + 12.5 us for the 13.6 k-row layer, 5 + 5.4 us for the root layer).  The ub workgroups need the
// query bias this same launch produces, so each recomputes it into LDS with the arithmetic of the workgroups that store it.
#ifndef FLID_PRELUDE_EXP
#define FLID_PRELUDE_EXP 0   // timing experiments only (results wrong): 1 no packing, 2 no transposes / gathers, 3 no merge tiles and no ub workgroups,
                             // 4 no zero fill, 5 no ub workgroups, 6 no merge tiles
#endif
struct PreludeArgs {
    tgs::PackJobs pk;
    TrJobs tr;
    MergeJobs mj;
    int nb_pack, tr_gy, nb_merge;
    const float* Wk;
    int H, hd, dk;
    float* ub;
};
__device__ __forceinline__ void prelude_body(const PreludeArgs& a, int bid, float (*sm)[32][33], float* qbs) {
    // (the merge tiles are the longest dependent chains of the launch: they take the lowest workgroup numbers and start first)
    int b = bid - a.nb_merge;
    if (b >= 0) {
        if (b < a.nb_pack) { if (FLID_PRELUDE_EXP != 1) tgs::pack_body(a.pk, b, a.nb_pack); return; }
        b -= a.nb_pack;
        if (FLID_PRELUDE_EXP != 2) transpose_many_body(a.tr, b & 63, b >> 6, 64, sm[0]);
        return;
    }
    if (FLID_PRELUDE_EXP == 3) return;
    b = bid;
    if (FLID_PRELUDE_EXP == 5 && b >= a.mj.total_tiles) return;
    if (FLID_PRELUDE_EXP == 6 && b < a.mj.total_tiles) return;
    if (b >= a.mj.total_tiles) {
        // the query-bias rows of the head(s) this workgroup's 32 outputs belong to; 8 rows of a wave in flight at a time (one row at a
        // time was 68 exposed load round trips: the launch took 35 us longer than the three it replaced)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int64_t j0 = (int64_t)(b - a.mj.total_tiles) * 32;
        const int h0 = (int)(j0 / a.dk), h1 = (int)min<int64_t>((j0 + 31) / a.dk, a.H - 1);
        const int i_lo = h0 * a.hd, i_hi = (h1 + 1) * a.hd;
        float* cbs = &sm[0][0][0];                       // cos(b_t), T <= 1024 (host-checked: dq <= 1024)
        for (int t = threadIdx.x; t < a.tr.mv_cols; t += 256) cbs[t] = a.tr.te_b ? tg::cos_phase(a.tr.te_b[t]) : a.tr.mv_x[t];
        __syncthreads();
        // 16 rows of a wave and two column steps in flight at a time: 32 independent loads per round trip (same order of every row's sum
        // as the workgroups that store the bias)
        // 17 rows of a wave and two column steps in flight at a time (a head's 136 rows in two round trips; same order of every row's
        // sum as the workgroups that store the bias).  With 8 rows and one column step these 28 workgroups were ten dependent round
        // trips long, then five more for ub: the launch's long pole (20.4 us with them, 14.6 without: ablation, round 5).
        constexpr int QR = 17;
        for (int i = i_lo + QR * wave; i < i_hi; i += 4 * QR) {
            float acc[QR];
#pragma unroll
            for (int r = 0; r < QR; ++r) acc[r] = 0.f;
            for (int t = lane; t < a.tr.mv_cols; t += 128) {
                const int t1 = t + 64;
                const bool two = t1 < a.tr.mv_cols;
                const float cb0 = cbs[t], cb1 = two ? cbs[t1] : 0.f;
                float w0[QR], w1[QR];
#pragma unroll
                for (int r = 0; r < QR; ++r) {
                    const int ir = min(i + r, i_hi - 1);
                    w0[r] = a.tr.mv_W[(int64_t)ir * a.tr.mv_ld + t];
                    w1[r] = two ? a.tr.mv_W[(int64_t)ir * a.tr.mv_ld + t1] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < QR; ++r) acc[r] = fmaf(w1[r], cb1, fmaf(w0[r], cb0, acc[r]));
            }
#pragma unroll
            for (int r = 0; r < QR; ++r) {
                const float v = tg::wave_sum(acc[r]);
                if (lane == 0 && i + r < i_hi) qbs[i + r] = v;
            }
        }
        __syncthreads();
    }
    merge_weights_body(a.mj, b, a.Wk, qbs, a.H, a.hd, a.dk, a.ub, sm[0], sm[1], sm[2]);
}
__global__ void __launch_bounds__(256) layer_prelude_kernel(PreludeArgs a) {
    __shared__ float sm[3][32][33];
    __shared__ float qbs[1024];
    prelude_body(a, (int)blockIdx.x, sm, qbs);
}
// The preludes of TWO layers of a step in one launch (the upper layer's depends on the weights and the row ids only, like the lower one's:
// as a launch of its own, 6-7 us of a 1 200-row layer's forward), plus -- workgroups behind them -- a zero fill (the step's gradient
// block: one memset launch less).  Workgroups [0, nb_a): layer a, [nb_a, nb_a + nb_b): layer b, the rest: zero z4 float4s.
__global__ void __launch_bounds__(256) layer_prelude2_kernel(PreludeArgs a, PreludeArgs b, int nb_a, int nb_b, float4* __restrict__ z, int64_t z4) {
    __shared__ float sm[3][32][33];
    __shared__ float qbs[1024];
    const int bid = (int)blockIdx.x;
    if (bid < nb_a) { prelude_body(a, bid, sm, qbs); return; }
    if (bid < nb_a + nb_b) { prelude_body(b, bid - nb_a, sm, qbs); return; }
    const int64_t nz = (int64_t)gridDim.x - nb_a - nb_b;
    if (FLID_PRELUDE_EXP == 4) return;
    for (int64_t i = ((int64_t)bid - nb_a - nb_b) * 256 + threadIdx.x; i < z4; i += nz * 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

using tg::UBR;
using tg::ub_bwd_body;

// The layer's backward TAIL in one launch: three independent finishing steps that used to be a launch each (4.5-12 us apiece for
// a few hundred KB of work).  Workgroups [0, n_head): the constant-part gradients of the query -- ub_bwd (merged projection,
// head_mode 1: one workgroup per UBR query rows) or wq_time (head_mode 2: (T / 64) x (dq / 16) workgroups); the rest: the slab sums.
struct TailArgs {
    int head_mode, n_head, wq_gx, wq_nb;       // wq_nb > 0: v is a slab of partial sums (wq_time_slab_body)
    const float *v, *qb, *Wk, *Wq, *cosb;        // v = dub (mode 1) or sum_rows dq (mode 2)
    int hd, dn, dq, dk, T;
    float *dWk, *dWq, *d_cosb;
    ColJob a, b;
    int groups_a, col_gx, col_ny;
};
__global__ void __launch_bounds__(256) layer_tail_kernel(TailArgs t) {
    __shared__ float red[4][64];
    __shared__ float red2[272];
    const int bid = (int)blockIdx.x;
    if (bid < t.n_head) {
        if (t.head_mode == 1) ub_bwd_body(bid, t.v, t.qb, t.Wk, t.Wq, t.cosb, t.hd, t.dn, t.dq, t.dk, t.T, t.dWk, t.dWq, t.d_cosb, &red[0][0]);
        else if (t.wq_nb > 0) tg::wq_time_slab_body(bid % t.wq_gx, bid / t.wq_gx, t.v, t.wq_nb, t.dq, t.cosb, t.T, t.Wq + t.dn, t.dWq + t.dn, t.dq, t.d_cosb, red2);
        else wq_time_body(bid % t.wq_gx, bid / t.wq_gx, t.v, t.dq, t.cosb, t.T, t.Wq + t.dn, t.dWq + t.dn, t.dq, t.d_cosb);
        return;
    }
    const int c = bid - t.n_head;
    colsum_seg2_body(t.a, t.b, t.groups_a, c % t.col_gx, c / t.col_gx, t.col_ny, red);
}

struct WT { float *Wk, *Wv, *W2, *W1a, *W1b, *Wr, *WqL, *P, *PT, *V, *VT, *ub; };
inline WT wt_layout(float* base, int H, int dn, int dq, int dk) {
    WT w;
    float* p = base;
    w.Wk = p; p += (int64_t)dk * dq;      // H blocks of (dk, hd)
    w.Wv = p; p += (int64_t)dk * dq;
    w.W2 = p; p += (int64_t)dn * dn;
    w.W1a = p; p += (int64_t)dq * dn;     // (dq, dn)
    w.W1b = p; p += (int64_t)dn * dn;
    w.Wr = p; p += (int64_t)dq * dq;
    w.WqL = p; p += (int64_t)dn * dq;      // (dn, dq)
    // merged projections (g_merged): P = [Wk_h^T Wq_h[:, :dn]]_h (H dk, dn), V = [Wr[:, h] Wv_h]_h (dq, H dk), their transposes,
    // and the constant part of u: ub_h = Wk_h^T (Wq_h[:, dn:] cos b)
    w.P = p; p += (int64_t)H * dk * dn;
    w.PT = p; p += (int64_t)H * dk * dn;
    w.V = p; p += (int64_t)H * dk * dq;
    w.VT = p; p += (int64_t)H * dk * dq;
    w.ub = p;
    return w;
}

// Packed weights of the chain kernels (tg_chain.hip), behind the transposed copies in the layer's wT block
struct PK { float *Wv, *Wr, *W1, *W2, *W2T, *W1aT, *W1bT, *WrT, *WvT, *Wq, *WkT, *Wk, *WqT; int64_t total; };
inline int64_t r4(int64_t n) { return (n + 3) / 4 * 4; }
inline PK pk_layout(float* base, int H, int dn, int dq, int dk) {
    PK k;
    const int hd = dq / H, hp = (hd + 15) / 16 * 16;
    const int yc = (dq + 31) / 32, rc = (dn + 31) / 32;
    float* p = base;
    k.Wv = p; p += H * tg::packed_floats(hd, dk);
    k.Wr = p; p += tg::packed_floats(dq, H * hp);
    k.W1 = p; p += tg::packed_floats(dn, 32 * (yc + rc));
    k.W2 = p; p += tg::packed_floats(dn, dn);
    const int hpb = (hd + 31) / 32 * 32;                 // backward: dctx in per-head blocks that start on a 32-k chunk
    k.W2T = p; p += tg::packed_floats(dn, dn);
    k.W1aT = p; p += tg::packed_floats(dq, dn);
    k.W1bT = p; p += tg::packed_floats(dn, dn);         // d raw = df1 W1[:, dq:] (layers whose raw rows carry a gradient: TGN)
    k.WrT = p; p += tg::packed_floats(H * hpb, dq);
    k.WvT = p; p += H * tg::packed_floats(dk, hpb);
    // query side of a short (not merged) layer (qu_fwd_kernel / dq_bwd_kernel): Wq[:, :dn]; per head Wk_h^T (K = hd padded to hpb) and
    // Wk_h; Wq[:, :dn]^T with K in per-head blocks of hp
    k.Wq = p; p += tg::packed_floats(dq, dn);
    k.WkT = p; p += H * tg::packed_floats(dk, hpb);
    k.Wk = p; p += H * tg::packed_floats(hd, dk);
    k.WqT = p; p += tg::packed_floats(dn, H * hp);
    k.total = p - base;
    return k;
}
inline int64_t wt_floats_plain(int dn, int dq, int dk) {
    const int64_t H = 2;
    return 2 * (int64_t)dk * dq + 2 * (int64_t)dn * dn + 2 * (int64_t)dq * dn + (int64_t)dq * dq + 2 * H * dk * dn + 2 * H * dk * dq + H * dk + 16;
}
bool g_chain = true;       // fused row-block chains (tg_chain.hip) where the layer's geometry allows

inline unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, tg::kMaxGridBlocks)); }

#define TG_TRY(expr) do { int _rc = (expr); if (_rc != TG_OK) return _rc; } while (0)

// Weight-gradient products run on a side stream: they depend on the main chain only through their two operands, and the
// main chain is alternately HBM-bound (fused attention backward) and latency-bound (one GEMM after another), so the matrix
// cores are free for them.  fork(): side waits for everything issued on main so far; join(): main waits for side.
struct SideStream {
    static constexpr int NS = 2;        // the fork groups alternate between two streams: each weight-gradient product is a few
                                        // hundred latency-bound workgroups, two of them side by side fill the chip better
    hipStream_t side[NS] = {nullptr, nullptr};
    hipEvent_t ev[32];
    int next = 0, cur = 0;
    bool ok = false;
    bool init() {
        if (ok) return true;
        // lowest priority: the side streams' products should fill what the main chain leaves idle, not share the CUs with it
        // evenly (three concurrent products each ran 2-3x slower, and the main-chain one is on the critical path)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        static const bool flat_prio = getenv("FLID_GEMM_TUNE") != nullptr && getenv("FLID_SIDE_PRIO_DEFAULT") != nullptr;
        for (auto& st : side)
            if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, flat_prio ? 0 : lo) != hipSuccess) return false;
        for (auto& e : ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        ok = true;
        return true;
    }
    hipEvent_t mark_deferred(int* rc) {                  // an event from the ring, to be recorded later by the issuing thread
        *rc = TG_OK;
        return ev[next++ & 31];
    }
    hipEvent_t mark(hipStream_t on, int* rc) {           // record "everything issued on `on` so far"
        hipEvent_t e = ev[next++ & 31];
        *rc = hipEventRecord(e, on) == hipSuccess ? TG_OK : TG_EHIP;
        return e;
    }
};

// The side stream's launches are ISSUED by a helper thread: a backward call is ~35 launches per layer, two thirds of them on
// the side stream, and at ~6 us of host time per launch the step had become bound by the issuing thread (host 1.4 ms vs GPU
// 1.5 ms per step, any host jitter showed in the headline).  Jobs run strictly in push order; drain() returns after the last
// one has been issued (not executed) so that the caller can join the streams.
class SideIssuer {
public:
    void push(std::function<int()> f) {
        {
            std::lock_guard<std::mutex> g(m_);
            if (!started_) { started_ = true; int dev = 0; (void)hipGetDevice(&dev); th_ = std::thread([this, dev] { run(dev); }); }
            q_.push_back(std::move(f));
        }
        cv_.notify_one();
    }
    int drain() {
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return q_.empty() && !busy_; });
        const int rc = err_;
        err_ = TG_OK;
        if (rc != TG_OK) tg::set_error(msg_);          // errors are thread-local: relay the issuing thread's text
        return rc;
    }
    ~SideIssuer() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_one();
        if (th_.joinable()) th_.join();
    }
private:
    void run(int dev) {
        (void)hipSetDevice(dev);
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return stop_ || !q_.empty(); });
            if (stop_ && q_.empty()) return;
            std::function<int()> f = std::move(q_.front());
            q_.pop_front();
            busy_ = true;
            g.unlock();
            const int rc = f();
            std::string msg = rc != TG_OK ? tg::get_error() : std::string();
            g.lock();
            busy_ = false;
            if (rc != TG_OK && err_ == TG_OK) { err_ = rc; msg_ = std::move(msg); }
            if (q_.empty()) done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<std::function<int()>> q_;
    std::thread th_;
    bool started_ = false, stop_ = false, busy_ = false;
    int err_ = TG_OK;
    std::string msg_;
};
SideIssuer g_issuer;
bool g_issue_thread = true;
// What a layer's forward decided (chain launch / short-layer query launch / merged projection), by the layer's weight buffer: the
// backward reads the saved activations accordingly (`res` holds the NORMALISED LayerNorm input behind a chain forward, `q` does not exist
// behind a merged one), so a switch flipped between the two calls (tg_set_layer_chain, tg_set_layer_merged, tg_set_gemm_mode, the row
// threshold) must be an error, not a silently different meaning.
std::mutex g_fwd_mode_mutex;
std::unordered_map<const void*, int> g_fwd_mode;
void note_forward_mode(const void* key, bool use_chain, bool use_qu, bool merged) {
    std::lock_guard<std::mutex> g(g_fwd_mode_mutex);
    if (g_fwd_mode.size() > 4096) g_fwd_mode.clear();          // (keys are long-lived arena addresses; a leak guard, not a cache policy)
    g_fwd_mode[key] = (use_chain ? 1 : 0) | (use_qu ? 2 : 0) | (merged ? 4 : 0);
}
int forward_mode(const void* key) {
    std::lock_guard<std::mutex> g(g_fwd_mode_mutex);
    const auto it = g_fwd_mode.find(key);
    return it == g_fwd_mode.end() ? -1 : it->second;
}

bool g_merged = true;      // merged projections (merge_weights_body); false = the reference's four separate products per layer
// the weight-space work of the merged form (one merge kernel forward, ~6 small launches backward) is a fixed cost per layer call:
// it pays from a few thousand rows on (TGAT layer 1: 12 k rows), not for the 1 200-row root layer or a TGN batch
int64_t kMergedMinRows = 4096;
SideStream g_side;
bool g_overlap = false;     // weight gradients on side streams under the main chain: measured 1-2 % SLOWER than issuing them in line
                            // (TGAT 1.066 vs 1.057 ms, link prediction 1.633 vs 1.598, TGN 0.904 vs 0.899) since the grouped launches;
                            // tg_set_overlap(1) turns it back on
bool g_wgrad_grouped = true;

}  // namespace

namespace {
// main waits for everything issued (also by the helper thread) on the side streams
int side_join(hipStream_t s) {
    if (!g_side.ok) return TG_OK;
    if (g_issue_thread) TG_TRY(g_issuer.drain());
    for (hipStream_t sd : g_side.side) {
        int rc = TG_OK;
        hipEvent_t e = g_side.mark(sd, &rc);
        TG_TRY(rc);
        TG_HIP_CHECK(hipStreamWaitEvent(s, e, 0));
    }
    return TG_OK;
}
}  // namespace

extern "C" int tg_side_join(void* stream) { return side_join((hipStream_t)stream); }

// [y | raw] laid out as one (R, dq + dn) buffer by the caller (y_ld = raw_ld = dq + dn, raw = y + dq)
static inline bool yr_joined(const tg_layer_desc* L, int dq) { return L->y_ld != 0 && L->raw == L->y + dq && L->raw_ld == L->y_ld; }

// mode 0: the whole forward; 1: the forward behind its prelude launch (the caller issued it: tg::layers_prelude); 2: only FILL *out with
// the prelude launch's arguments and workgroup count (nothing launched)
static int layer_fwd_impl(const tg_layer_desc* L, void* stream, int mode, PreludeArgs* out_pa, int64_t* out_blocks) {
    TG_REQUIRE(L, "tg_tgat_layer_fwd: null descriptor");
    const tg_attn_desc& a = L->attn;
    const int64_t R = a.m;
    const int H = a.heads, dn = a.dn, T = a.dt_dim, dq = dn + T, dk = dn + a.de + T, hd = dq / H;
    TG_REQUIRE(dq % H == 0, "The sum of node_feat_dim and time_feat_dim should be divided by num_heads!");
    TG_REQUIRE(dq <= 1024, "tg_tgat_layer_fwd: query dim > 1024 unsupported");
    if (R == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const tg_layer_params& P = L->params;
    // Transposed copies of the weights that the chain multiplies "from the right" (u = q Wk, and every dX = dY W of the
    // backward): with them EVERY product of the main chain has two k-contiguous operands and runs on the split-bf16 kernel.
    const WT wt = wt_layout(L->wT, H, dn, dq, dk);
    TG_REQUIRE(H <= 2, "tg_tgat_layer_fwd: the native layer path supports 1 or 2 heads");
    // everything behind the attention as ONE launch (tg_chain.hip) when the geometry and the alignment allow: its weights are packed here
    const PK pk = pk_layout(L->wT + r4(wt_floats_plain(dn, dq, dk)), H, dn, dq, dk);
    const int64_t ldy_c = L->y_ld ? L->y_ld : dq;
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool use_chain = g_chain && tg_get_gemm_mode() != 0 && tg::chain_shape_ok(H, dn, T, a.de) && a16(L->agg) && a16(L->ctx) && a16(L->res) && a16(L->y) && a16(L->raw) &&
                           a16(L->own) && a16(L->f1) && a16(L->out) && a16(L->cosb) && a16(P.br) && a16(P.b1) && a16(P.b2) && a16(P.ln_g) && a16(P.ln_b) &&
                           a16(L->wT) && L->own_ld % 4 == 0 && L->raw_ld % 4 == 0 && ldy_c % 4 == 0;
    PreludeArgs pa;                          // one launch for everything that depends on the weights alone
    pa.pk.n = 0; pa.pk.frag0[0] = 0; pa.nb_pack = 0;
    pa.mj.n = 0; pa.mj.total_tiles = 0;
    pa.Wk = P.Wk; pa.H = H; pa.hd = hd; pa.dk = dk; pa.ub = wt.ub;
    auto launch_prelude = [&](int n_tr, int gather_y, bool with_ub) -> int {
        pa.tr_gy = n_tr + 1 + gather_y;
        pa.nb_merge = pa.mj.total_tiles + (with_ub ? (int)(((int64_t)H * dk + 31) / 32) : 0);
        const int64_t blocks = (int64_t)pa.nb_merge + pa.nb_pack + 64 * pa.tr_gy;
        if (mode == 2) { *out_pa = pa; *out_blocks = blocks; return TG_OK; }
        if (mode == 1) return TG_OK;
        layer_prelude_kernel<<<(unsigned)blocks, 256, 0, s>>>(pa);
        return tg::launch_status("layer_prelude_kernel");
    };
    // the query side of a layer that does not take the merged projection (short layers) as one launch: q -> u
    const bool merged = g_merged && R >= kMergedMinRows;
    static const bool no_qu = getenv("FLID_GEMM_TUNE") && getenv("FLID_NO_QU") && atoi(getenv("FLID_NO_QU")) != 0;       // A/B timing
    const bool use_qu = !no_qu && use_chain && !merged && tg::qu_shape_ok(H, dn, T, a.de) && a16(L->q) && a16(L->u) && a16(L->qbias);
    if (mode != 1) note_forward_mode(L->wT, use_chain, use_qu, merged);
    if (use_chain) {
        const int hp = tg::chain_hp(H, dn, T), yc = (dq + 31) / 32, rc = (dn + 31) / 32;
        tg_pack_job jobs[20];
        int n = 0;
        for (int h = 0; h < H; ++h)
            jobs[n++] = tg_pack_job{P.Wv + (int64_t)h * hd * dk, dk, hd, dk, 0, pk.Wv + h * tg::packed_floats(hd, dk), 0, 0, 0, 0, 0, 0};
        jobs[n++] = tg_pack_job{P.Wr, dq, dq, H * hp, 0, pk.Wr, 0, dq, 0, 0, hd, hp};                       // K = per-head blocks of hp
        jobs[n++] = tg_pack_job{P.W1, (int64_t)dq + dn, dn, 32 * (yc + rc), 0, pk.W1, 0, dq + dn, 0, 0, dq, 32 * yc};   // K = [y | raw], each part padded to 32
        jobs[n++] = tg_pack_job{P.W2, dn, dn, dn, 0, pk.W2, 0, 0, 0, 0, 0, 0};
        // backward operands (the transposed weights its input-gradient products multiply with; weights change only at the optimizer step)
        const int hpb = tg::chain_hpb(H, dn, T);
        jobs[n++] = tg_pack_job{P.W2, dn, dn, dn, 1, pk.W2T, 0, 0, 0, 0, 0, 0};                                      // df1 = dout W2
        jobs[n++] = tg_pack_job{P.W1, (int64_t)dq + dn, dq, dn, 1, pk.W1aT, 0, 0, 0, 0, 0, 0};                      // dy = df1 W1[:, :dq]
        if (L->raw) jobs[n++] = tg_pack_job{P.W1 + dq, (int64_t)dq + dn, dn, dn, 1, pk.W1bT, 0, 0, 0, 0, 0, 0};       // d raw = df1 W1[:, dq:]
        jobs[n++] = tg_pack_job{P.Wr, dq, H * hpb, dq, 1, pk.WrT, dq, 0, hd, hpb, 0, 0};                             // dctx = dres Wr, columns in head blocks
        for (int h = 0; h < H; ++h)                                                                                  // dagg_h = dctx_h Wv_h
            jobs[n++] = tg_pack_job{P.Wv + (int64_t)h * hd * dk, dk, dk, hd, 1, pk.WvT + h * tg::packed_floats(dk, hpb), 0, 0, 0, 0, 0, 0};
        if (use_qu) {
            jobs[n++] = tg_pack_job{P.Wq, dq, dq, dn, 0, pk.Wq, 0, 0, 0, 0, 0, 0};                                       // q = own Wq[:, :dn]^T
            for (int h = 0; h < H; ++h)                                                                              // u_h = q_h Wk_h
                jobs[n++] = tg_pack_job{P.Wk + (int64_t)h * hd * dk, dk, dk, hpb, 1, pk.WkT + h * tg::packed_floats(dk, hpb), 0, hd, 0, 0, 0, 0};
            for (int h = 0; h < H; ++h)                                                                              // dq_h = du_h Wk_h^T
                jobs[n++] = tg_pack_job{P.Wk + (int64_t)h * hd * dk, dk, hd, dk, 0, pk.Wk + h * tg::packed_floats(hd, dk), 0, 0, 0, 0, 0, 0};
            jobs[n++] = tg_pack_job{P.Wq, dq, dn, H * hp, 1, pk.WqT, 0, dq, 0, 0, hd, hp};                                // d_own += dq Wq[:, :dn]
        }
        const int frags = tgs::pack_jobs_fill(pa.pk, n, jobs);
        TG_REQUIRE(frags >= 0, "tg_tgat_layer_fwd: packed-weight job table");
        pa.nb_pack = (int)std::min<int64_t>((frags + 3) / 4, 2048);
    }
    // the constant half of the query, qb = Wq[:, dn:] cos b, rides in the transposes' launch
    // ... together with cos(b) itself (when the caller asks: compute_cosb) and the gather of the layer's raw rows (gather_table)
    int gather_rows_y = 0;
    auto with_qbias = [&](TrJobs& jobs) {
        jobs.mv_W = P.Wq + dn; jobs.mv_x = L->cosb; jobs.mv_y = L->qbias; jobs.mv_rows = dq; jobs.mv_cols = T; jobs.mv_ld = dq;
        jobs.te_b = L->compute_cosb ? a.d_te_b : nullptr;
        jobs.cosb_out = L->compute_cosb ? const_cast<float*>(L->cosb) : nullptr;
        jobs.g_table = L->gather_table; jobs.g_idx = L->gather_idx; jobs.g_out = const_cast<float*>(L->raw);
        jobs.g_tld = L->gather_ld; jobs.g_old = L->raw_ld; jobs.g_n = R; jobs.g_cols = dn;
        gather_rows_y = L->gather_table ? (int)std::min<int64_t>(32, (R + 255) / 256) : 0;      // 64 workgroups x 4 rows per grid row
        jobs.g_rows_y = gather_rows_y;
    };
    if (merged) {
        // merged QUERY side (u = own P^T + ub: the q intermediate and one product per direction leave the chain); the value side keeps
        // the reference's two products (ctx_h = Wv_h agg_h, res = Wr ctx + br): its merged form V_h = Wr[:, h] Wv_h cost as much on the
        // main chain but needed a (dq x H dk) gradient product over all rows (89 us) plus a weight-space chain behind it
        TrJobs jobs;
        int n = 0;
        // (with the chain kernels the backward's transposed weights are PACKED by the launch above; only W1b -- the raw rows' input
        // gradient of a trainable base table, a launch of its own -- still wants a plain transposed copy)
        if (!use_chain) {
            jobs.j[n++] = TrJob{P.W2, wt.W2, dn, dn, dn, dn};
            jobs.j[n++] = TrJob{P.W1, wt.W1a, dn, dq, (int64_t)dq + dn, dn};
        }
        jobs.j[n++] = TrJob{P.W1 + dq, wt.W1b, dn, dn, (int64_t)dq + dn, dn};
        if (!use_chain) {
            for (int h = 0; h < H; ++h) jobs.j[n++] = TrJob{P.Wv + (int64_t)h * hd * dk, wt.Wv + (int64_t)h * dk * hd, hd, dk, dk, hd};
            jobs.j[n++] = TrJob{P.Wr, wt.Wr, dq, dq, dq, dq};
        }
        jobs.n = n;
        with_qbias(jobs);
        pa.tr = jobs;
        for (int h = 0; h < H; ++h)             // P_h (dk x dn): A(m = j, k) = Wk[h hd + k, j], B(k, n = i) = Wq[h hd + k, i]
            add_job(pa.mj, P.Wk + (int64_t)h * hd * dk, P.Wq + (int64_t)h * hd * dq, wt.P + (int64_t)h * dk * dn, wt.PT + (int64_t)h * dk,
                    dk, dn, hd, 1, dk, dq, 1, dn, H * dk, 0);
        TG_TRY(launch_prelude(n, gather_rows_y, true));
        if (mode == 2) return TG_OK;
        // u = own P^T + ub   (all heads in one product, K = dn)
        TG_TRY(tg_gemm_f32(0, 1, R, H * dk, dn, 1.f, L->own, L->own_ld, wt.P, dn, L->u, (int64_t)H * dk, wt.ub, 0, 0, stream));
        TG_TRY(tg_attn_fwd(&a, L->u, L->agg, L->prob, stream));
    } else {
        // Transposed copies of the weights that the chain multiplies "from the right" (u = q Wk, and every dX = dY W of the
        // backward): with them EVERY product of the main chain has two k-contiguous operands and runs on the split-bf16 kernel.
        TrJobs jobs;
        int n = 0;
        for (int h = 0; h < H && n < 4 && !use_qu; ++h) {
            jobs.j[n++] = TrJob{P.Wk + (int64_t)h * hd * dk, wt.Wk + (int64_t)h * dk * hd, hd, dk, dk, hd};
        }
        if (!use_chain) {
            for (int h = 0; h < H; ++h) jobs.j[n++] = TrJob{P.Wv + (int64_t)h * hd * dk, wt.Wv + (int64_t)h * dk * hd, hd, dk, dk, hd};
            jobs.j[n++] = TrJob{P.W2, wt.W2, dn, dn, dn, dn};
            jobs.j[n++] = TrJob{P.W1, wt.W1a, dn, dq, (int64_t)dq + dn, dn};
        }
        jobs.j[n++] = TrJob{P.W1 + dq, wt.W1b, dn, dn, (int64_t)dq + dn, dn};
        if (!use_chain) jobs.j[n++] = TrJob{P.Wr, wt.Wr, dq, dq, dq, dq};
        if (!use_qu) jobs.j[n++] = TrJob{P.Wq, wt.WqL, dq, dn, dq, dq};
        jobs.n = n;
        with_qbias(jobs);
        pa.tr = jobs;
        TG_TRY(launch_prelude(n, gather_rows_y, false));
        if (mode == 2) return TG_OK;
        if (use_qu) {
            TG_TRY(tg::qu_fwd(L, pk.Wq, pk.WkT, s));               // q = [own | cos b] Wq^T and u_h = Wk_h^T q_h in one launch
        } else {
            // q = [own | cos b] Wq^T : the constant half is a bias row
            TG_TRY(tg_gemm_f32(0, 1, R, dq, dn, 1.f, L->own, L->own_ld, P.Wq, dq, L->q, dq, L->qbias, 0, 0, stream));
            // u_h = Wk_h^T q_h
            TG_TRY(tg_gemm_f32_batched(0, 1, R, dk, hd, 1.f, L->q, dq, hd, wt.Wk, hd, (int64_t)dk * hd, L->u, (int64_t)H * dk, dk, H, nullptr, 0, 0, stream));
        }
        TG_TRY(tg_attn_fwd(&a, L->u, L->agg, L->prob, stream));
    }
    if (use_chain) return tg::chain_fwd(L, pk.Wv, pk.Wr, pk.W1, pk.W2, pk.total * 4, s);
    // ctx_h = Wv_h agg_h ; res = ctx Wr^T + br
    TG_TRY(tg_gemm_f32_batched(0, 1, R, hd, dk, 1.f, L->agg, (int64_t)H * dk, dk, P.Wv, dk, (int64_t)hd * dk, L->ctx, dq, hd, H, nullptr, 0, 0, stream));
    TG_TRY(tg_gemm_f32(0, 1, R, dq, dq, 1.f, L->ctx, dq, P.Wr, dq, L->res, dq, P.br, 0, 0, stream));
    const unsigned g = (unsigned)row_grid(R);
    const int64_t ldy = L->y_ld ? L->y_ld : dq;
    if (dq <= 64) ln_res_fwd_kernel<1><<<g, 256, 0, s>>>(L->res, L->own, L->own_ld, L->cosb, R, dn, dq, L->res_dropout_p, L->res_seed, P.ln_g, P.ln_b, L->y, ldy, L->mean, L->rstd);
    else if (dq <= 320) ln_res_fwd_kernel<5><<<g, 256, 0, s>>>(L->res, L->own, L->own_ld, L->cosb, R, dn, dq, L->res_dropout_p, L->res_seed, P.ln_g, P.ln_b, L->y, ldy, L->mean, L->rstd);
    else ln_res_fwd_kernel<16><<<g, 256, 0, s>>>(L->res, L->own, L->own_ld, L->cosb, R, dn, dq, L->res_dropout_p, L->res_seed, P.ln_g, P.ln_b, L->y, ldy, L->mean, L->rstd);
    TG_TRY(tg::launch_status("ln_res_fwd_kernel"));
    // merge: relu([y | raw] W1^T + b1) W2^T + b2
    const int64_t w1ld = dq + dn;
    if (yr_joined(L, dq)) {          // raw sits right behind y in one (R, dq + dn) buffer: fc1 is ONE product over K = dq + dn
        TG_TRY(tg_gemm_f32(0, 1, R, dn, dq + dn, 1.f, L->y, ldy, P.W1, w1ld, L->f1, dn, P.b1, 1, 0, stream));
    } else {
        TG_TRY(tg_gemm_f32(0, 1, R, dn, dq, 1.f, L->y, ldy, P.W1, w1ld, L->f1, dn, P.b1, 0, 0, stream));
        TG_TRY(tg_gemm_f32(0, 1, R, dn, dn, 1.f, L->raw, L->raw_ld, P.W1 + dq, w1ld, L->f1, dn, nullptr, 1, 1, stream));
    }
    TG_TRY(tg_gemm_f32(0, 1, R, dn, dn, 1.f, L->f1, dn, P.W2, dn, L->out, dn, P.b2, 0, 0, stream));
    return TG_OK;
}

extern "C" int tg_tgat_layer_fwd(const tg_layer_desc* L, void* stream) { return layer_fwd_impl(L, stream, 0, nullptr, nullptr); }

namespace tg {
// a step's forward with ONE prelude launch for its (one or two) layers, optionally zero-filling `zero_floats` floats at `zero` (16-byte
// aligned, a multiple of 4) in the same launch; then every layer's forward behind it, in order
int layers_forward(int n, const tg_layer_desc* const* Ls, float* zero, int64_t zero_floats, void* stream) {
    TG_REQUIRE(n >= 1 && n <= 2 && Ls, "layers_forward: one or two layers");
    TG_REQUIRE(!zero || ((reinterpret_cast<uintptr_t>(zero) & 15) == 0 && zero_floats % 4 == 0), "layers_forward: zero region alignment");
    if (n == 1 && !zero) return layer_fwd_impl(Ls[0], stream, 0, nullptr, nullptr);
    PreludeArgs pa[2];
    int64_t nb[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        if (Ls[i]->attn.m == 0) continue;
        TG_TRY(layer_fwd_impl(Ls[i], stream, 2, &pa[i], &nb[i]));
    }
    if (n == 1) { pa[1] = pa[0]; nb[1] = 0; }
    const int64_t z4 = zero ? zero_floats / 4 : 0;
    const int64_t nz = z4 > 0 ? std::min<int64_t>((z4 + 1023) / 1024, 1024) : 0;
    if (nb[0] + nb[1] + nz > 0) {
        layer_prelude2_kernel<<<(unsigned)(nb[0] + nb[1] + nz), 256, 0, (hipStream_t)stream>>>(pa[0], pa[1], (int)nb[0], (int)nb[1], reinterpret_cast<float4*>(zero), z4);
        TG_TRY(tg::launch_status("layer_prelude2_kernel"));
    }
    for (int i = 0; i < n; ++i) TG_TRY(layer_fwd_impl(Ls[i], stream, 1, nullptr, nullptr));
    return TG_OK;
}
}  // namespace tg

extern "C" int64_t tg_tgat_layer_wt_floats(int dn, int dq, int dk) {
    // transposed copies + merged projections (the native layer path supports 1 or 2 heads: sized for 2), then the packed weights of
    // the chain kernels (largest for H = 1: one head block of dq rows)
    const int64_t plain = r4(wt_floats_plain(dn, dq, dk));
    int64_t pk = 0;
    for (int H = 1; H <= 2; ++H) if (dq % H == 0) pk = std::max(pk, pk_layout(nullptr, H, dn, dq, dk).total);
    return plain + pk + 16;
}

extern "C" int64_t tg_tgat_layer_vec_floats(int dn, int dq, int dk, int heads) {
    const int64_t hk = (int64_t)heads * dk;
    (void)dq;
    return ((dq + hk + 3) / 4) * 4 + ((int64_t)hk * dn + 3) / 4 * 4 + 16;
}

extern "C" int64_t tg_tgat_layer_part_floats(int64_t rows, int dn, int dq, int dt_dim) {
    const int64_t a = ((rows + 15) / 16) * dn;                       // ReLU-mask slabs
    const int64_t b = std::max<int64_t>(row_grid(rows), tg::chain_blocks(rows)) * 4 * dq;   // LayerNorm slabs (one per workgroup of the chain
                                                                                             // kernel when it runs: uncapped, rows / 64 from 8 192 rows)
    const int64_t c = (int64_t)tg_attn_bwd_parts(rows) * 2 * dt_dim; // time-encoder slabs
    const int64_t d = ((rows + 15) / 16) * ((dq + 3) / 4 * 4);       // column sums of dq per 16-row block (layers whose query side is dq_bwd_kernel)
    return 16 + a + b + c + d + 4;                                            // disjoint regions: they are consumed concurrently
    // (with merged projections the caller appends dq * heads * dk + heads * dk * dn + 32 floats: dV and dP)
}

extern "C" int tg_tgat_layer_bwd(const tg_layer_desc* L, const tg_layer_bwd_desc* Bw, void* stream) {
    TG_REQUIRE(L && Bw, "tg_tgat_layer_bwd: null descriptor");
    const tg_attn_desc& a = L->attn;
    const int64_t R = a.m;
    const int H = a.heads, dn = a.dn, T = a.dt_dim, dq = dn + T, dk = dn + a.de + T, hd = dq / H;
    if (R == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const tg_layer_params& P = L->params;
    const tg_layer_grads& G = Bw->grads;
    const tg_layer_desc Lc = *L;                 // by-value copies for the closures the helper thread runs: with a deferred
    const tg_layer_bwd_desc Bc = *Bw;            // join they outlive this call (and the caller's descriptor structs)
    const int64_t w1ld = dq + dn;
    float* vec = Bw->vec;
    const WT wt = wt_layout(L->wT, H, dn, dq, dk);                 // filled by the forward call of this step
    // the products and the LayerNorm backward ahead of the attention as ONE launch (tg_chain.hip) -- under the same conditions as the
    // forward chain, whose prelude packed the transposed weights
    const PK pk = pk_layout(L->wT + r4(wt_floats_plain(dn, dq, dk)), H, dn, dq, dk);
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    float* dres_c = L->res_dropout_p > 0.f ? Bw->dres : Bw->dsum;
    const bool use_chain = g_chain && tg_get_gemm_mode() != 0 && tg::chain_shape_ok(H, dn, T, a.de) && a16(L->agg) && a16(L->ctx) && a16(L->res) &&
                           a16(L->y) && a16(L->raw) && a16(L->own) && a16(L->f1) && a16(L->out) && a16(L->cosb) && a16(P.br) && a16(P.b1) &&
                           a16(P.b2) && a16(P.ln_g) && a16(P.ln_b) && a16(L->wT) && L->own_ld % 4 == 0 && L->raw_ld % 4 == 0 &&
                           (L->y_ld ? L->y_ld : dq) % 4 == 0;                                    // = the forward's decision
    // (the forward chain leaves the NORMALISED LayerNorm input in `res`, which only the backward chain reads that way)
    TG_REQUIRE(!use_chain || (a16(Bw->dout) && a16(Bw->df1) && a16(dres_c) && a16(Bw->dctx) && a16(Bw->dagg) && a16(Bw->part) &&
                              (!Bw->d_own || (a16(Bw->d_own) && Bw->d_own_ld % 4 == 0))),
               "tg_tgat_layer_bwd: backward buffers must be 16-byte aligned (the layer's forward ran the chain kernel)");
    const bool merged = g_merged && R >= kMergedMinRows;
    static const bool no_qu = getenv("FLID_GEMM_TUNE") && getenv("FLID_NO_QU") && atoi(getenv("FLID_NO_QU")) != 0;
    const bool use_qu_fwd = !no_qu && use_chain && !merged && tg::qu_shape_ok(H, dn, T, a.de) && a16(L->q) && a16(L->u) && a16(L->qbias);
    TG_REQUIRE(!use_qu_fwd || (a16(Bw->du) && a16(Bw->dq) && (!Bw->d_own || (a16(Bw->d_own) && Bw->d_own_ld % 4 == 0))),
               "tg_tgat_layer_bwd: du / dq / d_own must be 16-byte aligned (the layer's forward ran the query-side chain launch)");
    const bool use_qu = use_qu_fwd;
    {
        const int fm = forward_mode(L->wT);
        TG_REQUIRE(fm < 0 || fm == ((use_chain ? 1 : 0) | (use_qu ? 2 : 0) | (merged ? 4 : 0)),
                   "tg_tgat_layer_bwd: the layer's forward ran in another mode (chain / query-side launch / merged projection): a switch "
                   "(tg_set_layer_chain, tg_set_layer_merged, tg_set_merged_min_rows, tg_set_gemm_mode) changed between forward and backward");
    }
    const bool overlap = g_overlap && g_side.init();
    // where everything that only feeds parameter gradients goes: re-pointed by every fork()
    void* wstream = stream;
    hipStream_t ws_ = s;
    // side(f): issue f's launches now, or hand them to the issuing thread (then every exit path drains it: the closures
    // refer to this call's descriptors)
    const bool threaded = overlap && g_issue_thread;
    const bool defer = overlap && Bw->defer_join != 0;             // the caller joins once, after its last layer (tg_side_join)
    struct Drain { bool on; ~Drain() { if (on) (void)g_issuer.drain(); } } drain_guard{threaded};    // disarmed on the success paths below
    // timing experiment (WRONG gradients: every weight-gradient launch is dropped): the main chain alone.  Read only in tuning
    // mode, like the other experiment knobs, and announced once -- a stray variable must not silently zero a training run's gradients.
    static const bool exp_skip_side = [] {
        const bool on = getenv("FLID_GEMM_TUNE") != nullptr && getenv("FLID_EXP_SKIP_SIDE") != nullptr;
        if (on) fprintf(stderr, "[flid_tg] FLID_EXP_SKIP_SIDE: weight-gradient launches are SKIPPED (timing experiment, gradients are wrong)\n");
        return on;
    }();
    auto side = [&](std::function<int()> f) -> int {
        if (exp_skip_side) return TG_OK;
        if (!threaded) return f();
        g_issuer.push(std::move(f));
        return TG_OK;
    };
    auto fork = [&](bool same_stream = false) -> int {             // side stream waits for everything issued on main so far
        if (!overlap) return TG_OK;
        int rc = TG_OK;
        hipEvent_t e = g_side.mark(s, &rc);
        if (rc != TG_OK) return rc;
        if (!same_stream) g_side.cur = (g_side.cur + 1) % SideStream::NS;
        hipStream_t sd = g_side.side[g_side.cur];
        wstream = (void*)sd;
        ws_ = sd;
        return side([e, sd] { return hipStreamWaitEvent(sd, e, 0) == hipSuccess ? TG_OK : TG_EHIP; });
    };
    // slab regions of `part` (each finished on the side stream while the main chain moves on)
    const int64_t relu_blocks = (R + 15) / 16;
    // (at most 768 workgroups = 3 per CU walk the rows: every workgroup leaves a slab of 4 dq column sums, and 3 400 of them were
    // 15 MB for the slab-sum launch to read: 11 us)
    const unsigned ln_grid = use_chain ? (unsigned)tg::chain_blocks(R) : (unsigned)std::min<int64_t>(row_grid(R), 768);
    const int attn_parts = tg_attn_bwd_parts(R);
    float* part_relu = Bw->part;
    float* part_ln = part_relu + relu_blocks * dn;
    float* part_attn = part_ln + (int64_t)ln_grid * 4 * dq;
    float* part_dq = part_attn + ((int64_t)attn_parts * 2 * T + 3) / 4 * 4;         // 16-byte aligned (dq % 4 == 0 where dq_bwd_kernel runs)
    const int64_t hk = (int64_t)H * dk;
    // zero-on-entry scratch (`vec`): [sum_rows dq (dq) or sum_rows du (H dk)] | dP (H dk, dn) (merged query side)
    float* vec_dq = vec;
    float* dub = vec;
    float* dPm = vec + ((dq + hk + 3) / 4) * 4;                 // gradient of the merged query projection, (H dk, dn)
    // Weight (and bias) gradients of up to 8 Linear layers in ONE launch (tg_wgrad_group: split-bf16 MFMA, bias
    // sums through a ones column, partial tiles folded with float atomics); shapes it does not cover fall back to one exact
    // product + one column sum per job.
    struct WJ { const float* A; int64_t lda; int M; const float* B; int64_t ldb; int N; float* C; int64_t ldc; float* cs; };
    // Without side streams every weight gradient of the layer waits for the attention backward and leaves in ONE launch (flush_wgrad):
    // a grouped launch costs ~24 us whatever its row count (launch, first-load latency, the atomic fold of slices x outputs), so
    // three launches per layer were 72 us of fixed cost for 93 us of work.  Their operands (dout, f1, df1, [y | raw], dres, ctx,
    // dctx, agg, du, own) are all distinct buffers that stay untouched until the call returns.
    std::vector<WJ> pending;
    // extra (optional): the layer's slab sums, as extra workgroups of the fold launch when the second form takes the jobs, else a launch
    auto issue_wgrad = [&](std::vector<WJ> jobs, const tg::ColExtra* extra = nullptr) -> int {
        hipStream_t st = ws_;
        void* stv = wstream;
        const bool has_extra = extra != nullptr;
        const tg::ColExtra ex = has_extra ? *extra : tg::ColExtra{};
        return side([=] {
            tg_wgrad_job q[8];
            const int n = (int)jobs.size();
            for (int i = 0; i < n; ++i) q[i] = tg_wgrad_job{jobs[i].A, jobs[i].lda, jobs[i].M, jobs[i].B, jobs[i].ldb, jobs[i].N, jobs[i].C, jobs[i].ldc, jobs[i].cs};
            // (both grouped forms are split-bf16 launches; the exact-product mode took the FIRST form here -- 122 + 30 us per 13.6 k-row layer
            // against 57 + 13 -- for no gain in exactness: it takes the second like the default mode.  tg_set_wgrad_grouped(0) is the
            // switch for one exact product + one column sum per gradient.)
            if (g_wgrad_grouped && tg::wgrad_group2(n, q, R, st, has_extra ? &ex : nullptr)) return tg::launch_status("wgrad kernel");
            if (has_extra) TG_TRY(colsum_seg2(ex.a, ex.b, st));
            if (has_extra && ex.wq_n > 0) {            // the time half of dWq that would have ridden in the fold launch
                TailArgs t{};
                t.head_mode = 2; t.n_head = ex.wq_n; t.wq_gx = ex.wq_gx; t.wq_nb = ex.wq_nb; t.v = ex.wq_sq; t.cosb = ex.wq_cosb; t.dq = ex.wq_dq; t.T = ex.wq_T;
                t.Wq = ex.wq_W; t.dWq = ex.wq_dW; t.dn = 0; t.d_cosb = ex.wq_dcosb;
                layer_tail_kernel<<<(unsigned)t.n_head, 256, 0, st>>>(t);
                TG_TRY(tg::launch_status("layer_tail_kernel"));
            }
            if (g_wgrad_grouped && tg::wgrad_group(n, q, R, st)) return tg::launch_status("wgrad kernel");
            for (int i = 0; i < n; ++i) {
                TG_TRY(tg_gemm_f32(1, 0, q[i].M, q[i].N, R, 1.f, q[i].A, q[i].lda, q[i].B, q[i].ldb, q[i].C, q[i].ldc, nullptr, 0, 1, stv));
                if (q[i].colsum_A) TG_TRY(colsum_seg(q[i].A, q[i].lda, R, q[i].M, seg1(q[i].colsum_A, q[i].M), st));
            }
            return (int)TG_OK;
        });
    };
    auto wgrad = [&](std::vector<WJ> jobs) -> int {
        if (overlap || !g_wgrad_grouped) return issue_wgrad(std::move(jobs));      // side streams: start as early as the operands exist
        pending.insert(pending.end(), jobs.begin(), jobs.end());
        return TG_OK;
    };
    auto flush_wgrad = [&](const tg::ColExtra* extra = nullptr) -> int {
        for (size_t i = 0; i < pending.size(); i += 8)
            TG_TRY(issue_wgrad(std::vector<WJ>(pending.begin() + i, pending.begin() + std::min(pending.size(), i + 8)),
                               i + 8 >= pending.size() ? extra : nullptr));
        pending.clear();
        return TG_OK;
    };
    // ---- merge layer -------------------------------------------------------------------------------------------------------
    // df1 = (f1 > 0) ? dout W2 : 0 -- the ReLU mask rides in the product's epilogue (one launch less per layer); widths the fused form
    // does not cover take the product and the mask kernel separately.  (db1 = sum_rows df1 comes out of the weight-gradient launch.)
    if (use_chain) {
        TG_TRY(tg::chain_bwd(L, Bw, dres_c, part_ln, pk.W2T, pk.W1aT, pk.WrT, pk.WvT, s, (Bw->d_raw && L->raw) ? pk.W1bT : nullptr));
    } else if (tg_get_gemm_mode() != 0 && dn % 4 == 0 && (reinterpret_cast<uintptr_t>(Bw->dout) & 15) == 0 && (reinterpret_cast<uintptr_t>(L->f1) & 15) == 0) {   // (the fused mask lives in the split-bf16 kernel)
        TG_TRY(tg_gemm_f32_nt_masked(R, dn, dn, Bw->dout, dn, wt.W2, dn, Bw->df1, dn, L->f1, dn, stream));
    } else {
        TG_TRY(tg_gemm_f32(0, 1, R, dn, dn, 1.f, Bw->dout, dn, wt.W2, dn, Bw->df1, dn, nullptr, 0, 0, stream));
        TG_REQUIRE(dn <= 1024, "tg_tgat_layer_bwd: node dim > 1024 unsupported");
        if (dn <= 256) relu_bwd_colsum_kernel<1><<<(unsigned)relu_blocks, 256, 0, s>>>(Bw->df1, L->f1, R, dn, 16, part_relu);
        else relu_bwd_colsum_kernel<4><<<(unsigned)relu_blocks, 256, 0, s>>>(Bw->df1, L->f1, R, dn, 16, part_relu);
        TG_TRY(tg::launch_status("relu_bwd_colsum_kernel"));
    }
    TG_TRY(fork());                           // dout, df1 are final: dW2 (+ db2), dW1 = df1^T [y | raw] (+ db1)
    const int64_t ldy = L->y_ld ? L->y_ld : dq;
    if (yr_joined(L, dq))
        TG_TRY(wgrad({WJ{Bc.dout, dn, dn, Lc.f1, dn, dn, G.W2, dn, G.b2},
                      WJ{Bc.df1, dn, dn, Lc.y, ldy, dq + dn, G.W1, w1ld, G.b1}}));
    else
        TG_TRY(wgrad({WJ{Bc.dout, dn, dn, Lc.f1, dn, dn, G.W2, dn, G.b2},
                      WJ{Bc.df1, dn, dn, Lc.y, ldy, dq, G.W1, w1ld, G.b1},
                      WJ{Bc.df1, dn, dn, Lc.raw, Lc.raw_ld, dn, G.W1 + dq, w1ld, nullptr}}));
    if (!use_chain) TG_TRY(tg_gemm_f32(0, 1, R, dq, dn, 1.f, Bw->df1, dn, wt.W1a, dn, Bw->dy, dq, nullptr, 0, 0, stream));
    // (the backward chain computes d raw = df1 W1[:, dq:] itself, beside dy: a 6 us launch of its own on TGN's 1 200 rows before)
    if (Bw->d_raw && !(use_chain && L->raw)) TG_TRY(tg_gemm_f32(0, 1, R, dn, dn, 1.f, Bw->df1, dn, wt.W1b, dn, Bw->d_raw, dn, nullptr, 0, 0, stream));
    // ---- residual + layer norm (+ dropout mask), all column sums in one slab -------------------------------------------------
    float* dres = dres_c;
    if (!use_chain) {
        const size_t lds = sizeof(float) * ROW_WAVES * 4 * dq;
        if (dq <= 64) ln_res_bwd_kernel<1><<<ln_grid, 256, lds, s>>>(L->res, L->own, L->own_ld, L->cosb, Bw->dy, R, dn, dq, L->res_dropout_p, L->res_seed, P.ln_g, L->mean, L->rstd, Bw->dsum, dres, part_ln, Bw->d_own, Bw->d_own_ld, Bw->d_own_accumulate);
        else if (dq <= 320) ln_res_bwd_kernel<5><<<ln_grid, 256, lds, s>>>(L->res, L->own, L->own_ld, L->cosb, Bw->dy, R, dn, dq, L->res_dropout_p, L->res_seed, P.ln_g, L->mean, L->rstd, Bw->dsum, dres, part_ln, Bw->d_own, Bw->d_own_ld, Bw->d_own_accumulate);
        else ln_res_bwd_kernel<16><<<ln_grid, 256, lds, s>>>(L->res, L->own, L->own_ld, L->cosb, Bw->dy, R, dn, dq, L->res_dropout_p, L->res_seed, P.ln_g, L->mean, L->rstd, Bw->dsum, dres, part_ln, Bw->d_own, Bw->d_own_ld, Bw->d_own_accumulate);
        TG_TRY(tg::launch_status("ln_res_bwd_kernel"));
    }
    // slab columns: [dgamma | dbeta | sum dsum (node half unused, time half = d cos(b) of the residual) | sum dres (= d br, which the
    // weight-gradient launch delivers through its ones column instead)]
    // head_mode 1: ub_bwd over dub; 2: wq_time over sum_rows dq (vec_dq); both with the slab sums behind them in the same launch
    auto slab_jobs = [&](ColJob& ja, ColJob& jb, int& groups_a, int& col_gx, int& col_ny) {
        ja = ColJob{part_ln, 4 * (int64_t)dq, (int64_t)ln_grid, 3 * dq, SegDst{}};
        ja.d.n = 5;
        ja.d.p[0] = G.ln_g;     ja.d.end[0] = dq;
        ja.d.p[1] = G.ln_b;     ja.d.end[1] = 2 * dq;
        ja.d.p[2] = nullptr;    ja.d.end[2] = 2 * dq + dn;
        ja.d.p[3] = Bw->d_cosb; ja.d.end[3] = 3 * dq;
        ja.d.p[4] = nullptr;    ja.d.end[4] = 4 * dq;
        jb = ColJob{part_attn, 2 * (int64_t)T, T > 0 ? (int64_t)attn_parts : 0, 2 * T, SegDst{}};
        jb.d.n = 2;
        jb.d.p[0] = Bw->d_tew; jb.d.end[0] = T;
        jb.d.p[1] = Bw->d_teb; jb.d.end[1] = 2 * T;
        const int ga = (ja.cols + 63) / 64, gb = jb.n > 0 ? (jb.cols + 63) / 64 : 0;
        const int64_t max_slices = std::max<int64_t>(64, std::min<int64_t>(512, 2048 / std::max(1, ga + gb)));
        groups_a = ga;
        col_gx = ga + gb;
        col_ny = (int)std::min<int64_t>(max_slices, std::max<int64_t>(1, std::max(ja.n, jb.n) / 32));
    };
    auto tail = [&](int head_mode, bool with_slab_sums = true) -> int {
        TailArgs t{};
        t.head_mode = head_mode;
        t.wq_gx = (T + 63) / 64;
        t.n_head = head_mode == 1 ? (dq + UBR - 1) / UBR : (T > 0 ? t.wq_gx * ((dq + WQT_ROWS - 1) / WQT_ROWS) : 0);
        t.v = head_mode == 1 ? dub : (use_qu ? part_dq : vec_dq);
        t.wq_nb = head_mode == 2 && use_qu ? (int)((R + 15) / 16) : 0;      // (the dq launch left per-block partial sums, not the vector)
        t.qb = Lc.qbias; t.Wk = P.Wk; t.Wq = P.Wq; t.cosb = Lc.cosb;
        t.hd = hd; t.dn = dn; t.dq = dq; t.dk = dk; t.T = T;
        t.dWk = G.Wk; t.dWq = G.Wq; t.d_cosb = Bc.d_cosb;
        slab_jobs(t.a, t.b, t.groups_a, t.col_gx, t.col_ny);
        if (!with_slab_sums) t.col_ny = 0;
        hipStream_t st = ws_;
        return side([=] {
            layer_tail_kernel<<<(unsigned)(t.n_head + t.col_gx * t.col_ny), 256, 0, st>>>(t);
            return tg::launch_status("layer_tail_kernel");
        });
    };
    // Only the short-layer form below (query-side launch, the time half of dWq in the fold) leaves nothing behind its grouped launch that
    // reads the launch's results inside this call: every other form cancels a deferral the caller asked for (tg::wgrad_defer_next).
    {
        static const bool no_fold_wq0 = getenv("FLID_GEMM_TUNE") && getenv("FLID_NO_FOLD_WQ") && atoi(getenv("FLID_NO_FOLD_WQ")) != 0;
        if (merged || !use_qu || T <= 0 || overlap || !g_wgrad_grouped || no_fold_wq0 || (Bw->finish_time_bias && T > 0)) tg::wgrad_defer_next(false);
    }
    if (merged) {
        // ---- output projection + value path (the reference's two products; weight gradients in one grouped launch) ------------------
        if (!use_chain) {
            TG_TRY(tg_gemm_f32(0, 1, R, dq, dq, 1.f, dres, dq, wt.Wr, dq, Bw->dctx, dq, nullptr, 0, 0, stream));
            TG_TRY(tg_gemm_f32_batched(0, 1, R, dk, hd, 1.f, Bw->dctx, dq, hd, wt.Wv, hd, (int64_t)dk * hd, Bw->dagg, hk, dk, H, nullptr, 0, 0, stream));
        }
        TG_TRY(fork());                       // dres / dsum, dctx and the LayerNorm slabs are final
        {
            std::vector<WJ> jobs;
            jobs.push_back(WJ{dres, dq, dq, Lc.ctx, dq, dq, G.Wr, dq, G.br});                                                    // dWr, d br
            for (int h = 0; h < H; ++h)                                                                                           // dWv_h = dctx_h^T agg_h
                jobs.push_back(WJ{Bc.dctx + h * hd, dq, hd, Lc.agg + (int64_t)h * dk, hk, dk, G.Wv + (int64_t)h * hd * dk, dk, nullptr});
            TG_TRY(wgrad(jobs));
        }
        // ---- fused attention backward -------------------------------------------------------------------------------------------------
        TG_TRY(tg_attn_bwd(&a, L->u, L->agg, L->prob, Bw->dagg, Bw->du, Bw->dfeat, Bw->dfeat_ld, Bw->pad_row, nullptr, 0, part_attn, stream));
        TG_TRY(fork());                       // du and the time-encoder slabs are final
        TG_TRY(wgrad({WJ{Bc.du, hk, (int)hk, Lc.own, Lc.own_ld, dn, dPm, dn, dub}}));                 // dP = du^T own, dub = sum_rows du
        if (!overlap && g_wgrad_grouped) {
            // everything left is weight space.  The slab sums ride in the weight gradients' fold launch (they were the larger part of the
            // tail launch, which could only start after the weight-space products although it does not depend on them).
            tg::ColExtra ce{};
            slab_jobs(ce.a, ce.b, ce.groups_a, ce.col_gx, ce.col_ny);
            TG_TRY(flush_wgrad(&ce));
            // P_h = Wk_h^T Wq_h[:, :dn] :  dWk_h += Wq_h[:, :dn] dP_h^T ;  dWq_h[:, :dn] += Wk_h dP_h ;  then the constant part (ub_bwd, which
            // also adds into dWk).  (One launch of 32 x 32 fp32 tiles for both products and ub_bwd was measured: 42 us against 7 + 9 + 5 --
            // its K = 444 product is 14 dependent chunk round trips per tile.)
            static const bool no_wtail = getenv("FLID_GEMM_TUNE") && getenv("FLID_NO_WSPACE_TAIL") && atoi(getenv("FLID_NO_WSPACE_TAIL")) != 0;   // A/B timing
            if (!no_wtail && tg_get_gemm_mode() != 0 && tg::wspace_tail(P.Wq, P.Wk, dPm, Lc.qbias, dub, Lc.cosb, G.Wk, G.Wq, Bc.d_cosb, H, hd, dn, dq, dk, T, s)) {
                TG_TRY(tg::launch_status("wspace_tail_kernel"));       // both products and ub_bwd in one launch (exact fp32, as before)
            } else {
                TG_TRY(tg_gemm_f32_batched(0, 1, hd, dk, dn, 1.f, P.Wq, dq, (int64_t)hd * dq, dPm, dn, (int64_t)dk * dn, G.Wk, dk, (int64_t)hd * dk, H, nullptr, 0, 1, stream));
                TG_TRY(tg_gemm_f32_batched(0, 0, hd, dn, dk, 1.f, P.Wk, dk, (int64_t)hd * dk, dPm, dn, (int64_t)dk * dn, G.Wq, dq, (int64_t)hd * dq, H, nullptr, 0, 1, stream));
                TG_TRY(tail(1, false));
            }
            if (Bw->d_own) TG_TRY(tg_gemm_f32(0, 1, R, dn, hk, 1.f, Bw->du, hk, wt.PT, hk, Bw->d_own, Bw->d_own_ld, nullptr, 0, 1, stream));
            drain_guard.on = false;
            if (Bw->finish_time_bias && T > 0) TG_TRY(tg_time_bias_finish(Bw->d_teb, a.d_te_b, Bw->d_cosb, T, stream));
            return TG_OK;
        }
        TG_TRY(flush_wgrad());
        {
            void* stv = wstream;
            // P_h = Wk_h^T Wq_h[:, :dn] :  dWk_h += Wq_h[:, :dn] dP_h^T ;  dWq_h[:, :dn] += Wk_h dP_h ;  then the constant part (ub)
            TG_TRY(side([=] { return tg_gemm_f32_batched(0, 1, hd, dk, dn, 1.f, P.Wq, dq, (int64_t)hd * dq, dPm, dn, (int64_t)dk * dn, G.Wk, dk, (int64_t)hd * dk, H, nullptr, 0, 1, stv); }));
            TG_TRY(side([=] { return tg_gemm_f32_batched(0, 0, hd, dn, dk, 1.f, P.Wk, dk, (int64_t)hd * dk, dPm, dn, (int64_t)dk * dn, G.Wq, dq, (int64_t)hd * dq, H, nullptr, 0, 1, stv); }));
        }
        TG_TRY(tail(1));                      // ub_bwd (after the two products above: it also adds into dWk) + the slab sums
        // ---- key / query path: d own = du P (+ the residual's share) ------------------------------------------------------------------
        if (Bw->d_own) {
            // (the residual's share is already there: ln_res_bwd_kernel)
            TG_TRY(tg_gemm_f32(0, 1, R, dn, hk, 1.f, Bw->du, hk, wt.PT, hk, Bw->d_own, Bw->d_own_ld, nullptr, 0, 1, stream));
        }
    } else {
        TG_TRY(fork());                           // dres / dsum and the LayerNorm slabs are final
        // ---- output projection ------------------------------------------------------------------------------------------------------
        if (!use_chain) TG_TRY(tg_gemm_f32(0, 1, R, dq, dq, 1.f, dres, dq, wt.Wr, dq, Bw->dctx, dq, nullptr, 0, 0, stream));
        // ---- value path -------------------------------------------------------------------------------------------------------------
        if (!use_chain) TG_TRY(tg_gemm_f32_batched(0, 1, R, dk, hd, 1.f, Bw->dctx, dq, hd, wt.Wv, hd, (int64_t)dk * hd, Bw->dagg, hk, dk, H, nullptr, 0, 0, stream));
        // ---- fused attention backward -------------------------------------------------------------------------------------------------
        TG_TRY(tg_attn_bwd(&a, L->u, L->agg, L->prob, Bw->dagg, Bw->du, Bw->dfeat, Bw->dfeat_ld, Bw->pad_row, nullptr, 0, part_attn, stream));
        // ---- key / query path --------------------------------------------------------------------------------------------------------
        // dq_h = du_h Wk_h^T and d_own += dq Wq[:, :dn] in one launch, which also leaves the column sums of dq per 16-row block in `part_dq`
        // (the weight-gradient launch's ones column delivered sum_rows dq before: the time half of dWq then had to wait for the fold)
        if (use_qu) TG_TRY(tg::dq_bwd(L, Bw, pk.Wk, pk.WqT, part_dq, s));
        else TG_TRY(tg_gemm_f32_batched(0, 1, R, hd, dk, 1.f, Bw->du, hk, dk, P.Wk, dk, (int64_t)hd * dk, Bw->dq, dq, hd, H, nullptr, 0, 0, stream));
        TG_TRY(fork());                           // dres, dctx, du, dq are final: the attention block's five weight gradients in one launch
        {
            std::vector<WJ> jobs;
            jobs.push_back(WJ{dres, dq, dq, Lc.ctx, dq, dq, G.Wr, dq, G.br});                                                    // dWr, d br
            for (int h = 0; h < H; ++h)                                                                                           // dWv_h = dctx_h^T agg_h
                jobs.push_back(WJ{Bc.dctx + h * hd, dq, hd, Lc.agg + (int64_t)h * dk, hk, dk, G.Wv + (int64_t)h * hd * dk, dk, nullptr});
            for (int h = 0; h < H; ++h)                                                                                           // dWk_h = q_h^T du_h
                jobs.push_back(WJ{Lc.q + h * hd, dq, hd, Bc.du + (int64_t)h * dk, hk, dk, G.Wk + (int64_t)h * hd * dk, dk, nullptr});
            jobs.push_back(WJ{Bc.dq, dq, dq, Lc.own, Lc.own_ld, dn, G.Wq, dq, use_qu ? nullptr : vec_dq});                       // dWq[:, :dn] (, sum_rows dq)
            TG_TRY(wgrad(jobs));
        }
        if (!overlap && g_wgrad_grouped) {        // the slab sums ride in the weight gradients' fold launch
            tg::ColExtra ce{};
            slab_jobs(ce.a, ce.b, ce.groups_a, ce.col_gx, ce.col_ny);
            static const bool no_fold_wq = getenv("FLID_GEMM_TUNE") && getenv("FLID_NO_FOLD_WQ") && atoi(getenv("FLID_NO_FOLD_WQ")) != 0;
            if (use_qu && T > 0 && !no_fold_wq) {
                // ... and so does the time half of dWq (sum_rows dq came out of the dq launch): no tail launch
                ce.wq_gx = (T + 63) / 64; ce.wq_n = ce.wq_gx * ((dq + WQT_ROWS - 1) / WQT_ROWS);
                ce.wq_dq = dq; ce.wq_T = T; ce.wq_sq = part_dq; ce.wq_nb = (int)((R + 15) / 16); ce.wq_cosb = Lc.cosb; ce.wq_W = P.Wq + dn; ce.wq_dW = G.Wq + dn;
                ce.wq_dcosb = Bc.d_cosb; ce.wq_ld = dq;
                TG_TRY(flush_wgrad(&ce));
            } else {
                TG_TRY(flush_wgrad(&ce));
                TG_TRY(tail(2, false));           // wq_time (the time half of dWq, d cos b)
            }
        } else {
            TG_TRY(flush_wgrad());
            TG_TRY(tail(2));                      // wq_time + the slab sums
        }
        if (Bw->d_own && !use_qu) {
            // (the residual's share is already there: ln_res_bwd_kernel)
            TG_TRY(tg_gemm_f32(0, 1, R, dn, dq, 1.f, Bw->dq, dq, wt.WqL, dq, Bw->d_own, Bw->d_own_ld, nullptr, 0, 1, stream));
        }
    }
    if (overlap && (!defer || Bw->finish_time_bias)) {
        drain_guard.on = false;
        TG_TRY(side_join(s));
    }
    drain_guard.on = false;                   // deferred join: the closures own copies of the descriptors (Lc, Bc)
    if (Bw->finish_time_bias && T > 0) TG_TRY(tg_time_bias_finish(Bw->d_teb, a.d_te_b, Bw->d_cosb, T, stream));
    return TG_OK;
}

extern "C" void tg_set_layer_chain(int on) { g_chain = on != 0; }
extern "C" void tg_set_overlap(int on) { g_overlap = (on & 1) != 0; g_issue_thread = (on & 2) == 0; }
extern "C" void tg_set_layer_merged(int on) { g_merged = on != 0; }
extern "C" void tg_set_wgrad_grouped(int on) { g_wgrad_grouped = on != 0; }
extern "C" void tg_set_merged_min_rows(int64_t rows) { kMergedMinRows = rows; }
