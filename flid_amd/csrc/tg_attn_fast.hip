// Production form of the gather-fused single-query temporal attention (tg_attn_fwd / tg_attn_bwd): same math and the same C entry
// points as tg_attn.hip, whose generic kernels stay as the fall-back for odd shapes.  Taken when
//     dn % 4 == de % 4 == 0, (dn + de) / 4 <= 128 chunks, dt_dim <= 128, k <= 64, heads <= 2, 16-byte aligned operands.
//
// replaces: models/modules.py:190-228 (neighbor side of MultiHeadAttention.forward) + the gathers of models/TGAT.py:110-129 /
//           models/MemoryModel.py:679-700, and their autograd.
//
// What bounds these kernels is bytes in flight, not arithmetic: one wave owns one attention instance and streams its k
// neighbor rows (node row + edge row, 2 x 688 B at the reference's 172-d features) from HBM / Infinity Cache.  The first
// version loaded a batch of rows, waited, computed, and only then loaded the next batch -- with 2 waves per SIMD that kept
// ~5 KB per SIMD in flight on average and ran at the latency-bound 2.0 (bwd) / 3.2 (fwd) TB/s.  Here
//   * the rows are SOFTWARE-PIPELINED through two register sets: the loads of batch i+1 are issued before batch i is used, so a
//     wave always has RB..2RB rows (5..11 KB) outstanding;
//   * the time encoding lives on its own lane mapping (column j on lane j mod 64, not in 25 of the 64 float4 slots), so a row
//     costs 2 phase evaluations per lane instead of 4, and the backward takes sin and cos from ONE reduction of the phase
//     (it used to re-evaluate the cosine and then sincos: 3 transcendental passes);
//   * the online-softmax rescale of the aggregate is skipped unless the running maximum actually moves (wave-uniform branch;
//     a new maximum appears ~3.6 times in 20 slots);
//   * neighbor-feature gradients leave the wave as contiguous 256-byte atomic instructions (transposed through LDS) instead of
//     four 16-byte-strided ones: a float-atomic request is 64 B at the memory side, the strided form issued 4x as many.
#include <math.h>
#include <stdlib.h>

#include "tg_common.h"

#ifndef FLID_ATTN_RBF
#define FLID_ATTN_RBF 2
#endif
#ifndef FLID_ATTN_RBB
#define FLID_ATTN_RBB 2
#endif
#ifndef FLID_ATTN_OCC
#define FLID_ATTN_OCC 1
#endif
namespace {

using tg::kWave;
constexpr int WPB = 4;                 // waves (= instances in flight) per workgroup

__device__ __forceinline__ void ld4(const float* __restrict__ p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void st4(float* __restrict__ p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void zero4(float (&v)[4]) { v[0] = v[1] = v[2] = v[3] = 0.f; }

// Lane l owns chunk l (16 bytes) of the node row AND chunk l of the edge row of every neighbor: each of the two loads of a slot then
// has ONE wave-uniform base (a scalar register pair) plus the lane's constant offset -- no per-lane pointer select, no 64-bit vector
// adds.  Rows of up to 64 chunks (256 floats) each; a lane past the row's end re-reads chunk 0: real, finite data that only ever meets
// a zero u / dagg in that lane and is never stored (a predicated load cost six instructions of mask handling and zero fill).
struct LaneMap {
    bool has[2];        // this lane owns a chunk of the node row [0] / edge row [1]
    int col[2];         // its first column inside the table row (0 when it owns none)
    int ucol[2];        // first column inside u / agg / du
};
__device__ __forceinline__ LaneMap lane_map(const tg_attn_desc& a, int lane) {
    LaneMap m;
    m.has[0] = 4 * lane < a.dn;
    m.has[1] = 4 * lane < a.de;
    m.col[0] = m.has[0] ? 4 * lane : 0;
    m.col[1] = m.has[1] ? 4 * lane : 0;
    m.ucol[0] = 4 * lane;
    m.ucol[1] = a.dn + 4 * lane;
    return m;
}

__device__ __forceinline__ void issue_row(const tg_attn_desc& a, const LaneMap& m, int64_t fi, int64_t ei, float (&z)[2][4]) {
    ld4(a.d_feat + fi * a.feat_ld + m.col[0], z[0]);
    ld4(a.d_edge + ei * a.edge_ld + m.col[1], z[1]);
}

__device__ __forceinline__ int rl(int v, int s) { return __builtin_amdgcn_readlane(v, s); }
__device__ __forceinline__ float rlf(float v, int s) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), s)); }
// lane s of v := the wave-uniform x.  v_writelane_b32 has no builtin, and on gfx9 it may name only one scalar register: the lane select
// goes through M0 (compiler-reserved, so saved and restored inside the statement).
__device__ __forceinline__ float wlf(float x, int s, float v) {
    const int xb = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x));
    int keep;
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(v), "=&s"(keep) : "s"(xb), "s"(s));
    return v;
}

inline int64_t fast_grid(int64_t m) { return tg::attn_grid_blocks(m); }

constexpr float kLog2e = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------------ forward
// The slot loop is software-pipelined through two register sets with STATIC counts: every trip issues the rows of the next batch
// unconditionally (past the end it re-reads the last slots, L1 hits) so that hipcc can wait with vmcnt(2 RB) instead of vmcnt(0) -- with
// a conditional issue it drained the batch it had just sent, every trip.
template <int H, int RB>
__global__ void __launch_bounds__(WPB* kWave, FLID_ATTN_OCC) attn_fwd_fast_kernel(tg_attn_desc a, const float* __restrict__ u, float* __restrict__ agg,
                                                                  float* __restrict__ prob) {
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // (uniform: row arithmetic on the scalar unit)
    const int nfe = a.dn + a.de, dk = nfe + a.dt_dim, k = a.k, T = a.dt_dim;
    const LaneMap m = lane_map(a, lane);
    const bool ht0 = lane < T, ht1 = lane + kWave < T;
    // a lane past T computes cos(0) = 1 against a zero u and stores nothing
    const float w0 = ht0 ? a.d_te_w[lane] : 0.f, b0 = ht0 ? a.d_te_b[lane] : 0.f;
    const float w1 = ht1 ? a.d_te_w[lane + kWave] : 0.f, b1 = ht1 ? a.d_te_b[lane + kWave] : 0.f;
    const bool two_t = T > kWave;
    const float scale2 = a.scale * kLog2e;          // scores in base-2 units: exp2 of a difference, no multiply per slot

    for (int64_t row = (int64_t)blockIdx.x * WPB + wave; row < a.m; row += (int64_t)gridDim.x * WPB) {
        const int64_t mo = row * k + lane;
        const bool sl = lane < k;
        const int my_f = sl ? a.d_feat_idx[mo] : 0, my_e = sl ? a.d_edge_idx[mo] : 0, my_n = sl ? a.d_nbr[mo] : 0;
        const float my_dt = sl ? a.d_dt[mo] : 0.f;
        float zA[RB][2][4], zB[RB][2][4];
        auto issue = [&](float (&zb)[RB][2][4], int sb) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int ss = sb + r < k ? sb + r : k - 1;
                issue_row(a, m, rl(my_f, ss), rl(my_e, ss), zb[r]);
            }
        };
        issue(zA, 0);
        float uh[H][2][4], ut[H][2], acc[H][2][4], at[H][2], mx[H], den[H], raw[H], keep[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            // the dropout decisions of this instance, slot s on lane s: ONE hash per lane and head instead of one per slot on every lane
            // (64-bit multiplies on the vector ALU: a third of the kernel's issue slots before)
            keep[h] = sl ? tg::dropout_keep_scale(a.seed, a.row0 + row, h, lane, a.dropout_p) : 0.f;
            const float* ur = u + (row * H + h) * dk;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (m.has[i]) ld4(ur + m.ucol[i], uh[h][i]); else zero4(uh[h][i]);
                zero4(acc[h][i]);
            }
            ut[h][0] = ht0 ? ur[nfe + lane] : 0.f;
            ut[h][1] = ht1 ? ur[nfe + lane + kWave] : 0.f;
            at[h][0] = at[h][1] = 0.f;
            mx[h] = -INFINITY; den[h] = 0.f; raw[h] = 0.f;
        }
        auto compute = [&](float (&zb)[RB][2][4], int sb) {
            if (sb >= k) return;                       // (wave-uniform) a whole batch past the end: only its loads were issued
            float zt[RB][2], part[RB * H];
            int nb[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int ss = sb + r < k ? sb + r : k - 1;
                nb[r] = rl(my_n, ss);
                const float dt = rlf(my_dt, ss);
                zt[r][0] = tg::cos_phase(fmaf(dt, w0, b0));
                zt[r][1] = 0.f;
                if (two_t) zt[r][1] = tg::cos_phase(fmaf(dt, w1, b1));
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float p = ut[h][0] * zt[r][0];
                    p = fmaf(ut[h][1], zt[r][1], p);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) p = fmaf(uh[h][i][e], zb[r][i][e], p);
                    part[r * H + h] = p;
                }
            }
            tg::wave_sum_n<RB * H>(part);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int s = sb + r;
                if (s >= k) break;                     // wave-uniform
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    raw[h] = wlf(part[r * H + h], s, raw[h]);         // the unscaled score of slot s, kept on lane s (-> prob at the end)
                    float sc = part[r * H + h] * scale2;
                    if (nb[r] == 0) sc = -1e10f * kLog2e;                               // modules.py:221
                    if (sc > mx[h]) {                  // wave-uniform: the running maximum moves, rescale what was gathered so far
                        const float corr = __builtin_amdgcn_exp2f(mx[h] - sc);
                        den[h] *= corr;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[h][i][e] *= corr;
                        at[h][0] *= corr; at[h][1] *= corr;
                        mx[h] = sc;
                    }
                    const float pe = __builtin_amdgcn_exp2f(sc - mx[h]);
                    den[h] += pe;
                    const float wgt = pe * rlf(keep[h], s);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[h][i][e] = fmaf(wgt, zb[r][i][e], acc[h][i][e]);
                    at[h][0] = fmaf(wgt, zt[r][0], at[h][0]);
                    at[h][1] = fmaf(wgt, zt[r][1], at[h][1]);
                }
            }
        };
        for (int sb = 0; sb < k; sb += 2 * RB) {
            issue(zB, sb + RB);
            compute(zA, sb);
            issue(zA, sb + 2 * RB);
            compute(zB, sb + RB);
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float inv = 1.f / den[h];
            if (sl) {
                const float sc = my_n == 0 ? -1e10f * kLog2e : raw[h] * scale2;
                prob[(row * H + h) * k + lane] = __builtin_amdgcn_exp2f(sc - mx[h]) * inv;
            }
            float* ar = agg + (row * H + h) * dk;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!m.has[i]) continue;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[h][i][e] * inv;
                st4(ar + m.ucol[i], o);
            }
            if (ht0) ar[nfe + lane] = at[h][0] * inv;
            if (ht1) ar[nfe + lane + kWave] = at[h][1] * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// d score_{h,n} = a'_{h,n} (dagg_h . z_n) - a_{h,n} (dagg_h . agg_h)      a' = dropped/scaled prob, a = softmax prob
// masked slots get no score gradient (masked_fill), but still pass d z through a'.
// DF: a gradient w.r.t. the gathered node rows is wanted (dfeat); DE: w.r.t. the gathered edge rows (dedge; stand-alone
// MultiHeadAttention.forward only -- the backbones' edge table carries no gradient, models/TGAT.py:26-29).
template <int H, int RB, int DFM, bool DE>
__global__ void __launch_bounds__(WPB* kWave, FLID_ATTN_OCC) attn_bwd_fast_kernel(tg_attn_desc a, const float* __restrict__ u, const float* __restrict__ agg,
        const float* __restrict__ prob, const float* __restrict__ dagg, float* __restrict__ du, float* __restrict__ dfeat, int64_t dfeat_ld,
        int64_t pad_row, float* __restrict__ dedge, int64_t dedge_ld, float* __restrict__ dte_part) {
    // DFM: 0 no feature gradient; 1 added into dfeat (float atomics); 2 written as one row per slot: dfeat is then the (m k, dn) slot-row
    // buffer and padded slots are skipped (tg::attn_bwd_slot_rows_next)
    constexpr bool DF = DFM == 1;
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // (uniform: row arithmetic on the scalar unit)
    const int nfe = a.dn + a.de, dk = nfe + a.dt_dim, k = a.k, T = a.dt_dim;
    const LaneMap m = lane_map(a, lane);
    const bool ht0 = lane < T, ht1 = lane + kWave < T;
    // a lane past T evaluates phase 0 against zero u / dagg: every gradient it forms is an exact zero
    const float w0 = ht0 ? a.d_te_w[lane] : 0.f, b0 = ht0 ? a.d_te_b[lane] : 0.f;
    const float w1 = ht1 ? a.d_te_w[lane + kWave] : 0.f, b1 = ht1 ? a.d_te_b[lane + kWave] : 0.f;
    const bool two_t = T > kWave;
    // LDS: [WPB][2 T] time-encoder partials | [WPB][dn] padded-slot feature gradient | [WPB][max(dn, de)] transpose scratch
    extern __shared__ __align__(16) float red[];
    const int wmax = a.dn > a.de ? a.dn : a.de;
    float* tr = red + WPB * (2 * T + a.dn) + wave * wmax;
    float gw[2] = {0.f, 0.f}, gb[2] = {0.f, 0.f};
    float dpad[4];
    zero4(dpad);

    for (int64_t row = (int64_t)blockIdx.x * WPB + wave; row < a.m; row += (int64_t)gridDim.x * WPB) {
        const int64_t mo = row * k + lane;
        const bool sl = lane < k;
        const int my_f = sl ? a.d_feat_idx[mo] : 0, my_e = sl ? a.d_edge_idx[mo] : 0, my_n = sl ? a.d_nbr[mo] : 0;
        const float my_dt = sl ? a.d_dt[mo] : 0.f;
        float zA[RB][2][4], zB[RB][2][4];
        auto issue = [&](float (&zb)[RB][2][4], int sb) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int ss = sb + r < k ? sb + r : k - 1;
                issue_row(a, m, rl(my_f, ss), rl(my_e, ss), zb[r]);
            }
        };
        issue(zA, 0);
        float my_p[H], my_pd[H];
        float uh[H][2][4], ut[H][2], dg[H][2][4], dgt[H][2], dacc[H][2][4], dat[H][2], cterm[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            my_p[h] = sl ? prob[(row * H + h) * k + lane] : 0.f;
            my_pd[h] = sl ? my_p[h] * tg::dropout_keep_scale(a.seed, a.row0 + row, h, lane, a.dropout_p) : 0.f;   // slot s on lane s, hashed once
            const int64_t o = (row * H + h) * dk;
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float ag[4];
                if (m.has[i]) {
                    if (DFM != 0 || DE) ld4(u + o + m.ucol[i], uh[h][i]);
                    ld4(dagg + o + m.ucol[i], dg[h][i]);
                    ld4(agg + o + m.ucol[i], ag);
                } else {
                    zero4(uh[h][i]); zero4(dg[h][i]); zero4(ag);
                }
                zero4(dacc[h][i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) p = fmaf(dg[h][i][e], ag[e], p);
            }
            ut[h][0] = ht0 ? u[o + nfe + lane] : 0.f;
            ut[h][1] = ht1 ? u[o + nfe + lane + kWave] : 0.f;
            dgt[h][0] = ht0 ? dagg[o + nfe + lane] : 0.f;
            dgt[h][1] = ht1 ? dagg[o + nfe + lane + kWave] : 0.f;
            p = fmaf(dgt[h][0], ht0 ? agg[o + nfe + lane] : 0.f, p);
            p = fmaf(dgt[h][1], ht1 ? agg[o + nfe + lane + kWave] : 0.f, p);
            dat[h][0] = dat[h][1] = 0.f;
            cterm[h] = p;
        }
        tg::wave_sum_n<H>(cterm);
        auto compute = [&](float (&zb)[RB][2][4], int sb) {
            if (sb >= k) return;                       // (wave-uniform)
            float zt[RB][2], sn[RB][2], dts[RB], part[RB * H];
            int nb[RB];
            int64_t fis[RB], eis[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int ss = sb + r < k ? sb + r : k - 1;
                nb[r] = rl(my_n, ss);
                fis[r] = rl(my_f, ss);
                eis[r] = rl(my_e, ss);
                dts[r] = rlf(my_dt, ss);
                tg::sincos_phase(fmaf(dts[r], w0, b0), &sn[r][0], &zt[r][0]);
                sn[r][1] = 0.f; zt[r][1] = 0.f;
                if (two_t) tg::sincos_phase(fmaf(dts[r], w1, b1), &sn[r][1], &zt[r][1]);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float p = dgt[h][0] * zt[r][0];
                    p = fmaf(dgt[h][1], zt[r][1], p);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) p = fmaf(dg[h][i][e], zb[r][i][e], p);
                    part[r * H + h] = p;
                }
            }
            tg::wave_sum_n<RB * H>(part);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int s = sb + r;
                if (s >= k) break;                     // wave-uniform
                float pd[H], dsc[H];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float pr = rlf(my_p[h], s);
                    pd[h] = rlf(my_pd[h], s);
                    dsc[h] = nb[r] == 0 ? 0.f : (pd[h] * part[r * H + h] - pr * cterm[h]) * a.scale;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) dacc[h][i][e] = fmaf(dsc[h], zb[r][i][e], dacc[h][i][e]);
                    dat[h][0] = fmaf(dsc[h], zt[r][0], dat[h][0]);
                    dat[h][1] = fmaf(dsc[h], zt[r][1], dat[h][1]);
                }
                // time encoder: d phase = -sin(phase) dz_time
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float dz = 0.f;
#pragma unroll
                    for (int h = 0; h < H; ++h) dz = fmaf(pd[h], dgt[h][j], fmaf(dsc[h], ut[h][j], dz));
                    const float dph = -sn[r][j] * dz;
                    gw[j] = fmaf(dts[r], dph, gw[j]);
                    gb[j] += dph;
                }
                if constexpr (DFM != 0 || DE) {
                    float dz[2][4];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = 0.f;
#pragma unroll
                            for (int h = 0; h < H; ++h) v = fmaf(pd[h], dg[h][i][e], fmaf(dsc[h], uh[h][i][e], v));
                            dz[i][e] = v;
                        }
                    if (DFM == 2) {
                        if (nb[r] != 0 && m.has[0]) st4(dfeat + (row * k + s) * (int64_t)a.dn + m.col[0], dz[0]);       // (nb: wave-uniform)
                    } else if (DF && nb[r] == 0 && pad_row >= 0) {         // wave-uniform: every padded slot gathers the same row
                        if (m.has[0])
#pragma unroll
                            for (int e = 0; e < 4; ++e) dpad[e] += dz[0][e];
                    } else if (DF) {
                        // transpose through LDS: lane l then owns columns l, l+64, ... and one atomic instruction covers 256
                        // contiguous bytes of the destination row
                        if (m.has[0]) st4(tr + m.col[0], dz[0]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        float* dst = dfeat + fis[r] * dfeat_ld;
                        for (int c = lane; c < a.dn; c += kWave) atomicAdd(dst + c, tr[c]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                    if constexpr (DE) {
                        if (m.has[1]) st4(tr + m.col[1], dz[1]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        float* dst = dedge + eis[r] * dedge_ld;
                        for (int c = lane; c < a.de; c += kWave) atomicAdd(dst + c, tr[c]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
            }
        };
        for (int sb = 0; sb < k; sb += 2 * RB) {
            issue(zB, sb + RB);
            compute(zA, sb);
            issue(zA, sb + 2 * RB);
            compute(zB, sb + RB);
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float* dr = du + (row * H + h) * dk;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (m.has[i]) st4(dr + m.ucol[i], dacc[h][i]);
            if (ht0) dr[nfe + lane] = dat[h][0];
            if (ht1) dr[nfe + lane + kWave] = dat[h][1];
        }
    }

    // block partial of (dw | db): waves -> LDS -> one slab row per workgroup (no atomics, deterministic)
    if (ht0) { red[wave * 2 * T + lane] = gw[0]; red[wave * 2 * T + T + lane] = gb[0]; }
    if (ht1) { red[wave * 2 * T + lane + kWave] = gw[1]; red[wave * 2 * T + T + lane + kWave] = gb[1]; }
    float* redp = red + WPB * 2 * T;
    if (DF && pad_row >= 0) {
        if (m.has[0]) st4(redp + wave * a.dn + m.col[0], dpad);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * T; j += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) s += red[w * 2 * T + j];
        dte_part[(int64_t)blockIdx.x * 2 * T + j] = s;
    }
    if (DF && pad_row >= 0) {
        for (int j = threadIdx.x; j < a.dn; j += blockDim.x) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) s += redp[w * a.dn + j];
            if (s != 0.f) atomicAdd(dfeat + pad_row * dfeat_ld + j, s);
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool shape_ok(const tg_attn_desc& a) {
    return a.dn % 4 == 0 && a.de % 4 == 0 && a.dn / 4 <= kWave && a.de / 4 <= kWave && a.de > 0 && a.dt_dim <= 2 * kWave && a.dt_dim % 4 == 0 &&
           a.k <= kWave && (a.heads == 1 || a.heads == 2) && a.feat_ld % 4 == 0 && a.edge_ld % 4 == 0 && aligned16(a.d_feat) &&
           aligned16(a.d_edge);
}

constexpr int kRBF = FLID_ATTN_RBF, kRBB = FLID_ATTN_RBB;
// (4 rows per batch for launches of a few thousand instances -- 8 rows in flight per wave at twice the staging registers -- measured
// SLOWER on the 1 200-instance root launch: forward 16.4 -> 19.1 us, backward 25.0 -> 26.6 us)

}  // namespace

namespace tg {

// returns TG_OK when launched, 1 when the shape is not covered (the caller falls back to the generic kernel)
int attn_fwd_fast(const tg_attn_desc& a, const float* u, float* agg, float* prob, hipStream_t s) {
    if (!shape_ok(a) || !aligned16(u) || !aligned16(agg)) return 1;
    const dim3 grid((unsigned)fast_grid(a.m)), block(WPB * kWave);
    if (a.heads == 1) attn_fwd_fast_kernel<1, kRBF><<<grid, block, 0, s>>>(a, u, agg, prob);
    else attn_fwd_fast_kernel<2, kRBF><<<grid, block, 0, s>>>(a, u, agg, prob);
    return launch_status("attn_fwd_fast_kernel");
}

int attn_bwd_fast(const tg_attn_desc& a, const float* u, const float* agg, const float* prob, const float* dagg, float* du,
                  float* dfeat, int64_t dfeat_ld, int64_t pad_row, float* dedge, int64_t dedge_ld, float* dte, hipStream_t s, float* slot_rows) {
    if (!shape_ok(a) || !aligned16(u) || !aligned16(agg) || !aligned16(dagg) || !aligned16(du)) return 1;
    if (dedge && !dfeat) return 1;
    if (slot_rows && (dedge || !dfeat || !aligned16(slot_rows))) return 1;
    const dim3 grid((unsigned)fast_grid(a.m)), block(WPB * kWave);
    const int wmax = a.dn > a.de ? a.dn : a.de;
    const size_t lds = sizeof(float) * WPB * (2 * a.dt_dim + a.dn + wmax);
#define FLID_LAUNCH(HH, RBB, DFF, DEE) attn_bwd_fast_kernel<HH, RBB, DFF, DEE><<<grid, block, lds, s>>>(a, u, agg, prob, dagg, du, dfeat, dfeat_ld, pad_row, dedge, dedge_ld, dte)
    if (slot_rows) {
        dfeat = slot_rows;
        if (a.heads == 1) FLID_LAUNCH(1, kRBB, 2, false); else FLID_LAUNCH(2, kRBB, 2, false);
    } else if (a.heads == 1) {
        if (dedge) FLID_LAUNCH(1, kRBB, 1, true); else if (dfeat) FLID_LAUNCH(1, kRBB, 1, false); else FLID_LAUNCH(1, kRBB, 0, false);
    } else {
        if (dedge) FLID_LAUNCH(2, kRBB, 1, true); else if (dfeat) FLID_LAUNCH(2, kRBB, 1, false); else FLID_LAUNCH(2, kRBB, 0, false);
    }
#undef FLID_LAUNCH
    return launch_status("attn_bwd_fast_kernel");
}

}  // namespace tg
