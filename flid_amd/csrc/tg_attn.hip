// Gather-fused single-query temporal attention (forward + backward), one wavefront per attention instance.
//
// replaces: models/modules.py:167-245 (MultiHeadAttention.forward, neighbor side) and the gathers that feed it,
//           models/TGAT.py:110-129 / models/MemoryModel.py:679-700.
//
// The query has length 1, so with u_h = Wk_h^T q_h (computed by the dense kernels)
//     score_{h,n} = scale * u_h . z_n            z_n = [feat[feat_idx_n] | edge[edge_idx_n] | cos(dt_n * w + b)]
//     agg_h       = sum_n dropout(softmax_n(score))_{h,n} * z_n          (ctx_h = Wv_h agg_h, dense kernel)
// is algebraically the reference's attention; each neighbor row is streamed from HBM exactly once per pass
// (online softmax: running max / denominator, no second pass, no LDS).  Lanes own columns of z: a row of
// dk floats is read as 16-byte chunks, lane l owning chunks l, l+64, ...  (688-B table rows -> 43 chunks).
#include <math.h>

#include <algorithm>

#include "tg_common.h"

namespace {

using tg::kWave;

template <int VEC> struct Chunk;
template <> struct Chunk<4> { using T = float4; };
template <> struct Chunk<1> { using T = float; };

template <int VEC>
__device__ __forceinline__ void load_chunk(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void store_chunk(float* __restrict__ p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *p = v[0];
}

struct Seg { int kind; int col; };   // kind 0 node feature, 1 edge feature, 2 time encoding, 3 none

__device__ __forceinline__ Seg classify(int col, int dn, int de, int dk) {
    if (col >= dk) return {3, 0};
    if (col < dn) return {0, col};
    if (col < dn + de) return {1, col - dn};
    return {2, col - dn - de};
}

constexpr int WAVES_PER_BLOCK = 4;

static_assert(WAVES_PER_BLOCK == 4, "tg::attn_grid_blocks assumes 4 instances per workgroup");
inline int64_t attn_grid(int64_t m) { return tg::attn_grid_blocks(m); }

// ------------------------------------------------------------------------------------------------ forward
// Slots are processed RB at a time: all RB rows' loads are issued first (memory-level parallelism: RB x CPL 16-byte loads
// per lane in flight), then the RB x H partial dot products are folded across the wave together.
constexpr int RB = 5;
// SPLIT = one attention instance per WORKGROUP: wave w takes the row batches w, w+4, ... and the four online-softmax states are
// merged through LDS.  For launches of a few thousand instances or fewer (the 1 200-row root layer, a TGN batch) there is about one
// wave per SIMD, so an instance's 20 rows are pure serial latency for that wave (28 / 68 us per launch forward / backward).
constexpr int64_t kSplitMaxRows = 4096;

template <int VEC, int CPL>
__device__ __forceinline__ void load_row(const tg_attn_desc& a, const Seg (&seg)[CPL], const float (&tw)[CPL][VEC],
                                         const float (&tb)[CPL][VEC], int64_t fi, int64_t ei, float dt, bool live,
                                         float (&z)[CPL][VEC]) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        if (live && seg[i].kind == 0) load_chunk<VEC>(a.d_feat + fi * a.feat_ld + seg[i].col, z[i]);
        else if (live && seg[i].kind == 1) load_chunk<VEC>(a.d_edge + ei * a.edge_ld + seg[i].col, z[i]);
        else if (live && seg[i].kind == 2) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) z[i][e] = tg::cos_phase(fmaf(dt, tw[i][e], tb[i][e]));
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) z[i][e] = 0.f;
        }
    }
}

template <int VEC, int CPL, int H, bool SPLIT = false>
__global__ void __launch_bounds__(WAVES_PER_BLOCK* kWave) attn_fwd_kernel(tg_attn_desc a, const float* __restrict__ u,
                                                                         float* __restrict__ agg, float* __restrict__ prob) {
    // merge area of the SPLIT form: per wave the running max / denominator of every head and the un-normalised aggregate
    __shared__ float s_md[SPLIT ? WAVES_PER_BLOCK : 1][2 * H];
    __shared__ float s_acc[SPLIT ? WAVES_PER_BLOCK : 1][SPLIT ? H * CPL * VEC * kWave : 1];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int dk = a.dn + a.de + a.dt_dim;
    const int k = a.k;

    Seg seg[CPL];
    float tw[CPL][VEC], tb[CPL][VEC];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        seg[i] = classify((lane + kWave * i) * VEC, a.dn, a.de, dk);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            tw[i][e] = seg[i].kind == 2 ? a.d_te_w[seg[i].col + e] : 0.f;
            tb[i][e] = seg[i].kind == 2 ? a.d_te_b[seg[i].col + e] : 0.f;
        }
    }

    const int64_t row_first = SPLIT ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    const int64_t row_step = SPLIT ? (int64_t)gridDim.x : (int64_t)gridDim.x * WAVES_PER_BLOCK;
    for (int64_t row = row_first; row < a.m; row += row_step) {
        float uh[H][CPL][VEC], acc[H][CPL][VEC];
        float mx[H], den[H], keep_score[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            mx[h] = -INFINITY; den[h] = 0.f; keep_score[h] = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                if (seg[i].kind != 3) load_chunk<VEC>(u + (row * H + h) * dk + (lane + kWave * i) * VEC, uh[h][i]);
                else { for (int e = 0; e < VEC; ++e) uh[h][i][e] = 0.f; }
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[h][i][e] = 0.f;
            }
        }
        for (int s0 = 0; s0 < k; s0 += kWave) {
            // lane s holds the metadata of slot s0+s: one coalesced read per array, broadcast per slot below
            const int sl = s0 + lane;
            const int64_t mo = row * k + sl;
            const int my_f = sl < k ? a.d_feat_idx[mo] : 0;
            const int my_e = sl < k ? a.d_edge_idx[mo] : 0;
            const int my_n = sl < k ? a.d_nbr[mo] : 0;
            const float my_dt = sl < k ? a.d_dt[mo] : 0.f;
            const int cnt = (k - s0) < kWave ? (k - s0) : kWave;
            for (int sb = SPLIT ? wave * RB : 0; sb < cnt; sb += SPLIT ? WAVES_PER_BLOCK * RB : RB) {
                float z[RB][CPL][VEC];
                int nb[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int s = sb + r;
                    const bool live = s < cnt;
                    const int ss = live ? s : 0;
                    const int64_t fi = __builtin_amdgcn_readlane(my_f, ss);
                    const int64_t ei = __builtin_amdgcn_readlane(my_e, ss);
                    nb[r] = __builtin_amdgcn_readlane(my_n, ss);
                    const float dt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_dt), ss));
                    load_row<VEC, CPL>(a, seg, tw, tb, fi, ei, dt, live, z[r]);
                }
                float part[RB * H];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        float p = 0.f;
#pragma unroll
                        for (int i = 0; i < CPL; ++i)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) p = fmaf(uh[h][i][e], z[r][i][e], p);
                        part[r * H + h] = p;
                    }
                tg::wave_sum_n<RB * H>(part);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int s = sb + r;
                    if (s >= cnt) break;               // wave-uniform
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        float sc = part[r * H + h] * a.scale;
                        if (nb[r] == 0) sc = -1e10f;                                    // modules.py:221
                        if (lane == s) keep_score[h] = sc;
                        const float mnew = fmaxf(mx[h], sc);
                        const float corr = __expf(mx[h] - mnew);
                        const float pe = __expf(sc - mnew);
                        den[h] = den[h] * corr + pe;
                        const float wgt = pe * tg::dropout_keep_scale(a.seed, a.row0 + row, h, s0 + s, a.dropout_p);
#pragma unroll
                        for (int i = 0; i < CPL; ++i)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) acc[h][i][e] = fmaf(wgt, z[r][i][e], acc[h][i][e] * corr);
                        mx[h] = mnew;
                    }
                }
            }
            if constexpr (SPLIT) {
                // k <= 64 (launcher): one metadata tile.  Merge the four waves' states, then every wave writes the probabilities
                // of the slots it owns and wave 0 the aggregate.
                if (lane == 0) {
#pragma unroll
                    for (int h = 0; h < H; ++h) { s_md[wave][2 * h] = mx[h]; s_md[wave][2 * h + 1] = den[h]; }
                }
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int i = 0; i < CPL; ++i)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) s_acc[wave][((h * CPL + i) * VEC + e) * kWave + lane] = acc[h][i][e];
                __syncthreads();
                float M[H], D[H], wsc[H][WAVES_PER_BLOCK];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    M[h] = -INFINITY;
#pragma unroll
                    for (int w = 0; w < WAVES_PER_BLOCK; ++w) M[h] = fmaxf(M[h], s_md[w][2 * h]);
                    D[h] = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
                        wsc[h][w] = s_md[w][2 * h + 1] > 0.f ? __expf(s_md[w][2 * h] - M[h]) : 0.f;      // a wave without rows: den = 0
                        D[h] += s_md[w][2 * h + 1] * wsc[h][w];
                    }
                }
                const bool mine = lane < cnt && (lane / RB) % WAVES_PER_BLOCK == wave;
#pragma unroll
                for (int h = 0; h < H; ++h)
                    if (mine) prob[(row * H + h) * k + lane] = __expf(keep_score[h] - M[h]) / D[h];
                if (wave == 0) {
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const float inv = 1.f / D[h];
#pragma unroll
                        for (int i = 0; i < CPL; ++i) {
                            if (seg[i].kind == 3) continue;
                            float o[VEC];
#pragma unroll
                            for (int e = 0; e < VEC; ++e) {
                                float t = 0.f;
#pragma unroll
                                for (int w = 0; w < WAVES_PER_BLOCK; ++w) t = fmaf(s_acc[w][((h * CPL + i) * VEC + e) * kWave + lane], wsc[h][w], t);
                                o[e] = t * inv;
                            }
                            store_chunk<VEC>(agg + (row * H + h) * dk + (lane + kWave * i) * VEC, o);
                        }
                    }
                }
                __syncthreads();                       // the merge area is reused by the next instance of this workgroup
                continue;                              // (SPLIT: the s0 loop has a single iteration)
            }
            // probabilities need the final max/denominator: exact here when k <= 64 (one tile); for longer rows the raw
            // scores are parked in `prob` and normalised after the loop.
            if (k <= kWave) {
#pragma unroll
                for (int h = 0; h < H; ++h)
                    if (lane < k) prob[(row * H + h) * k + lane] = __expf(keep_score[h] - mx[h]) / den[h];
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h)
                    if (sl < k) prob[(row * H + h) * k + sl] = keep_score[h];
            }
        }
        if constexpr (SPLIT) continue;
        if (k > kWave) {
#pragma unroll
            for (int h = 0; h < H; ++h)
                for (int sl = lane; sl < k; sl += kWave) {
                    const int64_t o = (row * H + h) * k + sl;
                    prob[o] = __expf(prob[o] - mx[h]) / den[h];
                }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float inv = 1.f / den[h];
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                if (seg[i].kind == 3) continue;
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = acc[h][i][e] * inv;
                store_chunk<VEC>(agg + (row * H + h) * dk + (lane + kWave * i) * VEC, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// d score_{h,n} = a'_{h,n} (dagg_h . z_n) - a_{h,n} (dagg_h . agg_h)      a' = dropped/scaled prob, a = softmax prob
// masked slots get no score gradient (masked_fill), but still pass d z through a'.
template <int VEC, int CPL, int H, int RBB = 4>
__global__ void __launch_bounds__(WAVES_PER_BLOCK* kWave) attn_bwd_kernel(tg_attn_desc a, const float* __restrict__ u,
        const float* __restrict__ agg, const float* __restrict__ prob, const float* __restrict__ dagg,
        float* __restrict__ du, float* __restrict__ dfeat, int64_t dfeat_ld, float* __restrict__ dte_part, int64_t pad_row,
        float* __restrict__ dedge, int64_t dedge_ld) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int dk = a.dn + a.de + a.dt_dim;
    const int k = a.k;
    extern __shared__ float red[];   // WAVES_PER_BLOCK * (2 * dt_dim + dn)
    // every padded slot gathers the SAME row (pad_row): its gradient is summed in registers and leaves the workgroup as one
    // row of atomics instead of thousands of adds onto one address (14x slower per the float-atomic contention rule)
    float dpad[CPL][VEC];
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) dpad[i][e] = 0.f;

    Seg seg[CPL];
    float tw[CPL][VEC], tb[CPL][VEC], gw[CPL][VEC], gb[CPL][VEC];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        seg[i] = classify((lane + kWave * i) * VEC, a.dn, a.de, dk);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            tw[i][e] = seg[i].kind == 2 ? a.d_te_w[seg[i].col + e] : 0.f;
            tb[i][e] = seg[i].kind == 2 ? a.d_te_b[seg[i].col + e] : 0.f;
            gw[i][e] = 0.f; gb[i][e] = 0.f;
        }
    }

    const int64_t row_first = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    const int64_t row_step = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    for (int64_t row = row_first; row < a.m; row += row_step) {
        float uh[H][CPL][VEC], dg[H][CPL][VEC], dacc[H][CPL][VEC];
        float cterm[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                float ag[VEC];
                if (seg[i].kind != 3) {
                    const int64_t o = (row * H + h) * dk + (lane + kWave * i) * VEC;
                    load_chunk<VEC>(u + o, uh[h][i]);
                    load_chunk<VEC>(dagg + o, dg[h][i]);
                    load_chunk<VEC>(agg + o, ag);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { uh[h][i][e] = 0.f; dg[h][i][e] = 0.f; ag[e] = 0.f; }
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) { p = fmaf(dg[h][i][e], ag[e], p); dacc[h][i][e] = 0.f; }
            }
            cterm[h] = p;
        }
        tg::wave_sum_n<H>(cterm);
        for (int s0 = 0; s0 < k; s0 += kWave) {
            const int sl = s0 + lane;
            const int64_t mo = row * k + sl;
            const int my_f = sl < k ? a.d_feat_idx[mo] : 0;
            const int my_e = sl < k ? a.d_edge_idx[mo] : 0;
            const int my_n = sl < k ? a.d_nbr[mo] : 0;
            const float my_dt = sl < k ? a.d_dt[mo] : 0.f;
            float my_p[H];
#pragma unroll
            for (int h = 0; h < H; ++h) my_p[h] = sl < k ? prob[(row * H + h) * k + sl] : 0.f;
            const int cnt = (k - s0) < kWave ? (k - s0) : kWave;
            for (int sb = 0; sb < cnt; sb += RBB) {
                float z[RBB][CPL][VEC];
                int nb[RBB];
                int64_t fis[RBB], eis[RBB];
                float dts[RBB];
#pragma unroll
                for (int r = 0; r < RBB; ++r) {
                    const int s = sb + r;
                    const bool live = s < cnt;
                    const int ss = live ? s : 0;
                    fis[r] = __builtin_amdgcn_readlane(my_f, ss);
                    eis[r] = __builtin_amdgcn_readlane(my_e, ss);
                    nb[r] = __builtin_amdgcn_readlane(my_n, ss);
                    dts[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_dt), ss));
                    load_row<VEC, CPL>(a, seg, tw, tb, fis[r], eis[r], dts[r], live, z[r]);
                }
                float part[RBB * H];
#pragma unroll
                for (int r = 0; r < RBB; ++r)
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        float p = 0.f;
#pragma unroll
                        for (int i = 0; i < CPL; ++i)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) p = fmaf(dg[h][i][e], z[r][i][e], p);
                        part[r * H + h] = p;
                    }
                tg::wave_sum_n<RBB * H>(part);
#pragma unroll
                for (int r = 0; r < RBB; ++r) {
                    const int s = sb + r;
                    if (s >= cnt) break;               // wave-uniform
                    float dz[CPL][VEC];
#pragma unroll
                    for (int i = 0; i < CPL; ++i)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) dz[i][e] = 0.f;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const float da = part[r * H + h];
                        const float pr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_p[h]), s));
                        const float pd = pr * tg::dropout_keep_scale(a.seed, a.row0 + row, h, s0 + s, a.dropout_p);
                        const float dsc = nb[r] == 0 ? 0.f : (pd * da - pr * cterm[h]) * a.scale;
#pragma unroll
                        for (int i = 0; i < CPL; ++i)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) {
                                dacc[h][i][e] = fmaf(dsc, z[r][i][e], dacc[h][i][e]);
                                dz[i][e] = fmaf(pd, dg[h][i][e], fmaf(dsc, uh[h][i][e], dz[i][e]));
                            }
                    }
#pragma unroll
                    for (int i = 0; i < CPL; ++i) {
                        if (seg[i].kind == 0 && dfeat) {
                            if (nb[r] == 0 && pad_row >= 0) {
#pragma unroll
                                for (int e = 0; e < VEC; ++e) dpad[i][e] += dz[i][e];
                            } else {
#pragma unroll
                                for (int e = 0; e < VEC; ++e) atomicAdd(dfeat + fis[r] * dfeat_ld + seg[i].col + e, dz[i][e]);
                            }
                        } else if (seg[i].kind == 1 && dedge) {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) atomicAdd(dedge + eis[r] * dedge_ld + seg[i].col + e, dz[i][e]);
                        } else if (seg[i].kind == 2) {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) {
                                float sn, cs;
                                tg::sincos_phase(fmaf(dts[r], tw[i][e], tb[i][e]), &sn, &cs);
                                const float dph = -sn * dz[i][e];
                                gw[i][e] = fmaf(dts[r], dph, gw[i][e]);
                                gb[i][e] += dph;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int i = 0; i < CPL; ++i)
                if (seg[i].kind != 3) store_chunk<VEC>(du + (row * H + h) * dk + (lane + kWave * i) * VEC, dacc[h][i]);
    }

    // block partial of (dw | db): waves -> LDS -> one slab row per workgroup (no atomics, deterministic)
    const int T = a.dt_dim;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
        if (seg[i].kind == 2)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                red[wave * 2 * T + seg[i].col + e] = gw[i][e];
                red[wave * 2 * T + T + seg[i].col + e] = gb[i][e];
            }
    float* redp = red + WAVES_PER_BLOCK * 2 * T;
    if (dfeat && pad_row >= 0) {
#pragma unroll
        for (int i = 0; i < CPL; ++i)
            if (seg[i].kind == 0)
#pragma unroll
                for (int e = 0; e < VEC; ++e) redp[wave * a.dn + seg[i].col + e] = dpad[i][e];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * T; j += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_PER_BLOCK; ++w) s += red[w * 2 * T + j];
        dte_part[(int64_t)blockIdx.x * 2 * T + j] = s;
    }
    if (dfeat && pad_row >= 0) {
        for (int j = threadIdx.x; j < a.dn; j += blockDim.x) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) s += redp[w * a.dn + j];
            if (s != 0.f) atomicAdd(dfeat + pad_row * dfeat_ld + j, s);
        }
    }
}

// one instance per workgroup (SPLIT) pays for launches that leave most SIMDs with a single wave
inline bool use_split(const tg_attn_desc& a, int vec, int cpl) {
    return vec == 4 && cpl == 2 && a.heads == 2 && a.m <= kSplitMaxRows && a.k <= kWave && a.k > RB;
}
inline int64_t split_grid(int64_t m) { return m < 1 ? 1 : (m > tg::kMaxGridBlocks ? tg::kMaxGridBlocks : m); }

template <int VEC, int CPL>
int launch_fwd(const tg_attn_desc& a, const float* u, float* agg, float* prob, hipStream_t s) {
    const dim3 grid((unsigned)attn_grid(a.m)), block(WAVES_PER_BLOCK * kWave);
    if constexpr (VEC == 4 && CPL == 2) {
        if (use_split(a, VEC, CPL)) {
            attn_fwd_kernel<4, 2, 2, true><<<dim3((unsigned)split_grid(a.m)), block, 0, s>>>(a, u, agg, prob);
            return tg::launch_status("attn_fwd_kernel");
        }
    }
    switch (a.heads) {
        case 1: attn_fwd_kernel<VEC, CPL, 1><<<grid, block, 0, s>>>(a, u, agg, prob); break;
        case 2: attn_fwd_kernel<VEC, CPL, 2><<<grid, block, 0, s>>>(a, u, agg, prob); break;
        case 4: attn_fwd_kernel<VEC, CPL, 4><<<grid, block, 0, s>>>(a, u, agg, prob); break;
        default: tg::set_error("tg_attn: heads must be 1, 2 or 4"); return TG_EINVAL;
    }
    return tg::launch_status("attn_fwd_kernel");
}

template <int VEC, int CPL>
int launch_bwd(const tg_attn_desc& a, const float* u, const float* agg, const float* prob, const float* dagg, float* du,
               float* dfeat, int64_t dfeat_ld, float* dte, int64_t pad_row, float* dedge, int64_t dedge_ld, hipStream_t s) {
    const dim3 grid((unsigned)attn_grid(a.m)), block(WAVES_PER_BLOCK * kWave);
    const size_t lds = sizeof(float) * WAVES_PER_BLOCK * (2 * a.dt_dim + a.dn);
    // (a one-instance-per-workgroup form of the backward was measured and dropped: 68 -> 65 us on the root layer, whose time is the 4 M
    // float atomics of the neighbor-feature gradient, not row latency)
    switch (a.heads) {
        case 1: attn_bwd_kernel<VEC, CPL, 1><<<grid, block, lds, s>>>(a, u, agg, prob, dagg, du, dfeat, dfeat_ld, dte, pad_row, dedge, dedge_ld); break;
        case 2: attn_bwd_kernel<VEC, CPL, 2><<<grid, block, lds, s>>>(a, u, agg, prob, dagg, du, dfeat, dfeat_ld, dte, pad_row, dedge, dedge_ld); break;
        case 4: attn_bwd_kernel<VEC, CPL, 4><<<grid, block, lds, s>>>(a, u, agg, prob, dagg, du, dfeat, dfeat_ld, dte, pad_row, dedge, dedge_ld); break;
        default: tg::set_error("tg_attn: heads must be 1, 2 or 4"); return TG_EINVAL;
    }
    return tg::launch_status("attn_bwd_kernel");
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// bit 0: forward on tg_attn_fast.hip, bit 1: backward, bit 2: the fast forward also where the one-instance-per-workgroup (SPLIT)
// form of the generic kernel would run (launches of <= 4096 instances).  tg_set_attn_fast() is for A/B tests and timing.  (An LDS-ring
// form of both kernels -- global_load_lds gathers, 3-4 x the bytes in flight -- ran 161 / 180 us against 94 / 106: commit 8606c5b.)
int g_fast = 7;

int check_desc(const tg_attn_desc* a) {
    TG_REQUIRE(a, "tg_attn: null descriptor");
    TG_REQUIRE(a->d_feat && a->d_feat_idx && a->d_edge && a->d_edge_idx && a->d_nbr && a->d_dt, "tg_attn: null pointer in descriptor");
    TG_REQUIRE(a->dt_dim == 0 || (a->d_te_w && a->d_te_b), "tg_attn: null time-encoder pointer");
    TG_REQUIRE(a->m >= 0 && a->k > 0 && a->dn > 0 && a->de >= 0 && a->dt_dim >= 0, "tg_attn: sizes");
    TG_REQUIRE(a->dn + a->de + a->dt_dim <= 1024, "tg_attn: key dimension > 1024 unsupported");
    TG_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "tg_attn: dropout_p");
    return TG_OK;
}

bool vec4_ok(const tg_attn_desc* a, const void* p0, const void* p1, const void* p2, const void* p3, const void* p4) {
    return a->dn % 4 == 0 && a->de % 4 == 0 && a->dt_dim % 4 == 0 && a->feat_ld % 4 == 0 && a->edge_ld % 4 == 0 &&
           aligned16(a->d_feat) && aligned16(a->d_edge) && aligned16(p0) && aligned16(p1) && aligned16(p2) &&
           aligned16(p3) && aligned16(p4);
}

}  // namespace

extern "C" int tg_attn_fwd(const tg_attn_desc* a, const float* d_u, float* d_agg, float* d_prob, void* stream) {
    if (int rc = check_desc(a)) return rc;
    TG_REQUIRE(d_u && d_agg && d_prob, "tg_attn_fwd: null pointer");
    if (a->m == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int dk = a->dn + a->de + a->dt_dim;
    // algorithmic bytes: k neighbor rows (node + edge) + 16 B slot metadata each, u in, agg out, prob out
    tg::ProfScope prof("attn_fwd", (double)a->m * (a->k * 4.0 * (a->dn + a->de) + a->k * 16.0 + 2.0 * a->heads * dk * 4 + a->heads * a->k * 4.0), s);
    if ((g_fast & 1) && ((g_fast & 4) || !use_split(*a, 4, 2))) {
        const int rc = tg::attn_fwd_fast(*a, d_u, d_agg, d_prob, s);
        if (rc != 1) return rc;
    }
    if (vec4_ok(a, d_u, d_agg, nullptr, nullptr, nullptr)) {
        const int c = (dk / 4 + 63) / 64;
        if (c <= 1) return launch_fwd<4, 1>(*a, d_u, d_agg, d_prob, s);
        if (c <= 2) return launch_fwd<4, 2>(*a, d_u, d_agg, d_prob, s);
        return launch_fwd<4, 4>(*a, d_u, d_agg, d_prob, s);
    }
    const int c = (dk + 63) / 64;
    if (c <= 1) return launch_fwd<1, 1>(*a, d_u, d_agg, d_prob, s);
    TG_REQUIRE(c <= 8, "tg_attn_fwd: unaligned rows wider than 512 floats unsupported");
    return launch_fwd<1, 8>(*a, d_u, d_agg, d_prob, s);
}

namespace {
thread_local float* t_slot_rows_next = nullptr;
thread_local bool t_slot_rows_taken = false;
}  // namespace
namespace tg {
void attn_bwd_slot_rows_next(float* rows) { t_slot_rows_next = rows; }
bool attn_bwd_slot_rows_taken() { return t_slot_rows_taken; }
}  // namespace tg

extern "C" int tg_attn_bwd_parts(int64_t m) { return (int)attn_grid(m); }

extern "C" void tg_set_attn_fast(int mask) { g_fast = mask; }

namespace {
__global__ void __launch_bounds__(256) dropped_scores_kernel(const float* __restrict__ prob, int64_t m, int heads, int k, float p, uint64_t seed,
                                                             float* __restrict__ out) {
    const int64_t total = m * heads * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % k), h = (int)((i / k) % heads);
        out[i] = prob[i] * tg::dropout_keep_scale(seed, i / ((int64_t)k * heads), h, s, p);
    }
}
}  // namespace

extern "C" int tg_attn_dropped_scores(const float* d_prob, int64_t m, int heads, int k, float dropout_p, uint64_t seed, float* d_out,
                                      void* stream) {
    TG_REQUIRE(d_prob && d_out && m >= 0 && heads > 0 && k > 0, "tg_attn_dropped_scores: arguments");
    if (m == 0) return TG_OK;
    const int64_t total = m * heads * k;
    dropped_scores_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, tg::kMaxGridBlocks), 256, 0, (hipStream_t)stream>>>(d_prob, m, heads, k, dropout_p, seed, d_out);
    return tg::launch_status("dropped_scores_kernel");
}

extern "C" int tg_attn_bwd(const tg_attn_desc* a, const float* d_u, const float* d_agg, const float* d_prob,
                           const float* d_dagg, float* d_du, float* d_dfeat, int64_t dfeat_ld, int64_t pad_feat_row,
                           float* d_dedge, int64_t dedge_ld, float* d_dte_part, void* stream) {
    float* slot_rows = t_slot_rows_next;        // (a request is good for ONE call, whoever serves it)
    t_slot_rows_next = nullptr;
    t_slot_rows_taken = false;
    if (int rc = check_desc(a)) return rc;
    TG_REQUIRE(d_u && d_agg && d_prob && d_dagg && d_du && d_dte_part, "tg_attn_bwd: null pointer");
    if (a->m == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int dk = a->dn + a->de + a->dt_dim;
    // as forward, plus dagg and agg in, du out
    tg::ProfScope prof("attn_bwd", (double)a->m * (a->k * 4.0 * (a->dn + a->de) + a->k * 16.0 + 4.0 * a->heads * dk * 4 + a->heads * a->k * 4.0), s);
    if ((g_fast & 2) && slot_rows && d_dfeat && !d_dedge) {
        const int rc = tg::attn_bwd_fast(*a, d_u, d_agg, d_prob, d_dagg, d_du, d_dfeat, dfeat_ld, pad_feat_row, d_dedge, dedge_ld, d_dte_part, s, slot_rows);
        if (rc != 1) { t_slot_rows_taken = rc == TG_OK; return rc; }
    }
    if (g_fast & 2) {
        const int rc = tg::attn_bwd_fast(*a, d_u, d_agg, d_prob, d_dagg, d_du, d_dfeat, dfeat_ld, pad_feat_row, d_dedge, dedge_ld, d_dte_part, s);
        if (rc != 1) return rc;
    }
    if (vec4_ok(a, d_u, d_agg, d_dagg, d_du, nullptr)) {
        const int c = (dk / 4 + 63) / 64;
        if (c <= 1) return launch_bwd<4, 1>(*a, d_u, d_agg, d_prob, d_dagg, d_du, d_dfeat, dfeat_ld, d_dte_part, pad_feat_row, d_dedge, dedge_ld, s);
        if (c <= 2) return launch_bwd<4, 2>(*a, d_u, d_agg, d_prob, d_dagg, d_du, d_dfeat, dfeat_ld, d_dte_part, pad_feat_row, d_dedge, dedge_ld, s);
        return launch_bwd<4, 4>(*a, d_u, d_agg, d_prob, d_dagg, d_du, d_dfeat, dfeat_ld, d_dte_part, pad_feat_row, d_dedge, dedge_ld, s);
    }
    const int c = (dk + 63) / 64;
    if (c <= 1) return launch_bwd<1, 1>(*a, d_u, d_agg, d_prob, d_dagg, d_du, d_dfeat, dfeat_ld, d_dte_part, pad_feat_row, d_dedge, dedge_ld, s);
    TG_REQUIRE(c <= 8, "tg_attn_bwd: unaligned rows wider than 512 floats unsupported");
    return launch_bwd<1, 8>(*a, d_u, d_agg, d_prob, d_dagg, d_du, d_dfeat, dfeat_ld, d_dte_part, pad_feat_row, d_dedge, dedge_ld, s);
}
