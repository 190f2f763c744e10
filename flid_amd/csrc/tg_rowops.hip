// Row-wise helpers around the dense kernels: gathers, fused add+LayerNorm (fwd/bwd), column sums, ReLU mask,
// time encoding.  All HBM-bound, 16-byte accesses where the shape allows.
//   tg_gather_rows        <- models/TGAT.py:87            node_raw_features[ids]
//   tg_add_layernorm_*    <- models/modules.py:238        layer_norm(output + residual), eps 1e-5
//   tg_time_encode        <- models/modules.py:28-40      cos(w t + b)
#include <math.h>
#include <stdlib.h>

#include <initializer_list>

#include "tg_common.h"

namespace {

using tg::kWave;
constexpr int ROW_WAVES = 4;

__host__ __device__ inline int64_t row_grid(int64_t n) {
    int64_t b = (n + ROW_WAVES - 1) / ROW_WAVES;
    return b < 1 ? 1 : (b > tg::kMaxGridBlocks ? tg::kMaxGridBlocks : b);
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ table, int64_t tld,
        const int32_t* __restrict__ idx, int64_t n, int cols, float* __restrict__ out, int64_t old, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * ROW_WAVES + wave; r < n; r += (int64_t)gridDim.x * ROW_WAVES) {
        const float* src = table + (int64_t)idx[r] * tld;
        float* dst = out + r * old;
        if (vec) {
            for (int c = lane * 4; c < cols; c += 256) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
        } else {
            for (int c = lane; c < cols; c += 64) dst[c] = src[c];
        }
    }
}

__global__ void __launch_bounds__(256) scatter_add_rows_kernel(const float* __restrict__ src, int64_t sld,
        const int32_t* __restrict__ idx, int64_t n, int cols, float* __restrict__ table, int64_t tld, const float* __restrict__ src2 = nullptr) {
    // src2 (optional, same layout as src): table[idx[r]] += src[r] + src2[r] -- two gradient streams into the same rows, one atomic each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * ROW_WAVES + wave; r < n; r += (int64_t)gridDim.x * ROW_WAVES) {
        float* dst = table + (int64_t)idx[r] * tld;
        const float* s = src + r * sld;
        const float* s2 = src2 ? src2 + r * sld : nullptr;
        for (int c = lane; c < cols; c += 64) atomicAdd(dst + c, s2 ? s[c] + s2[c] : s[c]);
    }
}

// y = LN(a + b): one wave per row, the row lives in registers (cols <= 64 * MAXC)
template <int MAXC>
__global__ void __launch_bounds__(256) add_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
        int cols, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
        float* __restrict__ mean, float* __restrict__ rstd, float drop_p = 0.f, uint64_t drop_seed = 0, float* __restrict__ sum_out = nullptr) {
    // drop_p > 0: b passes through dropout first (tg_dropout's mask of (drop_seed, flat index)); sum_out (optional) keeps a + dropout(b),
    // the residual stream a pre-LN block hands on
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * ROW_WAVES + wave; r < n; r += (int64_t)gridDim.x * ROW_WAVES) {
        float x[MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            float v = 0.f;
            if (c < cols) {
                const int64_t o = r * cols + c;
                v = a[o];
                if (b) {
                    float bv = b[o];
                    if (drop_p > 0.f) {
                        bv *= tg::res_keep_scale(drop_seed, o, drop_p);
                    }
                    v += bv;
                }
                if (sum_out) sum_out[o] = v;
            }
            x[i] = v;
            s += x[i];
        }
        const float mu = tg::wave_sum(s) / cols;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            const float d = c < cols ? x[i] - mu : 0.f;
            v = fmaf(d, d, v);
        }
        const float rs = rsqrtf(tg::wave_sum(v) / cols + 1e-5f);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) y[r * cols + c] = (x[i] - mu) * rs * gamma[c] + beta[c];
        }
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ; per-workgroup partials of dgamma / dbeta
template <int MAXC>
__global__ void __launch_bounds__(256) add_ln_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
        const float* __restrict__ dy, int64_t n, int cols, const float* __restrict__ gamma, const float* __restrict__ mean,
        const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ part,
        const float* __restrict__ dres = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0, float* __restrict__ dx_dropped = nullptr,
        int64_t part_ld = 0) {
    // dres (optional): dx = dres + dLN(dy) (the residual branch's gradient joins here); dx_dropped (optional): dropout(dx) with the mask
    // of (drop_seed, flat index) as tg_dropout draws it -- the gradient entering a dropout that sits in front of the residual sum
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ float red[];   // ROW_WAVES * 2 * cols
    float dgam[MAXC], dbet[MAXC], gm[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        dgam[i] = 0.f; dbet[i] = 0.f;
        const int c = lane + 64 * i;
        gm[i] = c < cols ? gamma[c] : 0.f;
    }
    for (int64_t r = (int64_t)blockIdx.x * ROW_WAVES + wave; r < n; r += (int64_t)gridDim.x * ROW_WAVES) {
        const float mu = mean[r], rs = rstd[r];
        float xh[MAXC], g[MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < cols;
            const float d = ok ? dy[r * cols + c] : 0.f;
            xh[i] = ok ? (a[r * cols + c] + (b ? b[r * cols + c] : 0.f) - mu) * rs : 0.f;
            g[i] = d * gm[i];
            s1 += g[i];
            s2 = fmaf(g[i], xh[i], s2);
            dgam[i] = fmaf(d, xh[i], dgam[i]);
            dbet[i] += d;
        }
        const float m1 = tg::wave_sum(s1) / cols, m2 = tg::wave_sum(s2) / cols;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) {
                const int64_t o = r * cols + c;
                float v = rs * (g[i] - m1 - xh[i] * m2);
                if (dres) v += dres[o];
                dx[o] = v;
                if (dx_dropped) {
                    dx_dropped[o] = v * tg::res_keep_scale(drop_seed, o, drop_p);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < cols) { red[wave * 2 * cols + c] = dgam[i]; red[wave * 2 * cols + cols + c] = dbet[i]; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * cols; j += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ROW_WAVES; ++w) s += red[w * 2 * cols + j];
        part[(int64_t)blockIdx.x * (part_ld > 0 ? part_ld : 2 * cols) + j] = s;
    }
}

// The same two passes for rows of at most 256 floats, a multiple of 4, 16-byte aligned (DyGFormer's 200-wide tokens): lane l owns
// columns [4 l, 4 l + 4) as ONE 16-byte access per operand, two rows per wave in flight, one 64-bit hash per four dropout decisions.
// (One float per lane and access, one row at a time: 35 us per pass over 38 400 x 200, 3.5 TB/s.)  Same grid, same wave -> row map
// and the same `part` layout as the kernels above.
__device__ __forceinline__ float4 ldv(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stv(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

__global__ void __launch_bounds__(256) add_ln_fwd4_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
        int cols, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
        float* __restrict__ mean, float* __restrict__ rstd, float drop_p, uint64_t drop_seed, float* __restrict__ sum_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = 4 * lane;
    const bool ok = c < cols;
    float gm[4], bt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[j] = ok ? gamma[c + j] : 0.f; bt[j] = ok ? beta[c + j] : 0.f; }
    const int64_t stride = (int64_t)gridDim.x * ROW_WAVES;
    const float inv = 1.f / (float)cols;
    for (int64_t r0 = (int64_t)blockIdx.x * ROW_WAVES + wave; r0 < n; r0 += 2 * stride) {
        const int64_t r1 = r0 + stride;
        const bool has1 = r1 < n;
        const int64_t o0 = r0 * cols + c, o1 = (has1 ? r1 : r0) * cols + c;
        float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0, b0 = x0, b1 = x0;
        if (ok) {
            x0 = ldv(a + o0); x1 = ldv(a + o1);
            if (b) { b0 = ldv(b + o0); b1 = ldv(b + o1); }
        }
        if (b && ok) {
            float k0[4], k1[4];
            tg::res_keep_scale4(drop_seed, o0, drop_p, k0);
            tg::res_keep_scale4(drop_seed, o1, drop_p, k1);
            x0 = make_float4(x0.x + b0.x * k0[0], x0.y + b0.y * k0[1], x0.z + b0.z * k0[2], x0.w + b0.w * k0[3]);
            x1 = make_float4(x1.x + b1.x * k1[0], x1.y + b1.y * k1[1], x1.z + b1.z * k1[2], x1.w + b1.w * k1[3]);
        }
        if (sum_out && ok) { stv(sum_out + o0, x0); if (has1) stv(sum_out + o1, x1); }
        const float mu0 = tg::wave_sum(sum4(x0)) * inv, mu1 = tg::wave_sum(sum4(x1)) * inv;
        float4 d0 = make_float4(x0.x - mu0, x0.y - mu0, x0.z - mu0, x0.w - mu0), d1 = make_float4(x1.x - mu1, x1.y - mu1, x1.z - mu1, x1.w - mu1);
        if (!ok) { d0 = make_float4(0.f, 0.f, 0.f, 0.f); d1 = d0; }
        const float v0 = fmaf(d0.x, d0.x, fmaf(d0.y, d0.y, fmaf(d0.z, d0.z, d0.w * d0.w)));
        const float v1 = fmaf(d1.x, d1.x, fmaf(d1.y, d1.y, fmaf(d1.z, d1.z, d1.w * d1.w)));
        const float rs0 = rsqrtf(tg::wave_sum(v0) * inv + 1e-5f), rs1 = rsqrtf(tg::wave_sum(v1) * inv + 1e-5f);
        if (ok) {
            stv(y + o0, make_float4(d0.x * rs0 * gm[0] + bt[0], d0.y * rs0 * gm[1] + bt[1], d0.z * rs0 * gm[2] + bt[2], d0.w * rs0 * gm[3] + bt[3]));
            if (has1) stv(y + o1, make_float4(d1.x * rs1 * gm[0] + bt[0], d1.y * rs1 * gm[1] + bt[1], d1.z * rs1 * gm[2] + bt[2], d1.w * rs1 * gm[3] + bt[3]));
        }
        if (lane == 0) {
            mean[r0] = mu0; rstd[r0] = rs0;
            if (has1) { mean[r1] = mu1; rstd[r1] = rs1; }
        }
    }
}

__global__ void __launch_bounds__(256) add_ln_bwd4_kernel(const float* __restrict__ a, const float* __restrict__ b,
        const float* __restrict__ dy, int64_t n, int cols, const float* __restrict__ gamma, const float* __restrict__ mean,
        const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ part,
        const float* __restrict__ dres, float drop_p, uint64_t drop_seed, float* __restrict__ dx_dropped, int64_t part_ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = 4 * lane;
    const bool ok = c < cols;
    extern __shared__ float red[];   // ROW_WAVES * 2 * cols
    float gm[4], dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) gm[j] = ok ? gamma[c + j] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * ROW_WAVES;
    const float inv = 1.f / (float)cols;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r0 = (int64_t)blockIdx.x * ROW_WAVES + wave; r0 < n; r0 += 2 * stride) {
        const int64_t r1 = r0 + stride;
        const bool has1 = r1 < n;
        const int64_t q1 = has1 ? r1 : r0;
        const int64_t o0 = r0 * cols + c, o1 = q1 * cols + c;
        float4 d0 = z, d1 = z, x0 = z, x1 = z, e0 = z, e1 = z;
        if (ok) {
            d0 = ldv(dy + o0); d1 = ldv(dy + o1);
            x0 = ldv(a + o0); x1 = ldv(a + o1);
            if (b) {
                const float4 t0 = ldv(b + o0), t1 = ldv(b + o1);
                x0 = make_float4(x0.x + t0.x, x0.y + t0.y, x0.z + t0.z, x0.w + t0.w);
                x1 = make_float4(x1.x + t1.x, x1.y + t1.y, x1.z + t1.z, x1.w + t1.w);
            }
            if (dres) { e0 = ldv(dres + o0); e1 = ldv(dres + o1); }
        }
        if (!has1) d1 = z;
        const float mu0 = mean[r0], rs0 = rstd[r0], mu1 = mean[q1], rs1 = rstd[q1];
        float xh0[4] = {(x0.x - mu0) * rs0, (x0.y - mu0) * rs0, (x0.z - mu0) * rs0, (x0.w - mu0) * rs0};
        float xh1[4] = {(x1.x - mu1) * rs1, (x1.y - mu1) * rs1, (x1.z - mu1) * rs1, (x1.w - mu1) * rs1};
        const float dv0[4] = {d0.x, d0.y, d0.z, d0.w}, dv1[4] = {d1.x, d1.y, d1.z, d1.w};
        float g0[4], g1[4], s10 = 0.f, s20 = 0.f, s11 = 0.f, s21 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!ok) { xh0[j] = 0.f; xh1[j] = 0.f; }
            g0[j] = dv0[j] * gm[j]; g1[j] = dv1[j] * gm[j];
            s10 += g0[j]; s11 += g1[j];
            s20 = fmaf(g0[j], xh0[j], s20); s21 = fmaf(g1[j], xh1[j], s21);
            dgam[j] = fmaf(dv0[j], xh0[j], dgam[j]); dbet[j] += dv0[j];
            dgam[j] = fmaf(dv1[j], xh1[j], dgam[j]); dbet[j] += dv1[j];
        }
        const float m10 = tg::wave_sum(s10) * inv, m20 = tg::wave_sum(s20) * inv, m11 = tg::wave_sum(s11) * inv, m21 = tg::wave_sum(s21) * inv;
        if (ok) {
            const float4 v0 = make_float4(rs0 * (g0[0] - m10 - xh0[0] * m20) + e0.x, rs0 * (g0[1] - m10 - xh0[1] * m20) + e0.y,
                                          rs0 * (g0[2] - m10 - xh0[2] * m20) + e0.z, rs0 * (g0[3] - m10 - xh0[3] * m20) + e0.w);
            stv(dx + o0, v0);
            if (dx_dropped) {
                float k[4];
                tg::res_keep_scale4(drop_seed, o0, drop_p, k);
                stv(dx_dropped + o0, make_float4(v0.x * k[0], v0.y * k[1], v0.z * k[2], v0.w * k[3]));
            }
            if (has1) {
                const float4 v1 = make_float4(rs1 * (g1[0] - m11 - xh1[0] * m21) + e1.x, rs1 * (g1[1] - m11 - xh1[1] * m21) + e1.y,
                                              rs1 * (g1[2] - m11 - xh1[2] * m21) + e1.z, rs1 * (g1[3] - m11 - xh1[3] * m21) + e1.w);
                stv(dx + o1, v1);
                if (dx_dropped) {
                    float k[4];
                    tg::res_keep_scale4(drop_seed, o1, drop_p, k);
                    stv(dx_dropped + o1, make_float4(v1.x * k[0], v1.y * k[1], v1.z * k[2], v1.w * k[3]));
                }
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[wave * 2 * cols + c + j] = dgam[j]; red[wave * 2 * cols + cols + c + j] = dbet[j]; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * cols; j += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ROW_WAVES; ++w) s += red[w * 2 * cols + j];
        part[(int64_t)blockIdx.x * (part_ld > 0 ? part_ld : 2 * cols) + j] = s;
    }
}

inline bool al16p(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
inline bool ln4_ok(int cols, std::initializer_list<const void*> ps) {
    if (cols > 256 || (cols & 3)) return false;
    for (const void* q : ps) if (q && !al16p(q)) return false;
    return true;
}

__global__ void __launch_bounds__(256) relu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        if (!(y[i] > 0.f)) dy[i] = 0.f;
}

__global__ void __launch_bounds__(256) time_encode_kernel(const float* __restrict__ t, int64_t n, const float* __restrict__ w,
        const float* __restrict__ b, int dim, int fused, float* __restrict__ out, const int32_t* __restrict__ mask_ids) {
    const int64_t total = n * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dim;
        const int j = (int)(i - r * dim);
        const float tv = t[r];
        float arg;
        if (fused) arg = fmaf(tv, w[j], b[j]);
        else arg = __fadd_rn(__fmul_rn(tv, w[j]), b[j]);   // keep the two roundings (no contraction)
        out[i] = (mask_ids && mask_ids[r] == 0) ? 0.f : tg::cos_phase(arg);     // DyGFormer.py:266 zeroes padded slots
    }
}

// backward of out = cos(fma(t, w, b)) (optionally masked): per-workgroup partials of dw_j = sum_i -sin(phase_ij) t_i g_ij and
// db_j = sum_i -sin(phase_ij) g_ij.  The phase is re-evaluated with the SAME single rounding as the forward (at phases of
// 1e6 rad one fp32 ulp is 0.1 rad, so a differently rounded phase would give an unrelated sine).
template <int MAXC>
__global__ void __launch_bounds__(256) time_encode_bwd_kernel(const float* __restrict__ t, const int32_t* __restrict__ mask_ids,
        int64_t n, const float* __restrict__ w, const float* __restrict__ b, int dim, const float* __restrict__ g,
        float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ float red[];   // 4 * 2 * dim
    float gw[MAXC], gb[MAXC], wv[MAXC], bv[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        gw[i] = 0.f; gb[i] = 0.f;
        wv[i] = c < dim ? w[c] : 0.f;
        bv[i] = c < dim ? b[c] : 0.f;
    }
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        if (mask_ids && mask_ids[r] == 0) continue;
        const float tv = t[r];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < dim) {
                float sn, cs;
                tg::sincos_phase(fmaf(tv, wv[i], bv[i]), &sn, &cs);
                const float dph = -sn * g[r * dim + c];
                gw[i] = fmaf(tv, dph, gw[i]);
                gb[i] += dph;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < dim) { red[wave * 2 * dim + c] = gw[i]; red[wave * 2 * dim + dim + c] = gb[i]; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * dim; j += blockDim.x)
        part[(int64_t)blockIdx.x * 2 * dim + j] = red[j] + red[2 * dim + j] + red[4 * dim + j] + red[6 * dim + j];
}

// scratch for the two-pass column sums, one per stream that uses them (the layer backward runs them on two streams)

}  // namespace

// Synthetic feature tables for graphs too large to generate on the host: element (row, col) is a pure
// function of (row, col, seed) -- uniform on [-sqrt(3), sqrt(3)) (unit variance), row 0 (the padding row) zero -- so that any row
// can be recomputed on the host for spot parity (flid_amd/synth.py: hash_features_host).
__global__ void __launch_bounds__(256) hash_features_kernel(float* __restrict__ out, int64_t ld, int64_t row0, int64_t nrows, int cols,
                                                            uint64_t seed) {
    const int64_t total = nrows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const int64_t row = row0 + r;
        const uint64_t key = seed ^ ((uint64_t)(row * cols + c) * 0x9E3779B97F4A7C15ULL);
        const float u = (float)(tg::mix32(key) & 0xFFFFFF) * (1.0f / 16777216.0f);
        out[r * ld + c] = row == 0 ? 0.f : (u - 0.5f) * 3.4641016f;
    }
}

extern "C" int tg_hash_features(float* d_out, int64_t ld, int64_t row0, int64_t nrows, int cols, uint64_t seed, void* stream) {
    TG_REQUIRE(d_out && nrows >= 0 && cols > 0 && ld >= cols && row0 >= 0, "tg_hash_features: arguments");
    if (nrows == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((nrows * cols + 255) / 256, tg::kMaxGridBlocks);
    hash_features_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_out, ld, row0, nrows, cols, seed);
    return tg::launch_status("hash_features_kernel");
}

// Adam update of ONE flat fp32 parameter (TGAT.flatten_parameters): the arithmetic of torch.optim.Adam (no amsgrad, L2 weight
// decay folded into the gradient, bias-corrected step) in a single element-wise pass -- torch's multi-tensor kernel spends 46 us
// on a single 1 M-element tensor (16 workgroups), this one ~5 us.
// tb_n > 0: elements [tb_off, tb_off + tb_n) are the time encoder's bias b, whose gradient still lacks what reached cos(b) (the
// encoding of a zero interval, models/TGAT.py:84-85): g -= sin(b) * d_cosb first, written back to the gradient block (= tg_time_bias_finish
// riding in this launch)
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float step_size, float omb1, float b2, float omb2,
                                                   float eps, float wd, float bc2_sqrt, int64_t tb_off, int tb_n, const float* __restrict__ d_cosb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (i >= tb_off && i < tb_off + tb_n) {
            gi -= sinf(p[i]) * d_cosb[i - tb_off];
            g[i] = gi;
        }
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = m[i] + omb1 * (gi - m[i]);                     // lerp, as torch's fused kernel
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

int tg::adam_time_bias(float* d_param, float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, int64_t n, double lr, double beta1,
                       double beta2, double eps, double weight_decay, int64_t step, int64_t tb_off, int tb_n, const float* d_cosb, void* stream) {
    TG_REQUIRE(d_param && d_grad && d_exp_avg && d_exp_avg_sq && n >= 0 && step >= 1, "tg_adam_f32: arguments");
    if (n == 0) return TG_OK;
    // scalars in double, as torch's host side computes them
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const int64_t blocks = std::min<int64_t>((n + 255) / 256, tg::kMaxGridBlocks);
    adam_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_param, d_grad, d_exp_avg, d_exp_avg_sq, n, step_size, omb1, (float)beta2, omb2,
                                                                    (float)eps, (float)weight_decay, bc2_sqrt, tb_off, tb_n, d_cosb);
    return tg::launch_status("adam_kernel");
}
extern "C" int tg_adam_f32(float* d_param, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, int64_t n, double lr, double beta1,
                double beta2, double eps, double weight_decay, int64_t step, void* stream) {
    return tg::adam_time_bias(d_param, const_cast<float*>(d_grad), d_exp_avg, d_exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, 0, 0, nullptr, stream);
}

// d_teb[j] -= sin(b[j]) * d_cosb[j]: the gradient that reached cos(b) (the time encoding of a zero interval, models/TGAT.py:84-85)
// handed on to b; one launch at the end of a backward pass instead of two element-wise torch kernels.
__global__ void time_bias_finish_kernel(float* __restrict__ d_teb, const float* __restrict__ b, const float* __restrict__ d_cosb, int T) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < T) d_teb[j] -= sinf(b[j]) * d_cosb[j];
}
extern "C" int tg_time_bias_finish(float* d_teb, const float* d_b, const float* d_cosb, int dim, void* stream) {
    TG_REQUIRE(d_teb && d_b && d_cosb && dim >= 0, "tg_time_bias_finish: arguments");
    if (dim == 0) return TG_OK;
    time_bias_finish_kernel<<<(dim + 127) / 128, 128, 0, (hipStream_t)stream>>>(d_teb, d_b, d_cosb, dim);
    return tg::launch_status("time_bias_finish_kernel");
}

// out[0] = scale * sum_i a[i] * w[i]: the scalar of a weighted-mean loss over an embedding block.  64 workgroups leave partial sums in
// a small workspace; the last one to arrive (agent-scope ticket) folds them in fixed order and re-arms the ticket -- one launch,
// deterministic (a single workgroup walking the block took 27 us: pure load latency; the last workgroup reading the 64 partial sums
// one after the other, 5 of this kernel's 10 us).
constexpr int WS_BLOCKS = 64;
__global__ void __launch_bounds__(256) weighted_sum_kernel(const float* __restrict__ a, const float* __restrict__ w, int64_t n, float scale,
                                                           float* __restrict__ out, float* __restrict__ part, unsigned int* __restrict__ ticket) {
    __shared__ float red[4];
    __shared__ bool last;
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = a4[i], y = w4[i];
        s = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, s))));
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s = fmaf(a[i], w[i], s);
    s = tg::wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {                            // the partial sums in one load per lane, folded by a fixed tree
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float t2 = (int)threadIdx.x < (int)gridDim.x ? __hip_atomic_load(part + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        t2 = tg::wave_sum(t2);
        if (threadIdx.x == 0) {
            out[0] = t2 * scale;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for the next launch (stream order)
        }
    }
}
extern "C" int tg_weighted_sum(const float* d_a, const float* d_w, int64_t n, float scale, float* d_out, void* stream) {
    TG_REQUIRE(d_a && d_w && d_out && n >= 0, "tg_weighted_sum: arguments");
    TG_REQUIRE(((reinterpret_cast<uintptr_t>(d_a) | reinterpret_cast<uintptr_t>(d_w)) & 15) == 0, "tg_weighted_sum: operands must be 16-byte aligned");
    // per-device workspace: WS_BLOCKS partial sums + the ticket (launches on one device are ordered by their stream; concurrent calls
    // on two streams of one device would share it -- the callers are the trainers' single loss reduction per step)
    static float* ws[16] = {};
    int dev = 0;
    TG_HIP_CHECK(hipGetDevice(&dev));
    TG_REQUIRE(dev >= 0 && dev < 16, "tg_weighted_sum: device index");
    if (!ws[dev]) {
        TG_HIP_CHECK(hipMalloc(&ws[dev], sizeof(float) * (WS_BLOCKS + 4)));
        TG_HIP_CHECK(hipMemset(ws[dev], 0, sizeof(float) * (WS_BLOCKS + 4)));
    }
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(WS_BLOCKS, (n / 4 + 255) / 256));
    weighted_sum_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(d_a, d_w, n, scale, d_out, ws[dev], reinterpret_cast<unsigned int*>(ws[dev] + WS_BLOCKS));
    return tg::launch_status("weighted_sum_kernel");
}

// loss[0] = mean_i BCE(sigmoid(z_i), y_i) with y_i = 1 for i < n_pos else 0, dz_i = (sigmoid(z_i) - y_i) / n: the link-prediction
// warm-up's sigmoid + nn.BCELoss (PTCL/EM_warmup.py:212-222) and its gradient w.r.t. the logits, one workgroup (n = 2 x batch).
// BCELoss clamps log() at -100; softplus(+-z) saturates the same way far beyond any logit a trained head produces.
__global__ void __launch_bounds__(256) bce_logits_kernel(const float* __restrict__ z, int64_t n_pos, int64_t n, float* __restrict__ loss,
                                                         float* __restrict__ dz) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv = 1.f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float v = z[i], y = i < n_pos ? 1.f : 0.f;
        const float sp = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));        // softplus(v) = -log(1 - sigmoid(v))
        float l = sp - y * v;                                             // = -[y log p + (1 - y) log(1 - p)]
        if (l > 100.f) l = 100.f;
        s += l;
        dz[i] = (1.f / (1.f + expf(-v)) - y) * inv;
    }
    s = tg::wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv;
}
extern "C" int tg_bce_logits(const float* d_z, int64_t n_pos, int64_t n, float* d_loss, float* d_dz, void* stream) {
    TG_REQUIRE(d_z && d_loss && d_dz && n > 0 && n_pos >= 0 && n_pos <= n, "tg_bce_logits: arguments");
    bce_logits_kernel<<<1, 256, 0, (hipStream_t)stream>>>(d_z, n_pos, n, d_loss, d_dz);
    return tg::launch_status("bce_logits_kernel");
}

// loss[0] = sum_i w_i CE(z_i, y_i) = sum_i w_i (logsumexp(z_i) - z_i[y_i]) over the rows with y_i >= 0, dz_i = w_i (softmax(z_i) - onehot(y_i))
// (0 for ignored rows): nn.CrossEntropyLoss(reduction='none') of the M-step with its ground-truth / pseudo-label masks and per-sample
// weights folded into w (PTCL/M_step.py:296-312).  One workgroup, a thread per row (C is the number of classes: a handful).
__global__ void __launch_bounds__(256) weighted_ce_kernel(const float* __restrict__ z, int64_t ldz, const int32_t* __restrict__ y,
                                                          const float* __restrict__ w, int64_t n, int C, float* __restrict__ loss,
                                                          float* __restrict__ dz, int64_t lddz) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float* zi = z + i * ldz;
        float* di = dz + i * lddz;
        const int yi = y[i];
        const float wi = (yi >= 0 && yi < C) ? w[i] : 0.f;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, zi[c]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(zi[c] - mx);
        const float lse = mx + logf(den), inv = 1.f / den;
        if (wi != 0.f) s += wi * (lse - zi[yi]);
        for (int c = 0; c < C; ++c) di[c] = wi * (expf(zi[c] - mx) * inv - (c == yi ? 1.f : 0.f));
    }
    s = tg::wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" int tg_weighted_ce(const float* d_z, int64_t ldz, const int32_t* d_labels, const float* d_weights, int64_t n, int classes,
                              float* d_loss, float* d_dz, int64_t lddz, void* stream) {
    TG_REQUIRE(d_z && d_labels && d_weights && d_loss && d_dz && n > 0 && classes > 0 && ldz >= classes && lddz >= classes, "tg_weighted_ce: arguments");
    weighted_ce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(d_z, ldz, d_labels, d_weights, n, classes, d_loss, d_dz, lddz);
    return tg::launch_status("weighted_ce_kernel");
}

extern "C" int tg_rowop_parts(int64_t n) { return (int)row_grid(n); }

extern "C" int tg_gather_rows(const float* d_table, int64_t table_ld, const int32_t* d_idx, int64_t n, int cols, float* d_out,
                              int64_t out_ld, void* stream) {
    TG_REQUIRE(d_table && d_idx && d_out && cols > 0 && n >= 0, "tg_gather_rows: arguments");
    if (n == 0) return TG_OK;
    const int vec = cols % 4 == 0 && table_ld % 4 == 0 && out_ld % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(d_table) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
    gather_rows_kernel<<<(unsigned)row_grid(n), 256, 0, (hipStream_t)stream>>>(d_table, table_ld, d_idx, n, cols, d_out, out_ld, vec);
    return tg::launch_status("gather_rows_kernel");
}

namespace tg {
int scatter_add_rows2(const float* d_src, const float* d_src2, int64_t src_ld, const int32_t* d_idx, int64_t n, int cols, float* d_table, int64_t table_ld,
                      hipStream_t s) {
    TG_REQUIRE(d_src && d_src2 && d_idx && d_table && cols > 0 && n >= 0, "scatter_add_rows2: arguments");
    if (n == 0) return TG_OK;
    scatter_add_rows_kernel<<<(unsigned)row_grid(n), 256, 0, s>>>(d_src, src_ld, d_idx, n, cols, d_table, table_ld, d_src2);
    return launch_status("scatter_add_rows_kernel");
}
}  // namespace tg

extern "C" int tg_scatter_add_rows(const float* d_src, int64_t src_ld, const int32_t* d_idx, int64_t n, int cols,
                                   float* d_table, int64_t table_ld, void* stream) {
    TG_REQUIRE(d_src && d_idx && d_table && cols > 0 && n >= 0, "tg_scatter_add_rows: arguments");
    if (n == 0) return TG_OK;
    scatter_add_rows_kernel<<<(unsigned)row_grid(n), 256, 0, (hipStream_t)stream>>>(d_src, src_ld, d_idx, n, cols, d_table, table_ld);
    return tg::launch_status("scatter_add_rows_kernel");
}

extern "C" int tg_add_layernorm_fwd(const float* d_a, const float* d_b, int64_t n, int cols, const float* d_gamma,
                                    const float* d_beta, float* d_y, float* d_mean, float* d_rstd, void* stream) {
    TG_REQUIRE(d_a && d_gamma && d_beta && d_y && d_mean && d_rstd, "tg_add_layernorm_fwd: null pointer");
    TG_REQUIRE(cols > 0 && cols <= 1024, "tg_add_layernorm_fwd: cols must be in 1..1024");
    if (n == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)row_grid(n);
    if (ln4_ok(cols, {d_a, d_b, d_y})) add_ln_fwd4_kernel<<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd, 0.f, 0, nullptr);
    else if (cols <= 64) add_ln_fwd_kernel<1><<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd);
    else if (cols <= 320) add_ln_fwd_kernel<5><<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd);
    else add_ln_fwd_kernel<16><<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd);
    return tg::launch_status("add_ln_fwd_kernel");
}

extern "C" int tg_add_layernorm_fwd_res(const float* d_a, const float* d_b, int64_t n, int cols, const float* d_gamma, const float* d_beta,
                                        float drop_p, uint64_t drop_seed, float* d_sum, float* d_y, float* d_mean, float* d_rstd, void* stream) {
    TG_REQUIRE(d_a && d_b && d_gamma && d_beta && d_y && d_mean && d_rstd, "tg_add_layernorm_fwd_res: null pointer");
    TG_REQUIRE(cols > 0 && cols <= 1024 && drop_p >= 0.f && drop_p < 1.f, "tg_add_layernorm_fwd_res: cols must be in 1..1024, p in [0, 1)");
    if (n == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)row_grid(n);
    if (ln4_ok(cols, {d_a, d_b, d_y, d_sum})) add_ln_fwd4_kernel<<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd, drop_p, drop_seed, d_sum);
    else if (cols <= 64) add_ln_fwd_kernel<1><<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd, drop_p, drop_seed, d_sum);
    else if (cols <= 320) add_ln_fwd_kernel<5><<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd, drop_p, drop_seed, d_sum);
    else add_ln_fwd_kernel<16><<<g, 256, 0, s>>>(d_a, d_b, n, cols, d_gamma, d_beta, d_y, d_mean, d_rstd, drop_p, drop_seed, d_sum);
    return tg::launch_status("add_ln_fwd_kernel");
}

extern "C" int tg_add_layernorm_bwd(const float* d_a, const float* d_b, const float* d_dy, int64_t n, int cols,
                                    const float* d_gamma, const float* d_mean, const float* d_rstd, float* d_dx,
                                    float* d_dgb_part, void* stream) {
    TG_REQUIRE(d_a && d_dy && d_gamma && d_mean && d_rstd && d_dx && d_dgb_part, "tg_add_layernorm_bwd: null pointer");
    TG_REQUIRE(cols > 0 && cols <= 1024, "tg_add_layernorm_bwd: cols must be in 1..1024");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)row_grid(n);
    const size_t lds = sizeof(float) * ROW_WAVES * 2 * cols;
    if (ln4_ok(cols, {d_a, d_b, d_dy, d_dx})) add_ln_bwd4_kernel<<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part, nullptr, 0.f, 0, nullptr, 0);
    else if (cols <= 64) add_ln_bwd_kernel<1><<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part);
    else if (cols <= 320) add_ln_bwd_kernel<5><<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part);
    else add_ln_bwd_kernel<16><<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part);
    return tg::launch_status("add_ln_bwd_kernel");
}

extern "C" int tg_add_layernorm_bwd_res(const float* d_a, const float* d_b, const float* d_dy, int64_t n, int cols, const float* d_gamma,
                                        const float* d_mean, const float* d_rstd, const float* d_dres, float* d_dx, float* d_dgb_part,
                                        float drop_p, uint64_t drop_seed, float* d_dx_dropped, int64_t part_ld, void* stream) {
    TG_REQUIRE(d_a && d_dy && d_gamma && d_mean && d_rstd && d_dx && d_dgb_part, "tg_add_layernorm_bwd_res: null pointer");
    TG_REQUIRE(part_ld == 0 || part_ld >= 2 * cols, "tg_add_layernorm_bwd_res: part_ld");
    TG_REQUIRE(cols > 0 && cols <= 1024 && drop_p >= 0.f && drop_p < 1.f, "tg_add_layernorm_bwd_res: cols must be in 1..1024, p in [0, 1)");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)row_grid(n);
    const size_t lds = sizeof(float) * ROW_WAVES * 2 * cols;
    if (ln4_ok(cols, {d_a, d_b, d_dy, d_dx, d_dres, d_dx_dropped})) add_ln_bwd4_kernel<<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part, d_dres, drop_p, drop_seed, d_dx_dropped, part_ld);
    else if (cols <= 64) add_ln_bwd_kernel<1><<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part, d_dres, drop_p, drop_seed, d_dx_dropped, part_ld);
    else if (cols <= 320) add_ln_bwd_kernel<5><<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part, d_dres, drop_p, drop_seed, d_dx_dropped, part_ld);
    else add_ln_bwd_kernel<16><<<g, 256, lds, s>>>(d_a, d_b, d_dy, n, cols, d_gamma, d_mean, d_rstd, d_dx, d_dgb_part, d_dres, drop_p, drop_seed, d_dx_dropped, part_ld);
    return tg::launch_status("add_ln_bwd_kernel");
}

// column sums folded with float atomics (one pass; the two-pass form spent as long in its fold kernel as in the sums for the
// 38 400-row inputs of DyGFormer)
__global__ void __launch_bounds__(256) colsum_atomic_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int cols, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < cols) {
        // four rows in flight per lane (the dependent add chain with one 256-byte row segment per round trip ran at 1.1 TB/s)
        const int64_t step = (int64_t)gridDim.y * 4;
        int64_t r = (int64_t)blockIdx.y * 4 + wave;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; r + 3 * step < n; r += 4 * step) {
            const float v0 = x[r * ld + c], v1 = x[(r + step) * ld + c], v2 = x[(r + 2 * step) * ld + c], v3 = x[(r + 3 * step) * ld + c];
            s += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; r < n; r += step) s += x[r * ld + c];
        s += s1 + s2 + s3;
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) atomicAdd(out + c, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

extern "C" int tg_colsum(const float* d_x, int64_t ld, int64_t n, int cols, float* d_out, int accumulate, void* stream) {
    TG_REQUIRE(d_x && d_out && cols > 0 && n >= 0 && ld >= cols, "tg_colsum: arguments");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) TG_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(float) * cols, s));
    if (n == 0) return TG_OK;
    const int col_groups = (cols + 63) / 64;
    // ~512 workgroups in all: every slice ends in one float atomic per column, and 512 slices adding into the same 64 addresses were most
    // of the launch (the column sums of a 38 400 x 200 operand: 21 us)
    static const int tune = (getenv("FLID_GEMM_TUNE") && getenv("FLID_COLSUM_WGS")) ? atoi(getenv("FLID_COLSUM_WGS")) : 0;
    const int64_t max_slices = std::max<int64_t>(16, std::min<int64_t>(512, (tune > 0 ? tune : 512) / col_groups));
    const int slices = (int)std::min<int64_t>(max_slices, std::max<int64_t>(1, n / 32));
    colsum_atomic_kernel<<<dim3(col_groups, slices), 256, 0, s>>>(d_x, ld, n, cols, d_out);
    return tg::launch_status("colsum_atomic_kernel");
}

extern "C" int tg_relu_bwd_inplace(float* d_dy, const float* d_y, int64_t numel, void* stream) {
    TG_REQUIRE(d_dy && d_y && numel >= 0, "tg_relu_bwd_inplace: arguments");
    if (numel == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((numel + 255) / 256, tg::kMaxGridBlocks);
    relu_bwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_dy, d_y, numel);
    return tg::launch_status("relu_bwd_kernel");
}

extern "C" int tg_time_encode_masked(const float* d_t, const int32_t* d_mask_ids, int64_t n, const float* d_w, const float* d_b,
                                     int dim, float* d_out, void* stream) {
    TG_REQUIRE(d_t && d_mask_ids && d_w && d_b && d_out && dim > 0 && n >= 0, "tg_time_encode_masked: arguments");
    if (n == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((n * dim + 255) / 256, tg::kMaxGridBlocks);
    time_encode_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_t, n, d_w, d_b, dim, 1, d_out, d_mask_ids);
    return tg::launch_status("time_encode_kernel");
}

extern "C" int tg_time_encode(const float* d_t, int64_t n, const float* d_w, const float* d_b, int dim, int fused_fma,
                              float* d_out, void* stream) {
    TG_REQUIRE(d_t && d_w && d_b && d_out && dim > 0 && n >= 0, "tg_time_encode: arguments");
    if (n == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((n * dim + 255) / 256, tg::kMaxGridBlocks);
    time_encode_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_t, n, d_w, d_b, dim, fused_fma, d_out, nullptr);
    return tg::launch_status("time_encode_kernel");
}

extern "C" int tg_time_encode_bwd(const float* d_t, const int32_t* d_mask_ids, int64_t n, const float* d_w, const float* d_b,
                                  int dim, const float* d_g, float* d_part, void* stream) {
    TG_REQUIRE(d_t && d_w && d_b && d_g && d_part && dim > 0 && dim <= 512 && n >= 0, "tg_time_encode_bwd: arguments (dim <= 512)");
    const unsigned g = (unsigned)row_grid(n);
    const size_t lds = sizeof(float) * 4 * 2 * dim;
    hipStream_t s = (hipStream_t)stream;
    if (dim <= 128) time_encode_bwd_kernel<2><<<g, 256, lds, s>>>(d_t, d_mask_ids, n, d_w, d_b, dim, d_g, d_part);
    else time_encode_bwd_kernel<8><<<g, 256, lds, s>>>(d_t, d_mask_ids, n, d_w, d_b, dim, d_g, d_part);
    return tg::launch_status("time_encode_bwd_kernel");
}
