// The constant part of the merged query projection's gradient as a device-side body (layer_tail_kernel of tg_layer.hip, the fused
// weight-space launch of tg_gemm_direct.hip).
#pragma once
#include "tg_common.h"

namespace tg {

// Gradient of the constant part of u (ub_h = Wk_h^T qb_h, qb = Wq[:, dn:] cos b), UBR query rows i = h hd + k per workgroup:
//   dqb_i = Wk[i, :] . dub_h ;  dWk[i, :] += qb_i dub_h ;  dWq[i, dn:] += dqb_i cos b ;  d cos b += Wq[i, dn:] dqb_i
// (one workgroup per row was 272 workgroups adding into the same T addresses of d cos b: 17 us of serialised float atomics; here the
// rows of a workgroup are summed first and the launch makes dq / UBR adds per address)
constexpr int UBR = 8;
__device__ __forceinline__ void ub_bwd_body(int blk, const float* __restrict__ dub, const float* __restrict__ qb, const float* __restrict__ Wk,
        const float* __restrict__ Wq, const float* __restrict__ cosb, int hd, int dn, int dq, int dk, int T, float* __restrict__ dWk,
        float* __restrict__ dWq, float* __restrict__ d_cosb, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blk * UBR;
    const int nwaves = (int)blockDim.x >> 6;
    for (int r = wave; r < UBR; r += nwaves) {                // a wave per row: dqb_i, and the rank-1 update of dWk's row
        const int i = i0 + r;
        if (i >= dq) break;                                    // (wave-uniform)
        const float* du = dub + (int64_t)(i / hd) * dk;
        const float qbi = qb[i];
        float part = 0.f;
        for (int j = lane; j < dk; j += 64) {
            const float d = du[j];
            part = fmaf(Wk[(int64_t)i * dk + j], d, part);
            if (dWk) dWk[(int64_t)i * dk + j] += qbi * d;
        }
        part = tg::wave_sum(part);
        if (lane == 0) red[r] = part;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float cb = cosb[t];
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < UBR; ++r) {
            const int i = i0 + r;
            if (i < dq) {
                const float dqb = red[r];
                dWq[(int64_t)i * dq + dn + t] += dqb * cb;
                acc = fmaf(Wq[(int64_t)i * dq + dn + t], dqb, acc);
            }
        }
        atomicAdd(d_cosb + t, acc);
    }
}


// The query projection sees [own | cos(b)]: with sq = sum_rows dq,
//   dWq[:, dn:] += sq (x) cos(b)      and      d cos(b) += sq^T Wq[:, dn:].
// Thread = one time column x 16 rows (independent loads), grid.y walks the rows.
constexpr int WQT_ROWS = 16;
__device__ __forceinline__ void wq_time_body(int bx, int by, const float* __restrict__ sq, int dq, const float* __restrict__ cosb, int T,
                                             const float* __restrict__ Wq_t, float* __restrict__ dWq_t, int64_t ld, float* __restrict__ d_cosb) {
    const int c = bx * 64 + (int)threadIdx.x;                    // the first 64 threads of the workgroup work
    if (threadIdx.x >= 64 || c >= T) return;
    const float cb = cosb[c];
    const int r0 = by * WQT_ROWS;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < WQT_ROWS; ++j) {
        const int r = r0 + j;
        if (r < dq) {
            const float v = sq[r];
            dWq_t[(int64_t)r * ld + c] += v * cb;
            acc = fmaf(v, Wq_t[(int64_t)r * ld + c], acc);
        }
    }
    atomicAdd(d_cosb + c, acc);
}

// The same with sq given as per-workgroup partial sums (slab[b * dq + r], b < nb: what dq_bwd_kernel leaves, one row per 16-row block --
// 75 workgroups adding into the same 272 addresses with float atomics cost the dq launch 16 us): all 256 threads sum the 16 rows this
// workgroup needs, then the first 64 work as above.  red: 272 floats of LDS.  Every thread of the workgroup must call.
__device__ __forceinline__ void wq_time_slab_body(int bx, int by, const float* __restrict__ slab, int nb, int dq, const float* __restrict__ cosb, int T,
                                                  const float* __restrict__ Wq_t, float* __restrict__ dWq_t, int64_t ld, float* __restrict__ d_cosb,
                                                  float* red) {
    const int t = (int)threadIdx.x, r0 = by * WQT_ROWS;
    {
        const int r = r0 + (t & 15);
        float s = 0.f;
        if (r < dq)
            for (int b = t >> 4; b < nb; b += 16) s += slab[(int64_t)b * dq + r];
        red[t] = s;
    }
    __syncthreads();
    if (t < 16) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k * 16 + t];
        red[256 + t] = v;
    }
    __syncthreads();
    const int c = bx * 64 + t;
    if (t >= 64 || c >= T) return;
    const float cb = cosb[c];
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < WQT_ROWS; ++j) {
        const int r = r0 + j;
        if (r < dq) {
            const float v = red[256 + j];
            dWq_t[(int64_t)r * ld + c] += v * cb;
            acc = fmaf(v, Wq_t[(int64_t)r * ld + c], acc);
        }
    }
    atomicAdd(d_cosb + c, acc);
}

}  // namespace tg
