// Slab column sums as a device-side body (layer_tail_kernel of tg_layer.hip, the fold launch of tg_wgrad.hip): see tg::ColJob.
#pragma once
#include "tg_common.h"

namespace tg {

// workgroup (bx, by) of a (groups_a + groups_b, ny) grid; red: 4 x 64 floats of LDS
__device__ __forceinline__ void colsum_seg2_body(const ColJob& a, const ColJob& b, int groups_a, int bx, int by, int ny, float (*red)[64]) {
    const bool first = bx < groups_a;
    const ColJob& j = first ? a : b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (bx - (first ? 0 : groups_a)) * 64 + lane;
    if ((int64_t)by * 4 >= j.n) return;                          // (uniform per workgroup) no rows for this slice
    float s = 0.f;
    if (c < j.cols)
        for (int64_t r = (int64_t)by * 4 + wave; r < j.n; r += (int64_t)ny * 4) s += j.x[r * j.ld + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || c >= j.cols) return;
    const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    int beg = 0;
    for (int i = 0; i < j.d.n; ++i) {
        if (c < j.d.end[i]) {
            if (j.d.p[i]) atomicAdd(j.d.p[i] + (c - beg), t);
            return;
        }
        beg = j.d.end[i];
    }
}

}  // namespace tg
