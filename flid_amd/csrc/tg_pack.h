// Weight packing for the split-bf16 row-block / chain products, as a device-side body shared by pack_weights_kernel
// (tg_gemm_rows.hip) and the layer prelude launch (tg_layer.hip).
// Packed operand of B (N x K, "row n = output column n"): [tile t = n / 16][step s = k / 32][plane hi, lo][lane 0..63][8 bf16]
// lane l of (t, s) holds B[16 t + (l & 15)][32 s + 8 (l >> 4) + 0..7]  -- the B fragment of v_mfma_f32_16x16x32_bf16 --
// zero beyond N or K.  trans = 0: B[n][k] = src[n * ld + k];  trans = 1: B[n][k] = src[k * ld + n] (the transposed weight that the
// input-gradient products multiply with).
#pragma once
#include "tg_common.h"
#include "tg_split.h"

namespace tgs {

struct PackJobs { tg_pack_job j[24]; int frag0[25]; int n; };

// host side: fills the table; returns the number of fragments (one wave handles one), or -1 on a bad job
inline int pack_jobs_fill(PackJobs& pj, int njobs, const tg_pack_job* jobs) {
    if (njobs < 0 || njobs > 24) return -1;
    pj.n = njobs;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].src || !jobs[i].dst || jobs[i].N <= 0 || jobs[i].K <= 0 || (reinterpret_cast<uintptr_t>(jobs[i].dst) & 15)) return -1;
        pj.j[i] = jobs[i];
        pj.frag0[i] = total;
        total += ((jobs[i].N + 15) / 16) * ((jobs[i].K + 31) / 32);
    }
    pj.frag0[njobs] = total;
    return total;
}

// workgroup `bx` of `nblocks` (256 threads: four waves, one fragment per wave and round)
__device__ __forceinline__ void pack_body(const PackJobs& jobs, int bx, int nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = jobs.frag0[jobs.n];
    for (int f = bx * 4 + wave; f < total; f += nblocks * 4) {
        int ji = 0;
        while (ji + 1 < jobs.n && f >= jobs.frag0[ji + 1]) ++ji;
        const tg_pack_job J = jobs.j[ji];
        const int S = (J.K + 31) / 32;
        const int fl = f - jobs.frag0[ji], t = fl / S, s = fl - t * S;
        const int np = 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
        int n = np;
        bool nok = np < J.N;
        if (J.n_pad > 0) { const int r = np % J.n_pad; nok = nok && r < J.n_len; n = (np / J.n_pad) * J.n_len + r; }
        nok = nok && n < (J.src_N > 0 ? J.src_N : J.N);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kp = k0 + q;
            int k = kp;
            bool ok = nok && kp < J.K;
            if (J.k_pad > 0) { const int r = kp % J.k_pad; ok = ok && r < J.k_len; k = (kp / J.k_pad) * J.k_len + r; }
            ok = ok && k < (J.src_K > 0 ? J.src_K : J.K);
            const int64_t o = J.trans ? (int64_t)k * J.ld + n : (int64_t)n * J.ld + k;
            v[q] = ok ? J.src[o] : 0.f;
        }
        uint2 h0, l0, h1, l1;
        split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
        split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
        uint4* dst = reinterpret_cast<uint4*>(J.dst) + ((int64_t)fl * 2) * 64 + lane;
        dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

}  // namespace tgs
