// Row-block products of the layer chain:  C[R, N] = A[R, K] * W^T (+ bias)(+ C)(ReLU)(ReLU-backward mask), split-bf16 ("bf16x3",
// see tg_gemm_bf16x3.hip for the arithmetic), for TALL operands against SMALL weights (N, K <= a few hundred; R = 1 k .. millions).
//
// replaces: aten::mm / addmm behind every nn.Linear of models/modules.py:54-69 (MergeLayer), :152-163, :199-235 (projections of
//           MultiHeadAttention) and their input gradients, as issued per layer by models/TGAT.py:131-142.
//
// Why a second product kernel.  The tile kernel of tg_gemm_bf16x3.hip cuts C into 128 x 96 tiles: a 13.6 k x 272 product is 321
// workgroups that each re-load and re-split the same 128 rows of A once per column tile (3-9x), stage the weights through LDS behind
// one barrier per 32-deep step, and keep ONE wave per SIMD busy with loads, splits, LDS stores, LDS reads and MFMAs in turn
// (profiles/r02: 92-130 TFLOP/s fp32-equivalent, neither HBM- nor MFMA-bound).  Here
//   * a workgroup owns 16 RB rows (64 by default) and ALL N columns: every element of A is loaded and split exactly once per product,
//     every element of C is written once, in whole row segments;
//   * the weights are PACKED once per step (pack_weights_kernel: split into bf16 hi / lo and stored fragment-major, so that the
//     operand of one 16-column tile and one 32-deep step is 1 KiB contiguous): each wave owns a block of column tiles and loads its
//     B fragments straight from L2 into registers, PF steps ahead -- no LDS traffic, no barrier for B;
//   * A passes through LDS in groups of 128 k (two buffers): one barrier per 4 steps x NTW tiles x RB row blocks x 3 MFMAs;
//   * the epilogue transposes through wave-private LDS and stores rows.
// MFMA: v_mfma_f32_16x16x32_bf16 (16-column granularity keeps the four waves' column blocks balanced at N = 172 / 272 / 444).
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "tg_common.h"
#include "tg_pack.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef FLID_ROWS_EXP
#define FLID_ROWS_EXP 0   // timing experiments only (results wrong): 1 no steady-state B loads, 2 no MFMAs, 3 no epilogue, 4 no steady-state A
#endif
constexpr int NTH = 256;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);     // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xFFFF0000u); }
__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
    hi.x = pack_bf16(v.x, v.y);
    hi.y = pack_bf16(v.z, v.w);
    lo.x = pack_bf16(v.x - bf16_lo_to_f32(hi.x), v.y - bf16_hi_to_f32(hi.x));
    lo.y = pack_bf16(v.z - bf16_lo_to_f32(hi.y), v.w - bf16_hi_to_f32(hi.y));
}

// ------------------------------------------------------------------------------------------------ weight packing
// (layout and body: tg_pack.h -- the layer prelude launch of tg_layer.hip runs the same body beside its other roles)
using tgs::PackJobs;
__global__ void __launch_bounds__(NTH) pack_weights_kernel(PackJobs jobs) { tgs::pack_body(jobs, (int)blockIdx.x, (int)gridDim.x); }

// ------------------------------------------------------------------------------------------------ the product
// GS = 32-deep steps per group (the barrier interval).  The k loop runs over HALVES: half h multiplies group h while everything group
// h + 1 needs is in flight -- its B fragments (a second register set, loaded at the top of the half), its A rows (registers -> LDS at
// the top of the half) -- and the A rows of group h + 3 start their trip from HBM.  Whatever a half waits for was issued a whole
// half earlier, so the conservative vmcnt(0) the compiler puts at a loop head costs nothing; the loop body is two halves with all
// register sets named statically, loads unconditional (a group past the end of K re-reads valid memory and multiplies zeros).
template <int RB, int GS>
struct Geo {
    static constexpr int ROWS = 16 * RB;
    static constexpr int CHS = ROWS * 64 + 64;      // bytes between the 32-k chunks of a plane (+64: the chunk stores of one row spread over banks)
    static constexpr int PLANE = GS * CHS;
    static constexpr int BUF = 2 * PLANE;           // hi plane, lo plane
    static constexpr int CPR = 8 * GS;              // float4 per row and group
    static constexpr int RPP = NTH / CPR;           // rows per pass of the 256 threads
    static constexpr int PER = (ROWS + RPP - 1) / RPP;   // float4 loads per thread and group
};

struct RowsArgs {
    const float* A; int64_t lda, strideA;
    const uint4* Bp; int64_t strideB;               // packed operand; batch stride in uint4
    float* C; int64_t ldc, strideC;
    const float* bias;                              // [N] or null (batch b reads bias + b * N)
    const float* mask; int64_t ldm;                 // ReLU backward: keep C where mask > 0
    int64_t R;
    int N, K, S, nt, tps, cpw;                      // S = 32-deep steps, nt = 16-column tiles, tps = tiles per column split, cpw = per wave
    int relu, accumulate;
};

// RB row blocks of 16 per workgroup; NTW = most column tiles a wave owns.  grid = (row blocks, batch, column splits)
template <int RB, int NTW, int GS>
__global__ void __launch_bounds__(NTH, 1) gemm_rows_kernel(RowsArgs a) {
    using G = Geo<RB, GS>;
    constexpr int PR = RB >= 2 ? 2 : 1;             // row blocks per epilogue pass
    constexpr int CW = NTW * 16 + 4;                // staging row stride (floats): 4 rows further = 16 banks further
    constexpr int LDS_BYTES = (2 * G::BUF > 4 * PR * 16 * CW * 4) ? 2 * G::BUF : 4 * PR * 16 * CW * 4;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * G::ROWS;
    const int batch = blockIdx.y;
    const float* __restrict__ A = a.A + batch * a.strideA;
    const uint4* __restrict__ Bp = a.Bp + batch * a.strideB;
    float* __restrict__ C = a.C + batch * a.strideC;
    const float* __restrict__ bias = a.bias ? a.bias + (int64_t)batch * a.N : nullptr;
    const int S = a.S;

    // ---- A: thread (r8 = tid / CPR, c4 = tid % CPR) loads floats [4 c4, 4 c4 + 4) of rows r8, r8 + RPP, ... of a group
    const int c4 = tid % G::CPR, r8 = tid / G::CPR;
    const float* aptr[G::PER];
    bool arow_ok[G::PER];
    int aoff[G::PER];
#pragma unroll
    for (int i = 0; i < G::PER; ++i) {
        const int r = r8 + G::RPP * i;
        int64_t rg = row0 + r;
        arow_ok[i] = rg < a.R && r < G::ROWS;
        if (rg > a.R - 1) rg = a.R - 1;
        aptr[i] = A + rg * a.lda;
        aoff[i] = (c4 >> 3) * G::CHS + (r < G::ROWS ? r : 0) * 64 + (((((c4 & 7) >> 1) ^ (((r >> 3) & 1) << 1))) << 4) + (c4 & 1) * 8;
    }
    // (the loads only: masking rows / columns past the end happens in writeA, after the wait the LDS stores need anyway -- a select
    // right behind the load made the compiler wait for every load as soon as it was issued)
    auto loadA = [&](float4 (&ra)[G::PER], int g) {
        const int k = 32 * GS * g + 4 * c4;
        const int ko = k < a.K ? k : 0;
#pragma unroll
        for (int i = 0; i < G::PER; ++i) ra[i] = *reinterpret_cast<const float4*>(aptr[i] + ko);
    };
    auto writeA = [&](int buf, const float4 (&ra)[G::PER], int g) {
        char* base = lds + buf * G::BUF;
        const bool kok = 32 * GS * g + 4 * c4 < a.K;
#pragma unroll
        for (int i = 0; i < G::PER; ++i) {
            const bool ok = kok && arow_ok[i];
            const float4 v = make_float4(ok ? ra[i].x : 0.f, ok ? ra[i].y : 0.f, ok ? ra[i].z : 0.f, ok ? ra[i].w : 0.f);
            uint2 hi, lo;
            split4(v, hi, lo);
            if (G::ROWS % G::RPP == 0 || r8 + G::RPP * i < G::ROWS) {
                *reinterpret_cast<uint2*>(base + aoff[i]) = hi;
                *reinterpret_cast<uint2*>(base + G::PLANE + aoff[i]) = lo;
            }
        }
    };
    // ---- B: this wave's column tiles [t0, t0 + cpw) of the split's range (clamped: a surplus tile repeats the last one, never stored)
    const int split0 = blockIdx.z * a.tps;
    int tend = split0 + a.tps;
    if (tend > a.nt) tend = a.nt;
    const int t0 = split0 + wave * a.cpw;
    const uint4* bptr[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int t = t0 + j;
        if (t > a.nt - 1) t = a.nt - 1;
        bptr[j] = Bp + (int64_t)t * S * 128 + lane;
    }
    auto loadB = [&](bf16x8 (&bh)[GS][NTW], bf16x8 (&bl)[GS][NTW], int g) {
#pragma unroll
        for (int sl = 0; sl < GS; ++sl) {
            const int s = g * GS + sl;
            const int sc = s < S ? s : S - 1;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                bh[sl][j] = __builtin_bit_cast(bf16x8, bptr[j][sc * 128]);
                bl[sl][j] = __builtin_bit_cast(bf16x8, bptr[j][sc * 128 + 64]);
            }
        }
    };
    f32x4 acc[RB][NTW];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A fragment of row block rb, step sl of the group: lane (r = l & 15, q = l >> 4) reads k = 8 q .. 8 q + 7 of row 16 rb + r
    const int laneoff = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
    auto compute = [&](int buf, const bf16x8 (&bh)[GS][NTW], const bf16x8 (&bl)[GS][NTW]) {
        const char* abuf = lds + buf * G::BUF + laneoff;
#pragma unroll
        for (int sl = 0; sl < GS; ++sl) {
            bf16x8 ah[RB], al[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                ah[rb] = *reinterpret_cast<const bf16x8*>(abuf + sl * G::CHS + rb * 1024);
                al[rb] = *reinterpret_cast<const bf16x8*>(abuf + G::PLANE + sl * G::CHS + rb * 1024);
            }
            if (FLID_ROWS_EXP != 2) {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rb], bh[sl][j], acc[rb][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rb], bl[sl][j], acc[rb][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rb], bh[sl][j], acc[rb][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const float4 x = __builtin_bit_cast(float4, ah[rb]), y = __builtin_bit_cast(float4, bh[sl][j]);
                        const float4 z = __builtin_bit_cast(float4, al[rb]), w = __builtin_bit_cast(float4, bl[sl][j]);
                        acc[rb][j][0] += x.x + y.y + z.z + w.w;
                    }
            }
        }
    };

    const int ngroups = (S + GS - 1) / GS;
    float4 ra0[G::PER], ra1[G::PER];
    bf16x8 b0h[GS][NTW], b0l[GS][NTW], b1h[GS][NTW], b1l[GS][NTW];
    loadA(ra0, 0);
    loadB(b0h, b0l, 0);
    loadA(ra1, 1);
    writeA(0, ra0, 0);
    loadA(ra0, 2);
    __syncthreads();
    for (int g = 0; g < ngroups; g += 2) {
        // half g: group g from LDS buffer 0 / B set 0
        if (FLID_ROWS_EXP != 4) { writeA(1, ra1, g + 1); loadA(ra1, g + 3); }
        if (FLID_ROWS_EXP != 1) loadB(b1h, b1l, g + 1);
        compute(0, b0h, b0l);
        __syncthreads();
        // half g + 1: group g + 1 from LDS buffer 1 / B set 1
        if (FLID_ROWS_EXP != 4) { writeA(0, ra0, g + 2); loadA(ra0, g + 4); }
        if (FLID_ROWS_EXP != 1) loadB(b0h, b0l, g + 2);
        if (g + 1 < ngroups) compute(1, b1h, b1l);          // (uniform; only LDS reads and MFMAs inside)
        __syncthreads();
    }

    // ---- epilogue: PR row blocks at a time through this wave's LDS region, then whole row segments (bias / += / ReLU / mask)
    float* st = reinterpret_cast<float*>(lds) + wave * (PR * 16 * CW);
    int ncw = (tend < t0 + a.cpw ? tend : t0 + a.cpw) * 16;  // columns this wave stores: [16 t0, min(N, 16 min(tend, t0 + cpw)))
    if (ncw > a.N) ncw = a.N;
    ncw -= t0 * 16;
    if (ncw < 0) ncw = 0;
    const int c4n = ncw >> 2;
    const int col0 = t0 * 16;
    const int rr = lane >> 5, lc = lane & 31;
#pragma unroll
    for (int pass = 0; pass < RB / PR; ++pass) {
#pragma unroll
        for (int rbl = 0; rbl < PR; ++rbl)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    st[(rbl * 16 + 4 * (lane >> 4) + r) * CW + j * 16 + (lane & 15)] = acc[pass * PR + rbl][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lc < c4n) {
            const int col = col0 + 4 * lc;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) b4 = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll 4
            for (int r = rr; r < PR * 16; r += 2) {
                const int64_t row = row0 + pass * PR * 16 + r;
                if (row >= a.R) break;
                float4 v = *reinterpret_cast<const float4*>(st + r * CW + 4 * lc);
                float* p = C + row * a.ldc + col;
                v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                if (a.accumulate) { const float4 o = *reinterpret_cast<const float4*>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (a.mask) {
                    const float4 y = *reinterpret_cast<const float4*>(a.mask + row * a.ldm + col);
                    v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f; v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
                }
                if (FLID_ROWS_EXP == 3) { if (v.x == 12345.678f) *p = v.y; continue; }
                *reinterpret_cast<float4*>(p) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int RB>
bool launch_rows(const RowsArgs& a, int nbatch, int nsplit, hipStream_t s) {
    const int64_t gx = (a.R + 16 * RB - 1) / (16 * RB);
    if (gx >= ((int64_t)1 << 31) || nbatch > 65535) return false;
    const dim3 grid((unsigned)gx, (unsigned)nbatch, (unsigned)nsplit);
    if (a.cpw <= 3) gemm_rows_kernel<RB, 3, 2><<<grid, NTH, 0, s>>>(a);
    else if (a.cpw <= 5) gemm_rows_kernel<RB, 5, 2><<<grid, NTH, 0, s>>>(a);
    else return false;
    return true;
}

}  // namespace

namespace tg {

int64_t packed_floats(int N, int K) { return (int64_t)((N + 15) / 16) * ((K + 31) / 32) * 512; }

int pack_weights(int njobs, const tg_pack_job* jobs, hipStream_t s) {
    TG_REQUIRE(njobs >= 0 && njobs <= 24, "tg_pack_weights: at most 24 jobs per launch");
    if (njobs == 0) return TG_OK;
    PackJobs pj;
    const int total = tgs::pack_jobs_fill(pj, njobs, jobs);
    TG_REQUIRE(total >= 0, "tg_pack_weights: bad job");
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 3) / 4, 2048);
    pack_weights_kernel<<<blocks, NTH, 0, s>>>(pj);
    return launch_status("pack_weights_kernel");
}

// true = launched; false = shape / alignment not covered (the caller takes tg_gemm_f32 on the unpacked weights)
bool gemm_rows_nt(int64_t R, int N, int K, const float* A, int64_t lda, int64_t strideA, const void* packed, int64_t packed_stride_floats,
                  float* C, int64_t ldc, int64_t strideC, int nbatch, const float* bias, int relu, int accumulate, const float* mask,
                  int64_t ldm, hipStream_t s) {
    if (R < 1 || N < 4 || K < 4 || nbatch < 1) return false;
    if (N % 4 || K % 4 || lda % 4 || ldc % 4 || strideA % 4 || strideC % 4 || packed_stride_floats % 4) return false;
    if (!al16(A) || !al16(C) || !al16(packed) || (bias && !al16(bias)) || (mask && (!al16(mask) || ldm % 4))) return false;
    if (mask && nbatch != 1) return false;
    // a workgroup covers at most 20 column tiles (5 per wave): wider outputs are cut into column splits (blockIdx.z), each of which
    // re-reads its rows of A (N = 444 as 2 x 14 tiles)
    const int nt = (N + 15) / 16, nsplit = (nt + 19) / 20, tps = (nt + nsplit - 1) / nsplit, cpw = (tps + 3) / 4;
    if (nsplit > 64) return false;
    RowsArgs a{A, lda, strideA, reinterpret_cast<const uint4*>(packed), packed_stride_floats / 4, C, ldc, strideC, bias, mask, ldm,
               R, N, K, (K + 31) / 32, nt, tps, cpw, relu, accumulate};
    ProfScope prof("gemm", 2.0 * R * N * K * nbatch, s);
    // 64-row blocks once they fill the chip; below that, shorter blocks spread the rows over more CUs (each workgroup streams the whole
    // packed operand whatever its row count, so the product's time is one workgroup's)
    bool ok;
    const int64_t work = R * nbatch * nsplit;
    if (work >= 64 * 160) ok = launch_rows<4>(a, nbatch, nsplit, s);
    else if (work >= 32 * 160) ok = launch_rows<2>(a, nbatch, nsplit, s);
    else ok = launch_rows<1>(a, nbatch, nsplit, s);
    return ok;
}

}  // namespace tg

extern "C" int64_t tg_packed_floats(int N, int K) { return tg::packed_floats(N, K); }

extern "C" int tg_pack_weights(int njobs, const tg_pack_job* jobs, void* stream) { return tg::pack_weights(njobs, jobs, (hipStream_t)stream); }

extern "C" int tg_gemm_rows_nt(int64_t R, int N, int K, const float* d_A, int64_t lda, const void* d_packed, float* d_C, int64_t ldc,
                               const float* d_bias, int relu, int accumulate, const float* d_mask, int64_t ldm, void* stream) {
    TG_REQUIRE(d_A && d_packed && d_C, "tg_gemm_rows_nt: null operand");
    if (R == 0) return TG_OK;
    if (!tg::gemm_rows_nt(R, N, K, d_A, lda, 0, d_packed, 0, d_C, ldc, 0, 1, d_bias, relu, accumulate, d_mask, ldm, (hipStream_t)stream)) {
        tg::set_error("invalid argument: tg_gemm_rows_nt: N, K, lda, ldc multiples of 4, 16-byte aligned operands");
        return TG_EINVAL;
    }
    return tg::launch_status("gemm_rows_kernel");
}
