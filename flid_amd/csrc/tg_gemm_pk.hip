// Row-panel products against PRE-SPLIT weights:  C[M, N] = A[M, K] * W^T (+ bias), split-bf16 ("bf16x3", see tg_gemm_bf16x3.hip for the
// arithmetic), for TALL activations against SMALL weights -- the products of DyGFormer's transformer blocks (38 400 rows against
// 200 x 200 .. 800 x 200 weights) in the native step (tg_dyg.hip).
//
// replaces: aten::addmm behind the nn.Linear layers and nn.MultiheadAttention's projections of models/DyGFormer.py:418-461 and their
//           input gradients.
//
// Why a second product kernel.  The tile kernel of tg_gemm_bf16x3.hip spends a K = 200 product in four costs of about equal size that
// ADD instead of overlapping (prologue latency, LDS traffic of both operands, the C write, the MFMAs: DESIGN.md section 6).  Here
//   * the A operand never touches LDS: a wave owns 32 rows, loads its MFMA fragments of a whole K chunk (<= 208) straight into registers
//     and splits them ONCE (the tile kernel re-stages and re-splits a row panel for every 96 columns);
//   * the weight is split and laid out in fragment order once per step (pack32_kernel); a workgroup's four waves share its 32-column
//     tiles through a three-buffer LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU, no LDS store
//     instructions), one tile ahead of the MFMAs, one barrier per tile;
//   * a tile of C leaves straight from the accumulators: one store instruction covers two 128-byte row segments.
// Two forms.  K <= 208, any N (gemm_pk_s_kernel): the A fragments of the whole contraction live in registers (loaded from global memory in
// fragment shape, once per unit of column tiles), the weight's 32-column tiles (26 KiB) pass through a three-buffer ring.  K > 208,
// N <= 224 (gemm_pk_l_kernel): the accumulators of all column tiles live in registers and the contraction passes in 32-deep stages of
// both operands through three-buffer rings -- A as raw fp32 rows in full 128-byte lines (swizzled through the source address, split when
// the fragments are read), the weight's stage slice of all tiles (28 KiB).
// Measured at 38 400 rows (tools/pk_bench.py, profiles/r04_pk_bench.txt; tile kernel in brackets): x 800 x 200 72 us (92), x 600 x 200 64
// (78), x 200 x 200 31 (35.5); x 200 x 800 65 (82), x 200 x 600 51 (64), x 200 x 496 46 (57).  What bounds them: without the C stores the
// 800-column product takes 50 us, with them 72-80 -- the 123 MB it writes are the floor (~4 TB/s); the deep form takes 37 us without its
// MFMAs and 65 with them, on three different A paths (fragment-shaped loads to registers one stage ahead; the same by hand-counted inline-asm
// loads two stages ahead; LDS-DMA of full lines two stages ahead, issued in front of or behind the stage's MFMAs): the matrix pipe's 28 us ADD to the memory time although the generated
// code waits only for the stage it reads (vmcnt(10), no compiler-inserted vmcnt(0) in the loop) -- not understood; DESIGN.md section 9.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "tg_common.h"
#include "tg_split.h"

namespace {

using tgs::bf16x8;
using tgs::f32x16;
using tgs::split4;

#ifndef FLID_PK_EXP
#define FLID_PK_EXP 0
#endif
#ifndef FLID_PK_STAMPS
#define FLID_PK_STAMPS 0   // 1: wave 0 of workgroup 0 of the deep form records cycle counts per loop phase (tools/pk_stamps.py)
#endif
#if FLID_PK_STAMPS
__device__ unsigned long long g_pk_stamps[8];
#define PK_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); g_pk_stamps[k] += t_ - t_last; t_last = t_; } } while (0)
#else
#define PK_STAMP(k) do {} while (0)
#endif
constexpr int PKS = 13;                       // 16-deep steps of one K chunk
constexpr int KC = 16 * PKS;                  // 208
constexpr int BLK = 1024;                     // one (step, plane) fragment block: 64 lanes x 8 bf16
constexpr int STAGE = 2 * PKS * BLK;          // one 32-column tile x one K chunk, hi and lo planes: 26 KiB
constexpr int NBUF = 3;

// packed operand: [tile t][step s][plane hi, lo][lane][8 bf16]; lane l of (t, s) holds W[32 t + (l & 31)][16 s + 8 (l >> 5) + 0..7] -- the
// B fragment of v_mfma_f32_32x32x16_bf16 -- zero beyond N or K.  Up to 32 weights per launch, one wave per fragment pair.
// (deep form, job.K > 208: [stage s = k / 32][tile t][step 2][plane hi, lo][lane][8 bf16] -- one 32-deep stage of ALL tiles contiguous, what
// gemm_pk_l_kernel's LDS-DMA copies per iteration; fragment pairs are numbered (s, t, step))
struct Pack32Jobs { tg_pack32_job j[32]; int frag0[33]; int n; };
__global__ void __launch_bounds__(256) pack32_kernel(Pack32Jobs jobs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = jobs.frag0[jobs.n];
    for (int f = blockIdx.x * 4 + wave; f < total; f += gridDim.x * 4) {
        int ji = 0;
        while (ji + 1 < jobs.n && f >= jobs.frag0[ji + 1]) ++ji;
        const tg_pack32_job J = jobs.j[ji];
        const int fl = f - jobs.frag0[ji];
        int t, k16;                                               // 32-column tile, 16-deep step of the contraction
        if (J.K > KC) { const int nt = (J.N + 31) / 32; k16 = (fl / (2 * nt)) * 2 + (fl & 1); t = (fl >> 1) % nt; }
        else { k16 = fl % PKS; t = fl / PKS; }
        const int n = 32 * t + (lane & 31), k0 = 16 * k16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
            const bool ok = n < J.N && k < J.K;
            v[q] = ok ? (J.trans ? J.src[(int64_t)k * J.ld + n] : J.src[(int64_t)n * J.ld + k]) : 0.f;
        }
        uint2 h0, l0, h1, l1;
        split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
        split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
        uint4* d = reinterpret_cast<uint4*>(J.dst) + ((int64_t)fl * 2) * 64 + lane;
        d[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        d[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// stage (chunk c, tile t) of the packed operand -> LDS buffer `buf`: 26 blocks of 1 KiB, wave w takes blocks w, w + 4, ... (every wave
// issues the same number of copies; the surplus ones repeat block 25 with identical bytes)
__device__ __forceinline__ void issue_stage(const uint4* __restrict__ Bp, int64_t stage_index, char* lds, int buf, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int blk = wave + 4 * i;
        if (blk > 2 * PKS - 1) blk = 2 * PKS - 1;
        const uint4* g = Bp + (stage_index * (2 * PKS) + blk) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds + buf * STAGE + blk * BLK), 16, 0, 0);
    }
}

// this lane's A fragments: rows fixed, k = 16 s + 8 h + 0..7 for s = 0..12; loads unconditional (an offset past K
// reads the row's start instead and is zeroed), split once
__device__ __forceinline__ void load_a(const float* __restrict__ arow, int K, int h, bf16x8 (&ah)[PKS], bf16x8 (&al)[PKS]) {
    float4 x[PKS], y[PKS];
#pragma unroll
    for (int s = 0; s < PKS; ++s) {
        const int k = 16 * s + 8 * h;
        x[s] = *reinterpret_cast<const float4*>(arow + (k < K ? k : 0));
        y[s] = *reinterpret_cast<const float4*>(arow + (k + 4 < K ? k + 4 : 0));
    }
#pragma unroll
    for (int s = 0; s < PKS; ++s) {
        const int k = 16 * s + 8 * h;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 h0, l0, h1, l1;
        split4(k < K ? x[s] : z, h0, l0);
        split4(k + 4 < K ? y[s] : z, h1, l1);
        ah[s] = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        al[s] = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
    }
}

__device__ __forceinline__ void mma_stage(const char* __restrict__ stage, int lane, const bf16x8 (&ah)[PKS], const bf16x8 (&al)[PKS], f32x16& acc) {
    const char* b = stage + lane * 16;
    // the tile's 26 fragment reads in front of its 39 MFMAs (a read pair in front of every three MFMAs exposed one LDS round trip each)
    bf16x8 bh[PKS], bl[PKS];
#pragma unroll
    for (int s = 0; s < PKS; ++s) {
        bh[s] = *reinterpret_cast<const bf16x8*>(b + s * 2 * BLK);
        bl[s] = *reinterpret_cast<const bf16x8*>(b + s * 2 * BLK + BLK);
    }
#pragma unroll
    for (int s = 0; s < PKS; ++s) {
        // (one accumulation chain: three chains -- one per split term, summed at the end -- issue faster in isolation, 33 against 52 cycles
        // per MFMA in tools/micro/mfma_rate.hip, but their 32 extra registers cost the kernel its second workgroup per CU: 68 -> 113 us;
        // two chains, which fit: 72 -> 74 us -- the chain is not what this kernel waits for)
        // (the weight fragment as operand A: the tile comes out TRANSPOSED in the accumulators -- lane = row, four consecutive registers
        // = four consecutive output columns -- so it leaves as 16-byte stores with no shuffle at all: store_tile_t)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s], al[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[s], ah[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s], ah[s], acc, 0, 0, 0);
    }
}

// A tile as the MFMA leaves it with the weight fragment as operand A: lane (i = l & 31, h = l >> 5) holds row row0 + i, register r the
// column col0 + (r & 3) + 8 (r >> 2) + 4 h -- four consecutive registers are four consecutive columns.  vec: N % 4 == 0 and 16-byte
// aligned rows of C: one 16-byte store per register group.  (Round 4 computed the tile the other way round and transposed it inside
// each lane quad with DPP broadcasts + selects: ~120 vector instructions per tile, 10 per MFMA of the short form -- PMC, round 5.)
__device__ __forceinline__ void store_tile_t(const f32x16& acc, float* __restrict__ C, int64_t ldc, int64_t row0, int64_t M, int col0, int N,
                                             const float* __restrict__ bias, int lane, int vec) {
    const int64_t row = row0 + (lane & 31);
    if (row >= M) return;
    float* crow = C + row * ldc;
    const int h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = col0 + 8 * g + 4 * h;
        if (vec) {
            if (col < N) {
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias) b4 = *reinterpret_cast<const float4*>(bias + col);
                *reinterpret_cast<float4*>(crow + col) = make_float4(acc[4 * g] + b4.x, acc[4 * g + 1] + b4.y, acc[4 * g + 2] + b4.z, acc[4 * g + 3] + b4.w);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col + q < N) crow[col + q] = acc[4 * g + q] + (bias ? bias[col + q] : 0.f);
        }
    }
}

// K <= 208: grid (row panels of 128, column units of `tpu` tiles)
__global__ void __launch_bounds__(256) gemm_pk_s_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K, const uint4* __restrict__ Bp, int N,
                                                        int ntiles, int tpu, float* __restrict__ C, int64_t ldc, const float* __restrict__ bias, int vec) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int t0 = blockIdx.y * tpu, t1 = t0 + tpu < ntiles ? t0 + tpu : ntiles;
    issue_stage(Bp, t0, lds, 0, wave, lane);
    if (t0 + 1 < t1) issue_stage(Bp, t0 + 1, lds, 1, wave, lane);
    int64_t row = row0 + (lane & 31);
    if (row > M - 1) row = M - 1;
    bf16x8 ah[PKS], al[PKS];
    load_a(A + row * lda, K, h, ah, al);
    // One barrier per tile, and no wait for the C stores: tile t's stores are issued one iteration late (behind the barrier of tile
    // t + 1), so the queue of outstanding vector-memory operations at the top of iteration u is, oldest first,
    //     [DMA of tile u + 1] [stores of tile u - 2 ... long done] [stores of tile u - 1] [DMA of tile u + 2 -- not yet issued]
    // i.e. at the wait: DMA(u) (7, oldest), stores(u - 2) (4), DMA(u + 1) (7).  vmcnt(7) lets 7 stay outstanding: loads retire in
    // order among themselves, so whichever way loads and stores interleave, 11 retirements include all 7 of DMA(u).
    f32x16 prev;
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // tile t has landed for every wave; every wave is done reading tile t - 1
        asm volatile("" ::: "memory");
        if (t > t0) {
            store_tile_t(prev, C, ldc, row0, M, 32 * (t - 1), N, bias, lane, vec);
        }
        if (t + 2 < t1) issue_stage(Bp, t + 2, lds, (t + 2 - t0) % NBUF, wave, lane);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_stage(lds + ((t - t0) % NBUF) * STAGE, lane, ah, al, acc);
        prev = acc;
    }
    store_tile_t(prev, C, ldc, row0, M, 32 * (t1 - 1), N, bias, lane, vec);
}

// ---- deep contraction (K > 208), narrow output (N <= 224): the accumulators of all (up to 7) column tiles stay in registers while the
// contraction passes in 32-deep stages; a stage = 32 k of this wave's 32 rows of A (raw fp32, 4 KiB, split when the fragments are read) +
// the 32-deep slice of ALL tiles of the pre-split weight (NT x 4 KiB), both by LDS-DMA into three-buffer rings.
constexpr int LSTEP = 4 * BLK;                // one tile's share of a stage: 2 steps x (hi, lo)

// LDS-DMA of one stage of the deep form.  The per-lane parts of every copy's address are 32-bit byte offsets computed ONCE (voff_b: this
// wave's blocks of the stage slice -- NT x 4 blocks of 1 KiB dealt over NWV waves, every wave the same number of copies, the surplus ones
// repeating the last block with identical bytes so that the counted waits hold for every wave; voff_a: row and swizzled chunk of each of the
// four A copies); a stage then only moves two wave-uniform base pointers (scalar ALU) -- computed per copy, the 64-bit address arithmetic
// was ~100 vector instructions per stage in front of the MFMAs, on SIMDs that carry one or two waves.
// A: 32 rows x 128 bytes of this wave's rows = four copies of 1 KiB, each lane one 16-byte chunk of a full 128-byte line
// (fragment-shaped loads straight to registers -- 64 scattered 16-byte pieces per instruction -- are address-processing bound).  The LDS
// image is row-major with the eight 16-byte slots of a row permuted through the SOURCE address (slot p of row r holds chunk p ^ (r & 7)),
// so that the fragment reads below are conflict-free.  In the last stage a chunk past K comes from a block of zeros (TAIL).
template <int NT, int NWV>
__device__ __forceinline__ void issue_stage_l(const char* __restrict__ bstage, const uint32_t (&voff_b)[(NT * 4 + NWV - 1) / NWV], char* ldsb, int wave) {
    constexpr int PER = (NT * 4 + NWV - 1) / NWV;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int blk = wave + NWV * i;
        if (blk > NT * 4 - 1) blk = NT * 4 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bstage + voff_b[i]),
                                         (__attribute__((address_space(3))) void*)(ldsb + blk * BLK), 16, 0, 0);
    }
}
template <bool TAIL>
__device__ __forceinline__ void issue_a_stage(const char* __restrict__ astage, const uint32_t (&voff_a)[4], int kleft, int chunk, char* region,
                                              const float* __restrict__ zeros) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const char* g = astage + voff_a[j];
        if (TAIL && 4 * chunk >= kleft) g = reinterpret_cast<const char*>(zeros);      // (chunk = this lane's swizzled chunk of row 8 j + ..: see the caller)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(region + j * BLK), 16, 0, 0);
    }
}
// lane (r = l & 31, h = l >> 5), step st: k = 16 st + 8 h + 0..7 = chunks 4 st + 2 h and + 1 of row r
__device__ __forceinline__ void read_a_stage(const char* __restrict__ region, int lane, bf16x8 (&ah)[2], bf16x8 (&al)[2]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int c0 = 4 * st + 2 * h;
        const float4 x = *reinterpret_cast<const float4*>(region + r * 128 + ((c0 ^ (r & 7)) << 4));
        const float4 y = *reinterpret_cast<const float4*>(region + r * 128 + (((c0 + 1) ^ (r & 7)) << 4));
        uint2 h0, l0, h1, l1;
        split4(x, h0, l0);
        split4(y, h1, l1);
        ah[st] = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        al[st] = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
    }
}

// NWV waves x 32 rows per workgroup (the launcher picks NWV so that the row panels fit the chip in one round: one workgroup per CU);
// three LDS buffers in rotation for both operands: stage i + 2 is issued at the top of stage i, the waits are counted by hand
template <int NT, int NWV>
__global__ void __launch_bounds__(64 * NWV, 1) gemm_pk_l_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K, const uint4* __restrict__ Bp,
                                                                int N, int nst, float* __restrict__ C, int64_t ldc, const float* __restrict__ bias, int vec,
                                                                const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 3 x NT x 4 KiB of B, then NWV x 3 x 4 KiB of A
    constexpr int PER = (NT * 4 + NWV - 1) / NWV;
    constexpr int OPS = PER + 4;                                    // vector-memory operations a wave issues per stage
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // (scalar: LDS destinations and block numbers stay on the scalar ALU)
    const int64_t row0 = (int64_t)blockIdx.x * (32 * NWV) + wave * 32;
    char* aring = lds + 3 * NT * LSTEP + wave * (3 * 4 * BLK);
    // per-lane address parts, once
    uint32_t voff_b[PER], voff_a[4];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int blk = wave + NWV * i;
        if (blk > NT * 4 - 1) blk = NT * 4 - 1;
        voff_b[i] = (uint32_t)((blk * 64 + lane) * 16);
    }
    // A copy j covers rows 8 j .. 8 j + 7 of the wave's 32; every one of them has (row & 7) = lane >> 3, so the swizzled chunk is the same
    // for the four copies
    const int chunk = (lane & 7) ^ ((lane >> 3) & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t row = row0 + 8 * j + (lane >> 3);
        if (row > M - 1) row = M - 1;
        voff_a[j] = (uint32_t)((row * lda + 4 * chunk) * 4);        // (the launcher checks M lda < 2^30)
    }
    const char* a0 = reinterpret_cast<const char*>(A);
    const char* b0 = reinterpret_cast<const char*>(Bp);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool ktail = (K & 31) != 0;                               // only the last stage can hold chunks past K
    auto issue = [&](int st) {
        issue_stage_l<NT, NWV>(b0 + (int64_t)st * (NT * LSTEP), voff_b, lds + (st % 3) * (NT * LSTEP), wave);
        char* region = aring + (st % 3) * (4 * BLK);
        if (ktail && st == nst - 1) issue_a_stage<true>(a0 + (int64_t)st * 128, voff_a, K - 32 * st, chunk, region, zeros);
        else issue_a_stage<false>(a0 + (int64_t)st * 128, voff_a, 32, chunk, region, zeros);
    };
    issue(0);
    if (nst > 1) issue(1);
#if FLID_PK_STAMPS
    unsigned long long t_last = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int q = 0; q < 8; ++q) g_pk_stamps[q] = 0;
#endif
    for (int i = 0; i < nst; ++i) {
        PK_STAMP(0);
        if (i + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(OPS) : "memory");      // everything but stage i + 1's copies has landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PK_STAMP(1);
        __builtin_amdgcn_s_barrier();                      // stage i has landed for every wave; every wave is done reading stage i - 1
        asm volatile("" ::: "memory");
        PK_STAMP(2);
        // (the A fragment reads in front of the copies' issue: behind 11-14 LDS-DMA instructions they took ~1 000 cycles per stage)
        bf16x8 ah[2], al[2];
        read_a_stage(aring + (i % 3) * (4 * BLK), lane, ah, al);
#if FLID_PK_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        PK_STAMP(3);
        if (i + 2 < nst) issue(i + 2);
        PK_STAMP(4);
        const char* b = lds + (i % 3) * (NT * LSTEP) + lane * 16;
        // all fragment reads of a step in front of its MFMAs (read-then-multiply per tile left one LDS round trip exposed in front of
        // every three MFMAs -- 14 per stage, with one wave per SIMD and nothing to hide them: the loop ran at a third of the matrix rate
        // with its copies switched off)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 bh[NT], bl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (FLID_PK_EXP == 10) { bh[t] = al[st]; bl[t] = ah[st]; continue; }     // (timing experiment: no B fragment reads)
                bh[t] = *reinterpret_cast<const bf16x8*>(b + t * LSTEP + st * 2 * BLK);
                bl[t] = *reinterpret_cast<const bf16x8*>(b + t * LSTEP + st * 2 * BLK + BLK);
            }
            // term-major: consecutive MFMAs go to different accumulators
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[t], al[st], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[t], ah[st], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[t], ah[st], acc[t], 0, 0, 0);
        }
#if FLID_PK_STAMPS
        asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[NT - 1][15]) : "memory");      // (the MFMAs' results are needed here)
#endif
        PK_STAMP(5);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        store_tile_t(acc[t], C, ldc, row0, M, 32 * t, N, bias, lane, vec);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace tg {

int64_t packed32_floats(int N, int K) {
    const int nt = (N + 31) / 32;
    if (K <= KC) return (int64_t)nt * (STAGE / 4);
    return nt <= 7 ? (int64_t)((K + 31) / 32) * nt * (LSTEP / 4) : -1;          // deep form: N <= 224
}

int pack32_weights(int njobs, const tg_pack32_job* jobs, hipStream_t s) {
    TG_REQUIRE(njobs >= 0 && njobs <= 32 && (njobs == 0 || jobs), "tg_pack32_weights: at most 32 jobs per launch");
    if (njobs == 0) return TG_OK;
    Pack32Jobs pj;
    pj.n = njobs;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        TG_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].N > 0 && jobs[i].K > 0 && al16(jobs[i].dst) && packed32_floats(jobs[i].N, jobs[i].K) > 0,
                   "tg_pack32_weights: bad job (K <= 208, or N <= 224)");
        pj.j[i] = jobs[i];
        pj.frag0[i] = total;
        const int nt = (jobs[i].N + 31) / 32;
        total += jobs[i].K <= KC ? nt * PKS : ((jobs[i].K + 31) / 32) * nt * 2;
    }
    pj.frag0[njobs] = total;
    pack32_kernel<<<(unsigned)std::min((total + 3) / 4, 2048), 256, 0, s>>>(pj);
    return launch_status("pack32_kernel");
}

// true = launched; false = shape / alignment not covered (the caller takes tg_gemm_f32 on the unpacked weight)
bool gemm_pk_nt(int64_t M, int N, int K, const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, const float* bias, hipStream_t s) {
    if (M < 1 || N < 1 || K < 4 || K % 4 || lda % 4 || !al16(A) || !al16(packed)) return false;
    const int ntiles = (N + 31) / 32;
    if (K > KC && ntiles > 7) return false;
    const int64_t panels = (M + 127) / 128;
    if (panels >= ((int64_t)1 << 31)) return false;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)gemm_pk_s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * STAGE) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_done = true;
    }
    const int vec = N % 4 == 0 && ldc % 4 == 0 && al16(C) && (!bias || al16(bias));
    ProfScope prof("gemm", 2.0 * M * N * K, s);
    if (K > KC) {
        if (M * lda >= ((int64_t)1 << 30)) return false;   // (32-bit byte offsets of the A copies)
        const int nst = (K + 31) / 32;
        const uint4* Bp = reinterpret_cast<const uint4*>(packed);
        const float* zeros = zero_block();                  // chunks past K are copied from it
        if (!zeros) return false;
        // waves (32 rows each) per workgroup: the fewest of 4 / 5 / 6 whose row panels fit the chip's 256 CUs in ONE round (one
        // workgroup per CU: 84 KiB of LDS for the weight's ring + 12 KiB per wave for its rows), else 6
        int nwv = 4;
        while (nwv < 6 && (M + 32 * nwv - 1) / (32 * nwv) > 256) ++nwv;
        const unsigned grid = (unsigned)((M + 32 * nwv - 1) / (32 * nwv));
#define PK_LW(NTV, NWVV) do { \
            static bool attr_l = false;      /* (one per instantiation: every expansion has its own) */ \
            if (!attr_l) { \
                if (hipFuncSetAttribute((const void*)gemm_pk_l_kernel<NTV, NWVV>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * NTV * LSTEP + NWVV * 12 * BLK) != hipSuccess) { \
                    (void)hipGetLastError(); return false; } \
                attr_l = true; } \
            gemm_pk_l_kernel<NTV, NWVV><<<grid, 64 * NWVV, 3 * NTV * LSTEP + NWVV * 12 * BLK, s>>>(A, lda, M, K, Bp, N, nst, C, ldc, bias, vec, zeros); } while (0)
#define PK_L(NTV) do { if (nwv == 4) PK_LW(NTV, 4); else if (nwv == 5) PK_LW(NTV, 5); else PK_LW(NTV, 6); } while (0)
        switch (ntiles) {
            case 1: PK_L(1); break; case 2: PK_L(2); break; case 3: PK_L(3); break; case 4: PK_L(4); break;
            case 5: PK_L(5); break; case 6: PK_L(6); break; default: PK_L(7); break;
        }
#undef PK_L
#undef PK_LW
        return true;
    }
    // units per panel u: rounds of the grid over the chip's 512 workgroup slots x (tiles per unit + the prologue, which costs about
    // six tiles' time: A rows from HBM, 26 splits per lane)
    int tpu = ntiles;
    int64_t best = -1;
    static const int force_u = (getenv("FLID_GEMM_TUNE") && getenv("FLID_PK_UNITS")) ? atoi(getenv("FLID_PK_UNITS")) : 0;
    for (int u = 1; u <= ntiles; ++u) {
        const int per = (ntiles + u - 1) / u;
        const int64_t cost = ((panels * ((ntiles + per - 1) / per) + 511) / 512) * (per + 6);
        if (force_u ? u == force_u : (best < 0 || cost < best)) { best = cost; tpu = per; }
    }
    const dim3 grid((unsigned)panels, (unsigned)((ntiles + tpu - 1) / tpu));
    gemm_pk_s_kernel<<<grid, 256, NBUF * STAGE, s>>>(A, lda, M, K, reinterpret_cast<const uint4*>(packed), N, ntiles, tpu, C, ldc, bias, vec);
    return true;
}

}  // namespace tg

#if FLID_PK_STAMPS
extern "C" int tg_pk_stamps_read(unsigned long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pk_stamps), 64) == hipSuccess ? 0 : -2; }
#endif
extern "C" int64_t tg_packed32_floats(int N, int K) { return tg::packed32_floats(N, K); }

extern "C" int tg_pack32_weights(int njobs, const tg_pack32_job* jobs, void* stream) { return tg::pack32_weights(njobs, jobs, (hipStream_t)stream); }

extern "C" int tg_gemm_pk_nt(int64_t M, int N, int K, const float* d_A, int64_t lda, const void* d_packed, float* d_C, int64_t ldc, const float* d_bias,
                             void* stream) {
    TG_REQUIRE(d_A && d_packed && d_C && ldc >= N, "tg_gemm_pk_nt: arguments");
    if (M == 0) return TG_OK;
    if (!tg::gemm_pk_nt(M, N, K, d_A, lda, d_packed, d_C, ldc, d_bias, (hipStream_t)stream)) {
        tg::set_error("tg_gemm_pk_nt: K <= 208 or N <= 224, K and lda multiples of 4, 16-byte aligned operands");
        return TG_ESHAPE;
    }
    return tg::launch_status("gemm_pk_s_kernel");
}
