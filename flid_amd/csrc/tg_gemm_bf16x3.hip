// Split-bf16 ("bf16x3") GEMM for the row-times-weight products of the main chain:  C[M,N] = A[M,K] * B[N,K]^T (+bias)(+C)(ReLU).
//
// Each fp32 operand element is split as x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); the product keeps
//     A_hi B_hi + A_hi B_lo + A_lo B_hi        (the dropped A_lo B_lo term and the residual x - hi - lo are <= 2^-17 relative)
// with fp32 accumulation inside v_mfma_f32_32x32x16_bf16.  Three bf16 MFMAs replace eight f32-input MFMAs per 16-deep K step
// (96 vs 512 cycles per 32x32 tile): on MI355X the f32-input matrix rate (157 TFLOP/s) was the bound of the whole path
// (SURVEY.md 8d "fp32 MFMA rate"), this lifts it by 5x for the products whose operands are k-contiguous.
// Measured accuracy on the full-dimension golden case: |emb - reference| = 1.6e-5 (exact-fp32 path: 7e-7; budget 1e-4).
//
// Workgroup = 4 waves stacked along M, block tile 128 x 96 (or 128 x 64); each wave owns 32 x 96 (three 32x32 tiles sharing the
// A fragment).
// Staging: global fp32 (full 128-B lines) -> split in registers -> LDS row = [32 x bf16 hi | 32 x bf16 lo] (+16 B pad, stride
// 144 B, conflict-free ds_read_b128) -> fragments.  Two LDS stages, one barrier per 32-deep stage, global loads of stage s+2
// in flight under stage s.
#include <math.h>
#include <stdlib.h>

#include "tg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;
constexpr int ROW_BYTES = 144;                 // 64 B hi + 64 B lo + 16 B pad
#ifndef FLID_BF_BM
#define FLID_BF_BM 128
#endif
constexpr int BM = FLID_BF_BM, NT = 2 * FLID_BF_BM;      // one wave per 32 rows of the block tile
#ifndef FLID_NT_SCHED
#define FLID_NT_SCHED 1
#endif
constexpr bool SCHED = FLID_NT_SCHED != 0;
#ifndef FLID_NT_EXP
#define FLID_NT_EXP 0   // timing experiments only (results wrong): 1 no steady-state global loads, 2 no MFMAs, 3 no split VALU
#endif
#ifndef FLID_NT_VPM
#define FLID_NT_VPM 6
#endif

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);     // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xFFFF0000u); }

// split 4 floats into 4 hi + 4 lo bf16 (two 8-byte words)
__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
    hi.x = pack_bf16(v.x, v.y);
    hi.y = pack_bf16(v.z, v.w);
    lo.x = pack_bf16(v.x - bf16_lo_to_f32(hi.x), v.y - bf16_hi_to_f32(hi.x));
    lo.y = pack_bf16(v.z - bf16_lo_to_f32(hi.y), v.w - bf16_hi_to_f32(hi.y));
}

// Loads past the end of K read this instead of being branched around: every issue() is then a fixed number of loads on every
// path, so the compiler can wait with s_waitcnt vmcnt(N) for exactly the older register set.  With guarded loads it fell back to
// vmcnt(0) before each LDS store, which also waited for the loads issued one stage ago -- the look-ahead was one stage, not two.
// (a 16-byte zero block in global memory, handed to the kernel as an ordinary pointer so that the select stays a global load)

// one panel of R rows x 32 k (k contiguous in memory); slot = (row, 4-float chunk)
template <int R>
struct Panel {
    static constexpr int TOTAL = R * 8;
    static constexpr int PER = TOTAL / NT;
    static_assert(TOTAL % NT == 0, "panel must tile the workgroup");
    const float* src[PER];
    int lds_off[PER];      // byte offset of the slot's hi word inside the panel
    int kcol[PER];

    __device__ __forceinline__ void init(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows, int64_t kbeg) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = threadIdx.x + j * NT;
            const int r = idx >> 3, c = idx & 7;
            int64_t row = row0 + r;
            if (row > nrows - 1) row = nrows - 1;           // clamped rows only feed outputs that are never stored
            src[j] = X + row * ld + kbeg + c * 4;
            lds_off[j] = r * ROW_BYTES + c * 8;
            kcol[j] = c * 4;
        }
    }
    // stage at float offset `elems` (= k0); `ok` = the stage exists at all (uniform)
    __device__ __forceinline__ void gload(int64_t elems, int64_t k0, int64_t kend, bool ok, const float* __restrict__ zeros,
                                          float4 (&reg)[PER]) const {
        const int64_t lim = ok ? kend : 0;           // (a scalar select, not a branch: the stage body must stay one basic block)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const float* p = (k0 + kcol[j] < lim) ? src[j] + elems : zeros;
            reg[j] = *reinterpret_cast<const float4*>(p);
        }
    }
    __device__ __forceinline__ void sstore(char* __restrict__ s, const float4 (&reg)[PER]) const {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            uint2 hi, lo;
            if (FLID_NT_EXP == 3) {
                hi = make_uint2(__builtin_bit_cast(uint32_t, reg[j].x), __builtin_bit_cast(uint32_t, reg[j].y));
                lo = make_uint2(__builtin_bit_cast(uint32_t, reg[j].z), __builtin_bit_cast(uint32_t, reg[j].w));
            } else {
                split4(reg[j], hi, lo);
            }
            *reinterpret_cast<uint2*>(s + lds_off[j]) = hi;
            *reinterpret_cast<uint2*>(s + lds_off[j] + 64) = lo;
        }
    }
};

// fragment of one 32-row tile for k-step ks (16 k): lane (r = l&31, h = l>>5) holds k = 16 ks + 8 h + 0..7
__device__ __forceinline__ void read_frag(const char* __restrict__ s, int tile_r0, int ks, bf16x8& hi, bf16x8& lo) {
    const int lane = threadIdx.x & 63;
    const char* p = s + (tile_r0 + (lane & 31)) * ROW_BYTES + (16 * ks + 8 * (lane >> 5)) * 2;
    hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
    lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p + 64));
}

// TNW = 32-column tiles per wave (block tile 128 x 32*TNW): 3 fits N = 272 / 172 / 444 (-> 288 / 192 / 480), 2 fits N <= 64k+..
template <int TNW>
__global__ void __launch_bounds__(NT) gemm_bf16x3_nt_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
        const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc, const float* __restrict__ bias, int relu,
        int accumulate, int gx, int gy, int64_t strideA, int64_t strideB, int64_t strideC, const float* __restrict__ zeros, int vec_c,
        const float* __restrict__ mask, int64_t ldm, int64_t K2, int64_t lda2, int64_t ldb2, const float* __restrict__ bias2) {
    constexpr int BNt = 32 * TNW;
    constexpr int FA = BM * ROW_BYTES, FB = BNt * ROW_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[2 * (FA + FB)];
    auto sA = [&](int i) -> char* { return lds + i * (FA + FB); };
    auto sB = [&](int i) -> char* { return lds + i * (FA + FB) + FA; };

    const int nwg = gx * gy, bid = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    const int by = swz / gx, bx = swz % gx;
    const int batch = blockIdx.z;
    A += batch * strideA; B += batch * strideB; C += batch * strideC;
    // K2 > 0: a PAIR of products with the same M and N but their own contraction depth, leading dimensions and bias (batch 1 takes the
    // second set): the two GRU gate products of a TGN step as one launch (gemm_bf16x3_nt_pair)
    if (K2 > 0) {
        if (batch == 1) { K = K2; lda = lda2; ldb = ldb2; bias = bias2; }
    } else if (bias) bias += batch * (int64_t)N;

    const int64_t bm = (int64_t)by * BM, bn = (int64_t)bx * BNt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bool live[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) live[t] = (bm + wave * 32 < M) && (bn + 32 * t < N);

    f32x16 acc[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    Panel<BM> pa;
    Panel<BNt> pb;
    pa.init(A, lda, bm, M, 0);
    pb.init(B, ldb, bn, N, 0);
    // two register sets: the loads of stage s+2 and s+3 are in flight while stage s computes (the stages are short -- 12..18
    // bf16 MFMAs -- so one stage of look-ahead does not cover an L2 round trip)
    float4 ra0[Panel<BM>::PER], rb0[Panel<BNt>::PER], ra1[Panel<BM>::PER], rb1[Panel<BNt>::PER];
    const int64_t nstage = (K + BK - 1) / BK;
    auto issue = [&](int64_t st, float4 (&ra)[Panel<BM>::PER], float4 (&rb)[Panel<BNt>::PER]) {
        pa.gload(st * BK, st * BK, K, st < nstage, zeros, ra);   // unconditional: a stage past the end loads zeros
        pb.gload(st * BK, st * BK, K, st < nstage, zeros, rb);
    };
    auto stage = [&](int64_t st, float4 (&ra)[Panel<BM>::PER], float4 (&rb)[Panel<BNt>::PER]) {
        // on entry LDS[st & 1] holds stage st and (ra, rb) hold stage st + 1
        const int cur = (int)(st & 1);
        bf16x8 ah[2], al[2], bh[TNW][2], bl[TNW][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            read_frag(sA(cur), wave * 32, ks, ah[ks], al[ks]);
#pragma unroll
            for (int t = 0; t < TNW; ++t) read_frag(sB(cur), 32 * t, ks, bh[t][ks], bl[t][ks]);
        }
        // One basic block per stage (no branches: the store after the last stage writes zeros into the idle buffer, tiles wholly
        // outside C multiply clamped rows), so that the scheduler can be told to run the next stage's split (VALU) and LDS
        // stores in the shadow of this stage's MFMAs instead of before them: a wave per SIMD has nobody else to overlap with.
        pa.sstore(sA(cur ^ 1), ra);
        pb.sstore(sB(cur ^ 1), rb);
        if (FLID_NT_EXP != 1) issue(st + 3, ra, rb);  // this register set is free again
#pragma unroll
        for (int ks = 0; ks < (FLID_NT_EXP == 2 ? 0 : 2); ++ks) {
#pragma unroll
            for (int t = 0; t < TNW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh[t][ks], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TNW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl[t][ks], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TNW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh[t][ks], acc[t], 0, 0, 0);
        }
        if (SCHED) {
            __builtin_amdgcn_sched_group_barrier(0x100, 8 + 4 * TNW - 4, 0);      // the fragment reads first
#pragma unroll
            for (int i = 0; i < 6 * TNW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // one MFMA ...
                __builtin_amdgcn_sched_group_barrier(0x002, FLID_NT_VPM, 0);      // ... six split / address instructions under it
                if (i % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // an LDS store
                if (i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // a global load
            }
        }
        __syncthreads();
    };
    if (nstage > 0) {
        issue(0, ra0, rb0);
        pa.sstore(sA(0), ra0);
        pb.sstore(sB(0), rb0);
        issue(1, ra0, rb0);
        issue(2, ra1, rb1);
        __syncthreads();
    }
    for (int64_t st = 0; st < nstage; st += 2) {
        stage(st, ra0, rb0);
        if (st + 1 < nstage) stage(st + 1, ra1, rb1);
    }

    // Epilogue through LDS: the MFMA C/D layout (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) would store
    // 4-byte elements, two 128-byte pieces per instruction, from ~1000 unrolled instructions; a one-stage launch spent most of
    // its 14 us there.  Each wave parks its 32 x BNt tile in its own LDS region and a short rolled loop writes whole
    // 16-byte-aligned row segments (bias / accumulate / ReLU applied on the way).
    __syncthreads();                                    // all waves are done reading the operand stages
    constexpr int CS = BNt + 8;                         // row stride in floats: 4 rows further = 32 banks further
    static_assert((NT / 64) * 32 * CS * 4 <= 2 * (FA + FB), "C staging must fit the operand stages");
    float* cs = reinterpret_cast<float*>(lds) + wave * 32 * CS;
    {
        const int rl = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int t = 0; t < TNW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * CS + t * 32 + rl] = acc[t][r];
    }
    __builtin_amdgcn_wave_barrier();                    // same wave reads back (LDS operations of a wave complete in order)
    constexpr int C4 = BNt / 4;
    const int64_t row0 = bm + wave * 32;
#pragma unroll 2
    for (int idx = lane; idx < 32 * C4; idx += 64) {
        const int r = idx / C4, c4 = idx - r * C4;
        const int64_t row = row0 + r, col = bn + c4 * 4;
        if (row >= M || col >= N) continue;
        float4 v = *reinterpret_cast<const float4*>(cs + r * CS + c4 * 4);
        float* p = C + row * ldc + col;
        if (vec_c) {                                    // N % 4 == 0, 16-byte aligned rows: the whole chunk is inside C
            if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
            if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (mask) {                               // ReLU backward fused: keep the entries whose forward output was positive
                const float4 y = *reinterpret_cast<const float4*>(mask + row * ldm + col);
                v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f; v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4*>(p) = v;
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (col + q >= N) break;
                float x = e[q] + (bias ? bias[col + q] : 0.f);
                if (accumulate) x += p[q];
                if (relu) x = fmaxf(x, 0.f);
                if (mask && !(mask[row * ldm + col + q] > 0.f)) x = 0.f;
                p[q] = x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight-gradient form  C[M,N] (+)= A^T B  with A stored (K x M) and B stored (K x N), both row-major (the contraction index is
// the ROW index of both operands: dW = dY^T X).  Each thread loads a 4(k) x 4(row) micro-tile with four coalesced 16-byte loads,
// transposes it in registers and writes 4 consecutive k per output row as one 8-byte word -- the LDS image is the same
// [row][32 x bf16 hi | 32 x bf16 lo] as above, so the fragment reads and the MFMA schedule are shared.  The contraction is
// split over blockIdx.z and folded with float atomics (C zeroed by the launcher).
template <int R>
struct PanelT {
    static constexpr int TILES = (R / 4) * 8;               // 4x4 micro-tiles: R/4 row groups x 8 k groups
    const float* src;          // micro-tile origin at k = 0
    int lds_off;               // byte offset of (row group, k group)
    int kcol;

    // threads beyond TILES redo slot (t mod TILES): same loads, same bytes into LDS -- no inactive-thread branches in the stage
    __device__ __forceinline__ void init(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows) {
        const int t = threadIdx.x % TILES;
        // consecutive threads -> consecutive k groups: their LDS stores (8 B each, same rows) fall into consecutive banks; with
        // consecutive ROWS per lane the 576-byte row-group stride put 8 lanes on every bank pair.  Global side: 8 k rows x 128 B
        // contiguous per instruction (whole cache lines either way).
        const int kg = t % 8, rg = t / 8;
        int64_t row = row0 + rg * 4;
        if (row > nrows - 4) row = nrows - 4;                   // nrows % 4 == 0 (checked by the launcher)
        if (row < 0) row = 0;
        src = X + (int64_t)(kg * 4) * ld + row;
        lds_off = (rg * 4) * ROW_BYTES + kg * 8;
        kcol = kg * 4;
    }
    // the four k rows of the micro-tile at stage offset k0; rows past kend (or a stage that does not exist) read the zero block
    __device__ __forceinline__ void gload(int64_t ld, int64_t k0, int64_t kend, bool ok, const float* __restrict__ zeros, float4 (&reg)[4]) const {
        const int64_t lim = ok ? kend : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* p = (k0 + kcol + i < lim) ? src + (k0 + i) * ld : zeros;
            reg[i] = *reinterpret_cast<const float4*>(p);
        }
    }
    __device__ __forceinline__ void sstore(char* __restrict__ s, const float4 (&reg)[4]) const {
        const float c0[4] = {reg[0].x, reg[1].x, reg[2].x, reg[3].x};   // row +0: k = 0..3
        const float c1[4] = {reg[0].y, reg[1].y, reg[2].y, reg[3].y};
        const float c2[4] = {reg[0].z, reg[1].z, reg[2].z, reg[3].z};
        const float c3[4] = {reg[0].w, reg[1].w, reg[2].w, reg[3].w};
        const float* cols[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint2 hi, lo;
            split4(make_float4(cols[r][0], cols[r][1], cols[r][2], cols[r][3]), hi, lo);
            *reinterpret_cast<uint2*>(s + lds_off + r * ROW_BYTES) = hi;
            *reinterpret_cast<uint2*>(s + lds_off + r * ROW_BYTES + 64) = lo;
        }
    }
};

// 1-D grid of tiles x slices (slice = batch * nsplit + split, a multiple of 8 of them): every tile of one K slice runs on the same
// XCD (hardware deals workgroups round-robin), as in the f32-input kernel.  Each workgroup leaves its partial tile in
// ws[slice][M][N] as whole 16-byte chunks; tg_gemm.hip folds the slices in fixed order.
template <int TNW>
__global__ void __launch_bounds__(NT) gemm_bf16x3_tn_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
        const float* __restrict__ B, int64_t ldb, float* __restrict__ ws, int gx, int ntiles, int64_t k_chunk, int nsplit,
        int64_t strideA, int64_t strideB, const float* __restrict__ zeros) {
    constexpr int BNt = 32 * TNW;
    constexpr int FA = BM * ROW_BYTES, FB = BNt * ROW_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[2 * (FA + FB)];
    auto sA = [&](int i) -> char* { return lds + i * (FA + FB); };
    auto sB = [&](int i) -> char* { return lds + i * (FA + FB) + FA; };
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int slice = xcd + 8 * (jj / ntiles), tile = jj % ntiles;
    const int by = tile / gx, bx = tile % gx;
    const int batch = slice / nsplit, split = slice % nsplit;
    A += batch * strideA; B += batch * strideB;
    const int64_t bm = (int64_t)by * BM, bn = (int64_t)bx * BNt;
    const int64_t kbeg = (int64_t)split * k_chunk, kend = (kbeg + k_chunk < K) ? kbeg + k_chunk : K;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    PanelT<BM> pa;
    PanelT<BNt> pb;
    pa.init(A, lda, bm, M);
    pb.init(B, ldb, bn, N);
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
    const int64_t nstage = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    auto issue = [&](int64_t st, float4 (&ra)[4], float4 (&rb)[4]) {
        pa.gload(lda, kbeg + st * BK, kend, st < nstage, zeros, ra);
        pb.gload(ldb, kbeg + st * BK, kend, st < nstage, zeros, rb);
    };
    auto stage = [&](int64_t st, float4 (&ra)[4], float4 (&rb)[4]) {
        const int cur = (int)(st & 1);
        bf16x8 ah[2], al[2], bh[TNW][2], bl[TNW][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            read_frag(sA(cur), wave * 32, ks, ah[ks], al[ks]);
#pragma unroll
            for (int t = 0; t < TNW; ++t) read_frag(sB(cur), 32 * t, ks, bh[t][ks], bl[t][ks]);
        }
        pa.sstore(sA(cur ^ 1), ra);                   // one basic block per stage, as in the NT kernel
        pb.sstore(sB(cur ^ 1), rb);
        issue(st + 3, ra, rb);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < TNW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh[t][ks], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TNW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl[t][ks], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TNW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh[t][ks], acc[t], 0, 0, 0);
        }
        if (SCHED) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4 + 4 * TNW, 0);
#pragma unroll
            for (int i = 0; i < 6 * TNW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                if (i % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __syncthreads();
    };
    if (nstage > 0) {
        issue(0, ra0, rb0);
        pa.sstore(sA(0), ra0);
        pb.sstore(sB(0), rb0);
        issue(1, ra0, rb0);
        issue(2, ra1, rb1);
        __syncthreads();
    }
    for (int64_t st = 0; st < nstage; st += 2) {
        stage(st, ra0, rb0);
        if (st + 1 < nstage) stage(st + 1, ra1, rb1);
    }
    // partial tile -> LDS -> whole 16-byte chunks of ws[slice] (N % 4 == 0, checked by the launcher)
    __syncthreads();
    constexpr int CS = BNt + 8;
    static_assert((NT / 64) * 32 * CS * 4 <= 2 * (FA + FB), "C staging must fit the operand stages");
    float* cs = reinterpret_cast<float*>(lds) + wave * 32 * CS;
    {
        const int rl = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int t = 0; t < TNW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * CS + t * 32 + rl] = acc[t][r];
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int C4 = BNt / 4;
    float* dst = ws + (int64_t)slice * M * N;
    const int64_t row0 = bm + wave * 32;
#pragma unroll 2
    for (int idx = lane; idx < 32 * C4; idx += 64) {
        const int r = idx / C4, c4 = idx - r * C4;
        const int64_t row = row0 + r, col = bn + c4 * 4;
        if (row >= M || col >= N) continue;
        *reinterpret_cast<float4*>(dst + row * N + col) = *reinterpret_cast<const float4*>(cs + r * CS + c4 * 4);
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Grouped weight-gradient launch: up to 8 products  C_j[M_j, N_j] += A_j^T B_j  over the SAME rows (the contraction index), e.g.
// all of a layer's dW that depend on one activation gradient, in ONE launch, with
//   * the bias gradient for free: if job.colsum is set, column N_j of the B panel reads as 1.0 (the padding of the last column
//     tile, N_j < gx * 32 TNW), so that output column is sum_rows A = the column sums of the activation gradient;
//   * no separate fold launch: every K slice adds its partial tile into C with float atomics, issued as contiguous 256-byte
//     wave instructions (C / colsum accumulate: the caller zeroes them -- the gradient block of a step is one zero fill).
// Same tiles, LDS image and schedule as gemm_bf16x3_tn_kernel; 1-D grid of (tiles of all jobs) x slices, slices in multiples of
// 8 so that all tiles of one K slice run on one XCD (its L2 pulls that row range of the operands once).
struct WgJob { const float* A; const float* B; float* C; float* colsum; int64_t lda, ldb, ldc; int M, N, gx, tile0; };
struct WgJobs { WgJob j[8]; int n, total_tiles; };

// K-major panel of R output rows x 32 k for a workgroup of NTH threads: (R / 4) * 8 micro-tiles of 4 (k) x 4 (rows), PER per thread
// (threads past the last micro-tile redo an earlier one: same loads, same bytes into LDS -- no inactive-thread branches in a stage).
// Loads are BUFFER loads: the descriptor of a stage covers exactly the k rows that exist (base = first row of the stage, size =
// what is left of the K slice), so rows past the end of the slice -- and the micro-tiles past the last output row, whose offset
// is set out of range -- read as zero from the bounds check.  No per-load address arithmetic, compares or selects in the stage
// (the pointer form cost ~50 of its ~170 VALU instructions, against 12 MFMAs), and the stage stays ONE basic block, which the
// interleaving hints below need: with a branch in it the compiler left all 12 MFMAs in a row behind the whole split.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int R, int NTH, bool INJECT>
struct PanelK {
    static constexpr int TILES = (R / 4) * 8;
    static constexpr int PER = (TILES + NTH - 1) / NTH;
    unsigned voff[PER][4];     // byte offset of the micro-tile's 4 k rows from the stage base (out of range = reads zero)
    int lds_off[PER];          // byte offset of (row group, k group)
    float onef[PER];           // 1.0 for the micro-tile that starts at the injected ones column (its loads are out of range)

    __device__ __forceinline__ void init(int64_t ld, int64_t row0, int64_t nrows, bool inject) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int t = (threadIdx.x + j * NTH) % TILES;
            const int kg = t % 8, rg = t / 8;
            const int64_t row = row0 + rg * 4;                      // nrows % 4 == 0 (checked by the launcher)
            onef[j] = (INJECT && inject && row == nrows) ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) voff[j][i] = row >= nrows ? 0x80000000u : (unsigned)((((int64_t)(kg * 4 + i)) * ld + row) * 4);
            lds_off[j] = (rg * 4) * ROW_BYTES + kg * 8;
        }
    }
    __device__ __forceinline__ void gload(__amdgpu_buffer_rsrc_t rs, float4 (&reg)[PER][4]) const {
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) reg[j][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j][i], 0, 0));
    }
    // 4x4 register transpose, split, 8-byte stores: 4 consecutive k of one output row per word; one call = output row r of micro-tile
    // j (the stage interleaves these calls with its MFMAs).  The ones column's k rows past the end of K are 1.0 too: they meet zero
    // rows of the other operand.
    __device__ __forceinline__ void sstore_row(char* __restrict__ s, const float4 (&reg)[PER][4], int j, int r) const {
        const float4 r0 = reg[j][0], r1 = reg[j][1], r2 = reg[j][2], r3 = reg[j][3];
        float4 v = r == 0 ? make_float4(r0.x, r1.x, r2.x, r3.x) : r == 1 ? make_float4(r0.y, r1.y, r2.y, r3.y)
                 : r == 2 ? make_float4(r0.z, r1.z, r2.z, r3.z) : make_float4(r0.w, r1.w, r2.w, r3.w);
        if (INJECT && r == 0) { v.x += onef[j]; v.y += onef[j]; v.z += onef[j]; v.w += onef[j]; }
        uint2 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<uint2*>(s + lds_off[j] + r * ROW_BYTES) = hi;
        *reinterpret_cast<uint2*>(s + lds_off[j] + r * ROW_BYTES + 64) = lo;
    }
    __device__ __forceinline__ void sstore(char* __restrict__ s, const float4 (&reg)[PER][4]) const {
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sstore_row(s, reg, j, r);
    }
    __device__ __forceinline__ void gload1(__amdgpu_buffer_rsrc_t rs, float4 (&reg)[PER][4], int j) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) reg[j][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j][i], 0, 0));
    }
};
// descriptor of stage `st` of an operand: base = first k row of the stage, size = the bytes left of the slice from there on
// (0 when the stage does not exist: every load then returns zero without touching memory).  All scalar arithmetic.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stage_rsrc(const float* base, int64_t ld, int64_t bytes, int64_t st) {
    const int64_t adv = st * BK * ld;
    int64_t rem = bytes - adv * 4;
    if (rem < 0) rem = 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + adv), 0, (int)rem, 0x00020000);
}

#ifndef FLID_WG_EXP
#define FLID_WG_EXP 0   // timing experiments only (results wrong): 1 = no atomic fold, 2 = no MFMAs, 3 = no fragment reads, 4 = no LDS stores
#endif
constexpr int WBM = 64, WNT = 128;     // weight-gradient workgroup: 2 waves x (32 x 32 TNW), block tile 64 x 32 TNW.  64 rows fit the
                                       // output heights of this path (172 -> 192, 272 -> 320, 136 -> 192; 128-row tiles padded them to
                                       // 256 / 384 / 256) and 46 KB of LDS lets 3 workgroups share a CU: measured 80 / 99 / 57 us ->
                                       // 52 / 73 / 50 us for the three launches of a 13.6 k-row layer.
template <int TNW>
__global__ void __launch_bounds__(WNT) gemm_bf16x3_wgrad_kernel(WgJobs jobs, int64_t K, int64_t k_chunk, int nslices) {
    constexpr int BNt = 32 * TNW;
    constexpr int FA = WBM * ROW_BYTES, FB = BNt * ROW_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[2 * (FA + FB)];
    auto sA = [&](int i) -> char* { return lds + i * (FA + FB); };
    auto sB = [&](int i) -> char* { return lds + i * (FA + FB) + FA; };
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    int slice, t;
    if (nslices >= 8) {                     // slices in multiples of 8: slice s lives on XCD s % 8
        slice = xcd + 8 * (jj / jobs.total_tiles);
        t = jj % jobs.total_tiles;
    } else {                                // 1, 2 or 4 slices: 8 / nslices tiles of the same slice side by side on as many XCDs
        slice = xcd % nslices;
        t = jj * (8 / nslices) + xcd / nslices;
        if (t >= jobs.total_tiles) return;
    }
    int ji = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) if (q < jobs.n && t >= jobs.j[q].tile0) ji = q;
    const WgJob J = jobs.j[ji];
    const int tile = t - J.tile0, by = tile / J.gx, bx = tile % J.gx;
    const int64_t M = J.M, N = J.N;
    const int64_t bm = (int64_t)by * WBM, bn = (int64_t)bx * BNt;
    const int64_t kbeg = (int64_t)slice * k_chunk, kend = (kbeg + k_chunk < K) ? kbeg + k_chunk : K;
    if (kbeg >= K) return;                                  // (uniform; a surplus slice has nothing to add)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[TNW];
#pragma unroll
    for (int tt = 0; tt < TNW; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;

    using PA = PanelK<WBM, WNT, false>;
    using PB = PanelK<BNt, WNT, true>;
    PA pa;
    PB pb;
    pa.init(J.lda, bm, M, false);
    pb.init(J.ldb, bn, N, J.colsum != nullptr);
    // THREE register sets: the loads of stages st + 2, st + 3, st + 4 are in flight while stage st computes -- a stage is short
    // (12-18 MFMAs, ~0.5 us) and a slice's operands come from MALL / HBM (1.5-2.5 us loaded latency): with two sets the kernel ran
    // at one stage per half latency, whatever its instruction schedule
    float4 ra0[PA::PER][4], rb0[PB::PER][4], ra1[PA::PER][4], rb1[PB::PER][4], ra2[PA::PER][4], rb2[PB::PER][4];
    const int64_t nstage = (kend - kbeg + BK - 1) / BK;
    const float* a_base = J.A + kbeg * J.lda;
    const float* b_base = J.B + kbeg * J.ldb;
    const int64_t a_bytes = ((kend - kbeg - 1) * J.lda + M) * 4, b_bytes = ((kend - kbeg - 1) * J.ldb + N) * 4;
    auto issue = [&](int64_t st, float4 (&ra)[PA::PER][4], float4 (&rb)[PB::PER][4]) {
        pa.gload(stage_rsrc(a_base, J.lda, a_bytes, st), ra);
        pb.gload(stage_rsrc(b_base, J.ldb, b_bytes, st), rb);
    };
    // One stage = one basic block, scheduled by hand: the fragment reads of stage st, then MFMA c followed by chunk c of the staging
    // of stage st + 1 (split + LDS stores of one output row of one micro-tile; the loads of stage st + 3 refill a micro-tile's
    // registers as soon as its fourth row is done).  A chunk is ~12 VALU instructions (48 cycles), an MFMA 32: both pipes stay busy
    // from ONE wave, where the compiler's own order (all splits, then all MFMAs back to back) left each idle half of the time.
    constexpr int NM = 6 * TNW, NCH = 4 * (PA::PER + PB::PER);
    static_assert(NM >= NCH, "every staging chunk needs an MFMA to hide under");
    auto stage = [&](int64_t st, float4 (&ra)[PA::PER][4], float4 (&rb)[PB::PER][4]) {
        const int cur = (int)(st & 1);
        bf16x8 ah[2], al[2], bh[TNW][2], bl[TNW][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (FLID_WG_EXP == 3) {
                const bf16x8 c = __builtin_bit_cast(bf16x8, make_uint4((unsigned)st, lane, ks, 1));
                ah[ks] = al[ks] = c;
#pragma unroll
                for (int tt = 0; tt < TNW; ++tt) bh[tt][ks] = bl[tt][ks] = c;
                continue;
            }
            read_frag(sA(cur), wave * 32, ks, ah[ks], al[ks]);
#pragma unroll
            for (int tt = 0; tt < TNW; ++tt) read_frag(sB(cur), 32 * tt, ks, bh[tt][ks], bl[tt][ks]);
        }
        const __amdgpu_buffer_rsrc_t rsa = stage_rsrc(a_base, J.lda, a_bytes, st + 4), rsb = stage_rsrc(b_base, J.ldb, b_bytes, st + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NM; ++c) {
            if (FLID_WG_EXP != 2) {
                const int ks = c / (3 * TNW), kind = (c % (3 * TNW)) / TNW, tt = c % TNW;
                if (kind == 0) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh[tt][ks], acc[tt], 0, 0, 0);
                else if (kind == 1) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl[tt][ks], acc[tt], 0, 0, 0);
                else acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh[tt][ks], acc[tt], 0, 0, 0);
            }
            if (c < NCH && FLID_WG_EXP != 4) {
                const int jt = c / 4, r = c % 4;                  // micro-tile (A's first, then B's), output row
                if (jt < PA::PER) {
                    pa.sstore_row(sA(cur ^ 1), ra, jt, r);
                    if (r == 3) pa.gload1(rsa, ra, jt);
                } else {
                    pb.sstore_row(sB(cur ^ 1), rb, jt - PA::PER, r);
                    if (r == 3) pb.gload1(rsb, rb, jt - PA::PER);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FLID_WG_EXP == 2) {                        // keep the fragment reads alive without the MFMAs
            auto fs = [](const bf16x8& v) { const float4 f = __builtin_bit_cast(float4, v); return (f.x + f.y) + (f.z + f.w); };
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc[0][ks] += fs(ah[ks]) + fs(al[ks]);
#pragma unroll
                for (int tt = 0; tt < TNW; ++tt) acc[tt][2 + ks] += fs(bh[tt][ks]) + fs(bl[tt][ks]);
            }
        }
        if (FLID_WG_EXP == 4) {
#pragma unroll
            for (int q = 0; q < PA::PER; ++q) { acc[0][q] += ra[q][0].x + ra[q][1].y + ra[q][2].z + ra[q][3].w; pa.gload1(rsa, ra, q); }
#pragma unroll
            for (int q = 0; q < PB::PER; ++q) { acc[0][q + 4] += rb[q][0].x + rb[q][1].y + rb[q][2].z + rb[q][3].w; pb.gload1(rsb, rb, q); }
        }
        __syncthreads();
    };
    issue(0, ra0, rb0);
    pa.sstore(sA(0), ra0);
    pb.sstore(sB(0), rb0);
    issue(1, ra0, rb0);
    issue(2, ra1, rb1);
    issue(3, ra2, rb2);
    __syncthreads();
    // (always three stages per trip -- a stage past the end multiplies zero rows: with conditional stages the compiler's wait-count
    // model gave up across the back edge and waited for all but 6 of the 24 loads in flight before the first stage of a trip)
    for (int64_t st = 0; st < nstage; st += 3) {
        stage(st, ra0, rb0);
        stage(st + 1, ra1, rb1);
        stage(st + 2, ra2, rb2);
    }
    // partial tile -> LDS -> float atomics into C, 64 consecutive columns of a row per wave instruction
    __syncthreads();
    constexpr int CS = BNt + 8;
    static_assert((WNT / 64) * 32 * CS * 4 <= 2 * (FA + FB), "C staging must fit the operand stages");
    float* cs = reinterpret_cast<float*>(lds) + wave * 32 * CS;
    {
        const int rl = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int tt = 0; tt < TNW; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * CS + tt * 32 + rl] = acc[tt][r];
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t row0 = bm + wave * 32;
    for (int idx = lane; idx < 32 * BNt; idx += 64) {
        const int r = idx / BNt, c = idx - r * BNt;
        const int64_t row = row0 + r, col = bn + c;
        if (row >= M) continue;
        const float v = cs[r * CS + c];
        if (FLID_WG_EXP == 1) { if (v == 12345.678f) J.C[0] = v; continue; }
        if (col < N) atomicAdd(J.C + row * J.ldc + col, v);
        else if (col == N && J.colsum) atomicAdd(J.colsum + row, v);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace tg {

// Grouped weight gradients (see gemm_bf16x3_wgrad_kernel).  false = a job's shape / alignment is not covered (nothing launched):
// the caller takes tg_gemm_f32 + tg_colsum per job.
bool wgrad_group(int njobs, const tg_wgrad_job* jobs, int64_t rows, hipStream_t s) {
    if (njobs < 1 || njobs > 8 || rows < 1) return false;
    // One column-tile width (64 or 96) and one K-slice count (a multiple of 8: XCD pinning) for the launch.  Swept on MI355X
    // (a slice-count sweep at 13.6 k rows, round 2): the best point of every launch of a layer has 500-700 workgroups (768 are resident at
    // once: 3 per CU at 46 KB of LDS) and the 64-wide tile unless it pads the outputs > 10 % more than the 96-wide one:
    //   dW2 + dW1a + dW1b: 33 tiles x 16 slices 52.6 us (x8 71.8, x24 58.6, x32 71.6);  dP: 42 x 16 51.6 us (x8 67.6, x24 54.9);
    //   dV: 70 x 8 83.5 us (x16 87.9).  More slices only multiply the atomic fold's traffic, fewer leave the chip half empty.
    int64_t tiles_c[2] = {0, 0};
    for (int c = 0; c < 2; ++c) {
        const int w = 32 * (c + 2);
        for (int i = 0; i < njobs; ++i) {
            const int64_t need = jobs[i].N + (jobs[i].colsum_A ? 1 : 0);
            tiles_c[c] += ((need + w - 1) / w) * ((jobs[i].M + WBM - 1) / WBM);
        }
    }
    int best_tnw = (double)tiles_c[0] * 64 <= (double)tiles_c[1] * 96 * 1.1 ? 2 : 3;
    const int64_t max_slices = std::max<int64_t>(8, rows / (4 * BK) / 8 * 8);
    // (re-swept with the buffer-load staging: 67 tiles x 16 slices 67.6 us against x 8 77.1 -- up to ~70 tiles take 16 slices)
    const int64_t per = 700 / std::max<int64_t>(1, tiles_c[best_tnw - 2]);
    int64_t best_slices = per >= 10 ? std::max<int64_t>(16, per / 8 * 8) : 8;
    if (best_slices > max_slices) best_slices = max_slices;
    // few rows (the root layer's 1 200): 8 slices are 5 stages each and an 8-fold atomic fold -- 2 or 4 slices of >= 256 rows, their
    // tiles spread over the XCDs instead (all six gradients of a 1 200-row layer: 42.9 us with 8 slices, 29.3 with 4, 28.6 with 2)
    if (rows < 2048) best_slices = rows >= 1024 ? 4 : (rows >= 512 ? 2 : 1);
    static const bool tuning = getenv("FLID_GEMM_TUNE") != nullptr;          // overrides are read only in tuning mode (tools/)
    if (tuning) {
        if (const char* e = getenv("FLID_WG_TNW")) { const int v = atoi(e); if (v == 2 || v == 3) best_tnw = v; }
        if (const char* e = getenv("FLID_WG_SLICES")) { const int v = atoi(e); if ((v >= 8 && v % 8 == 0) || v == 4 || v == 2 || v == 1) best_slices = v; }
        if (getenv("FLID_WG_VERBOSE")) fprintf(stderr, "[wgrad] jobs=%d rows=%lld tnw=%d slices=%lld\n", njobs, (long long)rows, best_tnw, (long long)best_slices);
    }
    const int tnw = best_tnw;
    WgJobs wj;
    wj.n = njobs;
    wj.total_tiles = 0;
    double flops = 0;
    for (int i = 0; i < njobs; ++i) {
        const tg_wgrad_job& q = jobs[i];
        if (!(q.A && q.B && q.C && q.M >= 4 && q.N >= 4 && q.M % 4 == 0 && q.N % 4 == 0 && q.lda % 4 == 0 && q.ldb % 4 == 0 && al16(q.A) && al16(q.B)))
            return false;
        if (q.lda < q.M || q.ldb < q.N || q.ldc < q.N) return false;
        const int64_t need = q.N + (q.colsum_A ? 1 : 0);
        const int gx = (int)((need + 32 * tnw - 1) / (32 * tnw)), gy = (q.M + WBM - 1) / WBM;
        wj.j[i] = WgJob{q.A, q.B, q.C, q.colsum_A, q.lda, q.ldb, q.ldc, q.M, q.N, gx, wj.total_tiles};
        wj.total_tiles += gx * gy;
        flops += 2.0 * q.M * q.N * rows;
    }
    const int64_t slices = best_slices;
    int64_t k_chunk = ((rows + slices - 1) / slices + BK - 1) / BK * BK;
    if (k_chunk < BK) k_chunk = BK;
    const int64_t blocks = slices >= 8 ? (int64_t)wj.total_tiles * slices : ((wj.total_tiles + 8 / slices - 1) / (8 / slices)) * 8;
    if (blocks >= ((int64_t)1 << 31)) return false;
    for (int i = 0; i < njobs; ++i)          // a K slice of either operand is addressed with 32-bit byte offsets (buffer loads)
        if ((k_chunk + 4 * BK) * std::max(jobs[i].lda, jobs[i].ldb) * 4 >= ((int64_t)1 << 31)) return false;
    ProfScope prof("gemm", flops, s);
    if (tnw == 3) gemm_bf16x3_wgrad_kernel<3><<<(unsigned)blocks, WNT, 0, s>>>(wj, rows, k_chunk, (int)slices);
    else gemm_bf16x3_wgrad_kernel<2><<<(unsigned)blocks, WNT, 0, s>>>(wj, rows, k_chunk, (int)slices);
    return true;
}

// returns true when the shape was handled; false = fall back to the exact f32-input kernel
bool gemm_bf16x3_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                    int64_t strideB, float* C, int64_t ldc, int64_t strideC, int nbatch, const float* bias, int relu, int accumulate,
                    hipStream_t s, const float* mask, int64_t ldm) {
    if (mask && nbatch != 1) return false;
    if (!(al16(A) && al16(B) && lda % 4 == 0 && ldb % 4 == 0 && K % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0)) return false;
    if (M < 1 || N < 1 || K < 8 || nbatch > 65535) return false;
    // column block 96 or 64: the smaller padded N wins (272 -> 288, 172 -> 192, 444 -> 480, 136 -> 192 either way), ties to 96
    const int64_t pad3 = (N + 95) / 96 * 96, pad2 = (N + 63) / 64 * 64;
    // (96-column blocks also where they pad up to 8 % more than 64-column ones: every block of columns re-stages and re-splits the whole
    // row panel, and fewer, wider blocks won 4-10 % at N = 600 / 800 / 888 -- tools/nt_tile_sweep.sh; 128-column blocks lost)
    int tnw = pad3 * 100 <= pad2 * 108 ? 3 : 2;
    static const int force_tnw = (getenv("FLID_GEMM_TUNE") && getenv("FLID_NT_TNW")) ? atoi(getenv("FLID_NT_TNW")) : 0;
    if (force_tnw == 2 || force_tnw == 3) tnw = force_tnw;
    const int64_t gx = (N + 32 * tnw - 1) / (32 * tnw), gy = (M + BM - 1) / BM;
    if (gx * gy >= ((int64_t)1 << 30)) return false;
    const float* zeros = zero_block();      // the kernel's out-of-range loads are pointed at it
    if (!zeros) return false;
    const int vec_c = N % 4 == 0 && ldc % 4 == 0 && strideC % 4 == 0 && al16(C) && (!bias || al16(bias)) && (!mask || (al16(mask) && ldm % 4 == 0));
    ProfScope prof("gemm", 2.0 * M * N * K * nbatch, s);
    const dim3 grid((unsigned)(gx * gy), 1, (unsigned)nbatch);
    if (tnw == 3)
        gemm_bf16x3_nt_kernel<3><<<grid, NT, 0, s>>>(M, N, K, A, lda, B, ldb, C, ldc, bias, relu, accumulate, (int)gx, (int)gy, strideA, strideB, strideC, zeros, vec_c, mask, ldm, 0, 0, 0, nullptr);
    else
        gemm_bf16x3_nt_kernel<2><<<grid, NT, 0, s>>>(M, N, K, A, lda, B, ldb, C, ldc, bias, relu, accumulate, (int)gx, (int)gy, strideA, strideB, strideC, zeros, vec_c, mask, ldm, 0, 0, 0, nullptr);
    return true;
}

// C1 = A1 B1^T + bias1 (K1 deep) and C2 = A2 B2^T + bias2 (K2 deep), both M x N, as ONE launch (grid z = 2): two dependent-latency chains of
// a few hundred workgroups each run side by side instead of one after the other.  false = not covered (the caller issues two products).
bool gemm_bf16x3_nt_pair(int64_t M, int64_t N, int64_t K1, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, float* C1, const float* bias1,
                         int64_t K2, const float* A2, int64_t lda2, const float* B2, int64_t ldb2, float* C2, const float* bias2, int64_t ldc, hipStream_t s) {
    if (!(al16(A1) && al16(B1) && al16(A2) && al16(B2) && lda1 % 4 == 0 && ldb1 % 4 == 0 && lda2 % 4 == 0 && ldb2 % 4 == 0 && K1 % 4 == 0 && K2 % 4 == 0)) return false;
    if (M < 1 || N < 1 || K1 < 8 || K2 < 8) return false;
    // (operands of one allocation each: the second problem is addressed as an element offset from the first)
    if ((reinterpret_cast<uintptr_t>(A1) | reinterpret_cast<uintptr_t>(A2) | reinterpret_cast<uintptr_t>(B1) | reinterpret_cast<uintptr_t>(B2) |
         reinterpret_cast<uintptr_t>(C1) | reinterpret_cast<uintptr_t>(C2)) & 3) return false;
    const int64_t pad3 = (N + 95) / 96 * 96, pad2 = (N + 63) / 64 * 64;
    const int tnw = pad3 * 100 <= pad2 * 108 ? 3 : 2;
    const int64_t gx = (N + 32 * tnw - 1) / (32 * tnw), gy = (M + BM - 1) / BM;
    if (gx * gy >= ((int64_t)1 << 30)) return false;
    const float* zeros = zero_block();
    if (!zeros) return false;
    const int vec_c = N % 4 == 0 && ldc % 4 == 0 && al16(C1) && al16(C2) && (!bias1 || al16(bias1)) && (!bias2 || al16(bias2));
    ProfScope prof("gemm", 2.0 * M * N * (K1 + K2), s);
    const dim3 grid((unsigned)(gx * gy), 1, 2);
    const int64_t sA = A2 - A1, sB = B2 - B1, sC = C2 - C1;
    if (tnw == 3)
        gemm_bf16x3_nt_kernel<3><<<grid, NT, 0, s>>>(M, N, K1, A1, lda1, B1, ldb1, C1, ldc, bias1, 0, 0, (int)gx, (int)gy, sA, sB, sC, zeros, vec_c, nullptr, 0, K2, lda2, ldb2, bias2);
    else
        gemm_bf16x3_nt_kernel<2><<<grid, NT, 0, s>>>(M, N, K1, A1, lda1, B1, ldb1, C1, ldc, bias1, 0, 0, (int)gx, (int)gy, sA, sB, sC, zeros, vec_c, nullptr, 0, K2, lda2, ldb2, bias2);
    return true;
}

// Partial products of C = A^T B (A: K x M, B: K x N, row-major) for nbatch x nsplit K slices into ws[slice][M][N]; the caller
// (tg_gemm.hip) chose the split, owns the workspace and folds it.  false = shape not handled (use the f32-input kernel).
bool gemm_bf16x3_tn_partials(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                             int64_t strideB, float* ws, int nbatch, int nsplit, int64_t k_chunk, hipStream_t s) {
    if (!(al16(A) && al16(B) && lda % 4 == 0 && ldb % 4 == 0 && M % 4 == 0 && N % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0)) return false;
    if (M < 4 || N < 4 || K < 64 || !ws || (nbatch * nsplit) % 8 != 0 || k_chunk % BK != 0) return false;
    const float* zeros = zero_block();
    if (!zeros) return false;
    const int64_t pad3 = (N + 95) / 96 * 96, pad2 = (N + 63) / 64 * 64;
    const int tnw = pad3 <= pad2 ? 3 : 2;
    const int64_t gx = (N + 32 * tnw - 1) / (32 * tnw), gy = (M + BM - 1) / BM;
    const int64_t blocks = gx * gy * nbatch * nsplit;
    if (blocks >= ((int64_t)1 << 31)) return false;
    if (tnw == 3)
        gemm_bf16x3_tn_kernel<3><<<(unsigned)blocks, NT, 0, s>>>(M, N, K, A, lda, B, ldb, ws, (int)gx, (int)(gx * gy), k_chunk, nsplit, strideA, strideB, zeros);
    else
        gemm_bf16x3_tn_kernel<2><<<(unsigned)blocks, NT, 0, s>>>(M, N, K, A, lda, B, ldb, ws, (int)gx, (int)(gx * gy), k_chunk, nsplit, strideA, strideB, zeros);
    return true;
}

}  // namespace tg
