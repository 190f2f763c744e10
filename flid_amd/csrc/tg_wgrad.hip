// Weight gradients of a layer, second form:  C_j[M_j, N_j] += A_j^T B_j  (A_j: rows x M_j, B_j: rows x N_j, row-major fp32; the
// contraction index is the ROW index of both operands), up to 8 products over the same rows in ONE launch + one fold launch.
//
// replaces: the autograd weight / bias gradients of every nn.Linear of a temporal-attention layer (models/modules.py:54-69, :152-163,
//           :235) -- same contract as the first form (gemm_bf16x3_wgrad_kernel in tg_gemm_bf16x3.hip), which stays as the fall-back.
//
// What the first form paid for (profiles/r02: 117 us for the six gradients of a 13.6 k-row layer, matrix pipe 15 % busy):
//   * 64 x 64 output tiles: every operand element was staged (and split fp32 -> bf16 hi / lo) by 3-7 different workgroups;
//   * a 4 x 4 register transpose per micro-tile, because the MFMA wants 8 consecutive k per lane and the operands are k-major;
//   * 16 K slices folded with float atomics: 16 x 600 k atomic adds per launch at the chip's ~1.3 TB/s atomic rate.
// Here
//   * a workgroup owns a 192 x 256 output tile (8 waves as 2 x 4, each 96 x 64 = six 32 x 32 accumulators): the five products of a
//     layer are 16 tiles, each operand element is staged 1-3 times;
//   * operands go to LDS as they lie in memory ([k][column] bf16 hi / lo planes, 8-byte stores of 4 consecutive columns) and come
//     back through ds_read_b64_tr_b16, the transposing LDS read of gfx950: no register transpose, no VALU besides the split;
//   * K slices write their partial tiles with plain stores into a workspace; a second, tiny launch adds the slices in fixed order
//     into C (deterministic, no atomics).  The bias gradient rides along as before: a column of ones appended to B.
// Split-bf16 arithmetic as in tg_gemm_bf16x3.hip: hi*hi + hi*lo + lo*hi in v_mfma_f32_32x32x16_bf16, fp32 accumulation.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "tg_common.h"
#include "tg_colsum.h"
#include "tg_tail.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TMB = 6, TNB = 8;                 // tile extent in 32-blocks
constexpr int TM = 32 * TMB, TN = 32 * TNB;     // 192 x 256
constexpr int CK = 32;                          // contraction rows per chunk
constexpr int WNT = 512;
constexpr int AST = TM * 2 + 64, BST = TN * 2 + 64;      // LDS row strides (bytes): +64 spreads the 4 rows of a transposed read over all banks
constexpr int A_PLANE = CK * AST, B_PLANE = CK * BST;
constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
constexpr int NA = TM / 4 * CK / WNT, NB = TN / 4 * CK / WNT;      // float4 per thread and chunk: 3 + 4
static_assert(TM / 4 * CK % WNT == 0 && TN / 4 * CK % WNT == 0, "staging must tile the workgroup");
#ifndef FLID_WG2_SCHED
#define FLID_WG2_SCHED 1
#endif
#ifndef FLID_WG2_EXP
#define FLID_WG2_EXP 0   // timing experiments only (results wrong): 1 no steady-state loads, 2 no MFMAs, 3 no LDS stores
#endif

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
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

struct WTile {
    const float *A, *B;         // operand columns of this tile: A + m0, B + n0
    int64_t lda, ldb;
    int mext, next;             // valid columns of A / B in the tile (multiples of 4)
    int ones_col;               // column of the B tile that reads as 1.0 (the bias gradient), or -1
    int nw;                     // stored width = next (+ 4 if ones_col >= 0)
    int64_t slab_off;           // floats, inside one slice of the workspace
    float* C; int64_t ldc;      // fold: C[m0.., n0..] += ...; colsum[m0..] += column ones_col
    float* colsum;
    int trans;                  // the tile holds (B^T A): the fold writes element (m, n) to C[n * ldc + m] (job computed with its operands swapped)
    int nsl;                    // K slices of this tile's group (what the fold adds up)
};
constexpr int MAX_TILES = 34;   // (kernel arguments stay under 4 KB: 34 x 88 B + two ColExtra)
struct WTiles { WTile t[MAX_TILES]; int n; };
static_assert(sizeof(WTile) == 88, "WTile grew: re-check the kernel argument block");
// Up to two GROUPS of jobs in one launch: each group contracts over its own rows (a 13.6 k-row layer and the 1 200-row layer above it;
// TGN's layer and its GRU).  The grid is `cap` blocks per XCD (block b runs on XCD b % 8: observed, used for speed only); group 0 takes
// slices s = x, x + 8, ... of all its tiles on XCD x (a slice's row range is pulled into ONE L2), group 1 fills what is left, in order.
struct WGroups {
    int n, cap;
    int tile0[2], ntiles[2], nslices[2];
    int64_t rows[2], rps[2];
};

// 8 bf16 of one MFMA operand fragment from a [k][column] image: two transposing reads (k = 8 h + 0..3, 8 h + 4..7)
__device__ __forceinline__ bf16x8 frag_tr(const char* p, int stride) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * stride));
    const s16x8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(WNT, 2) wgrad2_kernel(WTiles tiles, WGroups gr, float* __restrict__ ws, int64_t slice_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int t, slice, gi = 0;
    if (gr.n == 1) {
        const int nslices = gr.nslices[0], nt = gr.ntiles[0];
        if ((nslices & 7) == 0) {                   // all tiles of a K slice on one XCD: its L2 pulls that row range of the operands once
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            t = j % nt;
            slice = xcd + 8 * (j / nt);
        } else {
            t = blockIdx.x % nt;
            slice = blockIdx.x / nt;
        }
    } else {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int nt0 = gr.ntiles[0], s0 = gr.nslices[0];
        const int c0 = nt0 * ((s0 + 7 - x) >> 3);                    // group 0's blocks on this XCD
        if (j < c0) {
            t = j % nt0;
            slice = x + 8 * (j / nt0);
        } else {
            int before = 0;                                          // group 1's blocks on the XCDs in front of this one
            for (int y = 0; y < x; ++y) before += gr.cap - nt0 * ((s0 + 7 - y) >> 3);
            const int idx = before + (j - c0), nt1 = gr.ntiles[1];
            if (idx >= nt1 * gr.nslices[1]) return;                  // (a block nobody needs: the grid is a whole number of blocks per XCD)
            gi = 1;
            t = gr.tile0[1] + idx % nt1;
            slice = idx / nt1;
        }
    }
    const int64_t R = gr.rows[gi], rows_per_slice = gr.rps[gi];
    const WTile T = tiles.t[t];
    const int64_t kbeg = (int64_t)slice * rows_per_slice;
    const int64_t kend = kbeg + rows_per_slice < R ? kbeg + rows_per_slice : R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunks = kend > kbeg ? (int)((kend - kbeg + CK - 1) / CK) : 0;

    // ---- staging map: A float4 f = tid + 512 i -> (row f / 48, chunk f % 48); B f -> (row f >> 6, chunk f & 63)
    const float* pa[NA];
    const float* pb[NB];
    int ra_row[NA], rb_row[NB], la[NA], lb[NB];
    bool ca_ok[NA], cb_ok[NB], ones[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + WNT * i, row = f / (TM / 4), c4 = f % (TM / 4);
        ra_row[i] = row;
        ca_ok[i] = 4 * c4 < T.mext;
        la[i] = row * AST + c4 * 8;
        pa[i] = T.A + (ca_ok[i] ? 4 * c4 : 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = tid + WNT * i, row = f / (TN / 4), c4 = f % (TN / 4);
        rb_row[i] = row;
        cb_ok[i] = 4 * c4 < T.next;
        ones[i] = 4 * c4 == T.ones_col;
        lb[i] = row * BST + c4 * 8;
        pb[i] = T.B + (cb_ok[i] ? 4 * c4 : 0);
    }
    auto load = [&](float4 (&xa)[NA], float4 (&xb)[NB], int c) {
        const int64_t k0 = kbeg + (int64_t)c * CK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int64_t k = k0 + ra_row[i];
            if (k > R - 1) k = R - 1;
            xa[i] = *reinterpret_cast<const float4*>(pa[i] + k * T.lda);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int64_t k = k0 + rb_row[i];
            if (k > R - 1) k = R - 1;
            xb[i] = *reinterpret_cast<const float4*>(pb[i] + k * T.ldb);
        }
    };
    // one staging unit = one float4 of the next chunk: mask, split, two 8-byte LDS stores (u < NA: operand A, else B)
    auto store_unit = [&](char* base, const float4 (&xa)[NA], const float4 (&xb)[NB], int64_t k0, int u) {
        if (u < NA) {
            const int i = u;
            const bool ok = ca_ok[i] && k0 + ra_row[i] < kend;
            const float4 v = make_float4(ok ? xa[i].x : 0.f, ok ? xa[i].y : 0.f, ok ? xa[i].z : 0.f, ok ? xa[i].w : 0.f);
            uint2 hi, lo;
            split4(v, hi, lo);
            if (FLID_WG2_EXP != 3) {
                *reinterpret_cast<uint2*>(base + la[i]) = hi;
                *reinterpret_cast<uint2*>(base + A_PLANE + la[i]) = lo;
            }
        } else {
            const int i = u - NA;
            const bool rok = k0 + rb_row[i] < kend;
            const bool ok = cb_ok[i] && rok;
            float4 v = make_float4(ok ? xb[i].x : 0.f, ok ? xb[i].y : 0.f, ok ? xb[i].z : 0.f, ok ? xb[i].w : 0.f);
            if (ones[i]) v.x = rok ? 1.f : 0.f;
            uint2 hi, lo;
            split4(v, hi, lo);
            if (FLID_WG2_EXP != 3) {
                *reinterpret_cast<uint2*>(base + 2 * A_PLANE + lb[i]) = hi;
                *reinterpret_cast<uint2*>(base + 2 * A_PLANE + B_PLANE + lb[i]) = lo;
            }
        }
    };
    auto store = [&](int stage, const float4 (&xa)[NA], const float4 (&xb)[NB], int c) {
        const int64_t k0 = kbeg + (int64_t)c * CK;
#pragma unroll
        for (int u = 0; u < NA + NB; ++u) store_unit(lds + stage * STAGE, xa, xb, k0, u);
    };

    // ---- fragments: wave (wm, wn) owns blocks [3 wm, 3 wm + 3) x [2 wn, 2 wn + 2)
    const int wm = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, h = g >> 1, mh = g & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int fa = (8 * h + q) * AST + (32 * 3 * wm + 16 * mh + 4 * p) * 2;
    const int fb = (8 * h + q) * BST + (32 * 2 * wn + 16 * mh + 4 * p) * 2;
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One half of the chunk pipeline (see gemm_rows_kernel: what a half waits for was issued a whole half earlier), ONE basic block
    // (a chunk past the end stores zeros and multiplies them), scheduled by hand: the 36 MFMAs of chunk c (stage `cur`) in 12 groups
    // of three; under 7 of the groups one staging unit of chunk c + 1 (~26 VALU instructions + 2 LDS stores into stage `nxt`), under
    // the others the loads of chunk c + 3 into the registers the units have freed.  The two waves of a SIMD leave every barrier
    // together: with all staging ahead of all products they used the vector and the matrix pipe in turn (68 us for a 13.6 k-row layer).
    auto half = [&](int cur, int nxt, float4 (&xa)[NA], float4 (&xb)[NB], int c_next, int c_load) {
        const char* base = lds + cur * STAGE;
        char* nbase = lds + nxt * STAGE;
        const int64_t k0n = kbeg + (int64_t)c_next * CK;
        const int64_t k0l = kbeg + (int64_t)c_load * CK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ah[i] = frag_tr(base + fa + ks * 16 * AST + i * 64, AST);
                al[i] = frag_tr(base + A_PLANE + fa + ks * 16 * AST + i * 64, AST);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = frag_tr(base + 2 * A_PLANE + fb + ks * 16 * BST + j * 64, BST);
                bl[j] = frag_tr(base + 2 * A_PLANE + B_PLANE + fb + ks * 16 * BST + j * 64, BST);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp = 0; grp < 6; ++grp) {
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const int mm = grp * 3 + e, term = mm / 6, i = (mm % 6) / 2, j = mm % 2;
                    if (FLID_WG2_EXP == 2) {
                        const float4 x = __builtin_bit_cast(float4, ah[i]), y = __builtin_bit_cast(float4, bh[j]);
                        const float4 z = __builtin_bit_cast(float4, al[i]), w = __builtin_bit_cast(float4, bl[j]);
                        acc[i][j][term] += x.x + y.y + z.z + w.w;
                    } else if (term == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    else if (term == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
                const int slot = ks * 6 + grp;                  // 0..11
                if (slot < NA + NB) {
                    store_unit(nbase, xa, xb, k0n, slot);
#if FLID_WG2_SCHED
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
#endif
                } else if (FLID_WG2_EXP != 1) {
                    // loads of chunk c_load: NA + NB float4 over the remaining 12 - (NA + NB) groups
                    constexpr int REM = 12 - (NA + NB);
                    const int r = slot - (NA + NB);
#pragma unroll
                    for (int u = 0; u < NA + NB; ++u) {
                        if (u * REM / (NA + NB) != r) continue;
                        if (u < NA) {
                            int64_t k = k0l + ra_row[u];
                            if (k > R - 1) k = R - 1;
                            xa[u] = *reinterpret_cast<const float4*>(pa[u] + k * T.lda);
                        } else {
                            int64_t k = k0l + rb_row[u - NA];
                            if (k > R - 1) k = R - 1;
                            xb[u - NA] = *reinterpret_cast<const float4*>(pb[u - NA] + k * T.ldb);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    float4 xa0[NA], xb0[NB], xa1[NA], xb1[NB];
    if (nchunks > 0) {
        load(xa0, xb0, 0);
        load(xa1, xb1, 1);
        store(0, xa0, xb0, 0);
        load(xa0, xb0, 2);
        __syncthreads();
        for (int c = 0; c < nchunks; c += 2) {
            half(0, 1, xa1, xb1, c + 1, c + 3);
            __syncthreads();
            half(1, 0, xa0, xb0, c + 2, c + 4);
            __syncthreads();
        }
    }
    // ---- partial tile -> workspace (plain stores, 128 contiguous bytes per half wave)
    float* dst = ws + (int64_t)slice * slice_stride + T.slab_off;
    const int col = lane & 31, rh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 32 * (2 * wn + j) + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * (3 * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * rh;
                if (m < T.mext && n < T.nw) dst[(int64_t)m * T.nw + n] = acc[i][j][r];
            }
        }
}

// C += sum over slices (fixed order); the ones column lands in colsum.  One element per thread, the slices' loads independent.
// Grid rows behind the tiles' (blockIdx.y >= tiles.n) are the caller's slab sums (tg::ColExtra): independent of the fold, they used to
// be a part of the layer's tail launch that could only start after it.
__global__ void __launch_bounds__(256) wgrad2_fold_kernel(WTiles tiles, const float* __restrict__ ws, int64_t slice_stride, tg::ColExtra ex0, tg::ColExtra ex1,
                                                          int rows0) {
    if ((int)blockIdx.y >= tiles.n) {
        __shared__ float red[4][64];
        __shared__ float red2[272];
        // the two groups' slab sums: grid rows [tiles.n, tiles.n + rows0) belong to the first, the rest to the second.  (Two branches, not
        // a reference picked by a condition: a select between two argument structs makes hipcc copy both to scratch.)
        auto run = [&](const tg::ColExtra& ex, int c) {
            const int ncol = ex.col_gx * ex.col_ny;
            if (c < ncol) tg::colsum_seg2_body(ex.a, ex.b, ex.groups_a, c % ex.col_gx, c / ex.col_gx, ex.col_ny, red);
            else if (c - ncol < ex.wq_n) {
                if (ex.wq_nb > 0) tg::wq_time_slab_body((c - ncol) % ex.wq_gx, (c - ncol) / ex.wq_gx, ex.wq_sq, ex.wq_nb, ex.wq_dq, ex.wq_cosb, ex.wq_T, ex.wq_W,
                                                        ex.wq_dW, ex.wq_ld, ex.wq_dcosb, red2);
                else tg::wq_time_body((c - ncol) % ex.wq_gx, (c - ncol) / ex.wq_gx, ex.wq_sq, ex.wq_dq, ex.wq_cosb, ex.wq_T, ex.wq_W, ex.wq_dW, ex.wq_ld, ex.wq_dcosb);
            }
        };
        if ((int)blockIdx.y < tiles.n + rows0) run(ex0, ((int)blockIdx.y - tiles.n) * (int)gridDim.x + (int)blockIdx.x);
        else run(ex1, ((int)blockIdx.y - tiles.n - rows0) * (int)gridDim.x + (int)blockIdx.x);
        return;
    }
    // four columns per thread: slab rows are multiples of four floats wide and 16-byte aligned (mext, nw, slab_off, slice_stride are
    // multiples of 4), so a thread has all its slices' 16-byte loads in flight at once (one float per thread: 23 us on the headline step)
    const WTile T = tiles.t[blockIdx.y];
    const int64_t total4 = (int64_t)T.mext * T.nw / 4;
    const int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e4 >= total4) return;
    const int64_t e = e4 * 4;
    const int m = (int)(e / T.nw), n = (int)(e - (int64_t)m * T.nw);
    if (n >= T.next && n != T.ones_col) return;
    const float4* p = reinterpret_cast<const float4*>(ws + T.slab_off + e);
    const int64_t stride4 = slice_stride / 4;
    const int nslices = T.nsl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sl = 0;
    for (; sl + 8 <= nslices; sl += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(sl + q) * stride4];
#pragma unroll
        for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
    for (; sl < nslices; ++sl) { const float4 v = p[(int64_t)sl * stride4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (n < T.next) {
        if (T.trans) {
            float* c = T.C + (int64_t)n * T.ldc + m;
            c[0] += s.x; c[T.ldc] += s.y; c[2 * T.ldc] += s.z; c[3 * T.ldc] += s.w;
        } else {
            float* c = T.C + (int64_t)m * T.ldc + n;
            c[0] += s.x; c[1] += s.y; c[2] += s.z; c[3] += s.w;
        }
    } else if (T.colsum) T.colsum[m] += s.x;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct Ws { hipStream_t stream; float* p; size_t floats; };
Ws g_ws[4] = {};
std::mutex g_ws_mutex;
float* workspace(size_t need, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Ws* w = nullptr;
    for (auto& c : g_ws) if (c.p && c.stream == s) { w = &c; break; }
    if (!w) for (auto& c : g_ws) if (!c.p) { w = &c; w->stream = s; break; }
    if (!w) return nullptr;
    if (need > w->floats) {
        if (w->p) { if (hipStreamSynchronize(s) != hipSuccess) return nullptr; (void)hipFree(w->p); w->p = nullptr; w->floats = 0; }
        const size_t want = need + need / 4;
        if (hipMalloc(&w->p, want * sizeof(float)) != hipSuccess) { w->p = nullptr; (void)hipGetLastError(); return nullptr; }
        w->floats = want;
    }
    return w->p;
}

bool g_wgrad2 = true;

}  // namespace

namespace tg {

namespace {
struct WGroupIn { int njobs; const tg_wgrad_job* jobs; int64_t rows; const ColExtra* extra; };

// the tiles of one group's jobs appended to wt; false = a job's shape / alignment is not covered or the tile table is full
bool build_tiles(const WGroupIn& gin, WTiles& wt, int64_t& slab, double& flops, int* swapped_sums, int& nswapped) {
    for (int i = 0; i < gin.njobs; ++i) {
        const tg_wgrad_job& q = gin.jobs[i];
        if (!(q.A && q.B && q.C && q.M >= 4 && q.N >= 4 && q.M % 4 == 0 && q.N % 4 == 0 && q.lda % 4 == 0 && q.ldb % 4 == 0 && al16(q.A) && al16(q.B)))
            return false;
        if (q.lda < q.M || q.ldb < q.N || q.ldc < q.N) return false;
        // A tile costs its full 192 x 256 whatever it covers, so a job runs with its operands SWAPPED (C^T = B^T A, written back
        // transposed by the fold) where that takes fewer tiles: 200 x 800 is 2 x 4 tiles as it stands, 5 x 1 swapped.  The ones column
        // can only sum the columns of the tile's FIRST operand: a swapped job's bias sum is a launch of its own (below).
        const int gm_s = ((q.M + 31) / 32 + TMB - 1) / TMB, gn_s = ((q.N + (q.colsum_A ? 4 : 0) + 31) / 32 + TNB - 1) / TNB;
        const int gm_t = ((q.N + 31) / 32 + TMB - 1) / TMB, gn_t = ((q.M + 31) / 32 + TNB - 1) / TNB;
        static const bool no_swap = getenv("FLID_GEMM_TUNE") && getenv("FLID_WG2_NOSWAP") && atoi(getenv("FLID_WG2_NOSWAP")) != 0;
        // (the fold's transposed write-back is strided: swapping pays from about a third fewer tiles on -- 888 x 172 as 4 tiles instead
        // of 5 was 20 us SLOWER on the headline step)
        const bool swap = !no_swap && gm_t * gn_t * 10 <= gm_s * gn_s * 7;
        const float *opA = swap ? q.B : q.A, *opB = swap ? q.A : q.B;
        const int64_t ldA = swap ? q.ldb : q.lda, ldB = swap ? q.lda : q.ldb;
        const int Mo = swap ? q.N : q.M, No = swap ? q.M : q.N;
        float* cs = swap ? nullptr : q.colsum_A;
        const int nfull = No + (cs ? 4 : 0);                                // the ones column takes a 4-column slot of its own
        const int mb = (Mo + 31) / 32, nb = (nfull + 31) / 32;
        const int gm = (mb + TMB - 1) / TMB, gn = (nb + TNB - 1) / TNB;
        const int em = (mb + gm - 1) / gm * 32, en = (nb + gn - 1) / gn * 32;   // balanced extents (multiples of 32)
        for (int a = 0; a < gm; ++a)
            for (int b = 0; b < gn; ++b) {
                if (wt.n >= MAX_TILES) return false;
                WTile& T = wt.t[wt.n++];
                const int m0 = a * em, n0 = b * en;
                T.A = opA + m0; T.B = opB + n0; T.lda = ldA; T.ldb = ldB;
                T.mext = std::min(em, Mo - m0);
                T.next = std::max(0, std::min(en, No - n0));
                const bool has_ones = cs && No >= n0 && No < n0 + en;
                T.ones_col = has_ones ? No - n0 : -1;
                T.nw = T.next + (has_ones ? 4 : 0);
                if (T.mext <= 0 || T.nw <= 0) { --wt.n; continue; }
                T.slab_off = slab;
                slab += (int64_t)T.mext * T.nw;
                T.trans = swap ? 1 : 0;
                T.nsl = 1;
                T.C = swap ? q.C + (int64_t)n0 * q.ldc + m0 : q.C + (int64_t)m0 * q.ldc + n0; T.ldc = q.ldc;
                T.colsum = has_ones ? cs + m0 : nullptr;
            }
        if (swap && q.colsum_A) swapped_sums[nswapped++] = i;
        flops += 2.0 * q.M * q.N * gin.rows;
    }
    return true;
}

// One or two groups as ONE product launch + ONE fold launch.  false = not covered / no workspace (nothing launched).
bool launch_groups(int ng, const WGroupIn* gin, hipStream_t s) {
    WTiles wt;
    wt.n = 0;
    WGroups gr{};
    gr.n = ng;
    int64_t slab = 0;
    double flops = 0;
    int swapped[2][8], nswapped[2] = {0, 0};
    for (int g = 0; g < ng; ++g) {
        gr.tile0[g] = wt.n;
        if (gin[g].njobs < 1 || gin[g].njobs > 8 || gin[g].rows < 1) return false;
        if (!build_tiles(gin[g], wt, slab, flops, swapped[g], nswapped[g])) return false;
        gr.ntiles[g] = wt.n - gr.tile0[g];
        gr.rows[g] = gin[g].rows;
        if (gr.ntiles[g] < 1) return false;
    }
    slab = (slab + 3) / 4 * 4;
    static const bool tuning = getenv("FLID_GEMM_TUNE") != nullptr;
    int64_t slices[2] = {1, 1};
    if (ng == 1) {
        // K slices: ~256 workgroups (one per CU: 128 KB of LDS each); a multiple of 8 pins each slice to an XCD
        int64_t sl = std::max<int64_t>(1, 256 / wt.n);
        if (sl >= 8) sl = sl / 8 * 8;
        const int64_t max_slices = std::max<int64_t>(1, gr.rows[0] / (2 * CK));
        if (sl > max_slices) sl = max_slices;
        if (tuning) if (const char* e = getenv("FLID_WG2_SLICES")) { const int v = atoi(e); if (v >= 1) sl = v; }
        int64_t rps = ((gr.rows[0] + sl - 1) / sl + CK - 1) / CK * CK;
        sl = (gr.rows[0] + rps - 1) / rps;
        if (sl >= 8 && sl % 8) sl = (sl + 7) / 8 * 8;      // keep the XCD pinning: round the slice count up to a multiple of 8 (empty tails are legal)
        slices[0] = sl;
        gr.rps[0] = rps;
        gr.cap = 0;
    } else {
        // 256 workgroups in all (one per CU: 128 KB of LDS each), all resident at once: the launch lasts as long as its longest workgroup,
        // i.e. the largest number of 32-row chunks any slice contracts over.  Try every slice count of the smaller group, give the rest
        // of the chip to the larger one, keep the split with the fewest chunks per workgroup.
        int64_t best = -1;
        for (int64_t s1 = 1; s1 <= 16; ++s1) {
            if (s1 > std::max<int64_t>(1, gr.rows[1] / (2 * CK)) || gr.ntiles[1] * s1 >= 256) break;
            int64_t s0 = (256 - gr.ntiles[1] * s1) / gr.ntiles[0];
            s0 = std::max<int64_t>(1, std::min<int64_t>(s0, std::max<int64_t>(1, gr.rows[0] / (2 * CK))));
            const int64_t c0 = ((gr.rows[0] + s0 - 1) / s0 + CK - 1) / CK, c1 = ((gr.rows[1] + s1 - 1) / s1 + CK - 1) / CK;
            const int64_t t = std::max(c0, c1);
            if (best < 0 || t < best) { best = t; slices[0] = s0; slices[1] = s1; }
        }
        for (int g = 0; g < 2; ++g) {
            gr.rps[g] = ((gr.rows[g] + slices[g] - 1) / slices[g] + CK - 1) / CK * CK;
            slices[g] = (gr.rows[g] + gr.rps[g] - 1) / gr.rps[g];
        }
        int cap = (int)((gr.ntiles[0] * slices[0] + gr.ntiles[1] * slices[1] + 7) / 8);
        cap = std::max<int>(cap, gr.ntiles[0] * (int)((slices[0] + 7) / 8));
        // (group 1's blocks must fit behind group 0's on the eight XCDs)
        int room = 0;
        for (int x = 0; x < 8; ++x) room += cap - gr.ntiles[0] * (int)((slices[0] + 7 - x) >> 3);
        if (room < gr.ntiles[1] * slices[1]) cap += (int)((gr.ntiles[1] * slices[1] - room + 7) / 8);
        gr.cap = cap;
    }
    for (int g = 0; g < ng; ++g) {
        gr.nslices[g] = (int)slices[g];
        for (int t = gr.tile0[g]; t < gr.tile0[g] + gr.ntiles[g]; ++t) wt.t[t].nsl = (int)slices[g];
    }
    const int64_t max_sl = std::max(slices[0], ng > 1 ? slices[1] : (int64_t)1);
    float* ws = workspace((size_t)(max_sl * slab), s);
    if (!ws) return false;
    ProfScope prof("gemm", flops, s);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    const unsigned grid = ng == 1 ? (unsigned)(wt.n * slices[0]) : (unsigned)(8 * gr.cap);
    wgrad2_kernel<<<grid, WNT, 2 * STAGE, s>>>(wt, gr, ws, slab);
    if (hipGetLastError() != hipSuccess) return false;
    const unsigned fold_gx = (TM * (TN + 4) / 4 + 255) / 256;
    ColExtra ex[2] = {ColExtra{}, ColExtra{}};
    unsigned extra_rows[2] = {0, 0};
    for (int g = 0; g < ng; ++g)
        if (gin[g].extra) { ex[g] = *gin[g].extra; extra_rows[g] = ((unsigned)(ex[g].col_gx * ex[g].col_ny + ex[g].wq_n) + fold_gx - 1) / fold_gx; }
    wgrad2_fold_kernel<<<dim3(fold_gx, (unsigned)wt.n + extra_rows[0] + extra_rows[1]), 256, 0, s>>>(wt, ws, slab, ex[0], ex[1], (int)extra_rows[0]);
    for (int g = 0; g < ng; ++g)
        for (int i = 0; i < nswapped[g]; ++i) {
            const tg_wgrad_job& q = gin[g].jobs[swapped[g][i]];
            if (tg_colsum(q.A, q.lda, gin[g].rows, q.M, q.colsum_A, 1, s) != TG_OK) return true;      // (launch errors surface at the caller's launch_status)
        }
    return true;
}

// A group kept back (wgrad_defer_next): it leaves with the NEXT group on the same stream -- the 1 200-row root layer's weight gradients
// in the launch of the 13.6 k-row layer below it, TGN's layer in its GRU's.  Such a group alone is a latency chain of ~19 + 12 us (product
// launch + fold) on a fraction of the chip; riding along it costs the big launch a few workgroups.
struct Deferred {
    bool valid = false;
    hipStream_t stream = nullptr;
    tg_wgrad_job jobs[8];
    int njobs = 0;
    int64_t rows = 0;
    ColExtra extra{};
    bool has_extra = false;
};
thread_local Deferred t_deferred;
thread_local bool t_defer_next = false;
}  // namespace

void wgrad_defer_next(bool on) { t_defer_next = on; }

// a kept-back group that nothing picked up leaves by itself
int wgrad_flush_deferred(hipStream_t s) {
    t_defer_next = false;
    if (!t_deferred.valid) return TG_OK;
    Deferred d = t_deferred;
    t_deferred.valid = false;
    const WGroupIn g{d.njobs, d.jobs, d.rows, d.has_extra ? &d.extra : nullptr};
    if (!launch_groups(1, &g, d.stream)) { set_error("tg_wgrad: a deferred group could not be launched"); return TG_ESHAPE; }
    (void)s;
    return launch_status("wgrad2_kernel");
}

// false = a job's shape / alignment is not covered or no workspace (nothing launched): the caller takes the first form
bool wgrad_group2(int njobs, const tg_wgrad_job* jobs, int64_t rows, hipStream_t s, const ColExtra* extra) {
    if (!g_wgrad2 || njobs < 1 || njobs > 8 || rows < 1) return false;
    if (t_defer_next) {
        t_defer_next = false;
        if (t_deferred.valid && wgrad_flush_deferred(s) != TG_OK) return false;
        // keep it only if it is launchable at all (a dry run of the tiling)
        WTiles wt; wt.n = 0;
        int64_t slab = 0; double fl = 0; int sw[8], nsw = 0;
        const WGroupIn probe{njobs, jobs, rows, extra};
        if (!build_tiles(probe, wt, slab, fl, sw, nsw)) return false;
        t_deferred.valid = true; t_deferred.stream = s; t_deferred.njobs = njobs; t_deferred.rows = rows;
        for (int i = 0; i < njobs; ++i) t_deferred.jobs[i] = jobs[i];
        t_deferred.has_extra = extra != nullptr;
        if (extra) t_deferred.extra = *extra;
        return true;
    }
    if (t_deferred.valid && t_deferred.stream == s) {
        Deferred d = t_deferred;
        t_deferred.valid = false;
        // the group with more work first (it gets the XCD-pinned slices)
        const WGroupIn mine{njobs, jobs, rows, extra}, kept{d.njobs, d.jobs, d.rows, d.has_extra ? &d.extra : nullptr};
        const WGroupIn both[2] = {rows >= d.rows ? mine : kept, rows >= d.rows ? kept : mine};
        static const bool no_merge = getenv("FLID_GEMM_TUNE") && getenv("FLID_WG2_NOMERGE") && atoi(getenv("FLID_WG2_NOMERGE")) != 0;
        if (!no_merge && launch_groups(2, both, s)) return true;
        if (!launch_groups(1, &kept, s)) return false;         // (too many tiles for one table: one after the other)
        return launch_groups(1, &mine, s);
    }
    if (t_deferred.valid && wgrad_flush_deferred(s) != TG_OK) return false;
    const WGroupIn g{njobs, jobs, rows, extra};
    return launch_groups(1, &g, s);
}

}  // namespace tg

extern "C" void tg_set_wgrad_form(int form) { g_wgrad2 = form != 1; }
