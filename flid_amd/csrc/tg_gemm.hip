// Dense fp32 GEMM on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bit-exact fmaf chain).
//
// replaces: aten::mm / addmm behind every nn.Linear of the hot path (models/modules.py:54-69, 152-163, 235) and
//           their autograd transposes.
//
// C[M,N] = alpha * op(A) * op(B) (+bias) (+C) (ReLU), optionally strided-batched (the two attention heads in one launch).
//
// Shape of the problem: M is the number of attention instances of one batch (1e3..3e4), N and K are 136..888.  The f32 MFMA
// retires one 32x32x2 step per 64 cycles per SIMD, so the kernel is matrix-pipe bound as soon as every one of the 1024 SIMDs
// holds 2+ waves -- the difficulty is tile quantisation, not bandwidth.  Hence: ONE 32x32 output tile per wavefront,
// TM x TN wavefronts per workgroup (32TM x 32TN block tile, chosen per call so that there are >= 4 workgroups per CU),
// operands staged through LDS in full 128-byte lines (BK = 32), two LDS stages and one barrier per stage; the global loads
// of stage s+2 are in flight under the MFMAs of stage s.
//
// Operand panels in LDS, P[r][kk] (r = row of op(A) / column of op(B), kk = 0..31):
//   k-contiguous source ("KC": A as M x K, B given as N x K): stored [r][kk], row stride 36 floats -> the four
//       ds_read_b128 of a lane (k = 16*half + 0..15) are bank-conflict free;
//   row-contiguous source ("MC": A^T, B given as K x N): stored [kk][r], read with conflict-free ds_read_b32.
// MFMA step q uses k = 16*(lane>>5) + q for BOTH operands (any bijection of k is a valid contraction order).
#include <stdlib.h>

#include <mutex>

#include "tg_common.h"

namespace tg {
// (mask, ldm): optional (M x N) matrix whose non-positive entries zero the output -- the ReLU backward fused into the product
bool gemm_direct_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                    int64_t strideB, float* C, int64_t ldc, int64_t strideC, int nbatch, const float* bias, int relu, int accumulate,
                    hipStream_t s, const float* mask, int64_t ldm, bool b_kc = true);
bool gemm_bf16x3_nt_pair(int64_t M, int64_t N, int64_t K1, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, float* C1, const float* bias1,
                         int64_t K2, const float* A2, int64_t lda2, const float* B2, int64_t ldb2, float* C2, const float* bias2, int64_t ldc, hipStream_t s);
bool gemm_bf16x3_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                    int64_t strideB, float* C, int64_t ldc, int64_t strideC, int nbatch, const float* bias, int relu, int accumulate,
                    hipStream_t s, const float* mask, int64_t ldm);
bool gemm_bf16x3_tn_partials(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                             int64_t strideB, float* ws, int nbatch, int nsplit, int64_t k_chunk, hipStream_t s);
}

namespace {

// 1 = row-times-weight products (A as M x K, B as N x K) run on the split-bf16 kernel (tg_gemm_bf16x3.hip); 0 = everything on
// the exact f32-input MFMA kernel below.
int g_gemm_mode = 1;
// per-thread override (-1 = none): a caller that wants exact products for ONE call (models/modules.py _exact_products) sets it around
// that call on its own thread -- flipping the process-wide mode raced with every other issuing thread (side-stream issuer, loaders)
thread_local int t_gemm_mode = -1;
inline int gemm_mode() { return t_gemm_mode >= 0 ? t_gemm_mode : g_gemm_mode; }

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int KC_STRIDE = BK + 4;

template <int R, bool KC> struct PanelFloats { static constexpr int value = KC ? R * KC_STRIDE : BK * R; };

// ---- global -> registers -> LDS, one panel of R rows x 32 k ---------------------------------------------------------------
// Branch-free in the steady state: rows beyond the matrix are CLAMPED to its last row (their products land in output rows /
// columns that are never stored), slot indices wrap around (a few threads re-load a slot another thread also loads; both
// write identical bytes to LDS), and only the last, partial K stage takes the predicated path.
template <int R, int NT, bool KC, bool VEC>
struct Panel {
    static constexpr int TOTAL = R * 8;                       // float4 slots
    static constexpr int PER = (TOTAL + NT - 1) / NT;

    const float* src[PER];     // address of the slot at k = kbeg
    int lds_off[PER];          // float offset inside the LDS panel
    int kcol[PER];             // k offset of the slot inside a stage (first of its 4 floats if KC)
    int rvalid[PER];           // non-VEC only: how many of the 4 consecutive rows exist (MC) / 1 (KC)

    __device__ __forceinline__ void init(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows, int64_t kbeg) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = (threadIdx.x + j * NT) % TOTAL;
            if constexpr (KC) {
                const int r = idx >> 3, c = (idx & 7) * 4;
                int64_t row = row0 + r;
                rvalid[j] = row < nrows;
                if (row > nrows - 1) row = nrows - 1;
                src[j] = X + row * ld + kbeg + c;
                lds_off[j] = r * KC_STRIDE + c;
                kcol[j] = c;
            } else {
                constexpr int Q = R / 4;                       // float4 per k row
                const int kr = idx / Q, c = (idx % Q) * 4;
                int64_t row = row0 + c;
                rvalid[j] = (int)((nrows - row) < 0 ? 0 : ((nrows - row) > 4 ? 4 : (nrows - row)));
                if constexpr (VEC) { if (row > nrows - 4) row = nrows - 4; }
                src[j] = X + (kbeg + kr) * ld + row;
                lds_off[j] = kr * R + c;
                kcol[j] = kr;
            }
        }
    }

    // stage starting at absolute k0 = kbeg + stage*32; `full` <=> k0 + 32 <= kend (wave-uniform)
    __device__ __forceinline__ void gload(int64_t ld, int64_t stage_elems, int64_t k0, int64_t kend, bool full, float4 (&reg)[PER]) const {
        if constexpr (VEC) {
            if (full) {
#pragma unroll
                for (int j = 0; j < PER; ++j) reg[j] = *reinterpret_cast<const float4*>(src[j] + stage_elems);
            } else {
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k0 + kcol[j] < kend) v = *reinterpret_cast<const float4*>(src[j] + stage_elems);
                    reg[j] = v;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* p = src[j] + stage_elems;
                if constexpr (KC) {
                    const int64_t kk = k0 + kcol[j];
                    if (rvalid[j]) {
                        if (kk + 0 < kend) v.x = p[0];
                        if (kk + 1 < kend) v.y = p[1];
                        if (kk + 2 < kend) v.z = p[2];
                        if (kk + 3 < kend) v.w = p[3];
                    }
                } else {
                    if (k0 + kcol[j] < kend) {
                        if (rvalid[j] > 0) v.x = p[0];
                        if (rvalid[j] > 1) v.y = p[1];
                        if (rvalid[j] > 2) v.z = p[2];
                        if (rvalid[j] > 3) v.w = p[3];
                    }
                }
                reg[j] = v;
            }
        }
    }

    __device__ __forceinline__ void sstore(float* __restrict__ s, const float4 (&reg)[PER]) const {
#pragma unroll
        for (int j = 0; j < PER; ++j) *reinterpret_cast<float4*>(s + lds_off[j]) = reg[j];
    }
};

// ---- LDS -> MFMA fragments (16 k-steps of one 32-row tile) -------------------------------------------------------------------
template <int R, bool KC>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, int tile_r0, float (&f)[16]) {
    const int lane = threadIdx.x & 63;
    const int rl = lane & 31, kh = lane >> 5;
    if constexpr (KC) {
        const float* p = s + (tile_r0 + rl) * KC_STRIDE + kh * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(p + 4 * g);
            f[4 * g + 0] = v.x; f[4 * g + 1] = v.y; f[4 * g + 2] = v.z; f[4 * g + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) f[q] = s[(kh * 16 + q) * R + tile_r0 + rl];
    }
}

template <bool A_KC, bool B_KC, bool VEC, int TM, int TN>
__global__ void __launch_bounds__(TM* TN * 64) gemm_tile_kernel(int64_t M, int64_t N, int64_t K, float alpha,
        const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
        const float* __restrict__ bias, int relu, int accumulate, int64_t k_chunk, int use_atomics, int gx, int gy_in,
        int64_t strideA, int64_t strideB, int64_t strideC, int nbatch, int inner, int64_t innerA, int64_t innerB, int64_t innerC,
        float* __restrict__ ws, int nslice_x, int vec_c) {
    constexpr int NT = TM * TN * 64, RA = 32 * TM, RB_ = 32 * TN;
    constexpr int FA = PanelFloats<RA, A_KC>::value, FB = PanelFloats<RB_, B_KC>::value;
    using PA = Panel<RA, NT, A_KC, VEC>;
    using PB = Panel<RB_, NT, B_KC, VEC>;
    __shared__ __attribute__((aligned(16))) float lds[2 * (FA + FB)];
    auto sA = [&](int i) -> float* { return lds + i * (FA + FB); };          // [A0 | B0 | A1 | B1]
    auto sB = [&](int i) -> float* { return lds + i * (FA + FB) + FA; };

    // workgroup id -> (row block, column block): XCD-aware bijective remap so that the column blocks of one row block (they
    // share the A rows) run on the same XCD's L2.  Slice = batch * splits + split.
    // Split contraction (nslice_x > 0, 1-D grid): every tile of one K slice shares that slice's rows of A and B, so a whole
    // slice is pinned to one XCD (hardware deals workgroups to XCDs round-robin): each L2 then pulls 1/8 of the operands from
    // memory once, instead of all 8 pulling nearly everything (the 272 x 444 x 12235 gradient moved ~280 MB, 60 us).
    int gy = gy_in;
    const int nwg = gx * (gy < 0 ? -gy : gy);
    int swz, zidx, nz;
    if (nslice_x > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;          // nslice_x % 8 == 0 (host)
        zidx = xcd + 8 * (j / nwg);
        swz = j % nwg;
        nz = nslice_x;
    } else {
        const int bid = blockIdx.x;
        const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
        swz = gy < 0 ? bid : (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
        zidx = blockIdx.z;
        nz = gridDim.z;
    }
    if (gy < 0) gy = -gy;
    const int by = swz / gx, bx = swz % gx;
    const int nsplit = nz / nbatch;
    const int batch = zidx / nsplit, split = zidx % nsplit;
    // two-level batch index: problem = (batch / inner, batch % inner), e.g. (sequence, head)
    const int bo = batch / inner, bi = batch % inner;
    A += bo * strideA + bi * innerA; B += bo * strideB + bi * innerB; C += bo * strideC + bi * innerC;
    if (bias) bias += batch * (int64_t)N;

    const int64_t bm = (int64_t)by * RA, bn = (int64_t)bx * RB_;
    const int64_t kbeg = (int64_t)split * k_chunk;
    const int64_t kend = (kbeg + k_chunk < K) ? kbeg + k_chunk : K;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tm = wave / TN, tn = wave % TN;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    PA pa;
    PB pb;
    pa.init(A, lda, bm, M, kbeg);
    pb.init(B, ldb, bn, N, kbeg);
    const int64_t stepA = A_KC ? BK : BK * lda, stepB = B_KC ? BK : BK * ldb;     // floats per K stage
    float4 ra[PA::PER], rb[PB::PER];
    const int64_t nstage = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const int64_t nfull = kend > kbeg ? (kend - kbeg) / BK : 0;
    if (nstage > 0) {
        pa.gload(lda, 0, kbeg, kend, nfull > 0, ra);
        pb.gload(ldb, 0, kbeg, kend, nfull > 0, rb);
        pa.sstore(sA(0), ra);
        pb.sstore(sB(0), rb);
        __syncthreads();
        if (nstage > 1) {
            pa.gload(lda, stepA, kbeg + BK, kend, nfull > 1, ra);
            pb.gload(ldb, stepB, kbeg + BK, kend, nfull > 1, rb);
        }
    }
    for (int64_t st = 0; st < nstage; ++st) {
        const int cur = (int)(st & 1);
        float fa[16], fb[16];
        read_frag<RA, A_KC>(sA(cur), tm * 32, fa);
        read_frag<RB_, B_KC>(sB(cur), tn * 32, fb);
        if (st + 1 < nstage) {                       // registers hold stage st+1 (issued one full MFMA phase ago)
            pa.sstore(sA(cur ^ 1), ra);
            pb.sstore(sB(cur ^ 1), rb);
        }
        if (st + 2 < nstage) {                       // in flight under the MFMAs below
            pa.gload(lda, (st + 2) * stepA, kbeg + (st + 2) * BK, kend, st + 2 < nfull, ra);
            pb.gload(ldb, (st + 2) * stepB, kbeg + (st + 2) * BK, kend, st + 2 < nfull, rb);
        }
        // (tiles wholly outside C multiply clamped rows too: a branch around the MFMAs made the compiler shuttle the accumulator
        // between AGPRs and VGPRs every stage -- 48 moves -- and those waves wait at the barrier either way)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[q], acc, 0, 0, 0);
        __syncthreads();
    }

    // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int rl = lane & 31, kh = lane >> 5;
    if (use_atomics == 2) ws += (int64_t)zidx * M * N;          // this (batch, split)'s partial product, folded by splitk_reduce
    if (use_atomics != 1 && vec_c) {
        // the block tile goes through LDS and leaves as whole 16-byte row chunks (plain result or split partial); the element-wise
        // path below costs ~500 unrolled instructions and 128-byte store pieces per wave
        constexpr int CS = RB_ + 8;                             // row stride in floats: 4 rows further = 32 banks further
        static_assert(RA * CS <= 2 * (FA + FB), "C staging must fit the operand stages");
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * CS + tn * 32 + rl] = alpha * acc[r];
        __syncthreads();
        constexpr int C4 = RB_ / 4;
        float* dst = use_atomics == 2 ? ws : C;
        const int64_t dld = use_atomics == 2 ? N : ldc;
        const bool plain = use_atomics == 0;
        for (int idx = threadIdx.x; idx < RA * C4; idx += NT) {
            const int r = idx / C4, c4 = idx - r * C4;
            const int64_t row = bm + r, col = bn + c4 * 4;
            if (row >= M || col >= N) continue;
            float4 v = *reinterpret_cast<const float4*>(lds + r * CS + c4 * 4);
            float* p = dst + row * dld + col;
            if (plain) {
                if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            *reinterpret_cast<float4*>(p) = v;
        }
        return;
    }
    const int64_t col = bn + tn * 32 + rl;
    if (col >= N) return;
    const float bv = (bias && split == 0 && use_atomics != 2) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = bm + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= M) continue;
        float v = alpha * acc[r] + bv;
        float* p = C + row * ldc + col;
        if (use_atomics == 2) {
            ws[row * N + col] = v;
        } else if (use_atomics) {
            atomicAdd(p, v);
        } else {
            if (accumulate) v += *p;
            if (relu) v = fmaxf(v, 0.f);
            *p = v;
        }
    }
}

// C (+)= sum over the splits of the partial products a split-contraction launch left in ws[batch][split][M][N] (+ bias).
// Fixed summation order: weight gradients come out reproducible, and the ~1e8/s float-atomic rate of the memory side no longer
// bounds them (a 272 x 444 gradient split 48 ways was 5.8 M atomics = 60 us; the partials are 23 MB of plain stores).
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int64_t M, int64_t N, int nsplit, int nbatch,
        float* __restrict__ C, int64_t ldc, int64_t strideC, int inner, int64_t innerC, const float* __restrict__ bias, int accumulate) {
    const int64_t mn = M * N, total = mn * nbatch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / mn, rem = i - b * mn;
        const int64_t row = rem / N, col = rem - row * N;
        const float* p = ws + b * nsplit * mn + rem;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = 0;
        for (; sp + 4 <= nsplit; sp += 4) {
            s0 += p[(int64_t)sp * mn]; s1 += p[(int64_t)(sp + 1) * mn]; s2 += p[(int64_t)(sp + 2) * mn]; s3 += p[(int64_t)(sp + 3) * mn];
        }
        for (; sp < nsplit; ++sp) s0 += p[(int64_t)sp * mn];
        float v = (s0 + s1) + (s2 + s3);
        if (bias) v += bias[b * N + col];
        float* c = C + (b / inner) * strideC + (b % inner) * innerC + row * ldc + col;
        if (accumulate) v += *c;
        *c = v;
    }
}

// partial-product workspace, one per launch stream (main chain / weight-gradient side stream), grown on demand
struct SplitWs { hipStream_t stream; float* p; size_t floats; };
SplitWs g_split_ws[4] = {};
constexpr size_t kSplitWsMaxFloats = (size_t)256 << 20;       // 1 GiB per stream; larger splits fall back to atomics

// tg_gemm_f32 is reachable from two host threads (the side-stream issuing thread of tg_layer.hip and the caller's): slot claims and
// growth of the per-stream workspaces are serialised
std::mutex g_ws_mutex;

float* split_workspace(size_t need, hipStream_t s) {
    if (need > kSplitWsMaxFloats) return nullptr;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    SplitWs* w = nullptr;
    for (auto& c : g_split_ws) if (c.p && c.stream == s) { w = &c; break; }
    if (!w) for (auto& c : g_split_ws) if (!c.p) { w = &c; w->stream = s; break; }
    if (!w) return nullptr;
    if (need > w->floats) {
        if (w->p) { if (hipStreamSynchronize(s) != hipSuccess) return nullptr; (void)hipFree(w->p); w->p = nullptr; w->floats = 0; }
        const size_t want = std::max<size_t>(need + need / 4, (size_t)8 << 20);
        if (hipMalloc(&w->p, want * sizeof(float)) != hipSuccess) { w->p = nullptr; (void)hipGetLastError(); return nullptr; }
        w->floats = want;
    }
    return w->p;
}

// B (K x N, row-major) -> Bt (N x K): 32 x 32 tiles through LDS.  Lets an `X W` product (input gradient of a Linear whose caller
// did not keep a transposed weight) run on the k-contiguous kernels.
__global__ void __launch_bounds__(256) transpose_kn_kernel(const float* __restrict__ B, int64_t ldb, int64_t K, int64_t N, float* __restrict__ Bt) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t k0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t k = k0 + ty + 8 * q, n = n0 + tx;
        tile[ty + 8 * q][tx] = (k < K && n < N) ? B[k * ldb + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t n = n0 + ty + 8 * q, k = k0 + tx;
        if (n < N && k < K) Bt[n * K + k] = tile[tx][ty + 8 * q];
    }
}
SplitWs g_tr_ws[4] = {};
float* transpose_workspace(size_t need, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    SplitWs* w = nullptr;
    for (auto& c : g_tr_ws) if (c.p && c.stream == s) { w = &c; break; }
    if (!w) for (auto& c : g_tr_ws) if (!c.p) { w = &c; w->stream = s; break; }
    if (!w) return nullptr;
    if (need > w->floats) {
        if (w->p) { if (hipStreamSynchronize(s) != hipSuccess) return nullptr; (void)hipFree(w->p); w->p = nullptr; w->floats = 0; }
        const size_t want = std::max<size_t>(need, (size_t)1 << 20);
        if (hipMalloc(&w->p, want * sizeof(float)) != hipSuccess) { w->p = nullptr; (void)hipGetLastError(); return nullptr; }
        w->floats = want;
    }
    return w->p;
}

struct Args {
    float* ws; int vec_c;
    int64_t M, N, K; float alpha; const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc;
    const float* bias; int relu, accumulate; int64_t k_chunk; int atomics, gx, gy; int64_t sA, sB, sC; int nbatch, splits;
    int inner; int64_t iA, iB, iC;
};

template <bool A_KC, bool B_KC, bool VEC, int TM, int TN>
void launch(const Args& a, hipStream_t s) {
    const unsigned tiles = (unsigned)(a.gx * (a.gy < 0 ? -a.gy : a.gy)), slices = (unsigned)(a.nbatch * a.splits);
    static const bool nopin = getenv("FLID_GEMM_TUNE") != nullptr && getenv("FLID_GEMM_NOPIN") != nullptr;
    const int nslice_x = (!nopin && a.splits > 1 && slices % 8 == 0) ? (int)slices : 0;
    const dim3 grid(nslice_x ? tiles * slices : tiles, 1, nslice_x ? 1 : slices);
    gemm_tile_kernel<A_KC, B_KC, VEC, TM, TN><<<grid, TM * TN * 64, 0, s>>>(a.M, a.N, a.K, a.alpha, a.A, a.lda, a.B, a.ldb, a.C,
        a.ldc, a.bias, a.relu, a.accumulate, a.k_chunk, a.atomics, a.gx, a.gy, a.sA, a.sB, a.sC, a.nbatch, a.inner, a.iA, a.iB, a.iC, a.ws, nslice_x, a.vec_c);
}

template <bool A_KC, bool B_KC>
void dispatch(bool vec, int tm, int tn, const Args& a, hipStream_t s) {
    if (!vec) { launch<A_KC, B_KC, false, 1, 2>(a, s); return; }
    if (tm == 4) {
        if (tn == 2) launch<A_KC, B_KC, true, 4, 2>(a, s);
        else launch<A_KC, B_KC, true, 4, 1>(a, s);
    } else if (tm == 2) {
        if (tn == 4) launch<A_KC, B_KC, true, 2, 4>(a, s);
        else if (tn == 3) launch<A_KC, B_KC, true, 2, 3>(a, s);
        else launch<A_KC, B_KC, true, 2, 2>(a, s);
    } else {
        if (tn == 4) launch<A_KC, B_KC, true, 1, 4>(a, s);
        else if (tn == 3) launch<A_KC, B_KC, true, 1, 3>(a, s);
        else launch<A_KC, B_KC, true, 1, 2>(a, s);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int gemm_impl(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda, int64_t strideA,
              const float* d_B, int64_t ldb, int64_t strideB, float* d_C, int64_t ldc, int64_t strideC, int nbatch,
              const float* d_bias, int relu, int accumulate, hipStream_t s, int inner = 1, int64_t innerA = 0, int64_t innerB = 0,
              int64_t innerC = 0, const float* d_mask = nullptr, int64_t ldm = 0) {
    TG_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nbatch >= 1, "tg_gemm_f32: negative size");
    if (M == 0 || N == 0) return TG_OK;
    TG_REQUIRE(d_A && d_B && d_C, "tg_gemm_f32: null pointer");
    TG_REQUIRE(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, "tg_gemm_f32: leading dimension too small");

    // A panel: k-contiguous when A is M x K (not transposed).  B panel: k-contiguous when B is given as N x K (tb).
    const bool a_kc = !ta, b_kc = tb != 0;
    // X W with a small W (K x N): transpose W once into a per-stream scratch and take the k-contiguous kernels (direct / split-bf16):
    // the f32-input tile kernel runs such products at 131 TFLOP/s at best, the split-bf16 one at 230 (DyGFormer's 38 400-row
    // input gradients, TGN's GRU).  Not for batched calls, nor when W is as large as the activations.
    if (gemm_mode() >= 1 && a_kc && !b_kc && inner == 1 && nbatch == 1 && alpha == 1.f && K % 4 == 0 && M >= 4 * N && K * N <= ((int64_t)4 << 20)) {
        if (float* bt = transpose_workspace((size_t)K * N, s)) {
            transpose_kn_kernel<<<dim3((unsigned)((N + 31) / 32), (unsigned)((K + 31) / 32)), 256, 0, s>>>(d_B, ldb, K, N, bt);
            return gemm_impl(0, 1, M, N, K, alpha, d_A, lda, 0, bt, K, 0, d_C, ldc, 0, 1, d_bias, relu, accumulate, s);
        }
    }
    static const bool no_direct = getenv("FLID_GEMM_TUNE") != nullptr && getenv("FLID_GEMM_NODIRECT") != nullptr;
    // few rows (the root layer's 2 x batch): exact fp32, contraction split over the 4 waves of a workgroup (tg_gemm_direct.hip)
    if (!no_direct && a_kc && inner == 1 && alpha == 1.f && (b_kc || !d_mask) &&
        tg::gemm_direct_nt(M, N, K, d_A, lda, strideA, d_B, ldb, strideB, d_C, ldc, strideC, nbatch, d_bias, relu, accumulate, s, d_mask, ldm, b_kc))
        return tg::launch_status("gemm_direct_nt_kernel");
    if (gemm_mode() >= 1 && a_kc && b_kc && inner == 1 && alpha == 1.f &&
        tg::gemm_bf16x3_nt(M, N, K, d_A, lda, strideA, d_B, ldb, strideB, d_C, ldc, strideC, nbatch, d_bias, relu, accumulate, s, d_mask, ldm))
        return tg::launch_status("gemm_bf16x3_nt_kernel");
    TG_REQUIRE(!d_mask, "tg_gemm_f32_nt_masked: operands must be 16-byte aligned with leading dimensions / K multiples of 4");
    bool vec = al16(d_A) && al16(d_B) && lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0 &&
               innerA % 4 == 0 && innerB % 4 == 0;
    vec = vec && (a_kc ? K % 4 == 0 : M % 4 == 0) && (b_kc ? K % 4 == 0 : N % 4 == 0);

    // Workgroup = 4 waves = 64 x 64 block tile (2 x 2 tiles of 32 x 32).  Measured on MI355X (tools/gemm_bench2.py,
    // M = 131072 / 12235, N = K = 256): 64x64 96 / 72 TFLOP/s, 32x128 84 / 70, 64x128 91 / 62, and every 3- or 6-wave shape
    // (32x96, 64x96) 42..69 -- waves per workgroup should divide evenly over the CU's 4 SIMDs.  Ragged edges cost only the idle
    // wave slots of the edge blocks (tiles wholly outside C skip their MFMAs).
    int tn = 2, tm = vec ? 2 : 1;
    static const bool tuning = getenv("FLID_GEMM_TUNE") != nullptr;      // env overrides are read only in tuning mode (tools/)
    if (tuning) {
        if (const char* e = getenv("FLID_GEMM_TM")) { const int v = atoi(e); if (vec && (v == 1 || v == 2 || v == 4)) tm = v; }
        if (const char* e = getenv("FLID_GEMM_TN")) { const int v = atoi(e); if (vec && v >= 1 && v <= 4) tn = v; }
    }
    if (tm == 4 && tn > 2) tn = 2;
    if (tm != 4 && tn == 1) tn = 2;
    const int64_t gx = (N + 32 * tn - 1) / (32 * tn), gy = (M + 32 * tm - 1) / (32 * tm);
    TG_REQUIRE(gx * gy < (int64_t)1 << 30, "tg_gemm_f32: grid too large");

    // Split the contraction only for weight-gradient shapes (A^T: K = number of rows, small output) so that forward products
    // stay bitwise reproducible; the partial products are folded with float atomics.
    int64_t splits = 1;
    // tiles the split is sized for: the split-bf16 weight-gradient kernel (mode 2) works on 128 x 96 / 128 x 64 tiles
    // mode 1 takes it for the larger outputs only (dP 888 x 172: 71 -> 57 us, dV 272 x 888: 102 -> 84 us; 172 x 272 and smaller lose)
    const bool tn_bf16 = (gemm_mode() == 2 || (gemm_mode() == 1 && M * N >= 65536)) && ta && !tb && inner == 1 && alpha == 1.f && !d_bias &&
                         !relu && M % 4 == 0 && N % 4 == 0;
    const int64_t tiles_for_split = tn_bf16 ? ((M + 127) / 128) * std::min((N + 95) / 96, (N + 63) / 64) : gx * gy;
    if (!relu && ta && tiles_for_split * nbatch < 512 && K >= 2 * BK) {
        // ~2 workgroups per CU, each with at least 8 K-stages: many short slices would only multiply the atomic traffic onto
        // a small output (a 172 x 172 gradient split 131 ways spent 48 us; 57 ways ...)
        const char* e = tuning ? getenv("FLID_GEMM_SPLIT_BLOCKS") : nullptr;
        const int64_t target = e ? atoi(e) : 512;
        splits = (target + tiles_for_split * nbatch - 1) / (tiles_for_split * nbatch);
        const char* e2 = tuning ? getenv("FLID_GEMM_MIN_STAGES") : nullptr;
        const int64_t min_stages = e2 ? atoi(e2) : 8;
        const int64_t max_splits = (K + min_stages * BK - 1) / (min_stages * BK);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    int64_t k_chunk = (K + splits - 1) / splits;
    k_chunk = (k_chunk + BK - 1) / BK * BK;
    if (k_chunk < BK) k_chunk = BK;
    splits = K == 0 ? 1 : (K + k_chunk - 1) / k_chunk;
    // slices (batch x split) in multiples of 8 so that each can be pinned to an XCD; surplus splits past K are empty (zero partials)
    if (splits > 1 && (splits * nbatch) % 8 != 0) {
        int64_t up = splits;
        while ((up * nbatch) % 8 != 0) ++up;
        if (tn_bf16) {
            // the split-bf16 weight-gradient kernel REQUIRES a multiple of 8 slices: re-cut K evenly over the rounded count (the row
            // count of a layer changes every step; 18 slices used to fall back to the f32-input tile kernel -- 102 us instead of 35
            // for the 272 x 888 gradient -- while 16 or 24 did not)
            splits = up;
            k_chunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
            if (k_chunk < BK) k_chunk = BK;
        } else if (up - splits <= 3) {
            splits = up;
        }
    }
    TG_REQUIRE(splits * nbatch <= 65535, "tg_gemm_f32: too many splits");
    // split contraction: partial products into a workspace + one fixed-order fold; float atomics only if no workspace is to be had
    int atomics = splits > 1;
    float* ws = nullptr;
    static const bool force_atomics = tuning && getenv("FLID_GEMM_ATOMICS") != nullptr;
    if (atomics && !force_atomics && (ws = split_workspace((size_t)nbatch * splits * M * N, s)) != nullptr) atomics = 2;
    if (atomics == 1 && !accumulate)
        for (int b = 0; b < nbatch; ++b)
            TG_HIP_CHECK(hipMemset2DAsync(d_C + (b / inner) * strideC + (b % inner) * innerC, ldc * sizeof(float), 0,
                                          N * sizeof(float), M, s));

    static const bool skip_launch = tuning && getenv("FLID_GEMM_SKIP") != nullptr;   // timing experiment: host cost without the kernel
    if (skip_launch) return TG_OK;
    tg::ProfScope prof("gemm", 2.0 * M * N * K * nbatch, s);
    const bool noswz = tuning && getenv("FLID_GEMM_NOSWZ") != nullptr;
    // 16-byte row chunks of C (or of the split workspace) are stored whole when every chunk lies inside the matrix and is aligned
    const int vec_c = N % 4 == 0 && (atomics == 2 || (ldc % 4 == 0 && strideC % 4 == 0 && innerC % 4 == 0 && al16(d_C) && (!d_bias || al16(d_bias))));
    const Args a{ws, vec_c, M, N, K, alpha, d_A, lda, d_B, ldb, d_C, ldc, d_bias, relu, accumulate, k_chunk, atomics, (int)gx, noswz ? -(int)gy : (int)gy,
                 strideA, strideB, strideC, nbatch, (int)splits, inner, innerA, innerB, innerC};
    // mode 2: the weight-gradient form (A^T B, split contraction) on the split-bf16 kernel, same workspace and fold
    if (tn_bf16 && atomics == 2 &&
        tg::gemm_bf16x3_tn_partials(M, N, K, d_A, lda, strideA, d_B, ldb, strideB, ws, nbatch, (int)splits, k_chunk, s)) {
        // (partials are in place)
    } else
    if (a_kc && b_kc) dispatch<true, true>(vec, tm, tn, a, s);
    else if (a_kc && !b_kc) dispatch<true, false>(vec, tm, tn, a, s);
    else if (!a_kc && b_kc) dispatch<false, true>(vec, tm, tn, a, s);
    else dispatch<false, false>(vec, tm, tn, a, s);
    if (atomics == 2) {
        const int64_t total = (int64_t)nbatch * M * N;
        const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
        splitk_reduce_kernel<<<blocks, 256, 0, s>>>(ws, M, N, (int)splits, nbatch, d_C, ldc, strideC, inner, innerC, d_bias, accumulate);
    }
    return tg::launch_status("gemm_tile_kernel");
}

}  // namespace

extern "C" int tg_gemm_f32(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda,
                           const float* d_B, int64_t ldb, float* d_C, int64_t ldc, const float* d_bias, int relu,
                           int accumulate, void* stream) {
    return gemm_impl(ta, tb, M, N, K, alpha, d_A, lda, 0, d_B, ldb, 0, d_C, ldc, 0, 1, d_bias, relu, accumulate, (hipStream_t)stream);
}

extern "C" int tg_gemm_f32_batched(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda,
                                   int64_t stride_a, const float* d_B, int64_t ldb, int64_t stride_b, float* d_C, int64_t ldc,
                                   int64_t stride_c, int batch, const float* d_bias, int relu, int accumulate, void* stream) {
    return gemm_impl(ta, tb, M, N, K, alpha, d_A, lda, stride_a, d_B, ldb, stride_b, d_C, ldc, stride_c, batch, d_bias, relu,
                     accumulate, (hipStream_t)stream);
}

extern "C" int tg_gemm_f32_batched2(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda,
                                    int64_t outer_a, int64_t inner_a, const float* d_B, int64_t ldb, int64_t outer_b,
                                    int64_t inner_b, float* d_C, int64_t ldc, int64_t outer_c, int64_t inner_c, int outer, int inner,
                                    int accumulate, void* stream) {
    TG_REQUIRE(outer >= 1 && inner >= 1 && (int64_t)outer * inner <= 65535, "tg_gemm_f32_batched2: batch counts");
    return gemm_impl(ta, tb, M, N, K, alpha, d_A, lda, outer_a, d_B, ldb, outer_b, d_C, ldc, outer_c, outer * inner, nullptr, 0,
                     accumulate, (hipStream_t)stream, inner, inner_a, inner_b, inner_c);
}

extern "C" int tg_gemm_f32_nt_masked(int64_t M, int64_t N, int64_t K, const float* d_A, int64_t lda, const float* d_B, int64_t ldb, float* d_C,
                                     int64_t ldc, const float* d_Y, int64_t ldy, void* stream) {
    TG_REQUIRE(d_Y && ldy >= N, "tg_gemm_f32_nt_masked: mask matrix");
    return gemm_impl(0, 1, M, N, K, 1.f, d_A, lda, 0, d_B, ldb, 0, d_C, ldc, 0, 1, nullptr, 0, 0, (hipStream_t)stream, 1, 0, 0, 0, d_Y, ldy);
}

namespace tg {
// Two products of the same M x N with k-contiguous operands (X W^T + b) as one launch where both would take the split-bf16 tile kernel
// anyway (default product mode, more tiles than the direct kernel serves); false = the caller issues them one by one
bool gemm_pair_nt(int64_t M, int64_t N, int64_t K1, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, float* C1, const float* bias1,
                  int64_t K2, const float* A2, int64_t lda2, const float* B2, int64_t ldb2, float* C2, const float* bias2, int64_t ldc, hipStream_t s) {
    if (gemm_mode() < 1 || ((M + 31) / 32) * ((N + 31) / 32) <= 800) return false;
    return gemm_bf16x3_nt_pair(M, N, K1, A1, lda1, B1, ldb1, C1, bias1, K2, A2, lda2, B2, ldb2, C2, bias2, ldc, s);
}
}  // namespace tg

extern "C" int tg_wgrad_group(int njobs, const tg_wgrad_job* jobs, int64_t rows, void* stream) {
    TG_REQUIRE(jobs && njobs >= 1 && njobs <= 8 && rows >= 0, "tg_wgrad_group: arguments");
    if (rows == 0) return TG_OK;
    if (!tg::wgrad_group2(njobs, jobs, rows, (hipStream_t)stream) && !tg::wgrad_group(njobs, jobs, rows, (hipStream_t)stream)) {
        tg::set_error("tg_wgrad_group: shape / alignment not covered (M, N, lda, ldb multiples of 4, 16-byte aligned operands)");
        return TG_ESHAPE;
    }
    return tg::launch_status("gemm_bf16x3_wgrad_kernel");
}

extern "C" void tg_set_gemm_mode(int mode) { g_gemm_mode = mode; }
extern "C" int tg_get_gemm_mode(void) { return gemm_mode(); }
extern "C" void tg_set_gemm_mode_thread(int mode) { t_gemm_mode = mode; }
extern "C" int tg_get_gemm_mode_thread(void) { return t_gemm_mode; }
