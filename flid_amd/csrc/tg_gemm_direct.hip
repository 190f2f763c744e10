// Small-M products C = A B^T (+ bias, relu) for the root layer of the TGAT stack (M = 2 x batch = 1200 rows) and other short
// operands.  With so few rows the LDS-tiled kernels run a handful of workgroups whose K loop is a chain of dependent
// global -> LDS -> MFMA stages: 16..30 us for 0.3 GFLOP (rocprof, r01_v3).  Here the chain is cut instead of pipelined:
//   * one workgroup per 32 x 32 tile of C, its NW = 4 / 8 / 16 waves split the contraction NW ways (fixed order, so results are
//     reproducible): NW is the smallest that gives every wave ONE batch of loads (<= 8 chunks of 8 k) -- a second batch is a second
//     exposed round trip to L2 (K = 272 on 4 waves: 8 + 1 chunks; measured 12.5 us against 8.6 us for K = 172);
//   * no LDS staging: both operands are k-contiguous, so lane (i, h) loads float4s of row i at k = 8c + 4h straight into the
//     A/B operand registers of v_mfma_f32_32x32x2_f32 -- the k order inside a product is free as long as A and B agree;
//   * every load of a wave's slice is issued before its first MFMA (<= 8 chunks = 64 k per batch), the four partial tiles are
//     folded through LDS.
// Exact fp32 (f32-input MFMA).  Operands are L2 resident at these sizes, the kernel is latency-, not bandwidth-bound.
// BT = false: B is given as K x N (n contiguous, "NN" products such as dWq_h += Wk_h dP_h): lane (n, h) then reads the four k of
// its half chunk as four scalar loads, each coalesced over the 32 lanes of a half wave.
#include <stdlib.h>

#include "tg_common.h"

namespace tg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int U = 8;   // chunks (of 8 k) in flight per wave and operand: 16 float4 = 64 VGPRs

template <bool BT, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_direct_nt_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A,
                                                             int64_t lda, int64_t sA, const float* __restrict__ B, int64_t ldb,
                                                             int64_t sB, float* __restrict__ C, int64_t ldc, int64_t sC,
                                                             const float* __restrict__ bias, int relu, int accumulate, int gx, int vec_c,
                                                             const float* __restrict__ mask, int64_t ldm) {
    __shared__ float red[NW][32][36];               // stride 36 floats: 16-byte aligned rows for the fold's float4 reads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t tn = blockIdx.x % gx, tm = blockIdx.x / gx;
    A += blockIdx.y * sA;
    B += blockIdx.y * sB;
    C += blockIdx.y * sC;
    if (bias) bias += blockIdx.y * N;
    int64_t row = tm * 32 + i, col = tn * 32 + i;
    row = row < M ? row : M - 1;                    // clamped rows are loaded but never stored
    col = col < N ? col : N - 1;
    const int64_t k8 = (K + 7) >> 3;                // chunks of 8 k; K % 4 == 0, so each half chunk is wholly in or out
    const int64_t per = (k8 + NW - 1) / NW;
    const int64_t c0 = w * per, c1 = (c0 + per < k8) ? c0 + per : k8;
    const float* ap = A + row * lda + 4 * h;
    const float* bp = BT ? B + col * ldb + 4 * h : B + (int64_t)4 * h * ldb + col;

    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t c = c0; c < c1; c += U) {
        float4 av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = (c + u) * 8 + 4 * h;
            const bool ok = (c + u < c1) && (k < K);
            const int64_t off = ok ? (c + u) * 8 : 0;
            av[u] = *reinterpret_cast<const float4*>(ap + off);
            if (BT) {
                bv[u] = *reinterpret_cast<const float4*>(bp + off);
            } else {
                const float* q = bp + off * ldb;
                bv[u] = make_float4(q[0], q[ldb], q[2 * ldb], q[3 * ldb]);
            }
            if (!ok) { av[u] = make_float4(0.f, 0.f, 0.f, 0.f); bv[u] = av[u]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    }

    // every wave parks its partial tile (C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); then each of
    // the 256 threads folds one 4-column chunk in fixed wave order and writes it as one 16-byte store
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
    __syncthreads();
    if (threadIdx.x >= 256) return;
    const int rr = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const int64_t orow = tm * 32 + rr, ocol = tn * 32 + c4;
    if (orow >= M || ocol >= N) return;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float t = red[0][rr][c4 + q];
#pragma unroll
        for (int x = 1; x < NW; ++x) t += red[x][rr][c4 + q];
        v[q] = t;
    }
    float* p = C + orow * ldc + ocol;
    if (vec_c) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + ocol); o.x += b4.x; o.y += b4.y; o.z += b4.z; o.w += b4.w; }
        if (accumulate) { const float4 c = *reinterpret_cast<const float4*>(p); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        if (mask) {                                   // ReLU backward fused: keep the entries whose forward output was positive
            const float4 y = *reinterpret_cast<const float4*>(mask + orow * ldm + ocol);
            o.x = y.x > 0.f ? o.x : 0.f; o.y = y.y > 0.f ? o.y : 0.f; o.z = y.z > 0.f ? o.z : 0.f; o.w = y.w > 0.f ? o.w : 0.f;
        }
        *reinterpret_cast<float4*>(p) = o;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ocol + q >= N) break;
            float x = v[q] + (bias ? bias[ocol + q] : 0.f);
            if (accumulate) x += p[q];
            if (relu) x = fmaxf(x, 0.f);
            if (mask && !(mask[orow * ldm + ocol + q] > 0.f)) x = 0.f;
            p[q] = x;
        }
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// true = launched (or nothing to do); false = shape not handled here, the caller falls through to the tiled kernels
bool gemm_direct_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                    int64_t strideB, float* C, int64_t ldc, int64_t strideC, int nbatch, const float* bias, int relu, int accumulate,
                    hipStream_t s, const float* mask, int64_t ldm, bool b_kc) {
    if (K <= 0 || K % 4 || lda % 4 || strideA % 4 || !al16(A)) return false;
    if (b_kc ? (ldb % 4 || strideB % 4 || !al16(B)) : K < 8) return false;       // (K x N form: the guard rows 4..7 must exist)
    const int64_t gx = (N + 31) / 32, gy = (M + 31) / 32;
    // the point of this kernel is a chip that would otherwise be mostly idle: beyond ~3 workgroups per CU the tiled split-bf16
    // kernel's operand reuse wins (1 200 x 888 x 136 in 1 064 tiles: 16 us here, ~9 us there; swept 200 .. 2048: 800 is the step's
    // optimum, 0.945 -> 0.931 ms)
    static const int64_t max_tiles = (getenv("FLID_GEMM_TUNE") && getenv("FLID_DIRECT_MAX_TILES")) ? atoll(getenv("FLID_DIRECT_MAX_TILES")) : 800;
    if (gx * gy * nbatch > max_tiles || nbatch > 65535 || K > 4096) return false;
    const int vec_c = N % 4 == 0 && ldc % 4 == 0 && strideC % 4 == 0 && al16(C) && (!bias || al16(bias)) && (!mask || (al16(mask) && ldm % 4 == 0));
    if (mask && nbatch != 1) return false;
    ProfScope prof("gemm", 2.0 * M * N * K * nbatch, s);
    const int64_t k8 = (K + 7) / 8;
    int nw = 4;
    static const int force_nw = (getenv("FLID_GEMM_TUNE") && getenv("FLID_DIRECT_NW")) ? atoi(getenv("FLID_DIRECT_NW")) : 0;
    if (force_nw) nw = force_nw;
    else while (nw < 16 && (k8 + nw - 1) / nw > U) nw *= 2;
    const dim3 grid((unsigned)(gx * gy), (unsigned)nbatch);
#define FLID_DIRECT_LAUNCH(BTV, NWV)                                                                                                 \
    hipLaunchKernelGGL((gemm_direct_nt_kernel<BTV, NWV>), grid, dim3(64 * NWV), 0, s, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, \
                       strideC, bias, relu, accumulate, (int)gx, vec_c, mask, ldm)
    if (b_kc) {
        if (nw == 4) FLID_DIRECT_LAUNCH(true, 4);
        else if (nw == 8) FLID_DIRECT_LAUNCH(true, 8);
        else FLID_DIRECT_LAUNCH(true, 16);
    } else {
        if (nw == 4) FLID_DIRECT_LAUNCH(false, 4);
        else if (nw == 8) FLID_DIRECT_LAUNCH(false, 8);
        else FLID_DIRECT_LAUNCH(false, 16);
    }
#undef FLID_DIRECT_LAUNCH
    return true;
}

}  // namespace tg
