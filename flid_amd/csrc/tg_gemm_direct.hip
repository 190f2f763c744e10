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
#include "tg_tail.h"

namespace tg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int U = 8;   // chunks (of 8 k) in flight per wave and operand: 16 float4 = 64 VGPRs

// (bx, by) = tile number and batch index of the workgroup; r1a / r1b (optional, vec_c form only): C += r1a[row] * r1b[col], a rank-one
// term riding in the epilogue (both indexed inside the batch: + by * M / + by * N)
template <bool BT, int NW>
__device__ __forceinline__ void direct_tile(int bx, int by, float (*red)[32][36], int64_t M, int64_t N, int64_t K, const float* __restrict__ A,
                                            int64_t lda, int64_t sA, const float* __restrict__ B, int64_t ldb, int64_t sB,
                                            float* __restrict__ C, int64_t ldc, int64_t sC, const float* __restrict__ bias, int relu,
                                            int accumulate, int gx, int vec_c, const float* __restrict__ mask, int64_t ldm,
                                            const float* __restrict__ r1a = nullptr, const float* __restrict__ r1b = nullptr) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t tn = bx % gx, tm = bx / gx;
    A += by * sA;
    B += by * sB;
    C += by * sC;
    if (bias) bias += by * N;
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
        if (r1a) {
            const float ra = r1a[by * M + orow];
            const float4 rb = *reinterpret_cast<const float4*>(r1b + by * N + ocol);
            o.x = fmaf(ra, rb.x, o.x); o.y = fmaf(ra, rb.y, o.y); o.z = fmaf(ra, rb.z, o.z); o.w = fmaf(ra, rb.w, o.w);
        }
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

template <bool BT, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_direct_nt_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A,
                                                             int64_t lda, int64_t sA, const float* __restrict__ B, int64_t ldb,
                                                             int64_t sB, float* __restrict__ C, int64_t ldc, int64_t sC,
                                                             const float* __restrict__ bias, int relu, int accumulate, int gx, int vec_c,
                                                             const float* __restrict__ mask, int64_t ldm) {
    __shared__ float red[NW][32][36];               // stride 36 floats: 16-byte aligned rows for the fold's float4 reads
    direct_tile<BT, NW>((int)blockIdx.x, (int)blockIdx.y, red, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, bias, relu, accumulate, gx, vec_c, mask, ldm);
}

// The weight-space end of a merged-projection layer's backward in ONE launch (three launches before: 7 + 9 + 10 us, each a single
// latency chain on a few dozen workgroups).  With dP_h = du_h^T own (dk x dn, from the weight-gradient launch) and dub = sum_rows du:
//   workgroups [0, nA):        dWk_h (hd x dk) += Wq_h[:, :dn] dP_h^T  +  qb_h (x) dub_h      (the rank-one term = the dWk part of ub_bwd)
//   workgroups [nA, nA + nB):  dWq_h[:, :dn] (hd x dn) += Wk_h dP_h
//   the rest:                  ub_bwd without its dWk update: dWq[:, dn:] += dqb (x) cos b, d cos b += Wq[:, dn:]^T dqb
struct WspaceTail {
    const float *Wq, *Wk, *dP, *qb, *dub, *cosb;
    float *dWk, *dWq, *d_cosb;
    int H, hd, dn, dq, dk, T;
    int gxA, nA, gxB, nB;
};
__global__ __launch_bounds__(512) void wspace_tail_kernel(WspaceTail t) {
    __shared__ float red[8][32][36];
    const int bid = (int)blockIdx.x;
    if (bid < t.nA) {
        const int per = t.nA / t.H;
        direct_tile<true, 8>(bid % per, bid / per, red, t.hd, t.dk, t.dn, t.Wq, t.dq, (int64_t)t.hd * t.dq, t.dP, t.dn, (int64_t)t.dk * t.dn,
                             t.dWk, t.dk, (int64_t)t.hd * t.dk, nullptr, 0, 1, t.gxA, 1, nullptr, 0, t.qb, t.dub);
    } else if (bid < t.nA + t.nB) {
        const int b = bid - t.nA, per = t.nB / t.H;
        direct_tile<false, 8>(b % per, b / per, red, t.hd, t.dn, t.dk, t.Wk, t.dk, (int64_t)t.hd * t.dk, t.dP, t.dn, (int64_t)t.dk * t.dn,
                              t.dWq, t.dq, (int64_t)t.hd * t.dq, nullptr, 0, 1, t.gxB, 1, nullptr, 0);
    } else {
        ub_bwd_body(bid - t.nA - t.nB, t.dub, t.qb, t.Wk, t.Wq, t.cosb, t.hd, t.dn, t.dq, t.dk, t.T, nullptr, t.dWq, t.d_cosb, &red[0][0][0]);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// false = shapes / alignment not covered (nothing launched: the caller issues the two products and the tail launch)
bool wspace_tail(const float* Wq, const float* Wk, const float* dP, const float* qb, const float* dub, const float* cosb, float* dWk, float* dWq,
                 float* d_cosb, int H, int hd, int dn, int dq, int dk, int T, hipStream_t s) {
    if (dn % 4 || dk % 4 || dq % 4 || dn < 8 || dk < 8 || dn > 8 * 64 || dk > 8 * 64) return false;       // one batch of <= 8 chunks per wave
    if (!(al16(Wq) && al16(Wk) && al16(dP) && al16(dWk) && al16(dWq) && al16(dub))) return false;
    WspaceTail t{Wq, Wk, dP, qb, dub, cosb, dWk, dWq, d_cosb, H, hd, dn, dq, dk, T, 0, 0, 0, 0};
    t.gxA = (dk + 31) / 32; t.nA = t.gxA * ((hd + 31) / 32) * H;
    t.gxB = (dn + 31) / 32; t.nB = t.gxB * ((hd + 31) / 32) * H;
    const int nC = (dq + UBR - 1) / UBR;
    ProfScope prof("gemm", 4.0 * H * hd * dk * dn, s);
    wspace_tail_kernel<<<(unsigned)(t.nA + t.nB + nC), 512, 0, s>>>(t);
    return true;
}

namespace {
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
