// Dense fp32 GEMM on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bit-exact fmaf chain).
//
// replaces: aten::mm / addmm behind every nn.Linear of the hot path (models/modules.py:54-69, 152-163, 235) and
//           their autograd transposes.
//
// C[M,N] = alpha * op(A) * op(B) (+bias) (+C) (ReLU).  Workgroup = 4 waves stacked along M, tile 128 x (32*TN);
// each wave owns a 32 x (32*TN) strip = TN accumulator tiles of 16 VGPRs.  K advances in steps of 16 through one LDS
// stage; the next stage's global loads are issued before the MFMAs of the current one (register prefetch).
//
// Operand panels in LDS, P[r][kk] (r = row of op(A) / column of op(B), kk = 0..15):
//   k-contiguous source ("KC": A not transposed, B given as N x K): stored [r][kk] with row stride 20 floats so the
//       two ds_read_b128 per lane (k = 8*half + 0..7) are bank-conflict free;
//   row-contiguous source ("MC": A^T, B given as K x N): stored [kk][r] (stride 128), read with ds_read_b32.
// MFMA step s uses k = 8*(lane>>5) + s for BOTH operands, any such bijection of k is a valid contraction order.
#include "tg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int KC_STRIDE = BK + 4;
constexpr int MC_STRIDE = 128;
constexpr int PANEL_FLOATS = BM * KC_STRIDE;   // 2560 >= 16 * 128

// ---- global -> registers ---------------------------------------------------------------------------
// KC panel: ROWS x 16 floats, k contiguous in memory.  256 threads: thread t -> float4 (t&3) of rows (t>>2) + 64 j.
template <int ROWS, bool VEC>
__device__ __forceinline__ void load_kc(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                        int64_t kend, float4 (&r)[2]) {
    const int t = threadIdx.x;
    const int kc = (t & 3) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = (t >> 2) + 64 * j;
        const int64_t row = row0 + rr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < ROWS && row < nrows) {
            const float* p = X + row * ld + k0 + kc;
            if constexpr (VEC) {
                if (k0 + kc < kend) v = *reinterpret_cast<const float4*>(p);
            } else {
                if (k0 + kc + 0 < kend) v.x = p[0];
                if (k0 + kc + 1 < kend) v.y = p[1];
                if (k0 + kc + 2 < kend) v.z = p[2];
                if (k0 + kc + 3 < kend) v.w = p[3];
            }
        }
        r[j] = v;
    }
}
template <int ROWS>
__device__ __forceinline__ void store_kc(float* __restrict__ s, const float4 (&r)[2]) {
    const int t = threadIdx.x;
    const int kc = (t & 3) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = (t >> 2) + 64 * j;
        if (rr < ROWS) *reinterpret_cast<float4*>(s + rr * KC_STRIDE + kc) = r[j];
    }
}

// MC panel: 16 x ROWS floats, panel-row index contiguous in memory.  thread t -> float4 (t&31) of k rows (t>>5) + 8 j.
template <int ROWS, bool VEC>
__device__ __forceinline__ void load_mc(const float* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                        int64_t kend, float4 (&r)[2]) {
    const int t = threadIdx.x;
    const int c = (t & 31) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t kk = k0 + (t >> 5) + 8 * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < ROWS && kk < kend) {
            const float* p = X + kk * ld + row0 + c;
            if constexpr (VEC) {
                if (row0 + c < nrows) v = *reinterpret_cast<const float4*>(p);
            } else {
                if (row0 + c + 0 < nrows) v.x = p[0];
                if (row0 + c + 1 < nrows) v.y = p[1];
                if (row0 + c + 2 < nrows) v.z = p[2];
                if (row0 + c + 3 < nrows) v.w = p[3];
            }
        }
        r[j] = v;
    }
}
template <int ROWS>
__device__ __forceinline__ void store_mc(float* __restrict__ s, const float4 (&r)[2]) {
    const int t = threadIdx.x;
    const int c = (t & 31) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (c < ROWS) *reinterpret_cast<float4*>(s + ((t >> 5) + 8 * j) * MC_STRIDE + c) = r[j];
}

// ---- LDS -> MFMA fragments (8 k-steps of one 32-row tile) --------------------------------------------
template <bool KC>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, int tile_r0, float (&f)[8]) {
    const int lane = threadIdx.x & 63;
    const int rl = lane & 31, kh = lane >> 5;
    if constexpr (KC) {
        const float4 a = *reinterpret_cast<const float4*>(s + (tile_r0 + rl) * KC_STRIDE + kh * 8);
        const float4 b = *reinterpret_cast<const float4*>(s + (tile_r0 + rl) * KC_STRIDE + kh * 8 + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = s[(kh * 8 + q) * MC_STRIDE + tile_r0 + rl];
    }
}

template <bool A_KC, bool B_KC, bool VEC, int TN>
__global__ void __launch_bounds__(256) gemm_kernel(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
        int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
        const float* __restrict__ bias, int relu, int accumulate, int64_t k_chunk, int use_atomics) {
    constexpr int BN = 32 * TN;
    __shared__ __attribute__((aligned(16))) float sA[PANEL_FLOATS];
    __shared__ __attribute__((aligned(16))) float sB[B_KC ? BN * KC_STRIDE : 16 * MC_STRIDE];

    const int64_t bm = (int64_t)blockIdx.y * BM;
    const int64_t bn = (int64_t)blockIdx.x * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * k_chunk;
    const int64_t kend = (kbeg + k_chunk < K) ? kbeg + k_chunk : K;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;

    f32x16 acc[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float4 ra[2], rb[2];
    auto gload = [&](int64_t k0) {
        if constexpr (A_KC) load_kc<BM, VEC>(A, lda, bm, M, k0, kend, ra);
        else load_mc<BM, VEC>(A, lda, bm, M, k0, kend, ra);
        if constexpr (B_KC) load_kc<BN, VEC>(B, ldb, bn, N, k0, kend, rb);
        else load_mc<BN, VEC>(B, ldb, bn, N, k0, kend, rb);
    };

    if (kbeg < kend) gload(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();   // everyone finished reading the previous stage
        if constexpr (A_KC) store_kc<BM>(sA, ra); else store_mc<BM>(sA, ra);
        if constexpr (B_KC) store_kc<BN>(sB, rb); else store_mc<BN>(sB, rb);
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);   // in flight under the MFMAs below

        float fa[8], fb[TN][8];
        read_frag<A_KC>(sA, wave * 32, fa);
#pragma unroll
        for (int i = 0; i < TN; ++i) read_frag<B_KC>(sB, i * 32, fb[i]);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < TN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[i][q], acc[i], 0, 0, 0);
    }

    // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int rl = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int64_t col = bn + i * 32 + rl;
        if (col >= N) continue;
        const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = bm + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row >= M) continue;
            float v = alpha * acc[i][r] + bv;
            float* p = C + row * ldc + col;
            if (use_atomics) {
                atomicAdd(p, v);
            } else {
                if (accumulate) v += *p;
                if (relu) v = fmaxf(v, 0.f);
                *p = v;
            }
        }
    }
}


// ---- small problems: one wavefront = one 32x32 output tile over a K slice, operands straight from global (L2-resident) ----
// Used when the 128-row tiling would leave most of the 256 CUs idle (layer-2 shapes: a few hundred to ~1k rows, and their
// weight gradients).  No LDS, no barriers; the contraction is split across blockIdx.z and folded with float atomics.
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(64) gemm_small_kernel(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
        int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
        const float* __restrict__ bias, int relu, int accumulate, int64_t k_chunk, int use_atomics) {
    const int lane = threadIdx.x & 63;
    const int rl = lane & 31, kh = lane >> 5;
    const int64_t row = (int64_t)blockIdx.y * 32 + rl;      // A-operand row owned by this lane
    const int64_t col = (int64_t)blockIdx.x * 32 + rl;      // B-operand column owned by this lane
    const int64_t kbeg = (int64_t)blockIdx.z * k_chunk;
    const int64_t kend = (kbeg + k_chunk < K) ? kbeg + k_chunk : K;
    const bool row_ok = row < M, col_ok = col < N;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        float fa[8], fb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t kk = k0 + kh * 8 + q;
            const bool k_ok = kk < kend;
            fa[q] = (row_ok && k_ok) ? (A_KC ? A[row * lda + kk] : A[kk * lda + row]) : 0.f;
            fb[q] = (col_ok && k_ok) ? (B_KC ? B[col * ldb + kk] : B[kk * ldb + col]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[q], acc, 0, 0, 0);
    }
    if (!col_ok) return;
    const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t orow = (int64_t)blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (orow >= M) continue;
        float v = alpha * acc[r] + bv;
        float* p = C + orow * ldc + col;
        if (use_atomics) {
            atomicAdd(p, v);
        } else {
            if (accumulate) v += *p;
            if (relu) v = fmaxf(v, 0.f);
            *p = v;
        }
    }
}

template <bool A_KC, bool B_KC, bool VEC, int TN>
void launch(dim3 grid, hipStream_t s, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
            int64_t ldb, float* C, int64_t ldc, const float* bias, int relu, int accumulate, int64_t k_chunk, int atomics) {
    gemm_kernel<A_KC, B_KC, VEC, TN><<<grid, 256, 0, s>>>(M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, relu, accumulate,
                                                          k_chunk, atomics);
}

template <bool A_KC, bool B_KC>
void dispatch(bool vec, int tn, dim3 grid, hipStream_t s, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
              int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu, int accumulate,
              int64_t k_chunk, int atomics) {
#define TG_GO(V, T) launch<A_KC, B_KC, V, T>(grid, s, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, relu, accumulate, k_chunk, atomics)
    if (vec) {
        if (tn == 4) TG_GO(true, 4); else if (tn == 3) TG_GO(true, 3); else TG_GO(true, 2);
    } else {
        TG_GO(false, 2);
    }
#undef TG_GO
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int tg_gemm_f32(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda,
                           const float* d_B, int64_t ldb, float* d_C, int64_t ldc, const float* d_bias, int relu,
                           int accumulate, void* stream) {
    TG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "tg_gemm_f32: negative size");
    if (M == 0 || N == 0) return TG_OK;
    TG_REQUIRE(d_A && d_B && d_C, "tg_gemm_f32: null pointer");
    TG_REQUIRE(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, "tg_gemm_f32: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;

    // A panel: k-contiguous when A is M x K (not transposed).  B panel: k-contiguous when B is given as N x K (tb).
    const bool a_kc = !ta, b_kc = tb != 0;
    bool vec = al16(d_A) && al16(d_B) && lda % 4 == 0 && ldb % 4 == 0;
    vec = vec && (a_kc ? K % 4 == 0 : M % 4 == 0) && (b_kc ? K % 4 == 0 : N % 4 == 0);

    // column-tile width: least padded N, ties to the wider tile
    int tn = 2;
    if (vec) {
        int64_t best = -1;
        for (int c : {4, 3, 2}) {
            const int64_t padded = (N + 32 * c - 1) / (32 * c) * (32 * c);
            if (best < 0 || padded < best) { best = padded; tn = c; }
        }
    }
    const int64_t gx = (N + 32 * tn - 1) / (32 * tn), gy = (M + BM - 1) / BM;
    TG_REQUIRE(gy <= 65535 && gx <= 65535, "tg_gemm_f32: grid too large");

    // Small problems (fewer 128-row tiles than half the CUs and a short contraction): one wave per 32x32 tile.
    if (gx * gy < 128 && K < 2048) {
        const int64_t tx = (N + 31) / 32, ty = (M + 31) / 32;
        int64_t splits = 1;
        if (!relu && ta && tx * ty < 256) {   // only weight-gradient shapes (A^T): forward products stay bitwise reproducible
            splits = (512 + tx * ty - 1) / (tx * ty);
            const int64_t max_splits = (K + 31) / 32;
            if (splits > max_splits) splits = max_splits;
            if (splits < 1) splits = 1;
        }
        int64_t k_chunk = (K + splits - 1) / splits;
        k_chunk = (k_chunk + BK - 1) / BK * BK;
        if (k_chunk < BK) k_chunk = BK;
        splits = K == 0 ? 1 : (K + k_chunk - 1) / k_chunk;
        const int atomics = splits > 1;
        if (atomics && !accumulate) TG_HIP_CHECK(hipMemset2DAsync(d_C, ldc * sizeof(float), 0, N * sizeof(float), M, s));
        TG_REQUIRE(ty <= 65535 && tx <= 65535, "tg_gemm_f32: grid too large");
        dim3 grid((unsigned)tx, (unsigned)ty, (unsigned)splits);
#define TG_SMALL(AK, BKC) gemm_small_kernel<AK, BKC><<<grid, 64, 0, s>>>(M, N, K, alpha, d_A, lda, d_B, ldb, d_C, ldc, d_bias, relu, accumulate, k_chunk, atomics)
        if (a_kc && b_kc) TG_SMALL(true, true);
        else if (a_kc && !b_kc) TG_SMALL(true, false);
        else if (!a_kc && b_kc) TG_SMALL(false, true);
        else TG_SMALL(false, false);
#undef TG_SMALL
        return tg::launch_status("gemm_small_kernel");
    }

    // split the contraction when the output has too few tiles to fill 256 CUs (weight-gradient shapes: K = rows)
    int64_t splits = 1;
    if (!relu && ta && gx * gy < 256 && K >= 2048) {
        splits = (512 + gx * gy - 1) / (gx * gy);
        const int64_t max_splits = K / 256;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    int64_t k_chunk = (K + splits - 1) / splits;
    k_chunk = (k_chunk + BK - 1) / BK * BK;
    splits = K == 0 ? 1 : (K + k_chunk - 1) / k_chunk;
    const int atomics = splits > 1;
    if (atomics && !accumulate) TG_HIP_CHECK(hipMemset2DAsync(d_C, ldc * sizeof(float), 0, N * sizeof(float), M, s));

    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)splits);
    if (a_kc && b_kc) dispatch<true, true>(vec, tn, grid, s, M, N, K, alpha, d_A, lda, d_B, ldb, d_C, ldc, d_bias, relu, accumulate, k_chunk, atomics);
    else if (a_kc && !b_kc) dispatch<true, false>(vec, tn, grid, s, M, N, K, alpha, d_A, lda, d_B, ldb, d_C, ldc, d_bias, relu, accumulate, k_chunk, atomics);
    else if (!a_kc && b_kc) dispatch<false, true>(vec, tn, grid, s, M, N, K, alpha, d_A, lda, d_B, ldb, d_C, ldc, d_bias, relu, accumulate, k_chunk, atomics);
    else dispatch<false, false>(vec, tn, grid, s, M, N, K, alpha, d_A, lda, d_B, ldb, d_C, ldc, d_bias, relu, accumulate, k_chunk, atomics);
    return tg::launch_status("gemm_kernel");
}
