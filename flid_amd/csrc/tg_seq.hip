// Sequence-side kernels of the DyGFormer path (models/DyGFormer.py): neighbor co-occurrence counts, erf-GELU, row softmax,
// counter-based dropout, per-side patch means.  Dense products run on tg_gemm_f32(_batched).
//   tg_cooccurrence      <- models/DyGFormer.py:337-393  count_nodes_appearances (np.unique per row + python lambda)
//   tg_gelu_*            <- F.gelu (erf form) in models/DyGFormer.py:458
//   tg_softmax_*         <- softmax inside nn.MultiheadAttention (models/DyGFormer.py:454)
//   tg_dropout           <- nn.Dropout (models/DyGFormer.py:456-460), mask regenerated from (seed, index) in backward
//   tg_segment_mean_*    <- torch.mean over each side's patches (models/DyGFormer.py:185-187)
#include <math.h>
#include <initializer_list>

#include "tg_common.h"

namespace {

// one thread per (row, slot): counts of seq_a[row][slot] inside seq_a[row][:] and seq_b[row][:]  (widths <= a few dozen)
// (both sides in one launch: work items [0, n wa) are the source slots, the rest the destination slots)
__global__ void __launch_bounds__(256) cooc_kernel(const int32_t* __restrict__ a0, int64_t lda0, int wa0, const int32_t* __restrict__ b0,
        int64_t ldb0, int wb0, int64_t n, float* __restrict__ out_a /* (n, wa, 2) */, float* __restrict__ out_b /* (n, wb, 2) */) {
    const int64_t na = n * wa0, total = na + n * wb0;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += (int64_t)gridDim.x * blockDim.x) {
        const bool first = i0 < na;
        const int64_t i = first ? i0 : i0 - na;
        const int32_t* a = first ? a0 : b0;
        const int32_t* b = first ? b0 : a0;
        const int64_t lda = first ? lda0 : ldb0, ldb = first ? ldb0 : lda0;
        const int wa = first ? wa0 : wb0, wb = first ? wb0 : wa0;
        float* out = first ? out_a : out_b;
        const int64_t r = i / wa;
        const int s = (int)(i - r * wa);
        const int32_t v = a[r * lda + s];
        int ca = 0, cb = 0;
        if (v != 0) {                                   // padded id 0 counts as 0 (DyGFormer.py:387-391)
            for (int j = 0; j < wa; ++j) ca += a[r * lda + j] == v;
            for (int j = 0; j < wb; ++j) cb += b[r * ldb + j] == v;
        }
        // column order follows the reference: source rows are [in_src, in_dst], destination rows are [in_src, in_dst] too
        out[i * 2 + 0] = (float)(first ? ca : cb);
        out[i * 2 + 1] = (float)(first ? cb : ca);
    }
}

__global__ void __launch_bounds__(256) gelu_fwd_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    }
}
__global__ void __launch_bounds__(256) gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n,
                                                       float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}

// one wave per row, cols <= 64 * MAXC.  key_ids (optional, (n / rows_per_batch, cols) int32): column c of the rows of batch b is
// masked out (its score reads as -inf) where key_ids[b, c] == 0 -- nn.MultiheadAttention's key_padding_mask as the reference's
// TransformerEncoder builds it from the neighbor ids (models/modules.py:297-303).  A row whose keys are all masked is NaN, as there.
template <int MAXC>
__global__ void __launch_bounds__(256) softmax_fwd_kernel(const float* __restrict__ x, int64_t n, int cols, float* __restrict__ y,
                                                          const int32_t* __restrict__ key_ids, int64_t rows_per_batch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        float v[MAXC];
        float m = -INFINITY;
        const int32_t* km = key_ids ? key_ids + (r / rows_per_batch) * cols : nullptr;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = (c < cols && !(km && km[c] == 0)) ? x[r * cols + c] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = tg::wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) { v[i] = (lane + 64 * i) < cols ? expf(v[i] - m) : 0.f; s += v[i]; }
        s = tg::wave_sum(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) { const int c = lane + 64 * i; if (c < cols) y[r * cols + c] = v[i] * inv; }
    }
}
template <int MAXC>
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, int64_t n,
                                                          int cols, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        float p[MAXC], g[MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            p[i] = c < cols ? y[r * cols + c] : 0.f;
            g[i] = c < cols ? dy[r * cols + c] : 0.f;
            s = fmaf(p[i], g[i], s);
        }
        s = tg::wave_sum(s);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) { const int c = lane + 64 * i; if (c < cols) dx[r * cols + c] = p[i] * (g[i] - s); }
    }
}

// Element-wise passes over long tensors (38 400 x 800 in a DyGFormer block): four elements per lane and trip as 16-byte accesses, two
// trips in flight, ONE 64-bit hash for the four dropout decisions (tg::res_keep_scale4: the mask of every dropout of the sequence
// models -- tg_dropout, the fused passes below, the LayerNorm passes of tg_rowops.hip, the attention core of tg_seqattn.hip -- is that
// function of (seed, flat index)).  One element per lane and trip with a hash each ran gelu + dropout over 30.7 M elements in 62 us
// (4 TB/s, VALU- and latency-bound).  n % 4 != 0 or unaligned pointers: the element form.
template <class F4, class F1>
__device__ __forceinline__ void ew_loop(int64_t n, bool vec, F4 f4, F1 f1) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        int64_t i = tid;
        for (; i + nth < n4; i += 2 * nth) f4(i, i + nth);
        if (i < n4) f4(i, i);
    } else {
        for (int64_t i = tid; i < n; i += nth) f1(i);
    }
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float v) {
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    return cdf + v * pdf;
}
__device__ __forceinline__ float4 ld4(const float* p, int64_t i4) { return reinterpret_cast<const float4*>(p)[i4]; }
__device__ __forceinline__ void st4(float* p, int64_t i4, const float4& v) { reinterpret_cast<float4*>(p)[i4] = v; }

__global__ void __launch_bounds__(256) dropout_kernel(const float* __restrict__ x, int64_t n, float p, uint64_t seed, float* __restrict__ y, bool vec) {
    auto one = [&](int64_t i4) {
        const float4 v = ld4(x, i4);
        float k[4];
        tg::res_keep_scale4(seed, 4 * i4, p, k);
        return make_float4(v.x * k[0], v.y * k[1], v.z * k[2], v.w * k[3]);
    };
    ew_loop(n, vec, [&](int64_t a, int64_t b) { const float4 ra = one(a), rb = one(b); st4(y, a, ra); st4(y, b, rb); },
            [&](int64_t i) { y[i] = x[i] * tg::res_keep_scale(seed, i, p); });
}

// Fused element-wise passes of a transformer block (models/DyGFormer.py:448-461): each replaces two launches over the same elements.
//   gelu_dropout_fwd:  y = dropout(gelu(x))          gelu_dropout_bwd:  dx = gelu'(x) * dropout(dy)       (same mask: hash of seed, index)
//   dropout_add:       y = res + dropout(x)
__global__ void __launch_bounds__(256) gelu_dropout_fwd_kernel(const float* __restrict__ x, int64_t n, float p, uint64_t seed, float* __restrict__ y, bool vec) {
    auto one = [&](int64_t i4) {
        const float4 v = ld4(x, i4);
        float k[4];
        tg::res_keep_scale4(seed, 4 * i4, p, k);
        return make_float4(gelu_f(v.x) * k[0], gelu_f(v.y) * k[1], gelu_f(v.z) * k[2], gelu_f(v.w) * k[3]);
    };
    ew_loop(n, vec, [&](int64_t a, int64_t b) { const float4 ra = one(a), rb = one(b); st4(y, a, ra); st4(y, b, rb); },
            [&](int64_t i) { y[i] = gelu_f(x[i]) * tg::res_keep_scale(seed, i, p); });
}
__global__ void __launch_bounds__(256) gelu_dropout_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n, float p,
                                                               uint64_t seed, float* __restrict__ dx, bool vec) {
    auto one = [&](int64_t i4) {
        const float4 v = ld4(x, i4), d = ld4(dy, i4);
        float k[4];
        tg::res_keep_scale4(seed, 4 * i4, p, k);
        return make_float4(d.x * k[0] * dgelu_f(v.x), d.y * k[1] * dgelu_f(v.y), d.z * k[2] * dgelu_f(v.z), d.w * k[3] * dgelu_f(v.w));
    };
    ew_loop(n, vec, [&](int64_t a, int64_t b) { const float4 ra = one(a), rb = one(b); st4(dx, a, ra); st4(dx, b, rb); },
            [&](int64_t i) { dx[i] = dy[i] * tg::res_keep_scale(seed, i, p) * dgelu_f(x[i]); });
}
__global__ void __launch_bounds__(256) dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ res, int64_t n, float p,
                                                          uint64_t seed, float* __restrict__ y, bool vec) {
    auto one = [&](int64_t i4) {
        const float4 v = ld4(x, i4), r = ld4(res, i4);
        float k[4];
        tg::res_keep_scale4(seed, 4 * i4, p, k);
        return make_float4(r.x + v.x * k[0], r.y + v.y * k[1], r.z + v.z * k[2], r.w + v.w * k[3]);
    };
    ew_loop(n, vec, [&](int64_t a, int64_t b) { const float4 ra = one(a), rb = one(b); st4(y, a, ra); st4(y, b, rb); },
            [&](int64_t i) { y[i] = res[i] + x[i] * tg::res_keep_scale(seed, i, p); });
}

// x: (n, s, d); out[i, :] = mean_{j in [lo, hi)} x[i, j, :]
__global__ void __launch_bounds__(256) segment_mean_fwd_kernel(const float* __restrict__ x, int64_t n, int s, int d, int lo, int hi,
                                                               float* __restrict__ out) {
    const int64_t total = n * d;
    const float inv = 1.f / (float)(hi - lo);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        float acc = 0.f;
        for (int j = lo; j < hi; ++j) acc += x[(r * s + j) * d + c];
        out[i] = acc * inv;
    }
}
// dx[i, j, :] (j in [lo,hi)) = dout[i, :] / (hi - lo); other positions untouched
__global__ void __launch_bounds__(256) segment_mean_bwd_kernel(const float* __restrict__ dout, int64_t n, int s, int d, int lo, int hi,
                                                               float* __restrict__ dx) {
    const int64_t total = n * (hi - lo) * d;
    const float inv = 1.f / (float)(hi - lo);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const int64_t rj = i / d;
        const int j = lo + (int)(rj % (hi - lo));
        const int64_t r = rj / (hi - lo);
        dx[(r * s + j) * d + c] = dout[r * d + c] * inv;
    }
}

inline bool vec_ok(int64_t n, std::initializer_list<const void*> ps) {
    if (n & 3) return false;
    for (const void* q : ps) if (reinterpret_cast<uintptr_t>(q) & 15) return false;
    return true;
}
inline unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, tg::kMaxGridBlocks)); }

}  // namespace

extern "C" int tg_cooccurrence(const int32_t* d_a, int64_t lda, int wa, const int32_t* d_b, int64_t ldb, int wb, int64_t n,
                               float* d_out_a, float* d_out_b, void* stream) {
    TG_REQUIRE(d_a && d_b && d_out_a && d_out_b && wa > 0 && wb > 0 && n >= 0, "tg_cooccurrence: arguments");
    if (n == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    cooc_kernel<<<ew_grid(n * (wa + wb)), 256, 0, s>>>(d_a, lda, wa, d_b, ldb, wb, n, d_out_a, d_out_b);
    return tg::launch_status("cooc_kernel");
}

extern "C" int tg_gelu_fwd(const float* d_x, int64_t n, float* d_y, void* stream) {
    TG_REQUIRE(d_x && d_y && n >= 0, "tg_gelu_fwd: arguments");
    if (n == 0) return TG_OK;
    gelu_fwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(d_x, n, d_y);
    return tg::launch_status("gelu_fwd_kernel");
}
extern "C" int tg_gelu_bwd(const float* d_x, const float* d_dy, int64_t n, float* d_dx, void* stream) {
    TG_REQUIRE(d_x && d_dy && d_dx && n >= 0, "tg_gelu_bwd: arguments");
    if (n == 0) return TG_OK;
    gelu_bwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(d_x, d_dy, n, d_dx);
    return tg::launch_status("gelu_bwd_kernel");
}

static int softmax_fwd_impl(const float* d_x, int64_t n, int cols, float* d_y, const int32_t* d_key_ids, int64_t rows_per_batch, void* stream) {
    if (n == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, tg::kMaxGridBlocks));
    if (cols <= 64) softmax_fwd_kernel<1><<<g, 256, 0, s>>>(d_x, n, cols, d_y, d_key_ids, rows_per_batch);
    else if (cols <= 256) softmax_fwd_kernel<4><<<g, 256, 0, s>>>(d_x, n, cols, d_y, d_key_ids, rows_per_batch);
    else softmax_fwd_kernel<16><<<g, 256, 0, s>>>(d_x, n, cols, d_y, d_key_ids, rows_per_batch);
    return tg::launch_status("softmax_fwd_kernel");
}
extern "C" int tg_softmax_fwd(const float* d_x, int64_t n, int cols, float* d_y, void* stream) {
    TG_REQUIRE(d_x && d_y && n >= 0 && cols > 0 && cols <= 1024, "tg_softmax_fwd: cols must be in 1..1024");
    return softmax_fwd_impl(d_x, n, cols, d_y, nullptr, 1, stream);
}
extern "C" int tg_softmax_keymask_fwd(const float* d_x, int64_t n, int cols, const int32_t* d_key_ids, int64_t rows_per_batch, float* d_y,
                                      void* stream) {
    TG_REQUIRE(d_x && d_y && d_key_ids && n >= 0 && cols > 0 && cols <= 1024, "tg_softmax_keymask_fwd: cols must be in 1..1024");
    TG_REQUIRE(rows_per_batch > 0 && n % rows_per_batch == 0, "tg_softmax_keymask_fwd: n must be a multiple of rows_per_batch");
    return softmax_fwd_impl(d_x, n, cols, d_y, d_key_ids, rows_per_batch, stream);
}
extern "C" int tg_softmax_bwd(const float* d_y, const float* d_dy, int64_t n, int cols, float* d_dx, void* stream) {
    TG_REQUIRE(d_y && d_dy && d_dx && n >= 0 && cols > 0 && cols <= 1024, "tg_softmax_bwd: cols must be in 1..1024");
    if (n == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, tg::kMaxGridBlocks));
    if (cols <= 64) softmax_bwd_kernel<1><<<g, 256, 0, s>>>(d_y, d_dy, n, cols, d_dx);
    else if (cols <= 256) softmax_bwd_kernel<4><<<g, 256, 0, s>>>(d_y, d_dy, n, cols, d_dx);
    else softmax_bwd_kernel<16><<<g, 256, 0, s>>>(d_y, d_dy, n, cols, d_dx);
    return tg::launch_status("softmax_bwd_kernel");
}

extern "C" int tg_dropout(const float* d_x, int64_t n, float p, uint64_t seed, float* d_y, void* stream) {
    TG_REQUIRE(d_x && d_y && n >= 0 && p >= 0.f && p < 1.f, "tg_dropout: arguments");
    if (n == 0) return TG_OK;
    dropout_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(d_x, n, p, seed, d_y, vec_ok(n, {d_x, d_y}));
    return tg::launch_status("dropout_kernel");
}

extern "C" int tg_gelu_dropout_fwd(const float* d_x, int64_t n, float p, uint64_t seed, float* d_y, void* stream) {
    TG_REQUIRE(d_x && d_y && n >= 0 && p >= 0.f && p < 1.f, "tg_gelu_dropout_fwd: arguments");
    if (n == 0) return TG_OK;
    gelu_dropout_fwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(d_x, n, p, seed, d_y, vec_ok(n, {d_x, d_y}));
    return tg::launch_status("gelu_dropout_fwd_kernel");
}
extern "C" int tg_gelu_dropout_bwd(const float* d_x, const float* d_dy, int64_t n, float p, uint64_t seed, float* d_dx, void* stream) {
    TG_REQUIRE(d_x && d_dy && d_dx && n >= 0 && p >= 0.f && p < 1.f, "tg_gelu_dropout_bwd: arguments");
    if (n == 0) return TG_OK;
    gelu_dropout_bwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(d_x, d_dy, n, p, seed, d_dx, vec_ok(n, {d_x, d_dy, d_dx}));
    return tg::launch_status("gelu_dropout_bwd_kernel");
}
extern "C" int tg_dropout_add(const float* d_x, const float* d_res, int64_t n, float p, uint64_t seed, float* d_y, void* stream) {
    TG_REQUIRE(d_x && d_res && d_y && n >= 0 && p >= 0.f && p < 1.f, "tg_dropout_add: arguments");
    if (n == 0) return TG_OK;
    dropout_add_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(d_x, d_res, n, p, seed, d_y, vec_ok(n, {d_x, d_res, d_y}));
    return tg::launch_status("dropout_add_kernel");
}

extern "C" int tg_segment_mean_fwd(const float* d_x, int64_t n, int s, int d, int lo, int hi, float* d_out, void* stream) {
    TG_REQUIRE(d_x && d_out && n >= 0 && s > 0 && d > 0 && 0 <= lo && lo < hi && hi <= s, "tg_segment_mean_fwd: arguments");
    if (n == 0) return TG_OK;
    segment_mean_fwd_kernel<<<ew_grid(n * d), 256, 0, (hipStream_t)stream>>>(d_x, n, s, d, lo, hi, d_out);
    return tg::launch_status("segment_mean_fwd_kernel");
}
extern "C" int tg_segment_mean_bwd(const float* d_dout, int64_t n, int s, int d, int lo, int hi, float* d_dx, void* stream) {
    TG_REQUIRE(d_dout && d_dx && n >= 0 && s > 0 && d > 0 && 0 <= lo && lo < hi && hi <= s, "tg_segment_mean_bwd: arguments");
    if (n == 0) return TG_OK;
    segment_mean_bwd_kernel<<<ew_grid(n * (hi - lo) * d), 256, 0, (hipStream_t)stream>>>(d_dout, n, s, d, lo, hi, d_dx);
    return tg::launch_status("segment_mean_bwd_kernel");
}
