// Shared internals of libflid_tg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "flid_tg.h"

namespace tg {

void set_error(const std::string& s);
const float* zero_block();  // 256 zero bytes in device memory (out-of-range operand loads of the product kernels point here)
std::string get_error();   // this thread's last error text (errors are thread-local: the side-stream issuing thread relays its own)

#define TG_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::tg::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return TG_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

#define TG_REQUIRE(cond, msg)                                                                      \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::tg::set_error(std::string("invalid argument: ") + msg);                              \
            return TG_EINVAL;                                                                      \
        }                                                                                          \
    } while (0)

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        return TG_EHIP;
    }
    return TG_OK;
}

// One incidence of the time-sorted adjacency (16 B): a lookup of the newest k is one contiguous read.
struct __attribute__((aligned(16))) Incidence {
    int32_t nbr;
    int32_t eid;
    double t;
};

// Optional HIP-event timing of kernel families on the stream they are launched on (bench.py roofline leg):
// tg_profile_enable(1); run; tg_profile_collect("attn_bwd", ...).  A disabled scope costs one branch.
struct ProfScope {
    ProfScope(const char* tag, double units, hipStream_t s);
    ~ProfScope();
    const char* tag;
    double units;
    hipEvent_t a, b;     // null when the family is not being timed
    hipStream_t stream;
};

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kMaxGridBlocks = 2048; // 256 CUs x 8 resident blocks: grid-stride beyond this

// attention kernels (tg_attn.hip, tg_attn_fast.hip): 4 instances per workgroup; the grid is also the number of time-encoder
// gradient slabs (tg_attn_bwd_parts), so the generic and the fast kernels share it
constexpr int kAttnMaxBlocks = 4096;
inline int64_t attn_grid_blocks(int64_t m) {
    const int64_t b = (m + 3) / 4;
    return b < 1 ? 1 : (b > kAttnMaxBlocks ? kAttnMaxBlocks : b);
}
// tg_layer.hip: the forward of a step's one or two layers with ONE prelude launch (optionally zero-filling a region in it)
int layers_forward(int n, const tg_layer_desc* const* Ls, float* zero, int64_t zero_floats, void* stream);
// tg_rowops.hip: tg_adam_f32 that first finishes the time-encoder bias gradient (elements [tb_off, tb_off + tb_n) of the flat parameter)
int adam_time_bias(float* d_param, float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int64_t step, int64_t tb_off, int tb_n, const float* d_cosb, void* stream);
// tg_gemm_direct.hip: the weight-space end of a merged-projection layer's backward (two small products + the constant part's gradient)
// as one launch; false = not covered, nothing launched
bool wspace_tail(const float* Wq, const float* Wk, const float* dP, const float* qb, const float* dub, const float* cosb, float* dWk, float* dWq,
                 float* d_cosb, int H, int hd, int dn, int dq, int dk, int T, hipStream_t s);
// tg_pack.hip: packed (split-bf16, MFMA fragment order) weights of the chain kernels
int64_t packed_floats(int N, int K);
int pack_weights(int njobs, const tg_pack_job* jobs, hipStream_t s);
// (stepper-internal fusions of exported launches: tg_rowops.hip, tg_memory.hip)
int scatter_add_rows2(const float* d_src, const float* d_src2, int64_t src_ld, const int32_t* d_idx, int64_t n, int cols, float* d_table, int64_t table_ld,
                      hipStream_t s);
int tgn_persist_index(const float* d_rows, int64_t rows_ld, const int32_t* d_row_of, const int32_t* d_nodes, const int32_t* d_has, const float* d_msg_time,
                      float* d_memory, int64_t mem_ld, float* d_last_update, int64_t count, int d, int32_t* d_last_idx_ws, hipStream_t s);
int msg_scatter_last_indexed(const int32_t* d_nodes, const float* d_msgs, int64_t msg_ld, const float* d_t32, int64_t count, int width, float* d_table,
                             int64_t table_ld, int32_t* d_has, float* d_msg_time, int32_t* d_last_idx_ws, hipStream_t s);
int64_t packed32_floats(int N, int K);
int pack32_weights(int njobs, const tg_pack32_job* jobs, hipStream_t s);
bool gemm_pk_nt(int64_t M, int N, int K, const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, const float* bias, hipStream_t s);
// tg_chain.hip: everything behind the attention of a layer's forward as one launch
bool chain_shape_ok(int H, int dn, int T, int de);
int chain_hp(int H, int dn, int T);
int chain_fwd(const tg_layer_desc* L, const void* pWv, const void* pWr, const void* pW1, const void* pW2, int64_t packed_bytes, hipStream_t s);
int chain_hpb(int H, int dn, int T);
// the query side of a short (not merged) layer as one launch per direction: q -> u, dq -> d_own (tg_chain.hip)
bool qu_shape_ok(int H, int dn, int T, int de);
int qu_fwd(const tg_layer_desc* L, const void* pWq, const void* pWkT, hipStream_t s);
int dq_bwd(const tg_layer_desc* L, const tg_layer_bwd_desc* Bw, const void* pWk, const void* pWqT, float* dq_slab, hipStream_t s);
int64_t chain_blocks(int64_t rows);
int chain_bwd(const tg_layer_desc* L, const tg_layer_bwd_desc* Bw, float* dres, float* part, const void* pW2T, const void* pW1aT,
              const void* pWrT, const void* pWvT, hipStream_t s, const void* pW1bT = nullptr);
// Column sums of tall slab matrices added (float atomics) into up to 6 destination vectors: column c belongs to the first segment with
// c < end[i] and lands at p[i][c - begin_i]; a null p[i] drops the segment (device body: tg_colsum.h)
struct SegDst { float* p[6]; int end[6]; int n; };
struct ColJob { const float* x; int64_t ld, n; int cols; SegDst d; };
// two slab matrices summed by extra workgroups of another launch: a (col_gx x col_ny) grid of 256-thread workgroups, the first groups_a
// grid columns on `a`
// ... and, optionally (wq_n > 0 workgroups more), the time half of a query projection's weight gradient (tg_tail.h wq_time_body), which needs
// nothing of the fold either
struct ColExtra {
    ColJob a, b; int groups_a, col_gx, col_ny;
    int wq_n, wq_gx, wq_dq, wq_T; const float *wq_sq, *wq_cosb, *wq_W; float *wq_dW, *wq_dcosb; int64_t wq_ld;
    int wq_nb;                  // > 0: wq_sq is a slab of wq_nb partial rows (wq_time_slab_body), else the summed vector
};
// tg_wgrad.hip: big tiles + transposing LDS reads + slice fold; `extra` (optional): slab sums that ride in the fold launch
bool wgrad_group2(int njobs, const tg_wgrad_job* jobs, int64_t rows, hipStream_t s, const ColExtra* extra = nullptr);
// the NEXT wgrad_group2 call of this thread is kept back and leaves in the launch of the call after it on the same stream (one product
// launch + one fold for both groups); wgrad_flush_deferred launches a kept-back group that nothing picked up
void wgrad_defer_next(bool on);
int wgrad_flush_deferred(hipStream_t s);
bool wgrad_group(int njobs, const tg_wgrad_job* jobs, int64_t rows, hipStream_t s);   // tg_gemm_bf16x3.hip; false = shapes not covered
// The NEXT tg_attn_bwd of this thread that wants a feature gradient writes it as one row per neighbor slot (slot s of instance r at
// rows + (r k + s) dn, padded slots skipped) instead of adding it into dfeat with float atomics -- for a gradient table whose popular
// rows would serialise the atomics (TGN's compact `memory' + raw` table: tg_step.hip sums the rows with slot_rows_sum).  Only the fast
// backward kernel takes the request; attn_bwd_slot_rows_taken() says whether the last tg_attn_bwd did.
void attn_bwd_slot_rows_next(float* rows);
bool attn_bwd_slot_rows_taken();
// slots grouped by the table row they gather (any order inside a row): order[i] = slot, srow[i] = its row, i < *n_valid (slots with
// nbr == 0 left out); *d_rows (device) = table rows in use, the padding row is *d_rows; cnt: nrows_cap + 1 ints of scratch, rank: n ints
constexpr int64_t kSlotOrderMaxRows = 38000;        // LDS counters of build_slot_order: nrows_cap + 1 + 1 024 <= this
int build_slot_order(const int32_t* slot_row, const int32_t* nbr, int64_t n, int64_t nrows_cap, const int32_t* d_rows, int32_t* cnt, int32_t* rank,
                     int32_t* order, int32_t* srow, int32_t* n_valid, hipStream_t s);
// table[srow[i]] += rows[order[i]] over i < *n_valid: one wave per 16 entries, runs of equal rows summed in registers, one float-atomic
// row add per run
// (own_*: n_own further rows added by extra workgroups of the same launch: table[own_idx[r]] += own_a[r] + own_b[r], rows of dn floats)
int slot_rows_sum(const float* rows, int dn, const int32_t* order, const int32_t* srow, const int32_t* n_valid, int64_t n, float* table,
                  int64_t ld, hipStream_t s, const float* own_a = nullptr, const float* own_b = nullptr, const int32_t* own_idx = nullptr,
                  int64_t n_own = 0);
int attn_fwd_fast(const tg_attn_desc& a, const float* u, float* agg, float* prob, hipStream_t s);          // 1 = shape not covered
int attn_bwd_fast(const tg_attn_desc& a, const float* u, const float* agg, const float* prob, const float* dagg, float* du,
                  float* dfeat, int64_t dfeat_ld, int64_t pad_row, float* dedge, int64_t dedge_ld, float* dte, hipStream_t s, float* slot_rows = nullptr);
// tg_memory.hip: a positive TGN batch's new raw messages, only each node's last one, built straight into the pending-message table
int build_scatter_last(const float* d_mem, int64_t mem_ld, const float* d_last_update, const int32_t* d_a_ids, const int32_t* d_b_ids, const float* d_t32,
                       const float* d_edge, int64_t edge_ld, const int32_t* d_eids, const float* d_te_w, const float* d_te_b, int64_t n, int d, int de, int T,
                       float* d_table, int64_t table_ld, int32_t* d_has, float* d_msg_time, int32_t* d_last_idx_ws, hipStream_t s);
// tg_gemm.hip: two X W^T + b products of the same M x N as one launch (TGN's two GRU gate products); false = issue them one by one
bool gemm_pair_nt(int64_t M, int64_t N, int64_t K1, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, float* C1, const float* bias1,
                  int64_t K2, const float* A2, int64_t lda2, const float* B2, int64_t ldb2, float* C2, const float* bias2, int64_t ldc, hipStream_t s);
// ---- wave64 helpers -------------------------------------------------------------------------------
// DPP lane exchange inside the VALU (no LDS crossbar): quad swaps, 8- and 16-lane mirrors, then the two row broadcasts
// of the GFX9 wave64 reduction; the total lands in lane 63 and is read back as a wave-uniform scalar.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
    return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);         // row_half_mirror
    v = dpp_add<0x140>(v);         // row_mirror  -> every lane holds its 16-lane row sum
    v = dpp_add<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// N independent sums at once, step-major so the DPP latencies overlap
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0xB1>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x4E>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x141>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x140>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x142, 0xA>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x143, 0xC>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
}

// cos / sin of a float32 phase that may reach millions of radians (time deltas up to a month in seconds times w ~ 1).
// The phase is turned into revolutions with a two-term constant 1/(2 pi) = C_HI + C_LO in float32 only:
//   p = fl(phase * C_HI), e = fma(phase, C_HI, -p) (exact rounding error), q = fma(phase, C_LO, e)
//   fraction = (p - rint(p)) + q          -- p - rint(p) is exact, |q| <= ulp(p)
// (error ~1e-8 revolutions for |phase| < 2^22), then the hardware v_cos_f32 / v_sin_f32 (argument in revolutions) finish.
// Replaces the generic cosf() whose large-argument path costs ~100 VGPRs, and an earlier float64 reduction whose five
// double-rate instructions per element made the attention kernels VALU-bound; absolute error ~1e-6, far below the
// 0.06..0.25 rad float32 ulp of the phase itself.
__device__ __forceinline__ float phase_to_rev(float phase) {
    constexpr float C_HI = 0.15915494f;                 // fl32(1 / (2 pi))
    constexpr float C_LO = 6.4206383e-9f;               // 1 / (2 pi) - C_HI   (C_HI = 0.159154936671257019...)
    const float p = __fmul_rn(phase, C_HI);
    const float e = fmaf(phase, C_HI, -p);
    const float q = fmaf(phase, C_LO, e);
    return (p - rintf(p)) + q;
}
__device__ __forceinline__ float cos_phase(float phase) { return __builtin_amdgcn_cosf(phase_to_rev(phase)); }
__device__ __forceinline__ void sincos_phase(float phase, float* s, float* c) {
    const float r = phase_to_rev(phase);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// counter-based RNG for attention dropout: the same (seed, row, head, slot) gives the same bit in fwd and bwd
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)(x >> 11);
}
// Dropout of the residual path (models/modules.py:235): ONE 64-bit hash serves the 4 consecutive elements [4 g, 4 g + 4) of the
// flattened (row, column) index, 16 bits each (the keep probability is 1 - round(65536 p) / 65536: 0.899994 at p = 0.1).  The same
// function in the LayerNorm kernels (tg_layer.hip, element form) and in the chain kernels (tg_chain.hip, float4 form), forward and backward.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ void res_keep_scale4(uint64_t seed, int64_t idx4, float p, float (&k)[4]) {      // idx4 % 4 == 0
    if (p <= 0.f) { k[0] = k[1] = k[2] = k[3] = 1.f; return; }
    const uint64_t h = mix64(seed ^ ((uint64_t)(idx4 >> 2) * 0x9E3779B97F4A7C15ULL));
    const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
    const float s = 1.f / (1.f - p);
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    k[0] = (lo & 0xFFFFu) >= thr ? s : 0.f;
    k[1] = (lo >> 16) >= thr ? s : 0.f;
    k[2] = (hi & 0xFFFFu) >= thr ? s : 0.f;
    k[3] = (hi >> 16) >= thr ? s : 0.f;
}
__device__ __forceinline__ float res_keep_scale(uint64_t seed, int64_t idx, float p) {
    if (p <= 0.f) return 1.f;
    const uint64_t h = mix64(seed ^ ((uint64_t)(idx >> 2) * 0x9E3779B97F4A7C15ULL));
    const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
    return (uint32_t)((h >> (16 * (idx & 3))) & 0xFFFFu) >= thr ? 1.f / (1.f - p) : 0.f;
}
__device__ __forceinline__ float dropout_keep_scale(uint64_t seed, int64_t row, int head, int slot, float p) {
    if (p <= 0.f) return 1.f;
    uint64_t key = seed ^ ((uint64_t)row * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)(head * 1315423911u + slot) << 20);
    float u = (float)(mix32(key) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f / (1.0f - p) : 0.f;
}

}  // namespace tg
