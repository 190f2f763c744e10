// TGN memory path: GRU gate math and raw-message assembly.
//   tg_gru_gates_fwd/bwd  <- nn.GRUCell behind models/MemoryModel.py:531-543 (GRUMemoryUpdater), the two input products run on
//                            tg_gemm_f32; this is the element-wise gate stage and its gradient
//   tg_build_messages     <- models/MemoryModel.py:233-278 compute_new_node_raw_messages (identity message, python per-edge loop)
#include <math.h>

#include "tg_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gi, gh: (n, 3d) with gate order r | z | n (torch GRUCell layout); h, out: (n, d)
__global__ void __launch_bounds__(256) gru_gates_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
        const float* __restrict__ h, int64_t n, int d, float* __restrict__ out) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r_ = i / d;
        const int c = (int)(i - r_ * d);
        const float* a = gi + r_ * 3 * d;
        const float* b = gh + r_ * 3 * d;
        const float r = sigmoidf_(a[c] + b[c]);
        const float z = sigmoidf_(a[d + c] + b[d + c]);
        const float nn = tanhf(a[2 * d + c] + r * b[2 * d + c]);
        out[i] = (1.f - z) * nn + z * h[i];
    }
}

__global__ void __launch_bounds__(256) gru_gates_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
        const float* __restrict__ h, const float* __restrict__ dout, int64_t n, int d, float* __restrict__ dgi,
        float* __restrict__ dgh, float* __restrict__ dh) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r_ = i / d;
        const int c = (int)(i - r_ * d);
        const float* a = gi + r_ * 3 * d;
        const float* b = gh + r_ * 3 * d;
        const float r = sigmoidf_(a[c] + b[c]);
        const float z = sigmoidf_(a[d + c] + b[d + c]);
        const float ghn = b[2 * d + c];
        const float nn = tanhf(a[2 * d + c] + r * ghn);
        const float g = dout[i];
        const float dnn = g * (1.f - z);
        const float dz = g * (h[i] - nn);
        const float dpn = dnn * (1.f - nn * nn);
        const float dpr = dpn * ghn * r * (1.f - r);
        const float dpz = dz * z * (1.f - z);
        float* da = dgi + r_ * 3 * d;
        float* db = dgh + r_ * 3 * d;
        da[c] = dpr; db[c] = dpr;
        da[d + c] = dpz; db[d + c] = dpz;
        da[2 * d + c] = dpn; db[2 * d + c] = dpn * r;
        if (dh) dh[i] = g * z;
    }
}

// one wave per message row: [mem[a] | mem[b] | cos((t - last_update[a]) w + b) | edge[e]]
__global__ void __launch_bounds__(256) build_messages_kernel(const float* __restrict__ mem, int64_t mem_ld,
        const float* __restrict__ last_update, const int32_t* __restrict__ a_ids, const int32_t* __restrict__ b_ids,
        const float* __restrict__ t32, const float* __restrict__ edge, int64_t edge_ld, const int32_t* __restrict__ eids,
        const float* __restrict__ te_w, const float* __restrict__ te_b, int64_t n, int d, int de, int T, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int width = 2 * d + T + de;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const int64_t a = a_ids[r], b = b_ids[r], e = eids[r];
        const float dt = t32[r] - last_update[a];                       // float32 - float32, MemoryModel.py:255-257
        float* o = out + r * width;
        for (int c = lane; c < d; c += 64) { o[c] = mem[a * mem_ld + c]; o[d + c] = mem[b * mem_ld + c]; }
        for (int c = lane; c < T; c += 64) o[2 * d + c] = tg::cos_phase(fmaf(dt, te_w[c], te_b[c]));
        for (int c = lane; c < de; c += 64) o[2 * d + T + c] = edge[e * edge_ld + c];
    }
}

// persist the (already computed) GRU rows of the batch nodes that had a pending message: memory[node] = rows[row_of[i]],
// last_update[node] = time of that message (models/MemoryModel.py:214-231, :472-499).  One wave per batch entry; a node that
// occurs twice writes the same values twice.
__global__ void __launch_bounds__(256) tgn_persist_kernel(const float* __restrict__ rows, int64_t rows_ld, const int32_t* __restrict__ row_of,
        const int32_t* __restrict__ nodes, const int32_t* __restrict__ has, const float* __restrict__ msg_time, float* __restrict__ memory,
        int64_t mem_ld, float* __restrict__ last_update, int64_t count, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < count; i += (int64_t)gridDim.x * 4) {
        const int64_t v = nodes[i];
        if (!has[v]) continue;                                   // wave-uniform
        const float* src = rows + (int64_t)row_of[i] * rows_ld;
        for (int c = lane; c < d; c += 64) memory[v * mem_ld + c] = src[c];
        if (lane == 0) last_update[v] = msg_time[v];
    }
}

// "last message wins" (models/MemoryModel.py:312-320 reads only [-1] of a node's list; the lists are filled source role first, then
// destination role, :177-180): of the entries i = 0..count-1 naming the same node, the LARGEST i files its message.
__global__ void __launch_bounds__(256) msg_last_index_kernel(const int32_t* __restrict__ nodes, int64_t count, int32_t* __restrict__ last_idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        atomicMax(last_idx + nodes[i], (int32_t)i);
}
__global__ void __launch_bounds__(256) msg_scatter_last_kernel(const int32_t* __restrict__ nodes, const float* __restrict__ msgs, int64_t msg_ld,
        const float* __restrict__ t32, int64_t count, int width, float* __restrict__ table, int64_t table_ld, int32_t* __restrict__ has,
        float* __restrict__ msg_time, int32_t* __restrict__ last_idx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < count; i += (int64_t)gridDim.x * 4) {
        const int64_t v = nodes[i];
        if (last_idx[v] != (int32_t)i) continue;                 // wave-uniform: not the last entry of its node
        const float* src = msgs + i * msg_ld;
        for (int c = lane; c < width; c += 64) table[v * table_ld + c] = src[c];
        if (lane == 0) { has[v] = 1; msg_time[v] = t32[i]; last_idx[v] = -1; }      // the workspace is left all -1 for the next call
    }
}

// build_messages_kernel + msg_scatter_last_kernel as one launch: only a node's LAST message of the batch is ever read, so only those are built,
// straight into the node's row of the pending-message table (the winner index comes from tgn_persist_index_kernel's launch)
__global__ void __launch_bounds__(256) build_scatter_last_kernel(const float* __restrict__ mem, int64_t mem_ld,
        const float* __restrict__ last_update, const int32_t* __restrict__ a_ids, const int32_t* __restrict__ b_ids,
        const float* __restrict__ t32, const float* __restrict__ edge, int64_t edge_ld, const int32_t* __restrict__ eids,
        const float* __restrict__ te_w, const float* __restrict__ te_b, int64_t n, int d, int de, int T, float* __restrict__ table,
        int64_t table_ld, int32_t* __restrict__ has, float* __restrict__ msg_time, int32_t* __restrict__ last_idx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const int64_t a = a_ids[r];
        if (last_idx[a] != (int32_t)r) continue;                 // wave-uniform: not the last entry of its node
        const int64_t b = b_ids[r], e = eids[r];
        const float dt = t32[r] - last_update[a];                       // float32 - float32, MemoryModel.py:255-257
        float* o = table + a * table_ld;
        for (int c = lane; c < d; c += 64) { o[c] = mem[a * mem_ld + c]; o[d + c] = mem[b * mem_ld + c]; }
        for (int c = lane; c < T; c += 64) o[2 * d + c] = tg::cos_phase(fmaf(dt, te_w[c], te_b[c]));
        for (int c = lane; c < de; c += 64) o[2 * d + T + c] = edge[e * edge_ld + c];
        if (lane == 0) { has[a] = 1; msg_time[a] = t32[r]; last_idx[a] = -1; }      // the workspace is left all -1 for the next call
    }
}

// tgn_persist_kernel's rows and msg_last_index_kernel's winner search as roles of one grid (they are independent: the stepper's state
// advance is persist -> build messages -> scatter the last ones, and the winner index depends on the batch's node list alone)
__global__ void __launch_bounds__(256) tgn_persist_index_kernel(const float* __restrict__ rows, int64_t rows_ld, const int32_t* __restrict__ row_of,
        const int32_t* __restrict__ nodes, const int32_t* __restrict__ has, const float* __restrict__ msg_time, float* __restrict__ memory,
        int64_t mem_ld, float* __restrict__ last_update, int64_t count, int d, int persist_blocks, int32_t* __restrict__ last_idx) {
    if ((int)blockIdx.x >= persist_blocks) {
        const int64_t nb = (int64_t)gridDim.x - persist_blocks;
        for (int64_t i = ((int64_t)blockIdx.x - persist_blocks) * blockDim.x + threadIdx.x; i < count; i += nb * blockDim.x)
            atomicMax(last_idx + nodes[i], (int32_t)i);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < count; i += (int64_t)persist_blocks * 4) {
        const int64_t v = nodes[i];
        if (!has[v]) continue;                                   // wave-uniform
        const float* src = rows + (int64_t)row_of[i] * rows_ld;
        for (int c = lane; c < d; c += 64) memory[v * mem_ld + c] = src[c];
        if (lane == 0) last_update[v] = msg_time[v];
    }
}

// ---- the feature gradient of an attention layer as a segmented sum (tg::attn_bwd_slot_rows_next): the attention backward leaves one row
// per neighbor slot; slots are grouped by the table row they gathered (count / scan / fill on the sampler's side stream, while the
// previous batch computes), and one wave per 32 grouped slots adds up runs of equal rows in registers.  A popular node's row (the top
// item of a Reddit-shape batch is gathered by ~780 of 24 000 slots) then takes ~25 atomic row adds instead of 780 serialised ones.
// count and fill run as 1 024-thread workgroups with the row counters in LDS: a popular row costs one global atomic per WORKGROUP (one
// word takes ~90 atomics per us whoever sends them: 780 slot-wise adds to the top row were 9 us of the launch)
constexpr int SLOT_WG = 1024;
__global__ void __launch_bounds__(SLOT_WG) slot_count_kernel(const int32_t* __restrict__ slot_row, const int32_t* __restrict__ nbr, int64_t n,
                                                             const int32_t* __restrict__ d_rows, int32_t* __restrict__ cnt, int32_t* __restrict__ rank) {
    extern __shared__ int32_t hist[];
    const int m = *d_rows + 1;                        // rows in use (+ the padding row)
    for (int i = threadIdx.x; i < m; i += SLOT_WG) hist[i] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * SLOT_WG + threadIdx.x;
    const bool on = i < n && nbr[i] != 0;
    const int r = on ? slot_row[i] : 0;
    if (on) rank[i] = atomicAdd(&hist[r], 1);         // the slot's place among its workgroup's slots of that row
    __syncthreads();
    // the workgroup's slots of row r start at the value this add returns, once the scan has turned the counts into offsets: kept in hist
    for (int j = threadIdx.x; j < m; j += SLOT_WG)
        if (hist[j] != 0) hist[j] = atomicAdd(cnt + j, hist[j]);
    __syncthreads();
    if (on) rank[i] += hist[r];                       // offset inside the row, over all workgroups
}
// exclusive scan of cnt[0, m) in place, m = *d_rows + 1, ONE workgroup: the counts pass through LDS (coalesced both ways)
__global__ void __launch_bounds__(1024) slot_scan_kernel(int32_t* __restrict__ cnt, const int32_t* __restrict__ d_rows, int32_t* __restrict__ total) {
    extern __shared__ int32_t v[];                    // m values | 1 024 partial sums
    const int m = *d_rows + 1, t = threadIdx.x;
    int32_t* part = v + m;
    for (int i = t; i < m; i += 1024) v[i] = cnt[i];
    __syncthreads();
    const int per = (m + 1023) / 1024, lo = t * per, hi = lo + per < m ? lo + per : m;
    int32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += v[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {        // Hillis-Steele inclusive scan of the partial sums
        const int32_t x = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int32_t run = part[t] - sum;
    for (int i = lo; i < hi; ++i) { const int32_t c = v[i]; v[i] = run; run += c; }
    __syncthreads();
    for (int i = t; i < m; i += 1024) cnt[i] = v[i];
    if (t == 1023) *total = part[t];
}
__global__ void __launch_bounds__(256) slot_fill_kernel(const int32_t* __restrict__ slot_row, const int32_t* __restrict__ nbr, int64_t n,
                                                        const int32_t* __restrict__ cnt, const int32_t* __restrict__ rank, int32_t* __restrict__ order,
                                                        int32_t* __restrict__ srow) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (nbr[i] != 0) {
            const int32_t r = slot_row[i], pos = cnt[r] + rank[i];
            order[pos] = (int32_t)i;
            srow[pos] = r;
        }
}
// one wave per 16 grouped slots; lane l owns columns l, l + 64, ... (dn <= 256): a row is three or four coalesced loads, a run's sum
// as many contiguous float-atomic instructions
constexpr int SEG = 16;
// Workgroups past the segments' (blockIdx.x >= seg_blocks) add n_own further rows, one wave each: table[own_idx[r]] += own_a[r] + own_b[r]
// (the gradient w.r.t. the roots' own rows: the merge layer's and the query's share) -- a launch of its own before.
__global__ void __launch_bounds__(256) slot_rows_sum_kernel(const float* __restrict__ rows, int dn, const int32_t* __restrict__ order, const int32_t* __restrict__ srow,
                                                            const int32_t* __restrict__ n_valid, float* __restrict__ table, int64_t ld, int seg_blocks,
                                                            const float* __restrict__ own_a, const float* __restrict__ own_b, const int32_t* __restrict__ own_idx,
                                                            int64_t n_own) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= seg_blocks) {
        const int64_t r = ((int64_t)blockIdx.x - seg_blocks) * 4 + wave;
        if (r >= n_own) return;
        float* dst = table + (int64_t)own_idx[r] * ld;
        for (int c = lane; c < dn; c += 64) atomicAdd(dst + c, own_a[r * dn + c] + own_b[r * dn + c]);
        return;
    }
    const int64_t nv = *n_valid, first = ((int64_t)blockIdx.x * 4 + wave) * SEG;
    if (first >= nv) return;
    const int cnt = nv - first < SEG ? (int)(nv - first) : SEG;
    const int my_o = lane < cnt ? order[first + lane] : 0, my_r = lane < cnt ? srow[first + lane] : -1;
    float v[SEG][4];
#pragma unroll
    for (int e = 0; e < SEG; ++e) {                   // all rows of the segment in flight
        const int64_t o = __builtin_amdgcn_readlane(my_o, e < cnt ? e : cnt - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[e][q] = lane + 64 * q < dn ? rows[o * dn + lane + 64 * q] : 0.f;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < SEG; ++e) {
        if (e < cnt) {                                // (wave-uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += v[e][q];
            const int r = __builtin_amdgcn_readlane(my_r, e), rn = e + 1 < cnt ? __builtin_amdgcn_readlane(my_r, e + 1 < SEG ? e + 1 : e) : -1;
            if (rn != r) {                            // (wave-uniform) the run of row r ends here
                float* dst = table + (int64_t)r * ld + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (lane + 64 * q < dn) atomicAdd(dst + 64 * q, acc[q]);
                    acc[q] = 0.f;
                }
            }
        }
    }
}

}  // namespace

namespace tg {
int tgn_persist_index(const float* d_rows, int64_t rows_ld, const int32_t* d_row_of, const int32_t* d_nodes, const int32_t* d_has, const float* d_msg_time,
                      float* d_memory, int64_t mem_ld, float* d_last_update, int64_t count, int d, int32_t* d_last_idx_ws, hipStream_t s) {
    TG_REQUIRE(d_rows && d_row_of && d_nodes && d_has && d_msg_time && d_memory && d_last_update && d_last_idx_ws && count >= 0 && d > 0, "tgn_persist_index: arguments");
    if (count == 0) return TG_OK;
    const int pb = (int)std::min<int64_t>((count + 3) / 4, tg::kMaxGridBlocks), ib = (int)std::min<int64_t>((count + 255) / 256, 64);
    tgn_persist_index_kernel<<<(unsigned)(pb + ib), 256, 0, s>>>(d_rows, rows_ld, d_row_of, d_nodes, d_has, d_msg_time, d_memory, mem_ld, d_last_update, count,
                                                                  d, pb, d_last_idx_ws);
    return launch_status("tgn_persist_index_kernel");
}
// models/MemoryModel.py:233-278 (new raw messages) + :177-180, :312-320 (only the last one per node is ever read) behind
// tgn_persist_index: the winners' messages, built in place
int build_scatter_last(const float* d_mem, int64_t mem_ld, const float* d_last_update, const int32_t* d_a_ids, const int32_t* d_b_ids, const float* d_t32,
                       const float* d_edge, int64_t edge_ld, const int32_t* d_eids, const float* d_te_w, const float* d_te_b, int64_t n, int d, int de, int T,
                       float* d_table, int64_t table_ld, int32_t* d_has, float* d_msg_time, int32_t* d_last_idx_ws, hipStream_t s) {
    TG_REQUIRE(d_mem && d_last_update && d_a_ids && d_b_ids && d_t32 && d_edge && d_eids && d_te_w && d_te_b && d_table && d_has && d_msg_time && d_last_idx_ws,
               "build_scatter_last: null pointer");
    TG_REQUIRE(n >= 0 && d > 0 && de >= 0 && T > 0 && table_ld >= 2 * d + T + de, "build_scatter_last: sizes");
    if (n == 0) return TG_OK;
    build_scatter_last_kernel<<<(unsigned)std::min<int64_t>((n + 3) / 4, tg::kMaxGridBlocks), 256, 0, s>>>(d_mem, mem_ld, d_last_update, d_a_ids, d_b_ids, d_t32,
        d_edge, edge_ld, d_eids, d_te_w, d_te_b, n, d, de, T, d_table, table_ld, d_has, d_msg_time, d_last_idx_ws);
    return launch_status("build_scatter_last_kernel");
}
int build_slot_order(const int32_t* slot_row, const int32_t* nbr, int64_t n, int64_t nrows_cap, const int32_t* d_rows, int32_t* cnt, int32_t* rank,
                     int32_t* order, int32_t* srow, int32_t* n_valid, hipStream_t s) {
    TG_REQUIRE(slot_row && nbr && d_rows && cnt && rank && order && srow && n_valid && n >= 0 && nrows_cap > 0, "build_slot_order: arguments");
    TG_REQUIRE(nrows_cap + 1 + 1024 <= kSlotOrderMaxRows, "build_slot_order: more table rows than the LDS counters cover");
    TG_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(nrows_cap + 1), s));
    if (n == 0) { TG_HIP_CHECK(hipMemsetAsync(n_valid, 0, sizeof(int32_t), s)); return TG_OK; }
    static bool attr = false;
    if (!attr) {
        TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(slot_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSlotOrderMaxRows * 4));
        TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(slot_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSlotOrderMaxRows * 4));
        attr = true;
    }
    const size_t lds = sizeof(int32_t) * (size_t)(nrows_cap + 1);
    slot_count_kernel<<<(unsigned)((n + SLOT_WG - 1) / SLOT_WG), SLOT_WG, lds, s>>>(slot_row, nbr, n, d_rows, cnt, rank);
    slot_scan_kernel<<<1, 1024, lds + 1024 * sizeof(int32_t), s>>>(cnt, d_rows, n_valid);
    slot_fill_kernel<<<(unsigned)std::min<int64_t>((n + 255) / 256, tg::kMaxGridBlocks), 256, 0, s>>>(slot_row, nbr, n, cnt, rank, order, srow);
    return launch_status("slot_fill_kernel");
}
int slot_rows_sum(const float* rows, int dn, const int32_t* order, const int32_t* srow, const int32_t* n_valid, int64_t n, float* table,
                  int64_t ld, hipStream_t s, const float* own_a, const float* own_b, const int32_t* own_idx, int64_t n_own) {
    TG_REQUIRE(rows && order && srow && n_valid && table && dn > 0 && dn <= 256 && n >= 0, "slot_rows_sum: arguments");
    TG_REQUIRE(n_own == 0 || (own_a && own_b && own_idx), "slot_rows_sum: the own rows' operands");
    if (n == 0 && n_own == 0) return TG_OK;
    const int seg_blocks = (int)((n + 4 * SEG - 1) / (4 * SEG));
    slot_rows_sum_kernel<<<(unsigned)(seg_blocks + (n_own + 3) / 4), 256, 0, s>>>(rows, dn, order, srow, n_valid, table, ld, seg_blocks, own_a, own_b, own_idx, n_own);
    return launch_status("slot_rows_sum_kernel");
}
// tg_msg_scatter_last behind tgn_persist_index: the winner index is in the workspace already
int msg_scatter_last_indexed(const int32_t* d_nodes, const float* d_msgs, int64_t msg_ld, const float* d_t32, int64_t count, int width, float* d_table,
                             int64_t table_ld, int32_t* d_has, float* d_msg_time, int32_t* d_last_idx_ws, hipStream_t s) {
    TG_REQUIRE(d_nodes && d_msgs && d_t32 && d_table && d_has && d_msg_time && d_last_idx_ws && count >= 0 && width > 0, "msg_scatter_last_indexed: arguments");
    if (count == 0) return TG_OK;
    msg_scatter_last_kernel<<<(unsigned)std::min<int64_t>((count + 3) / 4, tg::kMaxGridBlocks), 256, 0, s>>>(d_nodes, d_msgs, msg_ld, d_t32, count, width,
        d_table, table_ld, d_has, d_msg_time, d_last_idx_ws);
    return launch_status("msg_scatter_last_kernel");
}
}  // namespace tg

extern "C" int tg_tgn_persist(const float* d_rows, int64_t rows_ld, const int32_t* d_row_of, const int32_t* d_nodes, const int32_t* d_has,
                              const float* d_msg_time, float* d_memory, int64_t mem_ld, float* d_last_update, int64_t count, int d,
                              void* stream) {
    TG_REQUIRE(d_rows && d_row_of && d_nodes && d_has && d_msg_time && d_memory && d_last_update && count >= 0 && d > 0, "tg_tgn_persist: arguments");
    if (count == 0) return TG_OK;
    tgn_persist_kernel<<<(unsigned)std::min<int64_t>((count + 3) / 4, tg::kMaxGridBlocks), 256, 0, (hipStream_t)stream>>>(d_rows, rows_ld,
        d_row_of, d_nodes, d_has, d_msg_time, d_memory, mem_ld, d_last_update, count, d);
    return tg::launch_status("tgn_persist_kernel");
}

extern "C" int tg_msg_scatter_last(const int32_t* d_nodes, const float* d_msgs, int64_t msg_ld, const float* d_t32, int64_t count, int width,
                                   float* d_table, int64_t table_ld, int32_t* d_has, float* d_msg_time, int32_t* d_last_idx_ws, void* stream) {
    TG_REQUIRE(d_nodes && d_msgs && d_t32 && d_table && d_has && d_msg_time && d_last_idx_ws && count >= 0 && width > 0, "tg_msg_scatter_last: arguments");
    if (count == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    msg_last_index_kernel<<<(unsigned)std::min<int64_t>((count + 255) / 256, tg::kMaxGridBlocks), 256, 0, s>>>(d_nodes, count, d_last_idx_ws);
    msg_scatter_last_kernel<<<(unsigned)std::min<int64_t>((count + 3) / 4, tg::kMaxGridBlocks), 256, 0, s>>>(d_nodes, d_msgs, msg_ld, d_t32, count,
        width, d_table, table_ld, d_has, d_msg_time, d_last_idx_ws);
    return tg::launch_status("msg_scatter_last_kernel");
}

extern "C" int tg_gru_gates_fwd(const float* d_gi, const float* d_gh, const float* d_h, int64_t n, int d, float* d_out, void* stream) {
    TG_REQUIRE(d_gi && d_gh && d_h && d_out && n >= 0 && d > 0, "tg_gru_gates_fwd: arguments");
    if (n == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((n * d + 255) / 256, tg::kMaxGridBlocks);
    gru_gates_fwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_gi, d_gh, d_h, n, d, d_out);
    return tg::launch_status("gru_gates_fwd_kernel");
}

extern "C" int tg_gru_gates_bwd(const float* d_gi, const float* d_gh, const float* d_h, const float* d_dout, int64_t n, int d,
                                float* d_dgi, float* d_dgh, float* d_dh, void* stream) {
    TG_REQUIRE(d_gi && d_gh && d_h && d_dout && d_dgi && d_dgh && n >= 0 && d > 0, "tg_gru_gates_bwd: arguments");
    if (n == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((n * d + 255) / 256, tg::kMaxGridBlocks);
    gru_gates_bwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_gi, d_gh, d_h, d_dout, n, d, d_dgi, d_dgh, d_dh);
    return tg::launch_status("gru_gates_bwd_kernel");
}

extern "C" int tg_build_messages(const float* d_mem, int64_t mem_ld, const float* d_last_update, const int32_t* d_a_ids,
                                 const int32_t* d_b_ids, const float* d_t32, const float* d_edge, int64_t edge_ld,
                                 const int32_t* d_eids, const float* d_te_w, const float* d_te_b, int64_t n, int d, int de, int T,
                                 float* d_out, void* stream) {
    TG_REQUIRE(d_mem && d_last_update && d_a_ids && d_b_ids && d_t32 && d_edge && d_eids && d_te_w && d_te_b && d_out,
               "tg_build_messages: null pointer");
    TG_REQUIRE(n >= 0 && d > 0 && de >= 0 && T > 0, "tg_build_messages: sizes");
    if (n == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((n + 3) / 4, tg::kMaxGridBlocks);
    build_messages_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_mem, mem_ld, d_last_update, d_a_ids, d_b_ids, d_t32,
        d_edge, edge_ld, d_eids, d_te_w, d_te_b, n, d, de, T, d_out);
    return tg::launch_status("build_messages_kernel");
}


// ---- the touched rows of a lazily updated TGN memory in one call (models/MemoryModel.py:117, :191-231, :501-543, :654-655) ------
// For the U distinct touched nodes uniq[r]: h = memory[uniq], x = pending message[uniq]; GRU cell; rows = has_message ? GRU : h (the
// updated memory, not persisted); base = rows + raw features (the layer-0 table of the embedding).  gi / gh are kept for backward.
namespace {
// two row gathers of different widths in one launch (grid.y = job)
__global__ void __launch_bounds__(256) gather2_kernel(const float* __restrict__ ta, int64_t lda, int ca, float* __restrict__ oa,
        const float* __restrict__ tb, int64_t ldb, int cb, float* __restrict__ ob, const int32_t* __restrict__ idx, int64_t n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* t = blockIdx.y ? tb : ta;
    float* o = blockIdx.y ? ob : oa;
    const int64_t ld = blockIdx.y ? ldb : lda;
    const int cols = blockIdx.y ? cb : ca;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const float* src = t + (int64_t)idx[r] * ld;
        float* dst = o + r * cols;
        for (int c = lane; c < cols; c += 64) dst[c] = src[c];
    }
}
// GRU gates + selection + base row: rows = has[uniq] ? (1 - z) n + z h : h ; base = rows + raw[uniq]
__global__ void __launch_bounds__(256) gru_select_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
        const int32_t* __restrict__ uniq, const int32_t* __restrict__ has, const float* __restrict__ raw, int64_t raw_ld, int64_t n, int d,
        float* __restrict__ rows, float* __restrict__ base) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r_ = i / d;
        const int c = (int)(i - r_ * d);
        const int32_t node = uniq[r_];
        float v = h[i];
        if (gi && has[node] > 0) {
            const float* a = gi + r_ * 3 * d;
            const float* b = gh + r_ * 3 * d;
            const float r = sigmoidf_(a[c] + b[c]);
            const float z = sigmoidf_(a[d + c] + b[d + c]);
            const float nn = tanhf(a[2 * d + c] + r * b[2 * d + c]);
            v = (1.f - z) * nn + z * v;
        }
        rows[i] = v;
        base[i] = v + raw[(int64_t)node * raw_ld + c];
    }
}
}  // namespace

extern "C" int tg_tgn_rows_fwd(const float* d_mem, int64_t mem_ld, const float* d_msg, int64_t msg_ld, const float* d_raw, int64_t raw_ld,
                               const int32_t* d_uniq, int64_t count, const int32_t* d_has, int d, int msg_dim, const float* d_w_ih,
                               const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, int pending, float* d_h_rows,
                               float* d_msg_rows, float* d_gi, float* d_gh, float* d_rows, float* d_base, void* stream) {
    TG_REQUIRE(d_mem && d_raw && d_uniq && d_has && d_h_rows && d_rows && d_base && count >= 0 && d > 0, "tg_tgn_rows_fwd: arguments");
    TG_REQUIRE(!pending || (d_msg && d_msg_rows && d_gi && d_gh && d_w_ih && d_w_hh && d_b_ih && d_b_hh && msg_dim > 0), "tg_tgn_rows_fwd: GRU arguments");
    if (count == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned gb = (unsigned)std::min<int64_t>((count + 3) / 4, tg::kMaxGridBlocks);
    gather2_kernel<<<dim3(gb, pending ? 2 : 1), 256, 0, s>>>(d_mem, mem_ld, d, d_h_rows, d_msg, msg_ld, msg_dim, d_msg_rows, d_uniq, count);
    if (pending) {
        // gi = msg W_ih^T + b_ih (616 deep) and gh = h W_hh^T + b_hh (172 deep): one launch where the tile kernel serves both (24 + 13 us
        // one after the other at ~3.8 k touched rows), else one by one
        if (!tg::gemm_pair_nt(count, 3 * d, msg_dim, d_msg_rows, msg_dim, d_w_ih, msg_dim, d_gi, d_b_ih, d, d_h_rows, d, d_w_hh, d, d_gh, d_b_hh, 3 * d, s)) {
            if (int rc = tg_gemm_f32(0, 1, count, 3 * d, msg_dim, 1.f, d_msg_rows, msg_dim, d_w_ih, msg_dim, d_gi, 3 * d, d_b_ih, 0, 0, stream)) return rc;
            if (int rc = tg_gemm_f32(0, 1, count, 3 * d, d, 1.f, d_h_rows, d, d_w_hh, d, d_gh, 3 * d, d_b_hh, 0, 0, stream)) return rc;
        }
    }
    const int64_t blocks = std::min<int64_t>((count * d + 255) / 256, tg::kMaxGridBlocks);
    gru_select_kernel<<<(unsigned)blocks, 256, 0, s>>>(pending ? d_gi : nullptr, d_gh, d_h_rows, d_uniq, d_has, d_raw, raw_ld, count, d, d_rows, d_base);
    return tg::launch_status("tg_tgn_rows_fwd");
}

namespace {
__global__ void __launch_bounds__(256) gru_gates_bwd_masked_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
        const float* __restrict__ h, const float* __restrict__ dout, const int32_t* __restrict__ uniq, const int32_t* __restrict__ has,
        int64_t n, int d, float* __restrict__ dgi, float* __restrict__ dgh) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r_ = i / d;
        const int c = (int)(i - r_ * d);
        const float* a = gi + r_ * 3 * d;
        const float* b = gh + r_ * 3 * d;
        const float r = sigmoidf_(a[c] + b[c]);
        const float z = sigmoidf_(a[d + c] + b[d + c]);
        const float ghn = b[2 * d + c];
        const float nn = tanhf(a[2 * d + c] + r * ghn);
        const float g = has[uniq[r_]] > 0 ? dout[i] : 0.f;          // rows without a pending message kept their old memory: no GRU gradient
        const float dnn = g * (1.f - z);
        const float dz = g * (h[i] - nn);
        const float dpn = dnn * (1.f - nn * nn);
        const float dpr = dpn * ghn * r * (1.f - r);
        const float dpz = dz * z * (1.f - z);
        float* da = dgi + r_ * 3 * d;
        float* db = dgh + r_ * 3 * d;
        da[c] = dpr; db[c] = dpr;
        da[d + c] = dpz; db[d + c] = dpz;
        da[2 * d + c] = dpn; db[2 * d + c] = dpn * r;
    }
}
}  // namespace

extern "C" int tg_gru_gates_bwd_masked(const float* d_gi, const float* d_gh, const float* d_h, const float* d_dout, const int32_t* d_uniq,
                                       const int32_t* d_has, int64_t n, int d, float* d_dgi, float* d_dgh, void* stream) {
    TG_REQUIRE(d_gi && d_gh && d_h && d_dout && d_uniq && d_has && d_dgi && d_dgh && n >= 0 && d > 0, "tg_gru_gates_bwd_masked: arguments");
    if (n == 0) return TG_OK;
    const int64_t blocks = std::min<int64_t>((n * d + 255) / 256, tg::kMaxGridBlocks);
    gru_gates_bwd_masked_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_gi, d_gh, d_h, d_dout, d_uniq, d_has, n, d, d_dgi, d_dgh);
    return tg::launch_status("gru_gates_bwd_masked_kernel");
}

// Host mirror of one positive batch's state advance (models/MemoryModel.py:155-180 with the assertion of :485-486): for the distinct
// batch nodes u[i] (any order) with the time new_t[i] of their last occurrence -- a node that holds a pending message gets it applied
// (its last-update time becomes the message's), then every node files a new message at new_t[i].  Nothing is changed when the
// assertion fails (TG_EINVAL, "Trying to update memory to time in the past!").  *next_violation = 1 when a filed message is older
// than its node's last update: the reference's NEXT get_updated_memories would raise on it.  Pure host code (numpy arrays in place).
extern "C" int tg_tgn_host_advance(const int64_t* u, const double* new_t, int64_t count, uint8_t* has, double* msg_time, float* last_update,
                                   int64_t num_nodes, int* next_violation) {
    TG_REQUIRE(u && new_t && has && msg_time && last_update && next_violation && count >= 0, "tg_tgn_host_advance: arguments");
    for (int64_t i = 0; i < count; ++i) {
        const int64_t v = u[i];
        TG_REQUIRE(v >= 0 && v < num_nodes, "tg_tgn_host_advance: node id out of range");
        TG_REQUIRE(!(has[v] && last_update[v] > (float)msg_time[v]), "Trying to update memory to time in the past!");
    }
    for (int64_t i = 0; i < count; ++i) {
        const int64_t v = u[i];
        if (has[v]) last_update[v] = (float)msg_time[v];
        has[v] = 1;
        msg_time[v] = new_t[i];
        if (last_update[v] > (float)new_t[i]) *next_violation = 1;
    }
    return TG_OK;
}
