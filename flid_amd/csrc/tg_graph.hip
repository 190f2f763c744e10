// Temporal adjacency (device CSR) + neighbor lookups.
//   tg_graph_create      <- utils/utils.py:283-302, :96-103   (reference: python lists of tuples, per-node sorted())
//   tg_sample_recent     <- utils/utils.py:130-147, :149-214 ('recent' branch :200-209)
//   tg_first_hop_window  <- utils/utils.py:254-273 + models/DyGFormer.py:196-245
#include <algorithm>
#include <numeric>
#include <math.h>
#include <vector>

#include <atomic>
#include <mutex>
#include <thread>

#include "tg_common.h"

namespace tg {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
std::string get_error() { return g_err; }
const float* zero_block() {
    static float* z = [] {
        float* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) { (void)hipGetLastError(); return (float*)nullptr; }
        return p;
    }();
    return z;
}

struct ProfRec { std::string tag; double units; hipEvent_t a, b; };
static int g_prof_mask = 0;   // bit 0 attn_fwd, bit 1 attn_bwd, bit 2 gemm, bit 3 tgn_advance
static int prof_bit(const char* tag) { return tag[0] == 't' ? 8 : tag[0] == 'g' ? 4 : (tag[5] == 'f' ? 1 : 2); }
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mutex;          // launches come from two host threads (tg_layer.hip: SideIssuer)
ProfScope::ProfScope(const char* tag_, double units_, hipStream_t s) : tag(tag_), units(units_), a(nullptr), b(nullptr), stream(s) {
    if (!(g_prof_mask & prof_bit(tag))) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, s);
}
ProfScope::~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, stream);
    std::lock_guard<std::mutex> g(g_prof_mutex);
    g_prof.push_back(ProfRec{tag, units, a, b});
}
}  // namespace tg

extern "C" void tg_profile_enable(int mask) { tg::g_prof_mask = mask; }

// Sum of elapsed ms / units / launches recorded under `tag` since the last reset.  Synchronises the device.
extern "C" int tg_profile_collect(const char* tag, double* ms, double* units, int64_t* count, int reset) {
    TG_HIP_CHECK(hipDeviceSynchronize());
    double m = 0, u = 0;
    int64_t c = 0;
    for (auto& r : tg::g_prof) {
        if (r.tag != tag) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { m += t; u += r.units; ++c; }
    }
    if (ms) *ms = m;
    if (units) *units = u;
    if (count) *count = c;
    if (reset) {
        std::vector<tg::ProfRec> keep;
        for (auto& r : tg::g_prof) {
            if (r.tag == tag) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
            else keep.push_back(r);
        }
        tg::g_prof.swap(keep);
    }
    return TG_OK;
}

struct tg_graph {
    int64_t num_rows = 0;
    int64_t num_entries = 0;
    int64_t* d_row_ptr = nullptr;
    tg::Incidence* d_inc = nullptr;
    double* d_cumw = nullptr;      // per incidence: running sum of the time-interval-aware sampling weights inside its row (tg_graph_set_time_weights)
};

extern "C" void tg_graph_destroy(tg_graph* g);
extern "C" const char* tg_last_error(void) { return tg::g_err.c_str(); }
extern "C" int tg_version(void) { return 1; }

namespace {
// run f(t) on nt host threads (t = 0 .. nt-1)
template <class F>
void parallel_for_threads(int nt, F f) {
    if (nt <= 1) { f(0); return; }
    std::vector<std::thread> th;
    th.reserve(nt);
    for (int t = 0; t < nt; ++t) th.emplace_back([=] { f(t); });
    for (auto& x : th) x.join();
}
}  // namespace

extern "C" int tg_graph_create(const int64_t* h_src, const int64_t* h_dst, const int64_t* h_eid, const double* h_t,
                               int64_t num_edges, int64_t num_rows, tg_graph** out) {
    TG_REQUIRE(out && num_edges >= 0 && num_rows > 0, "tg_graph_create: sizes");
    TG_REQUIRE(num_edges == 0 || (h_src && h_dst && h_eid && h_t), "tg_graph_create: null arrays");
    TG_REQUIRE(2 * num_edges < (int64_t)INT32_MAX * 2, "tg_graph_create: too many incidences");
    // Counting sort by owner keeps stream order inside a row; then (only if the stream is not chronological) a stable per-row sort by
    // time.  Multi-threaded on the host from 1 M edges on: the 10 M-node / 100 M-edge graph of SURVEY 8d config 5 took ~1 min on one
    // core.  Threads own disjoint NODE ranges (balanced by incidence count) and each scans the whole edge stream, so a row is filled
    // by one thread in stream order -- no atomics in the fill, same result as the sequential build.
    const int hw = (int)std::thread::hardware_concurrency();
    const int nt = num_edges < (1 << 20) ? 1 : std::max(1, std::min(hw > 0 ? hw : 8, 32));
    std::vector<int64_t> row_ptr(num_rows + 1, 0);
    std::vector<int> bad(nt, 0);
    bool chronological = true;
    {
        // degrees: per-thread edge chunks, atomic increments (relaxed) on the shared counters
        static_assert(sizeof(std::atomic<int64_t>) == sizeof(int64_t) && alignof(std::atomic<int64_t>) == alignof(int64_t), "lock-free 64-bit counters");
        std::atomic<int64_t>* cnt = reinterpret_cast<std::atomic<int64_t>*>(row_ptr.data());
        std::vector<int> unsorted(nt, 0);
        parallel_for_threads(nt, [&](int t) {
            const int64_t lo = num_edges * t / nt, hi = num_edges * (t + 1) / nt;
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t s = h_src[i], d = h_dst[i];
                if (s < 0 || s >= num_rows || d < 0 || d >= num_rows) { bad[t] = 1; return; }
                if (h_eid[i] < 0 || h_eid[i] > INT32_MAX) { bad[t] = 2; return; }
                cnt[s + 1].fetch_add(1, std::memory_order_relaxed);
                cnt[d + 1].fetch_add(1, std::memory_order_relaxed);
                if (i > 0 && h_t[i - 1] > h_t[i]) unsorted[t] = 1;
            }
        });
        for (int t = 0; t < nt; ++t) {
            TG_REQUIRE(bad[t] != 1, "tg_graph_create: node id out of range");
            TG_REQUIRE(bad[t] != 2, "tg_graph_create: edge id out of range");
            if (unsorted[t]) chronological = false;
        }
    }
    for (int64_t r = 0; r < num_rows; ++r) row_ptr[r + 1] += row_ptr[r];
    std::vector<tg::Incidence> inc(2 * num_edges);
    {
        // node ranges with about equal numbers of incidences
        std::vector<int64_t> cut(nt + 1, num_rows);
        cut[0] = 0;
        for (int t = 1; t < nt; ++t)
            cut[t] = std::lower_bound(row_ptr.begin(), row_ptr.end(), 2 * num_edges * t / nt) - row_ptr.begin();
        for (int t = 1; t <= nt; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
        cut[nt] = num_rows;
        std::vector<int64_t> cur(row_ptr.begin(), row_ptr.end() - 1);
        parallel_for_threads(nt, [&](int t) {
            const int64_t lo = cut[t], hi = cut[t + 1];
            if (lo >= hi) return;
            for (int64_t i = 0; i < num_edges; ++i) {   // source endpoint first, as the reference appends (:299-300)
                const int64_t sv = h_src[i], dv = h_dst[i];
                if (sv >= lo && sv < hi) inc[cur[sv]++] = tg::Incidence{(int32_t)dv, (int32_t)h_eid[i], h_t[i]};
                if (dv >= lo && dv < hi) inc[cur[dv]++] = tg::Incidence{(int32_t)sv, (int32_t)h_eid[i], h_t[i]};
            }
            if (!chronological)
                for (int64_t r = lo; r < hi; ++r)
                    std::stable_sort(inc.begin() + row_ptr[r], inc.begin() + row_ptr[r + 1],
                                     [](const tg::Incidence& a, const tg::Incidence& b) { return a.t < b.t; });
        });
    }
    tg_graph* g = new tg_graph();
    g->num_rows = num_rows;
    g->num_entries = 2 * num_edges;
    auto fail = [&](hipError_t e, const char* what) {      // nothing leaks on a failed allocation / copy
        tg::set_error(std::string(what) + ": " + hipGetErrorString(e));
        tg_graph_destroy(g);
        return TG_EHIP;
    };
    hipError_t e;
    if ((e = hipMalloc(&g->d_row_ptr, sizeof(int64_t) * (num_rows + 1))) != hipSuccess) return fail(e, "hipMalloc(row_ptr)");
    if ((e = hipMalloc(&g->d_inc, sizeof(tg::Incidence) * std::max<int64_t>(1, g->num_entries))) != hipSuccess) return fail(e, "hipMalloc(incidences)");
    if ((e = hipMemcpy(g->d_row_ptr, row_ptr.data(), sizeof(int64_t) * (num_rows + 1), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(row_ptr)");
    if (g->num_entries && (e = hipMemcpy(g->d_inc, inc.data(), sizeof(tg::Incidence) * g->num_entries, hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy(incidences)");
    *out = g;
    return TG_OK;
}

extern "C" void tg_graph_destroy(tg_graph* g) {
    if (!g) return;
    (void)hipFree(g->d_row_ptr);
    (void)hipFree(g->d_inc);
    (void)hipFree(g->d_cumw);
    delete g;
}
extern "C" int64_t tg_graph_num_rows(const tg_graph* g) { return g ? g->num_rows : -1; }
extern "C" int64_t tg_graph_num_entries(const tg_graph* g) { return g ? g->num_entries : -1; }

extern "C" int tg_graph_export(const tg_graph* g, int64_t* h_row_ptr, int32_t* h_nbr, int32_t* h_eid, double* h_t) {
    TG_REQUIRE(g && h_row_ptr && h_nbr && h_eid && h_t, "tg_graph_export: null");
    std::vector<tg::Incidence> inc(g->num_entries);
    TG_HIP_CHECK(hipMemcpy(h_row_ptr, g->d_row_ptr, sizeof(int64_t) * (g->num_rows + 1), hipMemcpyDeviceToHost));
    if (g->num_entries)
        TG_HIP_CHECK(hipMemcpy(inc.data(), g->d_inc, sizeof(tg::Incidence) * g->num_entries, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < g->num_entries; ++i) { h_nbr[i] = inc[i].nbr; h_eid[i] = inc[i].eid; h_t[i] = inc[i].t; }
    return TG_OK;
}

namespace {

// number of incidences of `row` with t strictly below `when` (numpy searchsorted side='left')
__device__ __forceinline__ int64_t history_end(const tg::Incidence* __restrict__ inc, int64_t lo, int64_t hi, double when) {
    int64_t a = lo, b = hi;
    while (a < b) {
        int64_t mid = (a + b) >> 1;
        if (inc[mid].t < when) a = mid + 1; else b = mid;
    }
    return a - lo;
}

// One query per (k-lane group): lane 0 of the group searches, then the group's lanes copy one slot each.
template <int GROUP>
__global__ void __launch_bounds__(256) sample_recent_kernel(const int64_t* __restrict__ row_ptr,
        const tg::Incidence* __restrict__ inc, int64_t num_rows, const int32_t* __restrict__ ids,
        const double* __restrict__ t64, const float* __restrict__ t32, int64_t n, int k,
        int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid, float* __restrict__ o_t, float* __restrict__ o_dt,
        int32_t* __restrict__ status) {
    const int lane = threadIdx.x % GROUP;
    const int64_t groups_per_block = blockDim.x / GROUP;
    for (int64_t q = blockIdx.x * groups_per_block + threadIdx.x / GROUP; q < n; q += (int64_t)gridDim.x * groups_per_block) {
        const int32_t v = ids[q];
        // hop >= 2 queries arrive as float32 and are promoted exactly (numpy semantics of utils.py:141)
        const double when = t64 ? t64[q] : (double)t32[q];
        int64_t cnt = 0, lo = 0;
        if (v < 0 || v >= num_rows) {
            if (lane == 0 && status) atomicExch(status, 1);
        } else {
            lo = row_ptr[v];
            if (lane == 0) cnt = history_end(inc, lo, row_ptr[v + 1], when);
            cnt = __shfl(cnt, 0, GROUP);
        }
        const int64_t take = cnt < k ? cnt : k;
        const int64_t first = lo + cnt - take;   // newest `take`, oldest first
        for (int s = lane; s < k; s += GROUP) {
            const int64_t j = s - (k - take);
            int32_t nb = 0, ed = 0;
            float tt = 0.f;
            if (j >= 0) {
                const tg::Incidence e = inc[first + j];
                nb = e.nbr; ed = e.eid; tt = (float)e.t;
            }
            const int64_t o = q * k + s;
            o_nbr[o] = nb; o_eid[o] = ed; o_t[o] = tt;
            if (o_dt) o_dt[o] = t64 ? (float)(when - (double)tt) : (t32[q] - tt);   // TGAT.py:120-125 dtype rules
        }
    }
}

__global__ void __launch_bounds__(256) first_hop_window_kernel(const int64_t* __restrict__ row_ptr,
        const tg::Incidence* __restrict__ inc, int64_t num_rows, const int32_t* __restrict__ ids,
        const double* __restrict__ t64, int64_t n, int max_len, int width,
        int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid, float* __restrict__ o_t, int32_t* __restrict__ o_len) {
    constexpr int GROUP = 32;
    const int lane = threadIdx.x % GROUP;
    const int64_t gpb = blockDim.x / GROUP;
    for (int64_t q = blockIdx.x * gpb + threadIdx.x / GROUP; q < n; q += (int64_t)gridDim.x * gpb) {
        const int32_t v = ids[q];
        const double when = t64[q];
        int64_t cnt = 0, lo = 0;
        if (v >= 0 && v < num_rows) {
            lo = row_ptr[v];
            if (lane == 0) cnt = history_end(inc, lo, row_ptr[v + 1], when);
            cnt = __shfl(cnt, 0, GROUP);
        }
        const int64_t keep = cnt < (max_len - 1) ? cnt : (max_len - 1);
        const int64_t first = lo + cnt - keep;
        for (int s = lane; s < width; s += GROUP) {
            int32_t nb = 0, ed = 0;
            float tt = 0.f;
            if (s == 0) { nb = v; tt = (float)when; }                       // DyGFormer.py:235-237
            else if (s <= keep) { const tg::Incidence e = inc[first + s - 1]; nb = e.nbr; ed = e.eid; tt = (float)e.t; }
            const int64_t o = q * width + s;
            o_nbr[o] = nb; o_eid[o] = ed; o_t[o] = tt;
        }
        if (lane == 0) o_len[q] = (int32_t)(keep + 1);
    }
}

// ---- distinct (node id, float32 time) pairs of a sampled level ----------------------------------------------------------------
// Open-addressing hash set in global memory for WHICH pairs are distinct; the compact row numbers are given out in the order of the
// pairs' FIRST OCCURRENCE in the input (round 5).  Before, the slot that claimed a key drew the next number from an atomic counter:
// the numbering followed the arrival order of the claims, two identical calls numbered their rows differently, and anything that
// depends on a row's POSITION (which workgroup a row lands in) could not be allowed to touch its value.  Three passes:
//   insert : claim / find the pair's slot; vals[slot] = min over its occurrences of the input index (atomicMin)
//   rank   : occurrence i is a first one iff vals[pos[i]] == i; per 4096-slot tile the first ones are ranked (wave ballots) and counted
//   number : a first occurrence's number = the counts of the tiles in front + its rank
//            (as ONE workgroup walking the tiles this took 60 us -- six dependent round trips per tile -- and a 16-wave workgroup parked
//            on a CU for that long stretched whichever main-stream launch had workgroups there: the root attention backward 23 -> 34 us)
//   lookup : every slot's row = the number of its pair
__device__ __forceinline__ uint32_t hash_pair(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (uint32_t)k;
}
__global__ void __launch_bounds__(256) dedupe_insert_kernel(const int32_t* __restrict__ ids, const float* __restrict__ t, int64_t n,
        uint32_t mask, unsigned long long* __restrict__ keys, int32_t* __restrict__ vals, int32_t* __restrict__ pos,
        int32_t* __restrict__ count_pad) {
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        const bool in = i < n;
        const int32_t id = in ? ids[i] : -1;
        const float tt = in ? t[i] : 0.f;
        const unsigned long long key = ((unsigned long long)(uint32_t)id << 32) | (unsigned long long)__float_as_uint(tt);
        // The padding pair (0, +0.0f) is a fifth of all slots: it bypasses the table (thousands of CAS on one slot would
        // serialise); the first padding lane of each wave (the wave's smallest index) reports it to count_pad[2]
        const bool is_pad = in && key == 0ULL;
        const unsigned long long pad_lanes = __ballot(is_pad);
        if (is_pad) {
            if ((int)(threadIdx.x & 63) == __ffsll((long long)pad_lanes) - 1) atomicMin(&count_pad[2], (int32_t)i);
            pos[i] = -1;
            continue;
        }
        if (!in) continue;
        uint32_t h = hash_pair(key) & mask;
        while (true) {
            const unsigned long long prev = atomicCAS(&keys[h], ~0ULL, key);
            if (prev == ~0ULL || prev == key) break;      // this slot holds the pair now
            h = (h + 1) & mask;
        }
        atomicMin(&vals[h], (int32_t)i);
        pos[i] = (int32_t)h;
    }
}
constexpr int kNumberThreads = 1024;
constexpr int kNumberTile = 4 * kNumberThreads;     // input slots per workgroup of the two numbering passes
constexpr int kNumberMaxBlocks = 1024;              // their totals live behind the hash set's values: vals[capacity + block]
// pass 1: rank[i] = rank of slot i among the first occurrences of its 4096-slot tile (-1: not a first occurrence); totals[tile]
__global__ void __launch_bounds__(kNumberThreads) dedupe_rank_kernel(int64_t n, const int32_t* __restrict__ pos, const int32_t* __restrict__ vals,
                                                                   const int32_t* __restrict__ count_pad, int32_t* __restrict__ rank,
                                                                   int32_t* __restrict__ totals) {
    __shared__ int32_t wtot[kNumberThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t pad_first = count_pad[2];
    const unsigned long long below = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
    const int64_t i = (int64_t)blockIdx.x * kNumberTile + 4 * tid;
    int32_t h[4];
    bool f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = i + e < n ? pos[i + e] : -2;
    int32_t lower = 0, total = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = h[e] == -2 ? false : (h[e] < 0 ? pad_first == (int32_t)(i + e) : vals[h[e]] == (int32_t)(i + e));
        const unsigned long long m = __ballot(f[e]);
        lower += __popcll(m & below);                            // first occurrences held by lower lanes (all of their four slots come first)
        total += __popcll(m);
    }
    if (lane == 0) wtot[wave] = total;
    __syncthreads();
    int32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kNumberThreads / 64; ++w) { const int32_t v = wtot[w]; all += v; before += w < wave ? v : 0; }
    int32_t r = before + lower;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i + e < n) rank[i + e] = f[e] ? r++ : -1;
    if (tid == 0) totals[blockIdx.x] = all;
}
// pass 2: the first occurrences take the numbers (totals of the tiles in front) + rank; a number overwrites the slot's first-occurrence
// index TAGGED (negative): nothing compares it with an index any more, the tag only keeps the lookup's meaning of the field explicit
__global__ void __launch_bounds__(kNumberThreads) dedupe_number_kernel(const int32_t* __restrict__ ids, const float* __restrict__ t, int64_t n,
        const int32_t* __restrict__ pos, int32_t* __restrict__ vals, const int32_t* __restrict__ rank, const int32_t* __restrict__ totals,
        int32_t* __restrict__ out_ids, float* __restrict__ out_t, int32_t* __restrict__ count_pad) {
    __shared__ int32_t part[kNumberThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t s = 0;
    for (int b = tid; b < (int)blockIdx.x; b += kNumberThreads) s += totals[b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    int32_t offset = 0;
#pragma unroll
    for (int w = 0; w < kNumberThreads / 64; ++w) offset += part[w];
    const int64_t i = (int64_t)blockIdx.x * kNumberTile + 4 * tid;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i + e >= n) break;
        const int32_t r = rank[i + e];
        if (r < 0) continue;
        const int32_t idx = offset + r, h = pos[i + e];
        if (h < 0) { count_pad[1] = idx; out_ids[idx] = 0; out_t[idx] = 0.f; }
        else { out_ids[idx] = ids[i + e]; out_t[idx] = t[i + e]; vals[h] = idx - 0x40000000; }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) count_pad[0] = offset + totals[blockIdx.x];
}
__global__ void __launch_bounds__(256) dedupe_lookup_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ vals, int64_t n,
                                                            int32_t offset, const int32_t* __restrict__ count_pad,
                                                            int32_t* __restrict__ inv) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        inv[i] = (pos[i] < 0 ? count_pad[1] : vals[pos[i]] + 0x40000000) + offset;
}

// empty hash set (no key, first occurrence = none) + (count, pad row, first padding slot) = (0, -1, none): one launch
__global__ void __launch_bounds__(256) dedupe_init_kernel(unsigned long long* __restrict__ keys, int32_t* __restrict__ vals, int64_t capacity,
                                                          int32_t* __restrict__ count_pad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * blockDim.x) { keys[i] = ~0ULL; vals[i] = 0x7fffffff; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { count_pad[0] = 0; count_pad[1] = -1; count_pad[2] = 0x7fffffff; }
}

inline int grid_for(int64_t work_items, int64_t per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, 8 * tg::kMaxGridBlocks));
}

}  // namespace

extern "C" int tg_sample_recent(const tg_graph* g, const int32_t* d_ids, const double* d_times64, const float* d_times32,
                                int64_t n, int k, int32_t* d_out_nbr, int32_t* d_out_eid, float* d_out_t, float* d_out_dt,
                                int32_t* d_status, void* stream) {
    TG_REQUIRE(g && d_ids && d_out_nbr && d_out_eid && d_out_t, "tg_sample_recent: null pointer");
    TG_REQUIRE((d_times64 != nullptr) != (d_times32 != nullptr), "tg_sample_recent: pass exactly one of times64/times32");
    TG_REQUIRE(k > 0, "Number of sampled neighbors for each node should be greater than 0!");
    TG_REQUIRE(n >= 0, "tg_sample_recent: n");
    if (n == 0) return TG_OK;
    hipStream_t s = (hipStream_t)stream;
    if (k <= 8)
        sample_recent_kernel<8><<<grid_for(n, 256 / 8), 256, 0, s>>>(g->d_row_ptr, g->d_inc, g->num_rows, d_ids, d_times64,
            d_times32, n, k, d_out_nbr, d_out_eid, d_out_t, d_out_dt, d_status);
    else
        sample_recent_kernel<32><<<grid_for(n, 256 / 32), 256, 0, s>>>(g->d_row_ptr, g->d_inc, g->num_rows, d_ids, d_times64,
            d_times32, n, k, d_out_nbr, d_out_eid, d_out_t, d_out_dt, d_status);
    return tg::launch_status("sample_recent_kernel");
}

extern "C" int tg_first_hop_window(const tg_graph* g, const int32_t* d_ids, const double* d_times64, int64_t n, int max_len,
                                   int width, int32_t* d_out_nbr, int32_t* d_out_eid, float* d_out_t, int32_t* d_out_len,
                                   void* stream) {
    TG_REQUIRE(g && d_ids && d_times64 && d_out_nbr && d_out_eid && d_out_t && d_out_len, "tg_first_hop_window: null pointer");
    TG_REQUIRE(max_len - 1 > 0, "Maximal number of neighbors for each node should be greater than 1!");
    TG_REQUIRE(width >= max_len || width >= 1, "tg_first_hop_window: width");
    if (n == 0) return TG_OK;
    first_hop_window_kernel<<<grid_for(n, 8), 256, 0, (hipStream_t)stream>>>(g->d_row_ptr, g->d_inc, g->num_rows, d_ids,
        d_times64, n, max_len, width, d_out_nbr, d_out_eid, d_out_t, d_out_len);
    return tg::launch_status("first_hop_window_kernel");
}

// GraphMixer's node encoder (models/GraphMixer.py:125-150): for every root, the feature rows of its `window` most recent neighbors
// before the query time, weighted by softmax(1 for a real neighbor, -1e10 for a padded slot) and then AVERAGED over all `window`
// slots -- i.e. (1 / window) (1 / nv) sum_valid X[nbr] with nv real neighbors; with none, every slot is the padding row with weight
// 1 / window.  The reference materialises (B, 2000, Dn) for this; here one workgroup walks a root's window and keeps the sum in
// registers: the ids of 256 entries at a time go through LDS, the rows are read 4 entries deep.
__global__ void __launch_bounds__(256) recent_window_mean_kernel(const int64_t* __restrict__ row_ptr, const tg::Incidence* __restrict__ inc,
        int64_t num_rows, const int32_t* __restrict__ ids, const double* __restrict__ t64, int64_t n, int window,
        const float* __restrict__ table, int64_t ld, int cols, float* __restrict__ out, int64_t out_ld) {
    __shared__ int32_t s_nbr[256];
    __shared__ int64_t s_cnt;
    __shared__ int s_nv;
    for (int64_t q = blockIdx.x; q < n; q += gridDim.x) {
        const int32_t v = ids[q];
        if (threadIdx.x == 0) {
            s_cnt = (v >= 0 && v < num_rows) ? history_end(inc, row_ptr[v], row_ptr[v + 1], t64[q]) : 0;
            s_nv = 0;
        }
        __syncthreads();
        const int64_t cnt = s_cnt, lo = (v >= 0 && v < num_rows) ? row_ptr[v] : 0;
        const int64_t take = cnt < window ? cnt : window;
        const int64_t first = lo + cnt - take;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};            // columns threadIdx.x + 256 j  (cols <= 1024)
        int nv_local = 0;
        for (int64_t base = 0; base < take; base += 256) {
            const int chunk = (int)((take - base) < 256 ? (take - base) : 256);
            __syncthreads();
            if ((int)threadIdx.x < chunk) {
                const int32_t nb = inc[first + base + threadIdx.x].nbr;
                s_nbr[threadIdx.x] = nb;
                nv_local += nb > 0;
            }
            __syncthreads();
            for (int e = 0; e < chunk; e += 4) {
                float x[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t nb = (e + u < chunk) ? s_nbr[e + u] : 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = threadIdx.x + 256 * j;
                        x[u][j] = (nb > 0 && c < cols) ? table[(int64_t)nb * ld + c] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += x[u][j];
            }
        }
        if (nv_local) atomicAdd(&s_nv, nv_local);
        __syncthreads();
        const int nv = s_nv;
        const float inv_w = 1.f / (float)window;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < cols) out[q * out_ld + c] = nv > 0 ? (acc[j] * (1.f / (float)nv)) * inv_w : table[c] * inv_w;   // table row 0 = the padding row
        }
        __syncthreads();
    }
}

extern "C" int tg_recent_window_mean(const tg_graph* g, const int32_t* d_ids, const double* d_times64, int64_t n, int window,
                                     const float* d_table, int64_t table_ld, int cols, float* d_out, int64_t out_ld, void* stream) {
    TG_REQUIRE(g && d_ids && d_times64 && d_table && d_out, "tg_recent_window_mean: null pointer");
    TG_REQUIRE(window > 0, "Number of sampled neighbors for each node should be greater than 0!");
    TG_REQUIRE(cols > 0 && cols <= 1024 && n >= 0, "tg_recent_window_mean: cols must be in 1..1024");
    if (n == 0) return TG_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(n, tg::kMaxGridBlocks);
    recent_window_mean_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g->d_row_ptr, g->d_inc, g->num_rows, d_ids, d_times64, n, window,
                                                                       d_table, table_ld, cols, d_out, out_ld);
    return tg::launch_status("recent_window_mean_kernel");
}

// ---- random sampling strategies on the device (a NON-bit-exact mode: the reference draws from numpy's RandomState on the host,
// utils/utils.py:176-199, and the bit-exact path does the same; here a counter-based generator replaces that stream) ------------
namespace {
__device__ __forceinline__ double u01(uint64_t seed, uint64_t ctr) {             // splitmix64 of (seed, counter) -> [0, 1)
    uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// 32 lanes per query, k <= 128 slots: draw k indices of the node's strictly-earlier history with replacement -- uniformly, or with
// the time-interval-aware probabilities softmax(p[:cnt]), p_i = e_i / sum_{j <= i} e_j, e_j = exp(tsf (t_j - t_last)) (utils.py:112-128,
// :183-186), through the row's running weight sums -- then order the k samples by (float32 time, draw) as the reference re-sorts
// them by time (:193-199).
constexpr int RS_KMAX = 128;
__global__ void __launch_bounds__(256) sample_random_kernel(const int64_t* __restrict__ row_ptr, const tg::Incidence* __restrict__ inc,
        const double* __restrict__ cumw, int64_t num_rows, const int32_t* __restrict__ ids, const double* __restrict__ t64,
        const float* __restrict__ t32, int64_t n, int k, uint64_t seed, int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid,
        float* __restrict__ o_t, float* __restrict__ o_dt, int32_t* __restrict__ status) {
    constexpr int GROUP = 32;
    __shared__ float s_t[8][RS_KMAX];
    __shared__ int64_t s_j[8][RS_KMAX];
    const int lane = threadIdx.x % GROUP, grp = threadIdx.x / GROUP;
    for (int64_t q0 = (int64_t)blockIdx.x * 8; q0 < n; q0 += (int64_t)gridDim.x * 8) {
        const int64_t q = q0 + grp;
        int64_t cnt = 0, lo = 0;
        double when = 0.0;
        if (q < n) {
            const int32_t v = ids[q];
            when = t64 ? t64[q] : (double)t32[q];
            if (v < 0 || v >= num_rows) {
                if (lane == 0 && status) atomicExch(status, 1);
            } else {
                lo = row_ptr[v];
                if (lane == 0) cnt = history_end(inc, lo, row_ptr[v + 1], when);
                cnt = __shfl(cnt, 0, GROUP);
            }
            const double wtot = (cumw && cnt > 0) ? cumw[lo + cnt - 1] : 0.0;
            for (int s = lane; s < k; s += GROUP) {
                int64_t j = -1;
                if (cnt > 0) {
                    const double u = u01(seed, (uint64_t)q * (uint64_t)k + (uint64_t)s);
                    if (wtot > 0.0) {                             // first index whose running weight exceeds u * total
                        const double target = u * wtot;
                        int64_t a = 0, b = cnt - 1;
                        while (a < b) {
                            const int64_t mid = (a + b) >> 1;
                            if (cumw[lo + mid] > target) b = mid; else a = mid + 1;
                        }
                        j = a;
                    } else {                                      // uniform (also: every weight of the prefix is zero -> softmax is uniform)
                        j = (int64_t)(u * (double)cnt);
                        if (j >= cnt) j = cnt - 1;
                    }
                }
                s_j[grp][s] = j;
                s_t[grp][s] = j >= 0 ? (float)inc[lo + j].t : 0.f;
            }
        }
        __syncthreads();
        if (q < n) {
            for (int s = lane; s < k; s += GROUP) {
                const int64_t j = s_j[grp][s];
                const float ts = s_t[grp][s];
                int r = 0;
                for (int x = 0; x < k; ++x) r += (s_t[grp][x] < ts) || (s_t[grp][x] == ts && x < s);
                const int64_t o = q * k + (j >= 0 ? r : s);
                int32_t nb = 0, ed = 0;
                if (j >= 0) { const tg::Incidence e = inc[lo + j]; nb = e.nbr; ed = e.eid; }
                o_nbr[o] = nb; o_eid[o] = ed; o_t[o] = ts;
                if (o_dt) o_dt[o] = t64 ? (float)(when - (double)ts) : (t32[q] - ts);
            }
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int tg_graph_set_time_weights(tg_graph* g, double time_scaling_factor) {
    TG_REQUIRE(g, "tg_graph_set_time_weights: null graph");
    std::vector<int64_t> rp(g->num_rows + 1);
    std::vector<tg::Incidence> inc(g->num_entries);
    TG_HIP_CHECK(hipMemcpy(rp.data(), g->d_row_ptr, sizeof(int64_t) * (g->num_rows + 1), hipMemcpyDeviceToHost));
    if (g->num_entries) TG_HIP_CHECK(hipMemcpy(inc.data(), g->d_inc, sizeof(tg::Incidence) * g->num_entries, hipMemcpyDeviceToHost));
    std::vector<double> cw(std::max<int64_t>(1, g->num_entries));
    for (int64_t v = 0; v < g->num_rows; ++v) {
        const int64_t lo = rp[v], hi = rp[v + 1];
        if (lo == hi) continue;
        const double tmax = inc[hi - 1].t;                       // rows are sorted by time: the maximum is the last entry
        double cum = 0.0, wsum = 0.0;
        for (int64_t j = lo; j < hi; ++j) {
            const double e = exp(time_scaling_factor * (inc[j].t - tmax));
            cum += e;
            const double pr = e / cum;                           // NaN (0 / 0) -> the reference's -1e10 -> weight 0
            const double w = (pr != pr) ? 0.0 : exp((double)(float)pr);      // softmax over float32(p): exp(p_i) / sum exp(p_j)
            wsum += w;
            cw[j] = wsum;
        }
    }
    if (!g->d_cumw) TG_HIP_CHECK(hipMalloc(&g->d_cumw, sizeof(double) * cw.size()));
    TG_HIP_CHECK(hipMemcpy(g->d_cumw, cw.data(), sizeof(double) * cw.size(), hipMemcpyHostToDevice));
    return TG_OK;
}

extern "C" int tg_sample_random(const tg_graph* g, const int32_t* d_ids, const double* d_times64, const float* d_times32, int64_t n, int k,
                                int weighted, uint64_t seed, int32_t* d_out_nbr, int32_t* d_out_eid, float* d_out_t, float* d_out_dt,
                                int32_t* d_status, void* stream) {
    TG_REQUIRE(g && d_ids && (d_times64 || d_times32) && d_out_nbr && d_out_eid && d_out_t, "tg_sample_random: null pointer");
    TG_REQUIRE(k > 0, "Number of sampled neighbors for each node should be greater than 0!");
    TG_REQUIRE(k <= RS_KMAX, "tg_sample_random: at most 128 sampled neighbors");
    TG_REQUIRE(!weighted || g->d_cumw, "tg_sample_random: call tg_graph_set_time_weights first");
    if (n <= 0) return TG_OK;
    sample_random_kernel<<<grid_for(n, 8), 256, 0, (hipStream_t)stream>>>(g->d_row_ptr, g->d_inc, weighted ? g->d_cumw : nullptr, g->num_rows,
        d_ids, d_times64, d_times32, n, k, seed, d_out_nbr, d_out_eid, d_out_t, d_out_dt, d_status);
    return tg::launch_status("sample_random_kernel");
}

// Host-side history counts (pure host code over the exported CSR, tg_graph_export): out[q] = number of incidences of ids[q] strictly
// before times[q] -- what find_neighbors_before (utils/utils.py:130-147) returns the length of.  DyGFormer needs the longest
// window of a batch BEFORE it can shape its launches (models/DyGFormer.py:196-245 pads to the batch maximum); reading the device
// kernel's lengths back stalled the host on the whole previous step.
extern "C" int tg_host_count_before(const int64_t* h_row_ptr, const double* h_t, int64_t num_rows, const int64_t* ids, const double* times,
                                    int64_t n, int64_t* out) {
    TG_REQUIRE(h_row_ptr && h_t && ids && times && out && n >= 0, "tg_host_count_before: arguments");
    for (int64_t q = 0; q < n; ++q) {
        const int64_t v = ids[q];
        if (v < 0 || v >= num_rows) { tg::set_error("list index out of range"); return TG_ERANGE; }
        int64_t lo = h_row_ptr[v], hi = h_row_ptr[v + 1];
        const int64_t base = lo;
        const double when = times[q];
        while (lo < hi) {                                   // searchsorted(side = 'left'): first index with t >= when
            const int64_t mid = (lo + hi) >> 1;
            if (h_t[mid] < when) lo = mid + 1; else hi = mid;
        }
        out[q] = lo - base;
    }
    return TG_OK;
}

extern "C" int64_t tg_dedupe_capacity(int64_t n) {
    int64_t c = 1024;
    while (c < 2 * n) c <<= 1;
    return c;
}

extern "C" int tg_dedupe_pairs(const int32_t* d_ids, const float* d_t, int64_t n, int64_t capacity, void* d_keys_ws,
                               int32_t* d_vals_ws, int32_t* d_pos_ws, int32_t row_offset, int32_t* d_out_ids, float* d_out_t,
                               int32_t* d_out_row, int32_t* d_count_pad, void* stream) {
    TG_REQUIRE(d_ids && d_t && d_keys_ws && d_vals_ws && d_pos_ws && d_out_ids && d_out_t && d_out_row && d_count_pad, "tg_dedupe_pairs: null pointer");
    TG_REQUIRE(n >= 0 && capacity >= 2 * n && (capacity & (capacity - 1)) == 0 && capacity <= ((int64_t)1 << 31), "tg_dedupe_pairs: capacity must be a power of two >= 2n");
    hipStream_t s = (hipStream_t)stream;
    TG_REQUIRE(n < 0x40000000, "tg_dedupe_pairs: more than 2^30 slots");
    dedupe_init_kernel<<<grid_for(capacity, 1024), 256, 0, s>>>((unsigned long long*)d_keys_ws, d_vals_ws, capacity, d_count_pad);   // pad row = -1 until seen
    if (n == 0) return tg::launch_status("dedupe_init_kernel");
    dedupe_insert_kernel<<<grid_for(n, 256), 256, 0, s>>>(d_ids, d_t, n, (uint32_t)(capacity - 1), (unsigned long long*)d_keys_ws, d_vals_ws,
                                                          d_pos_ws, d_count_pad);
    const unsigned nb = (unsigned)((n + kNumberTile - 1) / kNumberTile);
    TG_REQUIRE(nb <= (unsigned)kNumberMaxBlocks, "tg_dedupe_pairs: more than 4 M slots");
    int32_t* totals = d_vals_ws + capacity;           // (the values workspace is capacity + 1024 ints)
    // (the slots' rows double as the scratch of the ranks: the lookup pass overwrites them last)
    dedupe_rank_kernel<<<nb, kNumberThreads, 0, s>>>(n, d_pos_ws, d_vals_ws, d_count_pad, d_out_row, totals);
    dedupe_number_kernel<<<nb, kNumberThreads, 0, s>>>(d_ids, d_t, n, d_pos_ws, d_vals_ws, d_out_row, totals, d_out_ids, d_out_t, d_count_pad);
    dedupe_lookup_kernel<<<grid_for(n, 256), 256, 0, s>>>(d_pos_ws, d_vals_ws, n, row_offset, d_count_pad, d_out_row);
    return tg::launch_status("dedupe kernels");
}

#ifndef TG_TRY
#define TG_TRY(expr) do { int _rc = (expr); if (_rc != TG_OK) return _rc; } while (0)
#endif
// ---- graph-only part of one TGN batch in ONE call -----------------------------------------------------------------------------
// (what MemoryModel.prepare_batch_begin did with ~25 numpy / torch operations: 0.24 ms of host time per 600-edge batch, on a path
// whose GPU work is 0.55 ms).  Device blob, int32 units unless noted, n edges, m = hi - lo embedded edges (2 m roots):
//   [ root times (2m f64) | counterpart ids (2n) | edge ids twice (2n) | times twice (2n f32) | root ids (2m) | batch node ids (2n) | neighbor slots (2m k) ]
//   ^ off[0]               ^ off[1]              ^ off[2]              ^ off[3]              ^ off[4]        ^ off[5]              ^ off[6]      off[7] = end
// Everything before the neighbor slots is staged in pinned memory and copied with one async copy; the sampler writes the slots
// right behind, so [root ids | batch node ids | neighbor slots] IS the list whose distinct nodes the memory update touches.
extern "C" int tg_tgn_prepare_layout(int64_t n, int64_t m, int k, int64_t* off8) {
    TG_REQUIRE(off8 && n >= 0 && m >= 0 && m <= n && k > 0, "tg_tgn_prepare_layout: arguments");
    int64_t o = 0;
    off8[0] = o; o += 2 * (2 * m);          // f64 = two int32 units each
    off8[1] = o; o += 2 * n;
    off8[2] = o; o += 2 * n;
    off8[3] = o; o += 2 * n;
    off8[4] = o; o += 2 * m;
    off8[5] = o; o += 2 * n;
    off8[6] = o; o += 2 * m * k;
    off8[7] = o;
    return TG_OK;
}

extern "C" int tg_tgn_prepare_batch(const tg_graph* g, const int64_t* h_src, const int64_t* h_dst, const double* h_t, const int64_t* h_eid,
                                    int64_t n, int64_t lo, int64_t hi, int k, int64_t num_nodes, void* h_stage, int32_t* d_blob, int32_t* d_S_eid,
                                    float* d_S_t, float* d_S_dt, const float* d_zero_t, int64_t capacity, void* d_keys_ws,
                                    int32_t* d_vals_ws, int32_t* d_pos_ws, int32_t* d_uniq, float* d_uniq_t, int32_t* d_rowmap,
                                    int32_t* d_count_pad, int32_t* h_count_pad, int64_t* h_uniq_nodes, double* h_last_time, int64_t* h_num_uniq,
                                    void* stream) {
    TG_REQUIRE(g && h_src && h_dst && h_t && h_stage && d_blob && d_S_eid && d_S_t && d_S_dt && d_zero_t && d_keys_ws && d_vals_ws && d_pos_ws &&
               d_uniq && d_uniq_t && d_rowmap && d_count_pad && h_count_pad, "tg_tgn_prepare_batch: null pointer");
    TG_REQUIRE(k > 0, "Number of sampled neighbors for each node should be greater than 0!");
    TG_REQUIRE(n > 0 && 0 <= lo && lo < hi && hi <= n, "tg_tgn_prepare_batch: batch / shard bounds");
    const int64_t m = hi - lo;
    int64_t off[8];
    TG_TRY(tg_tgn_prepare_layout(n, m, k, off));
    for (int64_t i = 0; i < n; ++i)
        if (h_src[i] < 0 || h_src[i] >= num_nodes || h_dst[i] < 0 || h_dst[i] >= num_nodes) { tg::set_error("list index out of range"); return TG_ERANGE; }
    int32_t* st = reinterpret_cast<int32_t*>(h_stage);
    double* rt = reinterpret_cast<double*>(st + off[0]);
    float* t32 = reinterpret_cast<float*>(st + off[3]);
    for (int64_t i = 0; i < m; ++i) {
        rt[i] = h_t[lo + i]; rt[m + i] = h_t[lo + i];
        st[off[4] + i] = (int32_t)h_src[lo + i]; st[off[4] + m + i] = (int32_t)h_dst[lo + i];
    }
    for (int64_t i = 0; i < n; ++i) {
        st[off[1] + i] = (int32_t)h_dst[i]; st[off[1] + n + i] = (int32_t)h_src[i];            // the counterpart of [src role | dst role]
        const int32_t e = h_eid ? (int32_t)h_eid[i] : 0;
        st[off[2] + i] = e; st[off[2] + n + i] = e;
        t32[i] = (float)h_t[i]; t32[n + i] = (float)h_t[i];
        st[off[5] + i] = (int32_t)h_src[i]; st[off[5] + n + i] = (int32_t)h_dst[i];
    }
    // host mirror of the state advance: the distinct batch nodes (first-seen order) and, per node, the time of its LAST occurrence in
    // [src role | dst role] order -- the time its pending message will carry (MemoryModel.py:155-180)
    if (h_uniq_nodes && h_last_time && h_num_uniq) {
        int64_t cap2 = 64;
        while (cap2 < 4 * n) cap2 <<= 1;
        static thread_local std::vector<int64_t> keys, slot;
        keys.assign((size_t)cap2, -1);
        slot.assign((size_t)cap2, 0);
        int64_t cnt = 0;
        for (int64_t i = 0; i < 2 * n; ++i) {
            const int64_t v = i < n ? h_src[i] : h_dst[i - n];
            uint64_t h = ((uint64_t)v * 0x9E3779B97F4A7C15ULL) >> 20;
            for (;; ++h) {
                const int64_t j = (int64_t)(h & (uint64_t)(cap2 - 1));
                if (keys[j] == v) { h_last_time[slot[j]] = h_t[i < n ? i : i - n]; break; }
                if (keys[j] < 0) { keys[j] = v; slot[j] = cnt; h_uniq_nodes[cnt] = v; h_last_time[cnt] = h_t[i < n ? i : i - n]; ++cnt; break; }
            }
        }
        *h_num_uniq = cnt;
    }
    hipStream_t s = (hipStream_t)stream;
    TG_HIP_CHECK(hipMemcpyAsync(d_blob, h_stage, sizeof(int32_t) * (size_t)off[6], hipMemcpyHostToDevice, s));
    TG_TRY(tg_sample_recent(g, d_blob + off[4], reinterpret_cast<const double*>(d_blob + off[0]), nullptr, 2 * m, k, d_blob + off[6], d_S_eid, d_S_t,
                            d_S_dt, nullptr, stream));
    const int64_t total = off[7] - off[4];
    TG_TRY(tg_dedupe_pairs(d_blob + off[4], d_zero_t, total, capacity, d_keys_ws, d_vals_ws, d_pos_ws, 0, d_uniq, d_uniq_t, d_rowmap, d_count_pad, stream));
    TG_HIP_CHECK(hipMemcpyAsync(h_count_pad, d_count_pad, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return TG_OK;
}
