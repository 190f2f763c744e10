// The trainer's step on the recursive temporal-attention backbone as calls into ONE native object: the graph-only preparation of a
// batch (ids to the device, neighbor lookups, row sharing) on the object's own side stream, the forward of every layer, and the
// backward of every layer + the optimizer's update -- each a single C call that issues its launches back to back out of a pre-sized
// arena (no device allocation, no Python between launches).
//
// replaces the host side of: models/TGAT.py:50-144 (compute_src_dst_node_temporal_embeddings / compute_node_temporal_embeddings: the
// recursion, its sampler calls utils/utils.py:149-214 and index bookkeeping), the loss.backward() / optimizer.step() sequence of
// PTCL/EM_warmup.py:126-238 and PTCL/M_step.py:209-325 around it, and flid_amd/engine.py's Python form of the same (prepare_begin /
// prepare_finish / _native_forward / _native_backward), which stays as the autograd-facing path and as this one's test oracle.
#include <math.h>

#include <algorithm>
#include <vector>

#include "tg_common.h"

#ifndef TG_TRY
#define TG_TRY(expr) do { int _rc = (expr); if (_rc != TG_OK) return _rc; } while (0)
#endif

namespace {

inline int64_t r4(int64_t n) { return (n + 3) / 4 * 4; }
inline int64_t r64(int64_t n) { return (n + 63) / 64 * 64; }           // arena regions start on 256-byte lines

__global__ void __launch_bounds__(256) iota_kernel(int32_t* __restrict__ out, int64_t n, int32_t first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = first + (int32_t)i;
}

// one batch in preparation / in use
struct Slot {
    enum State { FREE = 0, BEGUN = 1, READY = 2, FORWARDED = 3 };
    State state = FREE;
    int64_t n = 0, count1 = 0;            // roots, distinct level-1 rows
    int32_t pad = -1;                     // frontier row of the padding pair (0, 0.0f), or -1
    // device (int32 units inside the arena)
    int32_t* ids_all = nullptr;           // (cap)            node id of every frontier row
    int32_t *S_nbr = nullptr, *S_eid = nullptr; float *S_t = nullptr, *S_dt = nullptr;     // (cap, k) slot lists
    double* times = nullptr;              // (max_roots)      query times of the roots
    float* uniq_t = nullptr;              // (max_roots k)    float32 query times of the level-1 rows
    int32_t* child = nullptr;             // (max_roots k)    frontier row of every root slot
    int32_t* count_pad = nullptr;         // (2)
    // pinned
    int32_t* h_ids = nullptr; double* h_times = nullptr; int32_t* h_count_pad = nullptr;
    hipEvent_t copied = nullptr, counted = nullptr, ready = nullptr, consumed = nullptr;
    bool consumed_pending = false;
    // TGN (tg_stepper_cfg.tgn): the blob of tg_tgn_prepare_batch and what hangs off it
    int64_t nb = 0, lo = 0, hi = 0, uniq_count = 0;      // batch edges, embedded shard [lo, hi), distinct touched nodes
    int64_t off[8] = {};
    int32_t *blob = nullptr, *uniq = nullptr, *rowmap = nullptr;      // (uniq_t above: the distinct nodes' all-zero times)
    int32_t *slot_order = nullptr, *slot_srow = nullptr, *slot_nvalid = nullptr;   // neighbor slots grouped by compact row (tg::build_slot_order)
    void* h_stage = nullptr;              // pinned: the staged blob + the (count, pad) words
    std::vector<int64_t> h_u; std::vector<double> h_newt; int64_t h_nu = 0;     // host mirror of the state advance
    bool has_eid = false;
};

struct TgnBuf {                           // arena regions of the memory stage (floats); U = touched nodes at most
    float *h_rows, *msg_rows, *gi, *gh, *rows, *base, *d_own, *d_raw, *dgi, *dgh, *msgs, *zero_t;
    float* slot_rows;                     // (max_roots k, dn): the attention backward's per-slot feature gradients
    int32_t *slot_cnt, *slot_rank;        // (max_u + 2), (max_roots k): scratch of build_slot_order
    int64_t max_u;
};

struct LayerBuf {                         // arena regions of one layer (floats)
    float *qbias, *q, *u, *agg, *prob, *ctx, *res, *y, *mean, *rstd, *f1, *wT, *out;
    float *df1, *dy, *dsum, *dres, *dctx, *dagg, *du, *dq, *part;
    int64_t max_rows, part_floats;
};

}  // namespace

struct tg_stepper {
    tg_stepper_cfg c;
    int dq, dk, hd;
    int64_t cap;                          // frontier rows at most: max_roots (1 + k) (two layers), max_roots (one layer)
    std::vector<Slot> slots;
    std::vector<LayerBuf> lay;
    std::vector<tg_layer_desc> desc;      // of the forward in flight (read by its backward)
    std::vector<int64_t> poff;            // offsets of [te_w, te_b, 11 per layer] in the flat parameter, then where they end
    int64_t ptotal = 0;                   // floats of the flat parameter (TGN: the GRU's four tensors behind the layers')
    float* cosb = nullptr;
    float* gblock = nullptr;              // [parameter gradients (flat layout) | extra | d cos b | vec per layer | lower layers' gradient rows]
    int64_t g_extra = 0, g_cosb = 0, g_vec = 0, vlen = 0, g_rows = 0, g_floats = 0;
    // dedupe workspace (the side stream's launches are ordered: one workspace)
    int64_t ded_cap = 0; void* ded_keys = nullptr; int32_t *ded_vals = nullptr, *ded_pos = nullptr;
    hipStream_t side = nullptr;
    void* pinned = nullptr;
    int fwd_slot = -1;
    int64_t fwd_rows[8];
    int64_t zeroed_floats = 0;            // floats of the gradient block the forward's prelude launch has zero-filled (0: none)
    TgnBuf tb{};
    int md = 0;                           // TGN message width 2 dn + dt_dim + de
    int64_t gru_off[4] = {};              // w_ih, w_hh, b_ih, b_hh inside the flat parameter
    bool fwd_pending = false;
};

namespace {

int64_t rows_of_layer(const tg_stepper* st, int l /* 1-based */, int64_t n, int64_t count1) {
    const int L = st->c.layers;
    return (L - l) == 0 ? n : n + count1;          // layers <= 2: the lower layer computes roots + level-1 rows
}

// TGN: the attention layer's gradient w.r.t. the compact table as per-slot rows + segmented sum (tg_memory.hip) instead of float atomics
// from the attention backward.  FLID_TGN_SLOT_ROWS=0 (with FLID_GEMM_TUNE): the atomics, for A/B runs.
const bool g_slot_rows = !(getenv("FLID_GEMM_TUNE") && getenv("FLID_TGN_SLOT_ROWS") && atoi(getenv("FLID_TGN_SLOT_ROWS")) == 0);

// arena layout: sizes only (base == nullptr) or pointers
struct Arena {
    float* base; int64_t off = 0;
    explicit Arena(float* b) : base(b) {}
    float* take(int64_t floats) { float* p = base ? base + off : nullptr; off += r64(floats); return p; }
};

// (the order build keeps one counter per table row in LDS: steppers sized for more rows than that -- a data-parallel rank's global batch
// of several thousand edges -- keep the attention backward's atomics)
inline bool slot_rows_on(const tg_stepper* st) { return g_slot_rows && st->c.tgn && st->tb.max_u + 2 + 1024 <= tg::kSlotOrderMaxRows; }

int layout(tg_stepper* st, float* base, int64_t* total) {
    const tg_stepper_cfg& c = st->c;
    const int L = c.layers, H = c.heads, dn = c.dn, T = c.dt_dim, dq = st->dq, dk = st->dk, k = c.k;
    Arena A(base);
    st->cap = L == 1 ? c.max_roots : c.max_roots * (1 + (int64_t)k);
    st->slots.resize((size_t)c.slots);
    if (c.tgn) {
        // a TGN batch: max_roots = 2 x (edges of a batch); every edge may be embedded (m = n).  The blob of tg_tgn_prepare_batch, the
        // slot lists behind it and the distinct-node lists, per slot
        const int64_t n = c.max_roots / 2;
        int64_t off[8];
        TG_TRY(tg_tgn_prepare_layout(n, n, k, off));
        const int64_t total = off[7] - off[4], mk = 2 * n * k;
        for (Slot& s : st->slots) {
            s.blob = reinterpret_cast<int32_t*>(A.take(off[7]));
            s.S_eid = reinterpret_cast<int32_t*>(A.take(mk)); s.S_t = A.take(mk); s.S_dt = A.take(mk);
            s.uniq = reinterpret_cast<int32_t*>(A.take(total)); s.uniq_t = A.take(total); s.rowmap = reinterpret_cast<int32_t*>(A.take(total));
            s.count_pad = reinterpret_cast<int32_t*>(A.take(4));
            s.slot_order = reinterpret_cast<int32_t*>(A.take(mk)); s.slot_srow = reinterpret_cast<int32_t*>(A.take(mk));
            s.slot_nvalid = reinterpret_cast<int32_t*>(A.take(4));
        }
        st->ded_cap = tg_dedupe_capacity(total);
        st->ded_keys = A.take(2 * st->ded_cap);
        st->ded_vals = reinterpret_cast<int32_t*>(A.take(st->ded_cap + 1024));
        st->ded_pos = reinterpret_cast<int32_t*>(A.take(total));
        TgnBuf& t = st->tb;
        const int64_t U = total, D = dn, MD = st->md;
        t.max_u = U;
        t.h_rows = A.take(U * D); t.msg_rows = A.take(U * MD); t.gi = A.take(U * 3 * D); t.gh = A.take(U * 3 * D); t.rows = A.take(U * D);
        t.base = A.take(U * D); t.d_own = A.take(c.max_roots * D); t.d_raw = A.take(c.max_roots * D); t.dgi = A.take(U * 3 * D);
        t.dgh = A.take(U * 3 * D); t.msgs = A.take(c.max_roots * MD); t.zero_t = A.take(U);
        t.slot_rows = A.take(mk * D); t.slot_cnt = reinterpret_cast<int32_t*>(A.take(U + 2)); t.slot_rank = reinterpret_cast<int32_t*>(A.take(mk));
    } else
    for (Slot& s : st->slots) {
        s.ids_all = reinterpret_cast<int32_t*>(A.take(st->cap));
        s.S_nbr = reinterpret_cast<int32_t*>(A.take(st->cap * k));
        s.S_eid = reinterpret_cast<int32_t*>(A.take(st->cap * k));
        s.S_t = A.take(st->cap * k);
        s.S_dt = A.take(st->cap * k);
        s.times = reinterpret_cast<double*>(A.take(2 * c.max_roots));
        s.uniq_t = A.take(c.max_roots * k);
        s.child = reinterpret_cast<int32_t*>(A.take(c.max_roots * k));
        s.count_pad = reinterpret_cast<int32_t*>(A.take(4));
    }
    if (!c.tgn) {
        st->ded_cap = tg_dedupe_capacity(c.max_roots * k);
        st->ded_keys = A.take(2 * st->ded_cap);
        st->ded_vals = reinterpret_cast<int32_t*>(A.take(st->ded_cap + 1024));
        st->ded_pos = reinterpret_cast<int32_t*>(A.take(c.max_roots * k));
    }
    st->cosb = A.take(T);
    st->lay.resize((size_t)L);
    const int64_t wt = tg_tgat_layer_wt_floats(dn, dq, dk);
    for (int l = 1; l <= L; ++l) {
        LayerBuf& b = st->lay[(size_t)l - 1];
        const int64_t R = l == L ? c.max_roots : st->cap;
        b.max_rows = R;
        b.qbias = A.take(dq); b.q = A.take(R * dq); b.u = A.take(R * H * dk); b.agg = A.take(R * H * dk); b.prob = A.take(R * H * k);
        b.ctx = A.take(R * dq); b.res = A.take(R * dq); b.y = A.take(R * (dq + dn)); b.mean = A.take(R); b.rstd = A.take(R);
        b.f1 = A.take(R * dn); b.wT = A.take(wt); b.out = A.take(R * dn);
        b.df1 = A.take(R * dn); b.dy = A.take(R * dq); b.dsum = A.take(R * dq); b.dres = A.take(R * dq); b.dctx = A.take(R * dq);
        b.dagg = A.take(R * H * dk); b.du = A.take(R * H * dk); b.dq = A.take(R * dq);
        b.part_floats = tg_tgat_layer_part_floats(R, dn, dq, T) + 32;
        b.part = A.take(b.part_floats);
    }
    // gradient block, zero-filled once per backward
    st->vlen = r4(tg_tgat_layer_vec_floats(dn, dq, dk, H));
    const int64_t npar = st->ptotal;
    st->g_extra = npar;
    st->g_cosb = st->g_extra + r4(c.extra_grad_floats);
    st->g_vec = st->g_cosb + r4(T);
    st->g_rows = st->g_vec + L * st->vlen;
    int64_t tot = st->g_rows;
    for (int l = L; l > 1; --l) tot += r4(st->lay[(size_t)l - 2].max_rows * dn);      // gradient rows of layer l - 1's output
    if (c.tgn) tot += r4(st->tb.max_u * dn);                                          // gradient of the compact layer-0 table (memory' + raw)
    st->g_floats = tot;
    st->gblock = A.take(tot);
    *total = A.off;
    return TG_OK;
}

int check_cfg(const tg_stepper_cfg* c) {
    TG_REQUIRE(c && c->graph && c->d_node && c->d_edge && c->d_param, "tg_stepper: null pointer in the configuration");
    TG_REQUIRE(c->layers == 1 || c->layers == 2, "tg_stepper: 1 or 2 attention layers (deeper recursions take the Python engine)");
    TG_REQUIRE(c->heads == 1 || c->heads == 2, "tg_stepper: 1 or 2 heads");
    TG_REQUIRE(c->k > 0, "Number of sampled neighbors for each node should be greater than 0!");
    TG_REQUIRE((c->dn + c->dt_dim) % c->heads == 0, "The sum of node_feat_dim and time_feat_dim should be divided by num_heads!");
    TG_REQUIRE(c->max_roots > 0 && c->slots >= 1 && c->slots <= 16, "tg_stepper: max_roots / slots");
    TG_REQUIRE(c->dn > 0 && c->de >= 0 && c->dt_dim > 0 && c->extra_grad_floats >= 0, "tg_stepper: dimensions");
    TG_REQUIRE(!c->tgn || (c->layers == 1 && c->max_roots % 2 == 0), "tg_stepper: the memory stage sits under ONE attention layer (the reference's TGN)");
    return TG_OK;
}

void param_offsets(tg_stepper* st) {
    const tg_stepper_cfg& c = st->c;
    const int dn = c.dn, T = c.dt_dim, dq = st->dq, dk = st->dk;
    std::vector<int64_t> sz = {T, T};
    for (int l = 0; l < c.layers; ++l) {
        const int64_t per[11] = {(int64_t)dq * dq, (int64_t)dq * dk, (int64_t)dq * dk, dq, dq, (int64_t)dq * dq, dq,
                                 (int64_t)dn * (dq + dn), dn, (int64_t)dn * dn, dn};
        sz.insert(sz.end(), per, per + 11);
    }
    st->poff.clear();
    int64_t o = 0;
    for (int64_t s : sz) { st->poff.push_back(o); o += r4(s); }
    st->poff.push_back(o);                                      // = where the layer parameters end
    st->md = 2 * dn + T + c.de;
    if (c.tgn) {                                                // nn.GRUCell(message_dim, memory_dim): weight_ih, weight_hh, bias_ih, bias_hh
        const int64_t gz[4] = {3 * (int64_t)dn * st->md, 3 * (int64_t)dn * dn, 3 * dn, 3 * dn};
        for (int i = 0; i < 4; ++i) { st->gru_off[i] = o; o += r4(gz[i]); }
    }
    st->ptotal = o;
}

tg_layer_params params_at(float* base, const std::vector<int64_t>& poff, int l /* 0-based */) {
    const int64_t* o = &poff[2 + (size_t)l * 11];
    return tg_layer_params{base + o[0], base + o[1], base + o[2], base + o[3], base + o[4], base + o[5], base + o[6], base + o[7],
                           base + o[8], base + o[9], base + o[10]};
}

}  // namespace

extern "C" int64_t tg_stepper_param_floats(const tg_stepper_cfg* cfg) {
    if (check_cfg(cfg) != TG_OK) return -1;
    tg_stepper st{};
    st.c = *cfg;
    st.dq = cfg->dn + cfg->dt_dim; st.dk = cfg->dn + cfg->de + cfg->dt_dim; st.hd = st.dq / cfg->heads;
    param_offsets(&st);
    return st.ptotal;
}

extern "C" int64_t tg_stepper_arena_floats(const tg_stepper_cfg* cfg) {
    if (check_cfg(cfg) != TG_OK) return -1;
    tg_stepper st{};
    st.c = *cfg;
    st.dq = cfg->dn + cfg->dt_dim; st.dk = cfg->dn + cfg->de + cfg->dt_dim; st.hd = st.dq / cfg->heads;
    param_offsets(&st);
    int64_t total = 0;
    if (layout(&st, nullptr, &total) != TG_OK) return -1;
    return total;
}

extern "C" void tg_stepper_destroy(tg_stepper* st) {
    if (!st) return;
    for (Slot& s : st->slots)
        for (hipEvent_t e : {s.copied, s.counted, s.ready, s.consumed})
            if (e) (void)hipEventDestroy(e);
    if (st->side) (void)hipStreamDestroy(st->side);
    if (st->pinned) (void)hipHostFree(st->pinned);
    delete st;
}

extern "C" int tg_stepper_create(const tg_stepper_cfg* cfg, float* d_arena, int64_t arena_floats, tg_stepper** out) {
    TG_TRY(check_cfg(cfg));
    TG_REQUIRE(d_arena && out && (reinterpret_cast<uintptr_t>(d_arena) & 255) == 0, "tg_stepper_create: the arena must be 256-byte aligned");
    TG_REQUIRE((reinterpret_cast<uintptr_t>(cfg->d_param) & 15) == 0, "tg_stepper_create: the flat parameter must be 16-byte aligned");
    tg_stepper* st = new tg_stepper{};
    st->c = *cfg;
    st->dq = cfg->dn + cfg->dt_dim; st->dk = cfg->dn + cfg->de + cfg->dt_dim; st->hd = st->dq / cfg->heads;
    param_offsets(st);
    int64_t total = 0;
    int rc = layout(st, d_arena, &total);
    if (rc == TG_OK && total > arena_floats) { tg::set_error("invalid argument: tg_stepper_create: arena smaller than tg_stepper_arena_floats()"); rc = TG_EINVAL; }
    if (rc == TG_OK && cfg->param_floats != st->ptotal) { tg::set_error("invalid argument: tg_stepper_create: param_floats != tg_stepper_param_floats()"); rc = TG_EINVAL; }
    if (rc != TG_OK) { delete st; return rc; }
    auto fail = [&](const char* what, hipError_t e) { tg::set_error(std::string(what) + ": " + hipGetErrorString(e)); tg_stepper_destroy(st); return TG_EHIP; };
    hipError_t e = hipStreamCreateWithFlags(&st->side, hipStreamNonBlocking);
    if (e != hipSuccess) return fail("hipStreamCreateWithFlags", e);
    // pinned staging per slot: ids (int32) | times (f64) | (count, pad); TGN: the staged blob of tg_tgn_prepare_batch | (count, pad)
    int64_t toff[8] = {};
    if (cfg->tgn) (void)tg_tgn_prepare_layout(cfg->max_roots / 2, cfg->max_roots / 2, cfg->k, toff);
    const int64_t per = cfg->tgn ? r64(toff[6]) * 4 + 64 : r4(cfg->max_roots) * 4 + cfg->max_roots * 8 + 64;
    e = hipHostMalloc(&st->pinned, (size_t)(per * cfg->slots), hipHostMallocDefault);
    if (e != hipSuccess) return fail("hipHostMalloc", e);
    char* hp = reinterpret_cast<char*>(st->pinned);
    for (Slot& s : st->slots) {
        if (cfg->tgn) {
            s.h_stage = hp;
            s.h_count_pad = reinterpret_cast<int32_t*>(hp + r64(toff[6]) * 4);
            s.h_u.resize((size_t)cfg->max_roots); s.h_newt.resize((size_t)cfg->max_roots);
        } else {
            s.h_times = reinterpret_cast<double*>(hp);
            s.h_ids = reinterpret_cast<int32_t*>(hp + cfg->max_roots * 8);
            s.h_count_pad = reinterpret_cast<int32_t*>(hp + cfg->max_roots * 8 + r4(cfg->max_roots) * 4);
        }
        hp += per;
        for (hipEvent_t* ev : {&s.copied, &s.counted, &s.ready, &s.consumed}) {
            e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
            if (e != hipSuccess) return fail("hipEventCreateWithFlags", e);
        }
    }
    if (cfg->tgn) {                       // the touched-node set's times: all zero, never written
        e = hipMemset(st->tb.zero_t, 0, sizeof(float) * (size_t)st->tb.max_u);
        if (e != hipSuccess) return fail("hipMemset", e);
    }
    *out = st;
    return TG_OK;
}

// offsets (floats, from the arena base the caller allocated) a binding needs to expose arena regions as its own tensors:
//   [0] gradient block  [1] floats in it that are parameter gradients (+ extra)  [2] embeddings of the last layer  [3] d cos b
extern "C" int tg_stepper_regions(const tg_stepper* st, const float* d_arena, int64_t* off4) {
    TG_REQUIRE(st && d_arena && off4, "tg_stepper_regions: null pointer");
    off4[0] = st->gblock - d_arena;
    off4[1] = st->g_cosb;
    off4[2] = st->lay.back().out - d_arena;
    off4[3] = st->gblock + st->g_cosb - d_arena;
    return TG_OK;
}

// ---- graph-only preparation of a batch (side stream; the host never waits for the GPU here) ---------------------------------------
// h_ids / h_times: n_roots host values (the trainers' numpy int64 ids / float64 times, PTCL/EM_warmup.py:128-130; several root lists
// of one batch -- [src | dst], [src | dst | negative dst] -- are passed concatenated, their times repeated).
extern "C" int tg_stepper_prepare_begin(tg_stepper* st, int slot, const int64_t* h_ids, const double* h_times, int64_t n_roots) {
    TG_REQUIRE(st && h_ids && h_times, "tg_stepper_prepare_begin: null pointer");
    TG_REQUIRE(!st->c.tgn, "tg_stepper_prepare_begin: a TGN stepper prepares with tg_stepper_tgn_prepare_begin");
    TG_REQUIRE(slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_prepare_begin: slot");
    TG_REQUIRE(n_roots > 0 && n_roots <= st->c.max_roots, "tg_stepper_prepare_begin: more roots than the stepper was sized for");
    Slot& s = st->slots[(size_t)slot];
    TG_REQUIRE(s.state == Slot::FREE, "tg_stepper_prepare_begin: the slot still holds a batch (finish its step or release it)");
    const int64_t nrows = tg_graph_num_rows(st->c.graph);
    for (int64_t i = 0; i < n_roots; ++i)
        if (h_ids[i] < 0 || h_ids[i] >= nrows) { tg::set_error("list index out of range"); return TG_ERANGE; }   // utils/utils.py:141
    const int k = st->c.k;
    hipStream_t sd = st->side;
    if (s.consumed_pending) {             // the main stream may still read this slot's lists (the step that used it last)
        TG_HIP_CHECK(hipStreamWaitEvent(sd, s.consumed, 0));
        s.consumed_pending = false;
    }
    if (s.n > 0) TG_HIP_CHECK(hipEventSynchronize(s.copied));      // the pinned block's previous copy has left (long ago)
    for (int64_t i = 0; i < n_roots; ++i) { s.h_ids[i] = (int32_t)h_ids[i]; s.h_times[i] = h_times[i]; }
    s.n = n_roots;
    TG_HIP_CHECK(hipMemcpyAsync(s.times, s.h_times, (size_t)n_roots * 8, hipMemcpyHostToDevice, sd));
    TG_HIP_CHECK(hipMemcpyAsync(s.ids_all, s.h_ids, (size_t)n_roots * 4, hipMemcpyHostToDevice, sd));
    TG_HIP_CHECK(hipEventRecord(s.copied, sd));
    TG_TRY(tg_sample_recent(st->c.graph, s.ids_all, s.times, nullptr, n_roots, k, s.S_nbr, s.S_eid, s.S_t, s.S_dt, nullptr, sd));
    s.count1 = 0;
    s.pad = -1;
    if (st->c.layers == 2) {
        const int64_t nk = n_roots * k;
        if (st->c.dedupe) {
            TG_TRY(tg_dedupe_pairs(s.S_nbr, s.S_t, nk, st->ded_cap, st->ded_keys, st->ded_vals, st->ded_pos, (int32_t)n_roots, s.ids_all + n_roots,
                                   s.uniq_t, s.child, s.count_pad, sd));
            TG_HIP_CHECK(hipMemcpyAsync(s.h_count_pad, s.count_pad, 8, hipMemcpyDeviceToHost, sd));
            TG_HIP_CHECK(hipEventRecord(s.counted, sd));
        } else {                          // the reference's row-for-row recursion: every slot its own row
            TG_HIP_CHECK(hipMemcpyAsync(s.ids_all + n_roots, s.S_nbr, (size_t)nk * 4, hipMemcpyDeviceToDevice, sd));
            TG_HIP_CHECK(hipMemcpyAsync(s.uniq_t, s.S_t, (size_t)nk * 4, hipMemcpyDeviceToDevice, sd));
            iota_kernel<<<(unsigned)std::min<int64_t>((nk + 255) / 256, 1024), 256, 0, sd>>>(s.child, nk, (int32_t)n_roots);
            TG_TRY(tg::launch_status("iota_kernel"));
            s.count1 = nk;
        }
    }
    s.state = Slot::BEGUN;
    return TG_OK;
}

// second half, a step later: reads the distinct-row count (pinned word; waits only if the side stream has not got there yet) and issues
// the level-1 lookups.  rows2[0] = roots, rows2[1] = distinct level-1 rows.
extern "C" int tg_stepper_prepare_finish(tg_stepper* st, int slot, int64_t* rows2) {
    TG_REQUIRE(st && slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_prepare_finish: slot");
    Slot& s = st->slots[(size_t)slot];
    TG_REQUIRE(s.state == Slot::BEGUN, "tg_stepper_prepare_finish: prepare_begin first");
    hipStream_t sd = st->side;
    const int k = st->c.k;
    if (st->c.tgn) {
        TG_HIP_CHECK(hipEventSynchronize(s.counted));
        s.uniq_count = s.h_count_pad[0];
        s.pad = s.h_count_pad[1];
        TG_REQUIRE(s.uniq_count >= 0 && s.uniq_count <= st->tb.max_u, "tg_stepper_prepare_finish: distinct-node count out of range");
        s.count1 = s.uniq_count;
    } else if (st->c.layers == 2) {
        if (st->c.dedupe) {
            TG_HIP_CHECK(hipEventSynchronize(s.counted));
            s.count1 = s.h_count_pad[0];
            s.pad = s.h_count_pad[1] >= 0 ? s.h_count_pad[1] + (int32_t)s.n : -1;
            TG_REQUIRE(s.count1 >= 0 && s.count1 <= s.n * k, "tg_stepper_prepare_finish: distinct-row count out of range");
        }
        if (s.count1 > 0)
            TG_TRY(tg_sample_recent(st->c.graph, s.ids_all + s.n, nullptr, s.uniq_t, s.count1, k, s.S_nbr + s.n * k, s.S_eid + s.n * k,
                                    s.S_t + s.n * k, s.S_dt + s.n * k, nullptr, sd));
    }
    TG_HIP_CHECK(hipEventRecord(s.ready, sd));
    s.state = Slot::READY;
    if (rows2) { rows2[0] = s.n; rows2[1] = s.count1; }
    return TG_OK;
}

extern "C" int tg_stepper_release(tg_stepper* st, int slot) {
    TG_REQUIRE(st && slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_release: slot");
    Slot& s = st->slots[(size_t)slot];
    if (s.state == Slot::FORWARDED) {                 // a forward without its backward: the main stream may still read the slot's lists
        if (st->fwd_slot == slot) st->fwd_slot = -1;
        TG_HIP_CHECK(hipDeviceSynchronize());
    }
    s.state = Slot::FREE;
    return TG_OK;
}

// The trainers swap the model's neighbor sampler between the train graph and the full graph every epoch (PTCL/EM_warmup.py:118, :296;
// PTCL/M_step.py:34, :200): the object follows.  Only between batches: no slot may hold a batch in preparation or a forward without
// its backward (tg_stepper_release first).  The new graph must cover the same id space (the arena is sized by roots, not by nodes).
extern "C" int tg_stepper_set_graph(tg_stepper* st, const tg_graph* graph) {
    TG_REQUIRE(st && graph, "tg_stepper_set_graph: null pointer");
    for (const Slot& s : st->slots) TG_REQUIRE(s.state == Slot::FREE, "tg_stepper_set_graph: a slot still holds a batch (release it first)");
    st->c.graph = graph;
    return TG_OK;
}

// device views of a prepared slot for tests / other consumers: ids_all, S_nbr, S_eid, S_t, S_dt, child (pointers), pad row
extern "C" int tg_stepper_slot_view(const tg_stepper* st, int slot, void** p6, int64_t* pad_row) {
    TG_REQUIRE(st && p6 && slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_slot_view: slot");
    const Slot& s = st->slots[(size_t)slot];
    p6[0] = s.ids_all; p6[1] = s.S_nbr; p6[2] = s.S_eid; p6[3] = s.S_t; p6[4] = s.S_dt; p6[5] = s.child;
    if (pad_row) *pad_row = s.pad;
    return TG_OK;
}

// ---- forward of every layer ------------------------------------------------------------------------------------------------------------
// seeds: 2 per layer (attention dropout, residual dropout), layer 1 first -- the order flid_amd/engine.py draws them in.
// *d_emb: (roots, dn) embeddings h^L inside the arena, valid until the next forward.
namespace {
// what the lowest layer reads: the node table by node id (TGAT), or a compact per-batch table through row maps (TGN)
struct Base { const float* table; int64_t ld; const int32_t* feat_idx0; const int32_t* gather_idx; };
int run_forward(tg_stepper* st, Slot& s, int slot, const Base& base, int training, const uint64_t* seeds, void* stream, float** d_emb,
                int64_t fill_extra = 0);
}  // namespace

extern "C" int tg_stepper_forward(tg_stepper* st, int slot, int training, const uint64_t* seeds, void* stream, float** d_emb) {
    TG_REQUIRE(st && slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_forward: slot");
    TG_REQUIRE(!st->c.tgn, "tg_stepper_forward: a TGN stepper runs tg_stepper_tgn_forward");
    Slot& s = st->slots[(size_t)slot];
    TG_REQUIRE(s.state == Slot::READY, "tg_stepper_forward: the slot holds no finished preparation");
    TG_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, s.ready, 0));
    return run_forward(st, s, slot, Base{st->c.d_node, st->c.node_ld, s.S_nbr, s.ids_all}, training, seeds, stream, d_emb);
}

namespace {
int run_forward(tg_stepper* st, Slot& s, int slot, const Base& base, int training, const uint64_t* seeds, void* stream, float** d_emb,
                int64_t fill_extra) {
    const tg_stepper_cfg& c = st->c;
    const int L = c.layers, H = c.heads, dn = c.dn, T = c.dt_dim, dq = st->dq, k = c.k;
    const float p_eff = training ? c.dropout_p : 0.f;              // (training == 2: train mode, the gradient block is left alone)
    TG_REQUIRE(p_eff == 0.f || seeds, "tg_stepper_forward: dropout needs seeds");
    float* te_w = c.d_param + st->poff[0];
    float* te_b = c.d_param + st->poff[1];
    st->desc.assign((size_t)L, tg_layer_desc{});
    const float* H_prev = nullptr;
    for (int l = 1; l <= L; ++l) {
        const int64_t R = rows_of_layer(st, l, s.n, s.count1);
        LayerBuf& b = st->lay[(size_t)l - 1];
        TG_REQUIRE(R <= b.max_rows, "tg_stepper_forward: more rows than the arena holds");
        st->fwd_rows[l - 1] = R;
        tg_layer_desc& d = st->desc[(size_t)l - 1];
        tg_attn_desc& a = d.attn;
        if (l == 1) { a.d_feat = base.table; a.feat_ld = base.ld; a.d_feat_idx = base.feat_idx0; }
        else { a.d_feat = H_prev; a.feat_ld = dn; a.d_feat_idx = s.child; }
        a.d_edge = c.d_edge; a.edge_ld = c.edge_ld; a.d_edge_idx = s.S_eid;
        a.d_nbr = s.S_nbr; a.d_dt = s.S_dt; a.d_te_w = te_w; a.d_te_b = te_b;
        a.m = R; a.k = k; a.heads = H; a.dn = dn; a.de = c.de; a.dt_dim = T;
        a.scale = (float)pow((double)st->hd, -0.5);        // float(head_dim ** -0.5), as the Python engine passes it
        a.dropout_p = p_eff; a.seed = p_eff > 0.f ? seeds[2 * (l - 1)] : 0; a.row0 = 0;
        d.params = params_at(c.d_param, st->poff, l - 1);
        d.raw = b.y + dq; d.raw_ld = dq + dn;
        if (l == 1) { d.own = d.raw; d.own_ld = d.raw_ld; } else { d.own = H_prev; d.own_ld = dn; }
        d.cosb = st->cosb;
        d.res_dropout_p = p_eff; d.res_seed = p_eff > 0.f ? seeds[2 * (l - 1) + 1] : 0;
        d.qbias = b.qbias; d.q = b.q; d.u = b.u; d.agg = b.agg; d.prob = b.prob; d.ctx = b.ctx; d.res = b.res; d.y = b.y;
        d.mean = b.mean; d.rstd = b.rstd; d.f1 = b.f1; d.out = b.out; d.wT = b.wT;
        d.y_ld = dq + dn;
        d.compute_cosb = l == 1;
        d.gather_table = base.table; d.gather_ld = base.ld; d.gather_idx = base.gather_idx;
        H_prev = b.out;
    }
    // In train mode the lowest layer's prelude launch also zero-fills the gradient block the backward accumulates into (parameter
    // gradients + scratch + the gradient rows of the lower layers' outputs + fill_extra floats behind them): one memset launch less.
    // (The upper layer keeps its own prelude launch right in front of its forward: folded into the first launch as well it ran 7 us
    // faster by itself, but the packed weights it writes were no longer L2-resident when the 1 200-row chains stream them 200-400 us
    // later -- chain_fwd / chain_bwd of the root layer 27 -> 32 and 31 -> 40 us, the step 7 us slower on the same box.)
    const tg_layer_desc* l0 = &st->desc[0];
    int64_t fill = 0;
    if (training == 1) {
        fill = st->g_rows + fill_extra;
        for (int l = L; l > 1; --l) fill += r4(st->fwd_rows[l - 2] * dn);
    }
    TG_TRY(tg::layers_forward(1, &l0, fill > 0 ? st->gblock : nullptr, fill, stream));
    for (int l = 2; l <= L; ++l) TG_TRY(tg_tgat_layer_fwd(&st->desc[(size_t)l - 1], stream));
    st->zeroed_floats = fill;
    s.state = Slot::FORWARDED;
    st->fwd_slot = slot;
    if (d_emb) *d_emb = st->lay.back().out;
    return TG_OK;
}
}  // namespace

// ---- backward of every layer (+ the optimizer's update) -------------------------------------------------------------------------------
// d_demb: (roots, dn) gradient of the loss w.r.t. the embeddings.  The gradient block [te_w | te_b | layer parameters ...], laid out like
// the flat parameter, is zero-filled and accumulated into here; *d_grad points at it (valid until the next backward).
// grad_ready (optional): called with each upper layer's finished block as soon as that layer's backward is queued (a data-parallel
// caller starts reducing it under the lower layers' backward); everything on side streams is joined first.
// adam (optional): the update of torch.optim.Adam on the flat parameter right behind the last layer's backward -- not with grad_ready
// (the caller reduces first and applies the update itself).
namespace {
// gradient w.r.t. the lowest layer's table (TGN: the compact `memory' + raw` table) -- null for TGAT, whose node table carries none
struct BaseGrad {
    float* d_table; int64_t pad_row; float* d_own; float* d_raw;
    // optional: the layer's feature gradient as per-slot rows + a segmented sum over the slot order (tg::attn_bwd_slot_rows_next)
    float* slot_rows = nullptr; const int32_t *order = nullptr, *srow = nullptr, *nvalid = nullptr; int64_t nslots = 0;
    // ... whose launch then also adds d_own + d_raw into the roots' rows own_idx[0, n_own) of the table; *own_done says that it did
    const int32_t* own_idx = nullptr; int64_t n_own = 0; bool* own_done = nullptr;
};

// the layers' backward calls; `fill_extra` floats behind the lower layers' gradient rows are zeroed with the gradient block
// defer_last: the LAST layer's weight gradients are kept back too (the caller has another grouped launch coming on this stream that carries
// them: TGN's GRU) -- the caller then owns the final tg::wgrad_flush_deferred
int run_backward(tg_stepper* st, Slot& s, const float* d_demb, void* stream, tg_grad_ready_fn grad_ready, void* user, bool fuse_tb,
                 const BaseGrad& bg, int64_t fill_extra, bool accumulate = false, bool defer_last = false) {
    const tg_stepper_cfg& c = st->c;
    const int L = c.layers, dn = c.dn;
    hipStream_t ms = (hipStream_t)stream;
    // ONE zero fill: parameter gradients + scratch + the gradient rows of the lower layers' outputs that this batch has
    int64_t fill = st->g_rows;
    for (int l = L; l > 1; --l) fill += r4(st->fwd_rows[l - 2] * dn);
    if (accumulate) {
        // the parameter gradients (and d cos b) of an earlier backward stay; the per-call scratch behind them starts from zero again
        TG_HIP_CHECK(hipMemsetAsync(st->gblock + st->g_vec, 0, sizeof(float) * (size_t)(fill + fill_extra - st->g_vec), ms));
    } else if (st->zeroed_floats < fill + fill_extra) {       // (else: zero-filled by the forward's prelude launch)
        TG_HIP_CHECK(hipMemsetAsync(st->gblock, 0, sizeof(float) * (size_t)(fill + fill_extra), ms));
    }
    st->zeroed_floats = 0;
    float* g = st->gblock;
    const float* dH = d_demb;
    int64_t rows_off = st->g_rows;
    int rc = TG_OK;
    for (int l = L; l >= 1 && rc == TG_OK; --l) {
        LayerBuf& b = st->lay[(size_t)l - 1];
        tg_layer_bwd_desc bw{};
        const tg_layer_params gp = params_at(g, st->poff, l - 1);
        bw.grads = tg_layer_grads{const_cast<float*>(gp.Wq), const_cast<float*>(gp.Wk), const_cast<float*>(gp.Wv), const_cast<float*>(gp.ln_g),
                                  const_cast<float*>(gp.ln_b), const_cast<float*>(gp.Wr), const_cast<float*>(gp.br), const_cast<float*>(gp.W1),
                                  const_cast<float*>(gp.b1), const_cast<float*>(gp.W2), const_cast<float*>(gp.b2)};
        bw.dout = dH;
        bw.df1 = b.df1; bw.dy = b.dy; bw.dsum = b.dsum; bw.dres = st->desc[(size_t)l - 1].res_dropout_p > 0.f ? b.dres : nullptr;
        bw.dctx = b.dctx; bw.dagg = b.dagg; bw.du = b.du; bw.dq = b.dq; bw.part = b.part;
        bw.vec = g + st->g_vec + (l - 1) * st->vlen;
        bw.d_cosb = g + st->g_cosb; bw.d_tew = g + st->poff[0]; bw.d_teb = g + st->poff[1];
        float* dH_prev = nullptr;
        if (l >= 2) {
            dH_prev = g + rows_off;
            rows_off += r4(st->fwd_rows[l - 2] * dn);
            bw.dfeat = dH_prev; bw.dfeat_ld = dn; bw.pad_row = s.pad;
            bw.d_own = dH_prev; bw.d_own_ld = dn; bw.d_own_accumulate = 1;
        } else {
            // TGAT: the node table carries no gradient (models/TGAT.py:26-29); TGN: the compact table does (the GRU's output)
            bw.dfeat = bg.d_table; bw.dfeat_ld = bg.d_table ? dn : 0; bw.pad_row = bg.d_table ? bg.pad_row : 0;
            bw.d_own = bg.d_own; bw.d_own_ld = bg.d_own ? dn : 0; bw.d_own_accumulate = 0;
            bw.d_raw = bg.d_raw;
        }
        bw.defer_join = (l > 1 && !grad_ready) ? 1 : 0;
        bw.finish_time_bias = (l == 1 && !fuse_tb) ? 1 : 0;
        // An upper layer's weight gradients (1 200 rows: a ~19 + 12 us latency chain on a fraction of the chip) ride in the grouped launch
        // of the layer below -- unless a data-parallel caller wants this layer's block now (grad_ready).
        tg::wgrad_defer_next(!grad_ready && (l > 1 || defer_last));
        const bool slot_mode = l == 1 && bg.d_table && bg.slot_rows;
        if (slot_mode) tg::attn_bwd_slot_rows_next(bg.slot_rows);
        rc = tg_tgat_layer_bwd(&st->desc[(size_t)l - 1], &bw, stream);
        tg::wgrad_defer_next(false);
        if (slot_mode) {
            tg::attn_bwd_slot_rows_next(nullptr);
            // (behind the layer's other launches: nothing of the layer reads the table's gradient)
            if (rc == TG_OK && tg::attn_bwd_slot_rows_taken()) {
                const bool own = bg.own_idx && bg.d_own && bg.d_raw && bg.n_own > 0;
                rc = tg::slot_rows_sum(bg.slot_rows, dn, bg.order, bg.srow, bg.nvalid, bg.nslots, bg.d_table, dn, (hipStream_t)stream,
                                       own ? bg.d_own : nullptr, own ? bg.d_raw : nullptr, own ? bg.own_idx : nullptr, own ? bg.n_own : 0);
                if (rc == TG_OK && own && bg.own_done) *bg.own_done = true;
            }
        }
        if (rc == TG_OK && grad_ready && l >= 2) {
            const int64_t lo = st->poff[2 + (size_t)(l - 1) * 11], hi = l < L ? st->poff[2 + (size_t)l * 11] : st->poff.back();
            grad_ready(user, g + lo, hi - lo);
        }
        dH = dH_prev;
    }
    const int rj = tg_side_join(stream);          // queued side-stream products must not outlive this call's operands
    const int rf = defer_last ? (int)TG_OK : tg::wgrad_flush_deferred((hipStream_t)stream);      // (nothing picked a kept-back group up: it leaves alone)
    return rc != TG_OK ? rc : (rj != TG_OK ? rj : rf);
}

int finish_backward(tg_stepper* st, Slot& s, int rc, void* stream, const tg_adam_args* adam, float** d_grad) {
    const tg_stepper_cfg& c = st->c;
    float* g = st->gblock;
    if (rc == TG_OK && adam)       // d b -= sin(b) d cos(b) rides in the update's launch
        rc = tg::adam_time_bias(c.d_param, g, adam->d_exp_avg, adam->d_exp_avg_sq, adam->n > 0 ? adam->n : st->ptotal, adam->lr, adam->beta1,
                                adam->beta2, adam->eps, adam->weight_decay, adam->step, st->poff[1], c.dt_dim, g + st->g_cosb, stream);
    // the slot's lists have been read for the last time once everything above has run
    if (hipEventRecord(s.consumed, (hipStream_t)stream) == hipSuccess) s.consumed_pending = true;
    s.state = Slot::FREE;
    st->fwd_slot = -1;
    if (d_grad) *d_grad = g;
    return rc;
}
}  // namespace

extern "C" int tg_stepper_backward(tg_stepper* st, int slot, const float* d_demb, void* stream, tg_grad_ready_fn grad_ready, void* user,
                                   const tg_adam_args* adam, float** d_grad) {
    TG_REQUIRE(st && d_demb && slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_backward: arguments");
    TG_REQUIRE(!st->c.tgn, "tg_stepper_backward: a TGN stepper runs tg_stepper_tgn_backward");
    Slot& s = st->slots[(size_t)slot];
    TG_REQUIRE(s.state == Slot::FORWARDED && st->fwd_slot == slot, "tg_stepper_backward: the last forward ran another slot");
    TG_REQUIRE(!(adam && grad_ready), "tg_stepper_backward: reduce first, then apply the update");
    const int rc = run_backward(st, s, d_demb, stream, grad_ready, user, adam != nullptr, BaseGrad{nullptr, 0, nullptr, nullptr}, 0);
    return finish_backward(st, s, rc, stream, adam, d_grad);
}

// ==== TGN: the memory stage around ONE attention layer ===================================================================================
// replaces the host side of models/MemoryModel.py:96-189 (compute_src_dst_node_temporal_embeddings: get_updated_memories over the
// touched nodes, the embedding, update_memories / store_node_raw_messages of a positive batch) for the fused trainers, as
// flid_amd/models/MemoryModel.py::train_step did from Python.  The flat parameter carries the GRU cell's four tensors behind the layer's.

// graph-only part of a batch on the side stream: tg_tgn_prepare_batch on the slot's buffers.  n edges; the embedded shard is [lo, hi).
extern "C" int tg_stepper_tgn_prepare_begin(tg_stepper* st, int slot, const int64_t* h_src, const int64_t* h_dst, const double* h_t,
                                            const int64_t* h_eid, int64_t n, int64_t lo, int64_t hi) {
    TG_REQUIRE(st && st->c.tgn && h_src && h_dst && h_t, "tg_stepper_tgn_prepare_begin: arguments");
    TG_REQUIRE(slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_tgn_prepare_begin: slot");
    TG_REQUIRE(n > 0 && 2 * n <= st->c.max_roots && 0 <= lo && lo < hi && hi <= n, "tg_stepper_tgn_prepare_begin: more edges than the stepper was sized for");
    Slot& s = st->slots[(size_t)slot];
    TG_REQUIRE(s.state == Slot::FREE, "tg_stepper_tgn_prepare_begin: the slot still holds a batch (finish its step or release it)");
    const int k = st->c.k;
    hipStream_t sd = st->side;
    if (s.consumed_pending) { TG_HIP_CHECK(hipStreamWaitEvent(sd, s.consumed, 0)); s.consumed_pending = false; }
    if (s.nb > 0) TG_HIP_CHECK(hipEventSynchronize(s.copied));
    TG_TRY(tg_tgn_prepare_layout(n, hi - lo, k, s.off));
    const int64_t total = s.off[7] - s.off[4];
    int64_t nu = 0;
    // (the times of the touched-node set are all zero: st->tb.zero_t is zero-filled at creation and never written)
    TG_TRY(tg_tgn_prepare_batch(st->c.graph, h_src, h_dst, h_t, h_eid, n, lo, hi, k, tg_graph_num_rows(st->c.graph), s.h_stage, s.blob, s.S_eid, s.S_t,
                                s.S_dt, st->tb.zero_t, tg_dedupe_capacity(total), st->ded_keys, st->ded_vals, st->ded_pos, s.uniq, s.uniq_t, s.rowmap,
                                s.count_pad, s.h_count_pad, s.h_u.data(), s.h_newt.data(), &nu, sd));
    TG_HIP_CHECK(hipEventRecord(s.copied, sd));
    TG_HIP_CHECK(hipEventRecord(s.counted, sd));
    s.h_nu = nu; s.nb = n; s.lo = lo; s.hi = hi; s.n = 2 * (hi - lo); s.has_eid = h_eid != nullptr;
    s.S_nbr = s.blob + s.off[6];
    // the neighbor slots grouped by the compact row they gather: what the backward's feature-gradient sum walks (tg::slot_rows_sum)
    if (slot_rows_on(st)) TG_TRY(tg::build_slot_order(s.rowmap + s.n + 2 * n, s.S_nbr, s.n * k, st->tb.max_u + 1, s.count_pad, st->tb.slot_cnt, st->tb.slot_rank,
                                                s.slot_order, s.slot_srow, s.slot_nvalid, sd));
    s.ids_all = s.rowmap;                 // the roots' rows of the compact table
    s.state = Slot::BEGUN;
    return TG_OK;
}

namespace {
inline bool any_pending(const tg_tgn_bank* b) {
    for (int64_t i = 0; i < b->num_nodes; ++i) if (b->h_has[i]) return true;
    return false;
}
int check_bank(const tg_tgn_bank* b, const tg_stepper* st) {
    TG_REQUIRE(b && b->d_mem && b->d_last_update && b->d_msg && b->d_has && b->d_msg_time && b->d_last_idx_ws && b->h_has && b->h_msg_time && b->h_last,
               "tg_stepper_tgn: null pointer in the memory bank");
    TG_REQUIRE(b->num_nodes == tg_graph_num_rows(st->c.graph) || b->num_nodes > 0, "tg_stepper_tgn: num_nodes");
    return TG_OK;
}
}  // namespace

// updated memory rows of the touched nodes (GRU on their pending messages, not persisted), `memory' + raw` as the layer's table, the layer
extern "C" int tg_stepper_tgn_forward(tg_stepper* st, int slot, const tg_tgn_bank* bank, int training, const uint64_t* seeds, void* stream,
                                      float** d_emb, int keep_grad) {
    TG_REQUIRE(st && st->c.tgn && slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_tgn_forward: arguments");
    TG_TRY(check_bank(bank, st));
    Slot& s = st->slots[(size_t)slot];
    TG_REQUIRE(s.state == Slot::READY, "tg_stepper_tgn_forward: the slot holds no finished preparation");
    TG_REQUIRE(!bank->past_violation, "Trying to update memory to time in the past!");          // models/MemoryModel.py:515-516
    const tg_stepper_cfg& c = st->c;
    const int D = c.dn;
    TG_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, s.ready, 0));
    const bool pending = any_pending(bank);
    st->fwd_pending = pending;
    TgnBuf& t = st->tb;
    float* P = c.d_param;
    TG_TRY(tg_tgn_rows_fwd(bank->d_mem, bank->mem_ld, bank->d_msg, bank->msg_ld, c.d_node, c.node_ld, s.uniq, s.uniq_count, bank->d_has, D, st->md,
                           P + st->gru_off[0], P + st->gru_off[1], P + st->gru_off[2], P + st->gru_off[3], pending ? 1 : 0, t.h_rows,
                           pending ? t.msg_rows : nullptr, pending ? t.gi : nullptr, pending ? t.gh : nullptr, t.rows, t.base, stream));
    const int64_t roots = s.n, nb2 = 2 * s.nb;
    // keep_grad: an earlier backward's gradients wait in the block for this batch's to be added (the warm-up's negative-then-positive
    // pair, PTCL/EM_warmup.py:159-175): no zero fill here
    return run_forward(st, s, slot, Base{t.base, D, s.rowmap + roots + nb2, s.rowmap}, keep_grad ? 2 : training, seeds, stream, d_emb,
                       pending ? r4(s.uniq_count * D) : 0);
}

// backward of the layer and of the GRU; positive != 0: the state advance of models/MemoryModel.py:155-180 (persist the batch nodes' GRU rows,
// build the new raw messages from the post-update state, file them last-message-wins; the host mirrors first: TG_EINVAL "Trying to update
// memory to time in the past!" leaves everything unchanged) BEFORE the update of the parameters, as the trainers order it.
extern "C" int tg_stepper_tgn_backward(tg_stepper* st, int slot, tg_tgn_bank* bank, const float* d_demb, int flags, void* stream,
                                       const tg_adam_args* adam, float** d_grad, tg_grad_ready_fn grad_ready, void* user) {
    const int positive = flags & 1;
    const bool accumulate = (flags & 2) != 0, more = (flags & 4) != 0;
    TG_REQUIRE(st && st->c.tgn && d_demb && slot >= 0 && slot < (int)st->slots.size(), "tg_stepper_tgn_backward: arguments");
    TG_TRY(check_bank(bank, st));
    Slot& s = st->slots[(size_t)slot];
    TG_REQUIRE(s.state == Slot::FORWARDED && st->fwd_slot == slot, "tg_stepper_tgn_backward: the last forward ran another slot");
    TG_REQUIRE(!positive || s.has_eid, "tg_stepper_tgn_backward: a positive batch needs its edge ids (prepare_begin)");
    const tg_stepper_cfg& c = st->c;
    const int D = c.dn, MD = st->md;
    const bool pending = st->fwd_pending;
    TgnBuf& t = st->tb;
    float* g = st->gblock;
    float* d_table = g + st->g_rows;                       // behind the (absent) lower layers' rows: zeroed with the block
    const int64_t U = s.uniq_count, roots = s.n, nb2 = 2 * s.nb;
    TG_REQUIRE(!(more && adam), "tg_stepper_tgn_backward: the update belongs to the LAST backward of a step");
    TG_REQUIRE(!(grad_ready && (adam || more)), "tg_stepper_tgn_backward: grad_ready is for the last backward of a step, before a reduction (no update here)");
    // (fuse_tb argument: true also when more backward calls follow -- d b is finished once, on the summed d cos b, by the last call)
    // (with pending messages the GRU's grouped weight-gradient launch follows and carries the layer's along)
    bool own_done = false;                                 // (the roots' own-row gradient went into the table inside run_backward's launches)
    int rc = run_backward(st, s, d_demb, stream, nullptr, nullptr, adam != nullptr || more,
                          pending ? (slot_rows_on(st) ? BaseGrad{d_table, s.pad, t.d_own, t.d_raw, t.slot_rows, s.slot_order, s.slot_srow, s.slot_nvalid, roots * st->c.k,
                                                                 s.rowmap, roots, &own_done}
                                                 : BaseGrad{d_table, s.pad, t.d_own, t.d_raw})
                                  : BaseGrad{nullptr, 0, nullptr, nullptr}, pending ? r4(U * D) : 0,
                          accumulate, pending && !grad_ready);
    // the attention + merge layer's block is final (the time encoder's two tensors in front of it are not: d b is finished below / by the
    // caller's update): a data-parallel caller starts reducing it under the GRU's backward and the state advance
    if (rc == TG_OK && grad_ready) grad_ready(user, g + st->poff[2], st->poff.back() - st->poff[2]);
    if (rc == TG_OK && pending) {
        // the merge layer's and the query's share of the gradient w.r.t. the roots' own rows, then the GRU
        if (!own_done) rc = tg::scatter_add_rows2(t.d_own, t.d_raw, D, s.rowmap, roots, D, d_table, D, (hipStream_t)stream);
        if (rc == TG_OK) rc = tg_gru_gates_bwd_masked(t.gi, t.gh, t.h_rows, d_table, s.uniq, bank->d_has, U, D, t.dgi, t.dgh, stream);
        if (rc == TG_OK) {
            const tg_wgrad_job jobs[2] = {{t.dgi, 3 * (int64_t)D, 3 * D, t.msg_rows, MD, MD, g + st->gru_off[0], MD, g + st->gru_off[2]},
                                          {t.dgh, 3 * (int64_t)D, 3 * D, t.h_rows, D, D, g + st->gru_off[1], D, g + st->gru_off[3]}};
            rc = tg_wgrad_group(2, jobs, U, stream);
        }
    }
    if (pending && !grad_ready) {                      // (safety net: an error path above may have left the layer's group kept back)
        const int rf = tg::wgrad_flush_deferred((hipStream_t)stream);
        if (rc == TG_OK) rc = rf;
    }
    if (rc == TG_OK && positive) {
        // (units: the edges whose state the advance files -- on a data-parallel rank the WHOLE global batch, section 7 of DESIGN.md)
        tg::ProfScope prof("tgn_advance", (double)(s.nb), (hipStream_t)stream);
        int viol = 0;
        rc = tg_tgn_host_advance(s.h_u.data(), s.h_newt.data(), s.h_nu, bank->h_has, bank->h_msg_time, bank->h_last, bank->num_nodes, &viol);
        if (rc == TG_OK) {
            if (viol) bank->past_violation = 1;
            const int32_t *batch_d = s.blob + s.off[5], *b_d = s.blob + s.off[1], *e_d = s.blob + s.off[2];
            const float* t32_d = reinterpret_cast<const float*>(s.blob + s.off[3]);
            // (the "last message wins" index of the batch's node list rides in the persist launch: it needs nothing of it)
            rc = tg::tgn_persist_index(t.rows, D, s.rowmap + roots, batch_d, bank->d_has, bank->d_msg_time, bank->d_mem, bank->mem_ld, bank->d_last_update, nb2,
                                       D, bank->d_last_idx_ws, (hipStream_t)stream);
            if (rc == TG_OK)      // the new raw messages: only each node's last one, built straight into its row of the pending-message table
                rc = tg::build_scatter_last(bank->d_mem, bank->mem_ld, bank->d_last_update, batch_d, b_d, t32_d, c.d_edge, c.edge_ld, e_d, c.d_param + st->poff[0],
                                            c.d_param + st->poff[1], nb2, D, c.de, c.dt_dim, bank->d_msg, bank->msg_ld, bank->d_has, bank->d_msg_time,
                                            bank->d_last_idx_ws, (hipStream_t)stream);
        }
    }
    return finish_backward(st, s, rc, stream, adam, d_grad);
}
