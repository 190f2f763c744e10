/* flid_tg.h -- C ABI of the MI355X (gfx950) temporal-graph embedding engine.
 *
 * The reference (3205914485/FLiD) has no native layer: its hot path is Python/PyTorch
 * (SURVEY.md 8b "Language / ABI").  These entry points are what a Python `ctypes`/cffi binding of that
 * path binds instead of the reference's per-node Python loops and ATen op chains; each one cites the
 * reference code it replaces (paths relative to the reference root).  Plain pointers and sizes only:
 * no torch types.  All `const T* d_*` / `T* d_*` arguments are DEVICE pointers; `stream` is a hipStream_t
 * passed as void*.  Every function returns 0 on success, a negative TG_E* code otherwise, and never
 * synchronises the stream unless stated.
 */
#ifndef FLID_TG_H
#define FLID_TG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_OK 0
#define TG_EINVAL -1   /* bad argument / unsupported shape */
#define TG_EHIP -2     /* a HIP runtime call failed; tg_last_error() has the text */
#define TG_ENOMEM -3
#define TG_ERANGE -4   /* an id beyond the graph: the reference's `IndexError: list index out of range` */
#define TG_ESHAPE -5   /* a valid request whose shape / alignment this entry point does not cover: nothing was launched, take the general form */

typedef struct tg_graph tg_graph; /* opaque: device CSR of time-sorted incidences */

const char* tg_last_error(void);
int tg_version(void);

/* optional HIP-event timing of the kernel families "attn_fwd", "attn_bwd" (units = algorithmic bytes), "gemm"
 * (units = flops) and "tgn_advance" (the state advance behind a positive TGN step, units = edges filed), recorded on the
 * launch stream; tg_profile_collect synchronises the device.
 * mask: bit 0 = attn_fwd, bit 1 = attn_bwd, bit 2 = gemm, bit 3 = tgn_advance (0 = off). */
void tg_profile_enable(int mask);
int tg_profile_collect(const char* tag, double* ms, double* units, int64_t* count, int reset);

/* ---- temporal adjacency ---------------------------------------------------------------------------
 * replaces utils/utils.py:283-302 get_neighbor_sampler + :73-110 NeighborSampler.__init__
 * HOST arrays of length num_edges.  Every edge is filed under both endpoints; per node the incidences are
 * stable-sorted by time (ties keep stream order).  num_rows = max node id + 1 (row 0 = padding node).
 * Device layout: row_ptr int64[num_rows+1]; entries {int32 nbr, int32 eid, double t} (16 B, AoS). */
int tg_graph_create(const int64_t* h_src, const int64_t* h_dst, const int64_t* h_eid, const double* h_t,
                    int64_t num_edges, int64_t num_rows, tg_graph** out);
void tg_graph_destroy(tg_graph* g);
int64_t tg_graph_num_rows(const tg_graph* g);
int64_t tg_graph_num_entries(const tg_graph* g);
/* copy the CSR back to HOST arrays (row_ptr: num_rows+1; others: num_entries) -- for the numpy-facing
 * sampler methods (uniform / time_interval_aware, get_all_first_hop_neighbors) and tests. */
int tg_graph_export(const tg_graph* g, int64_t* h_row_ptr, int32_t* h_nbr, int32_t* h_eid, double* h_t);

/* ---- neighbor sampling ----------------------------------------------------------------------------
 * replaces utils/utils.py:130-147 find_neighbors_before + :149-214 get_historical_neighbors ('recent').
 * For each of n queries (node id, time): the last k incidences strictly earlier than the query time,
 * right-aligned; unused leading slots are (0, 0, 0.0f).  Query times are float64 (d_times64) or, for hop>=2
 * exactly as the reference feeds them back (models/TGAT.py:110-111), float32 (d_times32); pass exactly one.
 * d_out_dt = query_time - neighbor_time in float32 (models/TGAT.py:120-125), may be NULL.
 * Outputs are (n, k) row-major.  A node id outside [0, num_rows) sets *d_status (int32, may be NULL) to 1
 * (the reference raises IndexError there, utils/utils.py:141). */
int tg_sample_recent(const tg_graph* g, const int32_t* d_ids, const double* d_times64, const float* d_times32,
                     int64_t n, int k, int32_t* d_out_nbr, int32_t* d_out_eid, float* d_out_t, float* d_out_dt,
                     int32_t* d_status, void* stream);

/* Distinct (node id, float32 time) pairs of a sampled level (flid_amd/engine.py row sharing): the embedding of a node at a
 * time is a function of that pair only, so repeated pairs inside a batch are computed once.  n input slots; outputs: the
 * distinct pairs (d_out_ids / d_out_t, at most n, in the order of their FIRST OCCURRENCE among the input slots: the same input gives the
 * same numbering, run after run), d_out_row[i] = row_offset + index of slot i's pair,
 * d_count_pad (4 ints): [0] = number of distinct pairs, [1] = index of the padding pair (0, 0.0f) or -1, [2..3] scratch.
 * Workspaces: keys (capacity x 8 B), vals (capacity + 1024 int32), pos (n int32); capacity = tg_dedupe_capacity(n) (power of two >= 2n);
 * n <= 4 M slots per call. */
int64_t tg_dedupe_capacity(int64_t n);
int tg_dedupe_pairs(const int32_t* d_ids, const float* d_t, int64_t n, int64_t capacity, void* d_keys_ws, int32_t* d_vals_ws,
                    int32_t* d_pos_ws, int32_t row_offset, int32_t* d_out_ids, float* d_out_t, int32_t* d_out_row,
                    int32_t* d_count_pad, void* stream);

/* DyGFormer first-hop window: replaces utils/utils.py:254-273 get_all_first_hop_neighbors +
 * models/DyGFormer.py:196-245 pad_sequences (cut to the newest max_len-1, self in slot 0, left-aligned).
 * Outputs are (n, width) row-major with width = max_len rounded up to a patch multiple by the caller;
 * d_out_len[i] = 1 + kept neighbors.  Slots beyond d_out_len are (0, 0, 0.0f). */
/* ---- random sampling strategies on the device: a NON-bit-exact mode (opt-in) ---------------------------------------------
 * The reference draws from numpy's RandomState on the host (utils/utils.py:176-199); the bit-exact path of the sampler mirror does the
 * same.  Here a counter-based generator (seed, query index, slot) replaces that stream: k draws with replacement from the node's
 * strictly-earlier history -- uniform (weighted = 0) or with the time-interval-aware probabilities softmax(p[:cnt]),
 * p = exp(tsf (t - t_last)) / cumsum(...) (utils.py:112-128, :183-186; tg_graph_set_time_weights(tsf) prepares their running sums) --
 * re-ordered by time as :193-199.  Nodes without history give all-zero rows.  k <= 128.  Output layout as tg_sample_recent. */
int tg_graph_set_time_weights(tg_graph* g, double time_scaling_factor);
int tg_sample_random(const tg_graph* g, const int32_t* d_ids, const double* d_times64, const float* d_times32, int64_t n, int k,
                     int weighted, uint64_t seed, int32_t* d_out_nbr, int32_t* d_out_eid, float* d_out_t, float* d_out_dt,
                     int32_t* d_status, void* stream);
/* host-side history lengths over the exported CSR (tg_graph_export): out[q] = number of incidences of ids[q] strictly before
 * times[q] (the prefix length of utils/utils.py:130-147).  No device work; TG_ERANGE for an id outside [0, num_rows). */
int tg_host_count_before(const int64_t* h_row_ptr, const double* h_t, int64_t num_rows, const int64_t* ids, const double* times, int64_t n,
                         int64_t* out);
int tg_first_hop_window(const tg_graph* g, const int32_t* d_ids, const double* d_times64, int64_t n, int max_len,
                        int width, int32_t* d_out_nbr, int32_t* d_out_eid, float* d_out_t, int32_t* d_out_len,
                        void* stream);
/* GraphMixer's node encoder, models/GraphMixer.py:125-150 (get_historical_neighbors(num_neighbors=time_gap) + softmax over the
 * validity mask + torch.mean over the time_gap slots): out[q] = (1 / window) (1 / nv) sum of the table rows of the nv most recent
 * neighbors (at most `window`) of d_ids[q] strictly before d_times64[q]; with no neighbor, table row 0 / window. */
int tg_recent_window_mean(const tg_graph* g, const int32_t* d_ids, const double* d_times64, int64_t n, int window,
                          const float* d_table, int64_t table_ld, int cols, float* d_out, int64_t out_ld, void* stream);

/* ---- time encoding --------------------------------------------------------------------------------
 * replaces models/modules.py:28-40 TimeEncoder.forward: out[i, j] = cos(fma(t[i], w[j], b[j])).
 * fused_fma=0 evaluates fl(fl(t*w)+b) instead (the reference's (B,1)-shaped call). */
int tg_time_encode(const float* d_t, int64_t n, const float* d_w, const float* d_b, int dim, int fused_fma,
                   float* d_out, void* stream);

/* same, but rows whose mask id is 0 come out as zeros (models/DyGFormer.py:263-266: padded slots of a sequence) */
int tg_time_encode_masked(const float* d_t, const int32_t* d_mask_ids, int64_t n, const float* d_w, const float* d_b, int dim,
                          float* d_out, void* stream);

/* backward w.r.t. (w, b): d_part (tg_rowop_parts(n), 2*dim) per-workgroup partial sums of (dw | db); d_mask_ids may be NULL */
int tg_time_encode_bwd(const float* d_t, const int32_t* d_mask_ids, int64_t n, const float* d_w, const float* d_b, int dim,
                       const float* d_g, float* d_part, void* stream);

/* ---- single-query temporal attention, gather-fused ------------------------------------------------
 * replaces the neighbor side of models/modules.py:167-245 MultiHeadAttention.forward together with the
 * gathers of models/TGAT.py:110-129.  Exact reassociation: score = (Wk^T q) . z, ctx = Wv (sum a z), so the
 * kernel streams each neighbor row z = [feat[feat_idx] | edge[edge_idx] | cos(dt*w+b)] once from HBM.
 *   d_u    (m, heads, dk)  query projected into key space, dk = dn + de + dt_dim
 *   d_agg  (m, heads, dk)  out: attention-weighted sum of z per head (after dropout)
 *   d_prob (m, heads, k)   out: softmax probabilities BEFORE dropout (saved for backward)
 * Masking: slots with nbr id 0 score -1e10 (modules.py:211-221); an all-padded row is uniform.
 * Dropout (modules.py:224): keep = hash(seed, row, head, slot) >= p, scaled 1/(1-p); p = 0 disables. */
typedef struct tg_attn_desc {
    const float* d_feat;   int64_t feat_ld;   const int32_t* d_feat_idx; /* (m*k) rows of d_feat */
    const float* d_edge;   int64_t edge_ld;   const int32_t* d_edge_idx; /* (m*k) rows of d_edge */
    const int32_t* d_nbr;  /* (m*k) neighbor node ids, 0 = padding */
    const float* d_dt;     /* (m*k) float32 time deltas */
    const float* d_te_w;   const float* d_te_b;
    int64_t m; int k; int heads; int dn; int de; int dt_dim;
    float scale; float dropout_p; uint64_t seed;
    int64_t row0;          /* instance index of row 0 in the dropout stream (non-zero when one call is split into row chunks) */
} tg_attn_desc;

int tg_attn_fwd(const tg_attn_desc* a, const float* d_u, float* d_agg, float* d_prob, void* stream);

/* backward of tg_attn_fwd.  d_dagg (m,heads,dk) in; d_du (m,heads,dk) out.
 * d_dfeat: if non-NULL, (rows of the feat source, ld dfeat_ld) gradient w.r.t. gathered feature rows, ADDED with
 * float atomics (rows may repeat); must be zeroed by the caller.  pad_feat_row >= 0 promises that every padded slot
 * (nbr id 0) gathers that one row: their gradients are pre-summed per workgroup instead of contending on one address.
 * d_dte_part: (parts, 2*dt_dim) per-workgroup partial sums of (dw | db); *parts is returned by tg_attn_bwd_parts().
 * d_dedge: if non-NULL (then d_dfeat must be given too), the same for the gathered edge rows (stand-alone
 * MultiHeadAttention.forward on materialised inputs; the backbones' edge table carries no gradient, models/TGAT.py:26-29).
 * dt_dim may be 0 (no time segment in z; d_te_w / d_te_b unused): the stand-alone forward passes [edge | time features] as its
 * edge rows.  tg_set_attn_fast(mask): bit 0 / 1 = forward / backward on the pipelined production kernels (tg_attn_fast.hip,
 * default 3), bit 2 = also for launches the generic one-instance-per-workgroup kernel would take; A/B tests and timing only. */
int tg_attn_bwd_parts(int64_t m);
int tg_attn_bwd(const tg_attn_desc* a, const float* d_u, const float* d_agg, const float* d_prob,
                const float* d_dagg, float* d_du, float* d_dfeat, int64_t dfeat_ld, int64_t pad_feat_row,
                float* d_dedge, int64_t dedge_ld, float* d_dte_part, void* stream);
void tg_set_attn_fast(int mask);
/* d_out (m, heads, k) = d_prob * keep-scale of the dropout stream (seed, row, head, slot) the attention kernels use: the scores
 * AFTER dropout that models/modules.py:224,242 returns (callers of the backbones discard them). */
int tg_attn_dropped_scores(const float* d_prob, int64_t m, int heads, int k, float dropout_p, uint64_t seed, float* d_out, void* stream);

/* ---- one whole temporal-attention layer per call -----------------------------------------------------
 * replaces, per layer, models/modules.py:167-245 + :58-69 as called from models/TGAT.py:132-142 (and MemoryModel.py:703-713)
 * and their autograd: the same kernels as above, launched natively in sequence, with dropout + the split residual
 * [own | cos b] fused into the LayerNorm kernels and every bias / LayerNorm / time-encoder column sum taken from slabs.
 * All buffers are caller-allocated device memory, row-major; shapes with R = attn.m rows, dq = dn + dt_dim, dk = dn + de + dt_dim:
 *   own (R, dn; own_ld)  query-side features h^(l-1);  raw (R, dn; raw_ld)  second input of the merge layer;  cosb (dt_dim) = cos(b)
 *   saved by forward for backward: qbias (dq), q (R,dq), u/agg (R,heads,dk), prob (R,heads,k), ctx/res/y (R,dq), mean/rstd (R),
 *   f1 (R,dn); out (R,dn) = h^l. */
typedef struct tg_layer_params { const float *Wq, *Wk, *Wv, *ln_g, *ln_b, *Wr, *br, *W1, *b1, *W2, *b2; } tg_layer_params;
typedef struct tg_layer_grads { float *Wq, *Wk, *Wv, *ln_g, *ln_b, *Wr, *br, *W1, *b1, *W2, *b2; } tg_layer_grads;
typedef struct tg_layer_desc {
    tg_attn_desc attn;
    tg_layer_params params;
    const float* own; int64_t own_ld;
    const float* raw; int64_t raw_ld;
    const float* cosb;
    float res_dropout_p; uint64_t res_seed;      /* dropout after residual_fc (modules.py:235) */
    float *qbias, *q, *u, *agg, *prob, *ctx, *res, *y, *mean, *rstd, *f1, *out;
    float* wT;   /* tg_tgat_layer_wt_floats(dn, dq, dk) floats: transposed weights, written by fwd, read by bwd of the same step */
    int64_t y_ld; /* row stride of y (0 = dq).  With y_ld = raw_ld = dq + dn and raw = y + dq the merge layer's input [y | raw]
                   * (modules.py:66 torch.cat) is one buffer: fc1 and its weight gradient are one product each instead of two */
    /* the layer's prelude launch (weight transposes + query bias) can also do, when set: */
    int compute_cosb;             /* cosb[t] = cos(te_b[t]) (attn.d_te_b), written to `cosb` before anything reads it: the time encoding of
                                   * a zero interval, models/TGAT.py:84-85 (otherwise the caller filled `cosb`) */
    const float* gather_table; int64_t gather_ld; const int32_t* gather_idx;   /* raw[r, :dn] = gather_table[gather_idx[r], :dn] (TGAT.py:77-79) */
} tg_layer_desc;
/* backward: dout (R, dn) in; this layer's parameter gradients are ADDED into grads.* and into d_cosb / d_tew / d_teb (dt_dim)
 * with float atomics: the caller zeroes them (one fill for the whole gradient block of a step) -- as torch accumulates into
 * .grad.  vec (tg_tgat_layer_vec_floats floats: column-sum scratch + the merged projections' gradients) must be zero on entry too.
 * dfeat / pad_row as in tg_attn_bwd; d_own (R, dn; ld) optional gradient w.r.t. own (accumulated into if d_own_accumulate); d_raw
 * (R, dn) optional gradient w.r.t. raw.
 * scratch: df1 (R,dn), dy/dsum/dres/dctx/dq (R,dq), dagg/du (R,heads,dk), part (tg_tgat_layer_part_floats floats of slabs).
 * All weight / bias gradients of the layer leave in ONE grouped launch (tg_wgrad_group) after the attention backward;
 * tg_set_wgrad_grouped(0) restores one exact product + one column sum per gradient (A/B tests). */
typedef struct tg_layer_bwd_desc {
    tg_layer_grads grads;
    const float* dout;
    float *df1, *dy, *dsum, *dres, *dctx, *dagg, *du, *dq, *part, *vec;
    float *d_cosb, *d_tew, *d_teb;
    float* dfeat; int64_t dfeat_ld; int64_t pad_row;
    float* d_own; int64_t d_own_ld; int d_own_accumulate;
    float* d_raw;
    int defer_join;   /* 1: do not wait for the side streams before returning; the caller calls tg_side_join() after its last
                       * layer and must keep every buffer named here untouched (and alive) until then */
    int finish_time_bias;  /* 1 (the caller's LAST layer): d_teb -= sin(b) * d_cosb at the end of this call (= tg_time_bias_finish: the
                            * gradient that reached cos(b) of models/TGAT.py:84-85 handed on to b), after joining the side streams */
} tg_layer_bwd_desc;
int tg_tgat_layer_fwd(const tg_layer_desc* layer, void* stream);
int64_t tg_tgat_layer_wt_floats(int dn, int dq, int dk);
int64_t tg_tgat_layer_part_floats(int64_t rows, int dn, int dq, int dt_dim);
int64_t tg_tgat_layer_vec_floats(int dn, int dq, int dk, int heads);
void tg_set_wgrad_grouped(int on);
void tg_set_merged_min_rows(int64_t rows);   /* layers with at least this many rows take the merged projections (default 4096) */
int tg_tgat_layer_bwd(const tg_layer_desc* layer, const tg_layer_bwd_desc* bwd, void* stream);
/* weight-gradient products of tg_tgat_layer_bwd go to an internal side stream and are issued by an internal helper thread
 * (default, on = 1); both are joined before the call returns.  on = 3: side stream, launches issued by the calling thread;
 * on = 0: everything on the caller's stream. */
void tg_set_overlap(int on);
/* 1 (default): the layer multiplies with merged projections computed once per call from the weights --
 * u_h = own (Wk_h^T Wq_h[:, :dn])^T + ub_h and res = agg (Wr[:, h] Wv_h)^T + br -- two products fewer per direction on the main
 * chain, same function up to fp32 reassociation; 0: the reference's q / u / ctx / res products.  Set before forward, keep for its backward. */
void tg_set_layer_merged(int on);
/* 1 (default): everything behind the attention of a layer's forward runs as ONE launch (tg_chain.hip: row blocks handed from product
 * to product through LDS) when the layer's geometry allows; 0: one launch per product / LayerNorm (A/B tests, and the fall-back) */
void tg_set_layer_chain(int on);
/* `stream` waits for everything tg_tgat_layer_bwd(defer_join = 1) put on the side streams (drains the helper thread first) */
int tg_side_join(void* stream);

/* ---- the trainers' step from one native object ---------------------------------------------------------
 * replaces the HOST side of models/TGAT.py:50-144 (the recursion, its sampler calls utils/utils.py:149-214 and its index bookkeeping)
 * together with the loss.backward() / optimizer.step() sequence the trainers run around it (PTCL/EM_warmup.py:126-238,
 * PTCL/M_step.py:209-325): a batch is PREPARED (ids to the device, neighbor lookups, row sharing: graph-only work on the object's own
 * side stream, one or two batches ahead), then one call runs the forward of every layer and one call the backward of every layer and
 * the Adam update -- launches issued back to back out of a caller-allocated arena, no device allocation, nothing between them.
 * The flat parameter is [time_encoder.w.weight | .bias | per layer: Wq Wk Wv ln_g ln_b Wr br W1 b1 W2 b2], every tensor starting on a
 * 16-byte boundary (tg_stepper_param_floats gives the length); the gradient block has the same layout. */
typedef struct tg_stepper tg_stepper;
typedef struct tg_stepper_cfg {
    const tg_graph* graph;
    const float* d_node; int64_t node_ld;        /* node table (row 0 = padding node), models/TGAT.py:26 */
    const float* d_edge; int64_t edge_ld;        /* edge table, models/TGAT.py:27 */
    int dn, de, dt_dim, heads, layers, k;        /* layers 1 or 2, heads 1 or 2 */
    int64_t max_roots;                           /* roots of one prepared batch at most (2 B, or 3 B for [src | dst | negative dst]) */
    int slots;                                   /* batches in preparation / in use at a time (a two-stage prefetch needs 3) */
    float* d_param; int64_t param_floats;
    float dropout_p;                             /* train-mode dropout of the attention weights and of the residual path */
    int dedupe;                                  /* 1: repeated (node, time) rows inside a batch are computed once (flid_amd/engine.py row sharing) */
    int64_t extra_grad_floats;                   /* room behind the parameter gradients for the caller's own (zero-filled with them) */
    int tgn;                                     /* 1: the memory stage of models/MemoryModel.py under ONE layer (tg_stepper_tgn_*): the flat parameter
                                                  * continues with nn.GRUCell's weight_ih (3 dn, 2 dn + dt_dim + de), weight_hh (3 dn, dn), bias_ih,
                                                  * bias_hh; d_node = the raw node features; max_roots = 2 x the edges of a batch */
} tg_stepper_cfg;
/* TGN state (models/MemoryModel.py:334-459 MemoryBank as flid_amd keeps it): memory (N, dn), last-update times (N), the pending-message
 * table (N, 2 dn + dt_dim + de: only a node's LAST message is ever read, :312-320) with its device flags / times, the all -1 workspace of
 * tg_msg_scatter_last, and the HOST mirrors of the scalars the reference's assertion needs (:485-486).  Passed per call: the trainers
 * re-create the bank's arrays (epoch reset, backup / reload). */
typedef struct tg_tgn_bank {
    float* d_mem; int64_t mem_ld; float* d_last_update;
    float* d_msg; int64_t msg_ld; int32_t* d_has; float* d_msg_time; int32_t* d_last_idx_ws;
    uint8_t* h_has; double* h_msg_time; float* h_last; int64_t num_nodes;
    int past_violation;                          /* in: the reference's next get_updated_memories would raise; out: set by a state advance */
} tg_tgn_bank;
typedef void (*tg_grad_ready_fn)(void* user, float* d_segment, int64_t floats);
typedef struct tg_adam_args {                    /* torch.optim.Adam's update (utils/utils.py:40-60 create_optimizer), as tg_adam_f32 */
    float *d_exp_avg, *d_exp_avg_sq; int64_t n;  /* n = 0: the whole flat parameter */
    double lr, beta1, beta2, eps, weight_decay; int64_t step;
} tg_adam_args;
int64_t tg_stepper_param_floats(const tg_stepper_cfg* cfg);
int64_t tg_stepper_arena_floats(const tg_stepper_cfg* cfg);
int tg_stepper_create(const tg_stepper_cfg* cfg, float* d_arena, int64_t arena_floats, tg_stepper** out);
void tg_stepper_destroy(tg_stepper* st);
/* offsets (floats from d_arena): [0] gradient block, [1] its length up to the end of the extra floats, [2] embeddings h^L, [3] d cos(b) */
int tg_stepper_regions(const tg_stepper* st, const float* d_arena, int64_t* off4);
/* HOST ids (int64) / times (float64), n_roots of each (several root lists of one batch concatenated, their times repeated).
 * TG_ERANGE for an id outside the graph.  Nothing waits for the GPU. */
int tg_stepper_prepare_begin(tg_stepper* st, int slot, const int64_t* h_ids, const double* h_times, int64_t n_roots);
/* a step later: reads the distinct-row count (pinned word) and issues the level-1 lookups; rows2 (optional) = {roots, distinct level-1 rows} */
int tg_stepper_prepare_finish(tg_stepper* st, int slot, int64_t* rows2);
int tg_stepper_release(tg_stepper* st, int slot);
/* the model's neighbor sampler was swapped (set_neighbor_sampler: models/TGAT.py:146-155, models/MemoryModel.py:717-726, called every
 * epoch by PTCL/EM_warmup.py:118, :296 and PTCL/M_step.py:34, :200): later batches are sampled from `graph`.  Every slot must be free. */
int tg_stepper_set_graph(tg_stepper* st, const tg_graph* graph);
/* device pointers of a prepared slot: {ids_all, S_nbr, S_eid, S_t, S_dt, child}; *pad_row = frontier row of the padding pair or -1 */
int tg_stepper_slot_view(const tg_stepper* st, int slot, void** p6, int64_t* pad_row);
/* seeds: 2 per layer (attention dropout, residual dropout), layer 1 first; may be NULL in eval mode.  *d_emb: (roots, dn) */
int tg_stepper_forward(tg_stepper* st, int slot, int training, const uint64_t* seeds, void* stream, float** d_emb);
int tg_stepper_backward(tg_stepper* st, int slot, const float* d_demb, void* stream, tg_grad_ready_fn grad_ready, void* user,
                        const tg_adam_args* adam, float** d_grad);
/* TGN (cfg.tgn): HOST arrays of one batch (n edges; h_eid may be NULL for a negative batch); the embedded shard is edges [lo, hi) (both
 * roles: 2 (hi - lo) roots), the state advance always covers the whole batch.  Then tg_stepper_prepare_finish, and: */
int tg_stepper_tgn_prepare_begin(tg_stepper* st, int slot, const int64_t* h_src, const int64_t* h_dst, const double* h_t, const int64_t* h_eid,
                                 int64_t n, int64_t lo, int64_t hi);
/* keep_grad != 0: the gradient block holds an earlier backward's gradients that this batch's will be added to (flags bit 1 below) */
int tg_stepper_tgn_forward(tg_stepper* st, int slot, const tg_tgn_bank* bank, int training, const uint64_t* seeds, void* stream, float** d_emb,
                           int keep_grad);
/* flags: bit 0 = positive batch: the state advance of models/MemoryModel.py:155-180 runs behind the backward, before the update;
 * bit 1 = ADD this batch's gradients to what the block holds (the warm-up embeds the negative pairs first, then the positive ones, and
 * backpropagates one loss over both: PTCL/EM_warmup.py:159-175, :212-231); bit 2 = another backward of the same step follows (the
 * time-encoder bias gradient is finished by the last one; no update here) */
int tg_stepper_tgn_backward(tg_stepper* st, int slot, tg_tgn_bank* bank, const float* d_demb, int flags, void* stream,
                            const tg_adam_args* adam, float** d_grad, tg_grad_ready_fn grad_ready, void* user);
/* (grad_ready, optional: called with the attention + merge layer's finished gradient block as soon as the layer's backward is queued --
 * the GRU's backward and the state advance follow -- so that a data-parallel caller overlaps its reduction with them) */

/* ---- optimizer step for the flat-parameter mode (the trainers' torch.optim.Adam, utils/utils.py:40-60 create_optimizer) ----
 * one element-wise pass over a flat fp32 parameter: exp_avg / exp_avg_sq updated in place, bias-corrected step `step` (>= 1),
 * L2 weight decay folded into the gradient.  No amsgrad. */
int tg_adam_f32(float* d_param, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, int64_t n, double lr, double beta1,
                double beta2, double eps, double weight_decay, int64_t step, void* stream);

/* d_teb[j] -= sin(d_b[j]) * d_cosb[j]: hands the gradient that reached cos(b) -- the encoding of a zero interval,
 * models/TGAT.py:84-85 -- on to the time encoder's bias (last launch of a backward pass). */
int tg_time_bias_finish(float* d_teb, const float* d_b, const float* d_cosb, int dim, void* stream);

/* d_out[0] = scale * sum_i d_a[i] * d_w[i]  (the scalar of a weighted-mean loss over an embedding block; the fused trainers'
 * stand-in for the reduction of PTCL/EM_warmup.py:222 / M_step.py:297-306).  Operands 16-byte aligned. */
int tg_weighted_sum(const float* d_a, const float* d_w, int64_t n, float scale, float* d_out, void* stream);

/* d_loss[0] = mean_i BCE(sigmoid(z_i), y_i), y_i = 1 for i < n_pos else 0; d_dz[i] = (sigmoid(z_i) - y_i) / n.
 * replaces `.sigmoid()` + nn.BCELoss + its backward in the link-prediction warm-up (PTCL/EM_warmup.py:212-222). */
int tg_bce_logits(const float* d_z, int64_t n_pos, int64_t n, float* d_loss, float* d_dz, void* stream);
/* replaces: nn.CrossEntropyLoss(reduction='none') + the ground-truth / pseudo-label masks and per-sample weights of the M-step
 * (PTCL/M_step.py:296-312), as ONE weighted sum: loss = sum_i w_i CE(z_i, y_i) over rows with 0 <= y_i < classes,
 * dz_i = w_i (softmax(z_i) - onehot(y_i)), 0 for ignored rows. */
int tg_weighted_ce(const float* d_z, int64_t ldz, const int32_t* d_labels, const float* d_weights, int64_t n, int classes,
                   float* d_loss, float* d_dz, int64_t lddz, void* stream);

/* C[M,N] = (Y > 0) ? A[M,K] B[N,K]^T : 0   -- the input gradient of `relu(x W^T)` with the ReLU mask applied in the product's
 * epilogue (Y = the forward output; models/modules.py:68 `self.act(self.fc1(...))` differentiated).  Operands 16-byte aligned,
 * lda / ldb / K multiples of 4. */
int tg_gemm_f32_nt_masked(int64_t M, int64_t N, int64_t K, const float* d_A, int64_t lda, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                          const float* d_Y, int64_t ldy, void* stream);

/* ---- PACKED weights of the row-block chain kernels (split-bf16 MFMA, tg_chain.hip / tg_pack.hip) -----------------
 * The chains that replace the aten::mm / addmm calls behind the nn.Linear layers of models/modules.py:54-69,152-163,199-235 (and their
 * input gradients) multiply 64-row blocks by weights given PACKED: tg_pack_weights splits an (N x K) fp32 weight into bf16 hi / lo and
 * stores it in MFMA fragment order, tg_packed_floats(N, K) floats per weight, once per optimizer step (tg_tgat_layer_fwd does it in its
 * prelude launch; this is the stand-alone entry point).
 *   trans = 0: W[n][k] = src[n * ld + k];  trans = 1: W[n][k] = src[k * ld + n] (the transposed weight of an input gradient). */
typedef struct tg_pack_job {
    const float* src; int64_t ld; int N, K, trans; void* dst;
    /* optional index maps (0 = identity): packed row n' -> source row (n' / n_pad) * n_len + n' % n_pad, zero where n' % n_pad >= n_len or
     * the source row >= src_N (0 = N); the same for columns with k_len / k_pad / src_K.  They let a fused chain keep an intermediate
     * in a padded layout (per-head blocks rounded up to MFMA tiles, [y | raw] with each part rounded up to 32) */
    int src_N, src_K, n_len, n_pad, k_len, k_pad;
} tg_pack_job;
int64_t tg_packed_floats(int N, int K);
int tg_pack_weights(int njobs, const tg_pack_job* jobs, void* stream);
/* Row-panel product against a PRE-SPLIT weight (csrc/tg_gemm_pk.hip): tg_pack32_weights splits up to 32 weights W (N x K with K <= 208, or N <= 224 with any K;
 * trans: given as K x N) into bf16 hi / lo in the fragment order of v_mfma_f32_32x32x16_bf16, tg_packed32_floats(N, K) floats each
 * (-1 for a weight that is both deeper than 208 and wider than 224), once per optimizer step; tg_gemm_pk_nt: C[M, N] = A[M, K] * W^T (+ bias[N]), A straight from global memory into
 * MFMA fragments (split once per row panel), the weight's 32-column tiles through an LDS ring filled by LDS-DMA.  Replaces aten::addmm
 * behind the nn.Linear layers of models/DyGFormer.py:418-461 and their input gradients (38 400 rows against 200..800-wide weights: a contraction of at most 208 with any
 * width, or a deeper one with at most 224 output columns, whose accumulators stay in registers while 32-deep stages pass).  K, lda multiples of 4, 16-byte aligned operands -- TG_ESHAPE otherwise (nothing launched: tg_gemm_f32). */
typedef struct tg_pack32_job { const float* src; int64_t ld; int32_t N, K, trans; void* dst; } tg_pack32_job;
int64_t tg_packed32_floats(int N, int K);
int tg_pack32_weights(int njobs, const tg_pack32_job* jobs, void* stream);
int tg_gemm_pk_nt(int64_t M, int N, int K, const float* d_A, int64_t lda, const void* d_packed, float* d_C, int64_t ldc, const float* d_bias,
                  void* stream);
/* ---- grouped weight gradients (split-bf16 MFMA) ------------------------------------------------------
 * replaces the autograd weight / bias gradients of the nn.Linear layers in models/modules.py:54-69,152-163,235 for one layer:
 * up to 8 products C_j[M_j, N_j] += A_j^T B_j over the same `rows` (A_j: rows x M_j, B_j: rows x N_j, row-major) in ONE launch;
 * colsum_A_j[M_j] += column sums of A_j when non-NULL (the bias gradient: A_j is the gradient of the layer's output).
 * C_j and colsum_A_j are ACCUMULATED into with float atomics (zero them, or pass a running gradient).  M_j, N_j, lda, ldb
 * multiples of 4, operands 16-byte aligned; returns TG_ESHAPE for other shapes (nothing launched: use tg_gemm_f32 + tg_colsum). */
typedef struct tg_wgrad_job {
    const float* A; int64_t lda; int M;
    const float* B; int64_t ldb; int N;
    float* C; int64_t ldc;
    float* colsum_A;
} tg_wgrad_job;
int tg_wgrad_group(int njobs, const tg_wgrad_job* jobs, int64_t rows, void* stream);
/* 2 (default): 192 x 256 output tiles, transposing LDS reads, K slices folded in fixed order by a second launch (deterministic);
 * 1: the first form (64 x 64 tiles, float-atomic fold) -- A/B tests and the fall-back for shapes form 2 does not cover */
void tg_set_wgrad_form(int form);

/* ---- synthetic feature tables (measurement only) ---------------------------
 * out[r, c] = f(row0 + r, c, seed), uniform with unit variance, row 0 = 0; the stand-in for the node / edge feature blobs
 * utils/DataLoader.py:238-246 loads, at sizes (10 M x 172, 100 M x 172) that are generated straight into HBM. */
int tg_hash_features(float* d_out, int64_t ld, int64_t row0, int64_t nrows, int cols, uint64_t seed, void* stream);

/* ---- dense fp32 (MFMA 32x32x2 f32, exact fp32) -----------------------------------------------------
 * replaces the aten::mm / addmm calls behind nn.Linear in models/modules.py:54-69,152-163,235.
 * C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+ bias[N]) (+ C if accumulate), then optional ReLU.
 * op(A) = A if !ta (A is M x K, lda) else A^T (A is K x M, lda); op(B) likewise (B is K x N or N x K).
 * All row-major. */
int tg_gemm_f32(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda,
                const float* d_B, int64_t ldb, float* d_C, int64_t ldc, const float* d_bias, int relu, int accumulate,
                void* stream);

/* Precision of the products:
 * mode 1 (default) = split-bf16 (three bf16 MFMAs per product term, fp32 accumulation, relative error ~4e-6 per product;
 *   |emb - reference| 1.6e-5 on the full-dimension golden case) for products whose operands are both k-contiguous (ta = 0, tb = 1:
 *   activations times a weight given as N x K) and for weight-gradient products (ta = 1, tb = 0) with M * N >= 65536; exact
 *   f32-input MFMA for everything else (few-row products, small weight gradients);
 * mode 2 = split-bf16 for every weight-gradient product as well;
 * mode 0 = exact fp32 (f32-input MFMA) everywhere. */
void tg_set_gemm_mode(int mode);
int tg_get_gemm_mode(void);
/* override for the CALLING THREAD only (mode as tg_set_gemm_mode; -1 = follow the process-wide mode): per-call precision without
 * touching what other issuing threads see */
void tg_set_gemm_mode_thread(int mode);
int tg_get_gemm_mode_thread(void);      /* the calling thread's override as set (-1 = none): save / restore around a scoped override */

/* strided-batched form: problem b uses A + b*stride_a, B + b*stride_b, C + b*stride_c (bias + b*N).  One launch for the
 * per-head products of models/modules.py:186-197 (head h = column / row block h of the projection weights). */
int tg_gemm_f32_batched(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda,
                        int64_t stride_a, const float* d_B, int64_t ldb, int64_t stride_b, float* d_C, int64_t ldc,
                        int64_t stride_c, int batch, const float* d_bias, int relu, int accumulate, void* stream);

/* two-level batch: problem (o, i), o < outer, i < inner, uses X + o*outer_x + i*inner_x.  The (sequence, head) products of
 * nn.MultiheadAttention (models/DyGFormer.py:454): Q K^T, P V and their gradients, straight on the packed qkv layout. */
int tg_gemm_f32_batched2(int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float* d_A, int64_t lda,
                         int64_t outer_a, int64_t inner_a, const float* d_B, int64_t ldb, int64_t outer_b, int64_t inner_b,
                         float* d_C, int64_t ldc, int64_t outer_c, int64_t inner_c, int outer, int inner, int accumulate,
                         void* stream);

/* ---- row-wise helpers ------------------------------------------------------------------------------ */
/* out[i, 0:cols] = table[idx[i], 0:cols]        (models/TGAT.py:87 node_raw_features[ids]) */
int tg_gather_rows(const float* d_table, int64_t table_ld, const int32_t* d_idx, int64_t n, int cols,
                   float* d_out, int64_t out_ld, void* stream);
/* table[idx[i], :] += src[i, :] with float atomics */
int tg_scatter_add_rows(const float* d_src, int64_t src_ld, const int32_t* d_idx, int64_t n, int cols,
                        float* d_table, int64_t table_ld, void* stream);
/* y = LayerNorm(a + b) * gamma + beta, eps 1e-5 (modules.py:238); d_b may be NULL (plain LayerNorm, DyGFormer.py:452,458).
 * Saves mean/rstd (n each) for backward. */
int tg_add_layernorm_fwd(const float* d_a, const float* d_b, int64_t n, int cols, const float* d_gamma,
                         const float* d_beta, float* d_y, float* d_mean, float* d_rstd, void* stream);
/* dx = dLN(dy); d_dgb_part (parts, 2*cols) partial sums of (dgamma | dbeta); parts = tg_rowop_parts(n). */
int tg_rowop_parts(int64_t n);
int tg_add_layernorm_bwd(const float* d_a, const float* d_b, const float* d_dy, int64_t n, int cols,
                         const float* d_gamma, const float* d_mean, const float* d_rstd, float* d_dx,
                         float* d_dgb_part, void* stream);
/* out[j] (+)= sum_i x[i, j]   (bias gradients, partial-slab reduction) */
int tg_colsum(const float* d_x, int64_t ld, int64_t n, int cols, float* d_out, int accumulate, void* stream);
/* dx = dy * (y > 0) in place on dy */
int tg_relu_bwd_inplace(float* d_dy, const float* d_y, int64_t numel, void* stream);

/* ---- TGN memory ------------------------------------------------------------------------------------
 * GRU gate stage of nn.GRUCell (models/MemoryModel.py:531-543): gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh come from
 * tg_gemm_f32, both (n, 3d) in torch's r|z|n order; out = (1-z) * tanh(gi_n + r*gh_n) + z*h. */
int tg_gru_gates_fwd(const float* d_gi, const float* d_gh, const float* d_h, int64_t n, int d, float* d_out, void* stream);
/* gradients w.r.t. gi, gh (n, 3d) and, if d_dh != NULL, h (n, d) given d_dout (n, d) */
int tg_gru_gates_bwd(const float* d_gi, const float* d_gh, const float* d_h, const float* d_dout, int64_t n, int d,
                     float* d_dgi, float* d_dgh, float* d_dh, void* stream);
/* raw identity messages of models/MemoryModel.py:233-278: out[i] = [mem[a_i] | mem[b_i] | cos((t_i - last_update[a_i]) w + b) |
 * edge[e_i]], width 2d + T + de; t and last_update are float32 as in the reference (:255-257). */
int tg_build_messages(const float* d_mem, int64_t mem_ld, const float* d_last_update, const int32_t* d_a_ids,
                      const int32_t* d_b_ids, const float* d_t32, const float* d_edge, int64_t edge_ld, const int32_t* d_eids,
                      const float* d_te_w, const float* d_te_b, int64_t n, int d, int de, int T, float* d_out, void* stream);
/* persist the GRU rows of the batch nodes that have a pending message (models/MemoryModel.py:214-231, :472-499):
 * for i < count with d_has[d_nodes[i]]: d_memory[d_nodes[i]] = d_rows[d_row_of[i]], d_last_update[d_nodes[i]] = d_msg_time[d_nodes[i]]. */
int tg_tgn_persist(const float* d_rows, int64_t rows_ld, const int32_t* d_row_of, const int32_t* d_nodes, const int32_t* d_has,
                   const float* d_msg_time, float* d_memory, int64_t mem_ld, float* d_last_update, int64_t count, int d, void* stream);
/* file the new raw messages in the (num_nodes, width) pending-message table, "last message wins" per node in entry order (the
 * reference appends source-role then destination-role messages and reads only [-1]: MemoryModel.py:177-180, :312-320); sets
 * d_has[node] = 1 and d_msg_time[node] = t32 of the winning entry.  d_last_idx_ws: (num_nodes) int32, all -1 on entry and on exit. */
int tg_msg_scatter_last(const int32_t* d_nodes, const float* d_msgs, int64_t msg_ld, const float* d_t32, int64_t count, int width,
                        float* d_table, int64_t table_ld, int32_t* d_has, float* d_msg_time, int32_t* d_last_idx_ws, void* stream);

/* ---- graph-only part of one TGN batch (prefetchable; host arrays in, everything issued on `stream`) ---------------------------
 * replaces, for a prepared batch, the host-side glue of models/MemoryModel.py:96-131 (ids to the device), the neighbor lookup of
 * :632-715 for the 2 m embedded roots (edges [lo, hi) of the batch, both roles) and the search for the distinct nodes whose memory
 * the call touches.  tg_tgn_prepare_layout gives the offsets (int32 units) of the device blob's sections:
 *   off[0] root times (2m f64) | off[1] counterpart ids (2n) | off[2] edge ids twice (2n) | off[3] times twice (2n f32) |
 *   off[4] root ids (2m) | off[5] batch node ids [src | dst] (2n) | off[6] neighbor slots (2m, k) | off[7] end
 * h_stage: pinned, 4 off[6] bytes; d_blob: 4 off[7] bytes.  d_uniq / d_uniq_t / d_rowmap (off[7] - off[4] each), d_count_pad,
 * the workspaces and `capacity` as tg_dedupe_pairs (over [root ids | batch node ids | neighbor slots], times all zero: d_zero_t);
 * h_count_pad (pinned, 2 ints) receives (count, padding row) with an async copy.  h_uniq_nodes / h_last_time (2n each, optional): the
 * distinct batch nodes and the time of each one's last occurrence in [src | dst] order (host mirror of :155-180), *h_num_uniq of them.
 * TG_ERANGE for an id outside [0, num_nodes). */
/* the touched rows of a lazily updated TGN memory in one call (models/MemoryModel.py:117, :191-231, :501-543, :654-655): for the
 * `count` distinct touched nodes d_uniq[r]: h = memory[node], x = pending message[node], GRU cell (nn.GRUCell, :531-543),
 * rows = has_message[node] ? GRU : h (the updated memory, not persisted), base = rows + raw features[node].  pending = 0 skips the GRU
 * (no node of the graph holds a message).  d_h_rows (count, d), d_msg_rows (count, msg_dim), d_gi / d_gh (count, 3 d) are kept for
 * backward; tg_gru_gates_bwd_masked is tg_gru_gates_bwd with the upstream gradient zeroed for rows without a message. */
int tg_tgn_rows_fwd(const float* d_mem, int64_t mem_ld, const float* d_msg, int64_t msg_ld, const float* d_raw, int64_t raw_ld,
                    const int32_t* d_uniq, int64_t count, const int32_t* d_has, int d, int msg_dim, const float* d_w_ih,
                    const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, int pending, float* d_h_rows, float* d_msg_rows,
                    float* d_gi, float* d_gh, float* d_rows, float* d_base, void* stream);
int tg_gru_gates_bwd_masked(const float* d_gi, const float* d_gh, const float* d_h, const float* d_dout, const int32_t* d_uniq,
                            const int32_t* d_has, int64_t n, int d, float* d_dgi, float* d_dgh, void* stream);
/* host mirror of one positive batch's state advance (models/MemoryModel.py:155-180, assertion of :485-486), numpy arrays in place:
 * nodes u[i] with a pending message get it applied (last_update = its time), then each files a new message at new_t[i].  TG_EINVAL
 * ("Trying to update memory to time in the past!") leaves everything unchanged; *next_violation = 1 when a filed message is older
 * than its node's last update.  No device work. */
int tg_tgn_host_advance(const int64_t* u, const double* new_t, int64_t count, uint8_t* has, double* msg_time, float* last_update,
                        int64_t num_nodes, int* next_violation);
int tg_tgn_prepare_layout(int64_t n, int64_t m, int k, int64_t* off8);
int tg_tgn_prepare_batch(const tg_graph* g, const int64_t* h_src, const int64_t* h_dst, const double* h_t, const int64_t* h_eid,
                         int64_t n, int64_t lo, int64_t hi, int k, int64_t num_nodes, void* h_stage, int32_t* d_blob, int32_t* d_S_eid,
                         float* d_S_t, float* d_S_dt, const float* d_zero_t, int64_t capacity, void* d_keys_ws, int32_t* d_vals_ws,
                         int32_t* d_pos_ws, int32_t* d_uniq, float* d_uniq_t, int32_t* d_rowmap, int32_t* d_count_pad,
                         int32_t* h_count_pad, int64_t* h_uniq_nodes, double* h_last_time, int64_t* h_num_uniq, void* stream);

/* ---- DyGFormer sequence side --------------------------------------------------------------------------
 * neighbor co-occurrence counts (models/DyGFormer.py:337-393): for every slot of the source rows (n, wa) and destination
 * rows (n, wb): how often that node id occurs in the source row and in the destination row -> out (n, w, 2) float32 in the
 * reference's column order [count in src, count in dst]; padded id 0 -> (0, 0). */
int tg_cooccurrence(const int32_t* d_src_ids, int64_t ld_src, int w_src, const int32_t* d_dst_ids, int64_t ld_dst, int w_dst,
                    int64_t n, float* d_out_src, float* d_out_dst, void* stream);
/* erf GELU (F.gelu default) and its gradient w.r.t. the pre-activation x */
int tg_gelu_fwd(const float* d_x, int64_t n, float* d_y, void* stream);
int tg_gelu_bwd(const float* d_x, const float* d_dy, int64_t n, float* d_dx, void* stream);
/* row softmax over the last dimension (cols <= 1024) and its backward from the probabilities */
int tg_softmax_fwd(const float* d_x, int64_t n, int cols, float* d_y, void* stream);
int tg_softmax_bwd(const float* d_y, const float* d_dy, int64_t n, int cols, float* d_dx, void* stream);
/* the same with nn.MultiheadAttention's key_padding_mask as models/modules.py:297-303 builds it for the TCL backbone: column c of
 * the rows_per_batch rows of batch b counts as -inf where d_key_ids[b * cols + c] == 0 (backward: tg_softmax_bwd, masked
 * probabilities are exactly 0).  Rows with every key masked are NaN, as in the reference. */
int tg_softmax_keymask_fwd(const float* d_x, int64_t n, int cols, const int32_t* d_key_ids, int64_t rows_per_batch, float* d_y, void* stream);
/* the attention core of nn.MultiheadAttention on the packed in-projection d_qkv (B, S, 3 d) = [Q | K | V], no masks
 * (models/DyGFormer.py:442-461): per (sequence, head) P = softmax(Q K^T / sqrt(d / heads)), O = dropout(P) V, as ONE launch -- and its
 * backward (dQ | dK | dV into d_dqkv) as one launch.  d_prob (B, heads, S, S) keeps P for the backward; the dropout mask is tg_dropout's
 * on that tensor's flat index.  Exact fp32 (f32-input MFMA).  TG_ESHAPE for S > 64 or d / heads > 100 (nothing launched). */
int tg_seq_attn_fwd(const float* d_qkv, int64_t B, int S, int d, int heads, float dropout_p, uint64_t seed, float* d_out, float* d_prob,
                    void* stream);
int tg_seq_attn_bwd(const float* d_qkv, const float* d_prob, const float* d_dout, int64_t B, int S, int d, int heads, float dropout_p,
                    uint64_t seed, float* d_dqkv, void* stream);
/* y[i] = x[i] / (1-p) if hash(seed, i) >= p else 0.  Applying it to dy with the same seed is the backward. */
int tg_dropout(const float* d_x, int64_t n, float p, uint64_t seed, float* d_y, void* stream);
/* fused element-wise passes of the pre-LN transformer block (models/DyGFormer.py:448-461), masks as tg_dropout(seed, index):
 * y = dropout(gelu(x)); dx = gelu'(x) * dropout(dy); y = res + dropout(x) */
int tg_gelu_dropout_fwd(const float* d_x, int64_t n, float p, uint64_t seed, float* d_y, void* stream);
int tg_gelu_dropout_bwd(const float* d_x, const float* d_dy, int64_t n, float p, uint64_t seed, float* d_dx, void* stream);
int tg_dropout_add(const float* d_x, const float* d_res, int64_t n, float p, uint64_t seed, float* d_y, void* stream);
/* out[i, :] = mean over positions [lo, hi) of x (n, s, d); backward writes dout/(hi-lo) into those positions of dx */
int tg_segment_mean_fwd(const float* d_x, int64_t n, int s, int d, int lo, int hi, float* d_out, void* stream);
int tg_segment_mean_bwd(const float* d_dout, int64_t n, int s, int d, int lo, int hi, float* d_dx, void* stream);
/* tg_add_layernorm_fwd with the residual sum in front of it: d_sum (optional) = d_a + dropout(d_b) (mask as tg_dropout(drop_seed) on the
 * flat index), d_y = LayerNorm(that sum) -- `outputs = inputs + dropout(h); norm(outputs)` of a pre-LN block (models/DyGFormer.py:448-461)
 * in one pass */
int tg_add_layernorm_fwd_res(const float* d_a, const float* d_b, int64_t n, int cols, const float* d_gamma, const float* d_beta, float drop_p,
                             uint64_t drop_seed, float* d_sum, float* d_y, float* d_mean, float* d_rstd, void* stream);
/* tg_add_layernorm_bwd with the residual branch joined: d_dx = d_dres (optional) + dLN(d_dy); d_dx_dropped (optional) = dropout(d_dx)
 * with tg_dropout's mask of (drop_seed, flat index) -- the gradient entering the dropout in front of a pre-LN block's residual sum; part_ld
 * (0 = 2 cols): row stride of d_dgb_part, so that several LayerNorms' partial sums sit side by side for ONE column-sum launch */
int tg_add_layernorm_bwd_res(const float* d_a, const float* d_b, const float* d_dy, int64_t n, int cols, const float* d_gamma,
                             const float* d_mean, const float* d_rstd, const float* d_dres, float* d_dx, float* d_dgb_part,
                             float drop_p, uint64_t drop_seed, float* d_dx_dropped, int64_t part_ld, void* stream);

/* ---- DyGFormer's training step as one native object (csrc/tg_dyg.hip) ----------------------------------------
 * replaces the host side of models/DyGFormer.py:60-194 compute_src_dst_node_temporal_embeddings (patch size 1) and the
 * loss.backward() / optimizer.step() of the trainers around it (PTCL/M_step.py:209-325): tg_dyg_forward issues every launch from the
 * id copy to the (2 B, dn) embeddings, tg_dyg_backward the whole backward into a gradient block in the flat parameter's layout and,
 * optionally, torch.optim.Adam's update.  poff: offsets (floats, multiples of 4) inside the flat parameter of
 *   [time_encoder.w.weight, .bias, co-occurrence encoder layer 0 weight, bias, layer 2 weight, bias,
 *    projection_layer node / edge / time / neighbor_co_occurrence (weight, bias each),
 *    per transformer block: in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, norm 0 weight, bias, norm 1 weight, bias,
 *    linear 0 weight, bias, linear 1 weight, bias;  output_layer.weight, output_layer.bias].
 * TG_ESHAPE from tg_dyg_arena_floats (-1) / tg_dyg_create for shapes the step does not cover (two sides of more than 32 positions, heads
 * wider than 100 columns): the autograd path takes those. */
typedef struct tg_dyg tg_dyg;
typedef struct tg_dyg_cfg {
    const tg_graph* graph;
    const float* d_node; int64_t node_ld; const float* d_edge; int64_t edge_ld; int64_t num_edge_rows;
    float* d_param; int64_t param_floats;
    int64_t poff[64];
    int32_t dn, de, dt_dim, channel, layers, heads, max_len, max_edges;      /* max_len = max_input_sequence_length, max_edges = B at most */
} tg_dyg_cfg;
int64_t tg_dyg_arena_floats(const tg_dyg_cfg* cfg);
int tg_dyg_create(const tg_dyg_cfg* cfg, float* d_arena, int64_t arena_floats, tg_dyg** out);
void tg_dyg_destroy(tg_dyg* st);
/* offsets (floats from d_arena): [0] gradient block (flat-parameter layout), [1] embeddings (2 B, dn): source rows, then destination rows */
int tg_dyg_regions(const tg_dyg* st, const float* d_arena, int64_t* off2);
/* models/DyGFormer.py:308-317 (set_neighbor_sampler): later batches read their histories from `graph` */
int tg_dyg_set_graph(tg_dyg* st, const tg_graph* graph);
/* HOST ids (int64) / times (float64) of B edges; ws / wd: this batch's side widths (longest history of the side + 1, at most max_len --
 * tg_host_count_before gives them without device work); dropout_p = 0 in eval mode, else seeds = 4 per block.  TG_ERANGE for an id
 * outside the graph.  Nothing waits for the GPU. */
int tg_dyg_forward(tg_dyg* st, const int64_t* h_src, const int64_t* h_dst, const double* h_t, int64_t B, int ws, int wd, float dropout_p,
                   const uint64_t* seeds, void* stream, float** d_emb);
/* d_demb (2 B, dn): gradient of the loss w.r.t. the embeddings of the forward in flight; adam (optional): the update, behind the backward */
int tg_dyg_backward(tg_dyg* st, const float* d_demb, void* stream, const tg_adam_args* adam, float** d_grad);

#ifdef __cplusplus
}
#endif
#endif /* FLID_TG_H */
