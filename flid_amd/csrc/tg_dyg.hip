// DyGFormer's training step as calls into ONE native object: ids to the device, the first-hop sequences of both sides, co-occurrence
// counts, the four channels' features assembled into one operand, patch projection, the transformer blocks, the per-side means and the
// output layer -- then the whole backward and the optimizer's update -- each direction a single C call that issues its launches back to
// back out of a pre-sized arena (no device allocation, no Python and no autograd graph between launches).
//
// replaces the host side of models/DyGFormer.py:60-194 (compute_src_dst_node_temporal_embeddings: get_all_first_hop_neighbors,
// pad_sequences :196-245, get_features :247-268, get_patches :270-306 at patch size 1, the projection / transformer / output layers) and
// the loss.backward() / optimizer.step() of the trainers around it (PTCL/M_step.py:209-325), and flid_amd/models/DyGFormer.py +
// flid_amd/seqops.py's autograd form of the same, which stays as the autograd-facing path and as this one's test oracle.
//
// Layout: a position p = b S + j of the batch's (B, S = ws + wd) token grid is source slot j of edge b for j < ws, destination slot
// j - ws otherwise (the order the transformer reads them, DyGFormer.py:164-174).  X (n, Kp) = [node row | edge row | time encoding |
// co-occurrence encoding | pad] per position; one product against the block-diagonal projection weight gives the (n, 4 C) tokens.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tg_common.h"

#ifndef TG_TRY
#define TG_TRY(expr) do { int _rc = (expr); if (_rc != TG_OK) return _rc; } while (0)
#endif

namespace {

inline int64_t r4(int64_t n) { return (n + 3) / 4 * 4; }
inline int64_t r64(int64_t n) { return (n + 63) / 64 * 64; }

// parameter tensors in the order of tg_dyg_cfg.poff
enum { P_TE_W = 0, P_TE_B, P_CO_W0, P_CO_B0, P_CO_W2, P_CO_B2, P_PN_W, P_PN_B, P_PE_W, P_PE_B, P_PT_W, P_PT_B, P_PC_W, P_PC_B, P_BLOCK0 };
enum { B_IN_W = 0, B_IN_B, B_OUT_W, B_OUT_B, B_LN1_G, B_LN1_B, B_LN2_G, B_LN2_B, B_FC1_W, B_FC1_B, B_FC2_W, B_FC2_B, B_COUNT };

struct AssembleArgs {
    const int32_t *nbr, *eid; const float* tt;          // (2 B, wmax): rows [0, B) source windows, [B, 2 B) destination windows
    const float *cnt_s, *cnt_d;                         // (B, ws, 2), (B, wd, 2)
    const double* t64;                                  // (>= B) query times
    const float* node; int64_t node_ld; const float* edge; int64_t edge_ld; int64_t num_edge_rows;
    const float *te_w, *te_b, *co_w0, *co_b0;
    int64_t B; int ws, wd, wmax, dn, de, T, C, Kp, Cp;  // Cp: row stride of hs (C rounded up to 4, pad columns zero)
    float *X, *hs, *dtv, *cnt; int32_t* mask;
};

// one wave per position: the two gathered rows, the masked time encoding, the first layer of the co-occurrence encoder on both counts
// (DyGFormer.py:259-266, :409-411: relu(c w0 + b0) of either count; their SUM goes through the second layer in one product, which is
// linear), and the per-position scalars the backward needs
__global__ void __launch_bounds__(256) dyg_assemble_kernel(AssembleArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = a.ws + a.wd;
    const int64_t n = a.B * S;
    const int o_e = a.dn, o_t = a.dn + a.de, o_c = o_t + a.T, Kx = o_c + a.C;
    const bool vec = (a.dn % 4 == 0) && (a.de % 4 == 0) && (a.node_ld % 4 == 0) && (a.edge_ld % 4 == 0);
    for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < n; p += (int64_t)gridDim.x * 4) {
        const int64_t b = p / S;
        const int j = (int)(p - b * S);
        const bool dst = j >= a.ws;
        const int jj = dst ? j - a.ws : j;
        const int64_t w = (dst ? a.B + b : b) * a.wmax + jj;
        const int32_t v = a.nbr[w], e = a.eid[w];
        const float t = a.tt[w];
        const float* cp = dst ? a.cnt_d + (b * a.wd + jj) * 2 : a.cnt_s + (b * a.ws + jj) * 2;
        const float c0 = cp[0], c1 = cp[1];
        const float dt = (float)(a.t64[b] - (double)t);                     // float64 - float32 -> float32 (DyGFormer.py:263)
        int64_t er = ((int64_t)e - 1) % a.num_edge_rows;                     // FLiD's `edge_ids - 1` gather (:261): id 0 wraps to the last row
        if (er < 0) er += a.num_edge_rows;
        float* x = a.X + p * a.Kp;
        const float* nr = a.node + (int64_t)v * a.node_ld;
        const float* ed = a.edge + er * a.edge_ld;
        if (vec) {
            for (int c = lane * 4; c < a.dn; c += 256) *reinterpret_cast<float4*>(x + c) = *reinterpret_cast<const float4*>(nr + c);
            for (int c = lane * 4; c < a.de; c += 256) *reinterpret_cast<float4*>(x + o_e + c) = *reinterpret_cast<const float4*>(ed + c);
        } else {
            for (int c = lane; c < a.dn; c += 64) x[c] = nr[c];
            for (int c = lane; c < a.de; c += 64) x[o_e + c] = ed[c];
        }
        for (int c = lane; c < a.T; c += 64) x[o_t + c] = v == 0 ? 0.f : tg::cos_phase(fmaf(dt, a.te_w[c], a.te_b[c]));     // :266 zeroes padded slots
        for (int c = lane; c < a.C; c += 64) {
            const float w0 = a.co_w0[c], b0 = a.co_b0[c];
            a.hs[p * a.Cp + c] = fmaxf(fmaf(c0, w0, b0), 0.f) + fmaxf(fmaf(c1, w0, b0), 0.f);
        }
        for (int c = a.C + lane; c < a.Cp; c += 64) a.hs[p * a.Cp + c] = 0.f;
        for (int c = Kx + lane; c < a.Kp; c += 64) x[c] = 0.f;
        if (lane == 0) { a.dtv[p] = dt; a.mask[p] = v; a.cnt[2 * p] = c0; a.cnt[2 * p + 1] = c1; }
    }
}

struct WbdArgs {
    const float* W[4]; const float* b[4]; int k[4], off[4];      // channel c: W[c] (C, k[c]) lands at columns off[c].. of rows c C..
    int C, Kp; const float* co_b2;
    float *Wbd, *bbd, *b2x2;
};
// the block-diagonal projection weight (4 C, Kp), its bias, and twice the co-occurrence encoder's output bias (both counts carry it)
__global__ void __launch_bounds__(256) dyg_wbd_kernel(WbdArgs a) {
    const int64_t total = (int64_t)4 * a.C * a.Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / a.Kp), k = (int)(i - (int64_t)r * a.Kp);
        const int c = r / a.C, rr = r - c * a.C;
        const int kk = k - a.off[c];
        a.Wbd[i] = (kk >= 0 && kk < a.k[c]) ? a.W[c][(int64_t)rr * a.k[c] + kk] : 0.f;
        if (k == 0) a.bbd[r] = a.b[c][rr];
        if (i < a.C) a.b2x2[i] = 2.f * a.co_b2[i];
    }
}

struct ProjGradArgs {
    float* G[4]; float* gb[4]; int k[4], off[4];
    int C, Kp, Cp; const float *dWbd, *dbbd; float* g_co_b2;
    const float *w2s, *b2s; float* g_co_w2;      // the co-occurrence encoder's second layer: (Cp x Cp) gradient and column sums out of the grouped launch
};
// diagonal blocks of the block-diagonal weight's gradient -> the four projection layers' gradients; the co-occurrence encoder's second
// layer out of its padded scratch: d W2 = (d cf)^T hs, d b2 = 2 x (column sums of d cf)
__global__ void __launch_bounds__(256) dyg_proj_grad_kernel(ProjGradArgs a) {
    const int64_t total = (int64_t)4 * a.C * a.Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / a.Kp), k = (int)(i - (int64_t)r * a.Kp);
        const int c = r / a.C, rr = r - c * a.C;
        const int kk = k - a.off[c];
        if (kk >= 0 && kk < a.k[c]) a.G[c][(int64_t)rr * a.k[c] + kk] = a.dWbd[i];
        if (k == 0) a.gb[c][rr] = a.dbbd[r];
        if (i < a.C) a.g_co_b2[i] = 2.f * a.b2s[i];
        if (i < (int64_t)a.C * a.C) { const int r2 = (int)(i / a.C), c2 = (int)(i - (int64_t)r2 * a.C); a.g_co_w2[i] = a.w2s[r2 * a.Cp + c2]; }
    }
}

// first layer of the co-occurrence encoder, backward: d w0[c] = sum_p d hs[p, c] (c0 [pre0 > 0] + c1 [pre1 > 0]),
// d b0[c] = sum_p d hs[p, c] ([pre0 > 0] + [pre1 > 0])   (channel width <= 256)
__global__ void __launch_bounds__(256) dyg_cooc_bwd_kernel(const float* __restrict__ dhs, const float* __restrict__ cnt, int64_t n, int C,
        const float* __restrict__ w0, const float* __restrict__ b0, float* __restrict__ g_w0, float* __restrict__ g_b0) {
    __shared__ float red[4][2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f}, wv[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; wv[i] = c < C ? w0[c] : 0.f; bv[i] = c < C ? b0[c] : 0.f; }
    for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < n; p += (int64_t)gridDim.x * 4) {
        const float c0 = cnt[2 * p], c1 = cnt[2 * p + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < C) {
                const float g = dhs[p * C + c];
                const float m0 = fmaf(c0, wv[i], bv[i]) > 0.f ? 1.f : 0.f, m1 = fmaf(c1, wv[i], bv[i]) > 0.f ? 1.f : 0.f;
                gw[i] = fmaf(g, c0 * m0 + c1 * m1, gw[i]);
                gb[i] = fmaf(g, m0 + m1, gb[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[wave][0][lane + 64 * i] = gw[i]; red[wave][1][lane + 64 * i] = gb[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        atomicAdd(g_w0 + c, red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c]);
        atomicAdd(g_b0 + c, red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c]);
    }
}

// gradient of the two per-side means (DyGFormer.py:185-187) at every position, and its dropped copy under the top block's last seed
__global__ void __launch_bounds__(256) dyg_segmean_bwd_kernel(const float* __restrict__ d_means, int64_t B, int S, int d, int ws, float p, uint64_t seed,
                                                              float* __restrict__ dx, float* __restrict__ dx_dropped, int vec) {
    const float is = 1.f / (float)ws, id = 1.f / (float)(S - ws);
    if (vec) {          // d % 4 == 0, 16-byte aligned operands: four columns per lane, one hash for their four dropout decisions
        const int d4 = d >> 2;
        const int64_t total4 = B * S * d4;
        for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * blockDim.x) {
            const int c4 = (int)(i4 % d4);
            const int64_t pos = i4 / d4, b = pos / S;
            const int j = (int)(pos - b * S);
            const float sc = j < ws ? is : id;
            const float4 m = *reinterpret_cast<const float4*>(d_means + (j < ws ? b : B + b) * d + 4 * c4);
            const float4 v = make_float4(m.x * sc, m.y * sc, m.z * sc, m.w * sc);
            reinterpret_cast<float4*>(dx)[i4] = v;
            if (dx_dropped) {
                float k[4];
                tg::res_keep_scale4(seed, 4 * i4, p, k);
                reinterpret_cast<float4*>(dx_dropped)[i4] = make_float4(v.x * k[0], v.y * k[1], v.z * k[2], v.w * k[3]);
            }
        }
        return;
    }
    const int64_t total = B * S * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const int64_t pos = i / d, b = pos / S;
        const int j = (int)(pos - b * S);
        const float v = j < ws ? d_means[b * d + c] * is : d_means[(B + b) * d + c] * id;
        dx[i] = v;
        if (dx_dropped) {
            dx_dropped[i] = v * tg::res_keep_scale(seed, i, p);
        }
    }
}

struct BlockBuf {
    float *y1, *m1, *r1, *qkv, *prob, *att, *ao, *o1, *y2, *m2, *r2, *h, *hgd, *out;
    // pre-split weights of the products whose contraction is the token width d (tg_gemm_pk.hip): in_proj, out_proj, linear 0 forward;
    // linear 1 and out_proj transposed for the input gradients d hgd = d f W2, d att = d ao Wo
    float *pk_in, *pk_out, *pk_fc1, *pk_fc2t, *pk_outt;
    // ... and of the deep ones with d output columns: linear 1 forward (K = 4 d), d y2 = d h W1 (K = 4 d), d y1 = d qkv Wqkv (K = 3 d)
    float *pk_fc2, *pk_fc1t, *pk_int;
};

struct Arena {
    float* base; int64_t off = 0;
    explicit Arena(float* b) : base(b) {}
    float* take(int64_t floats) { float* p = base ? base + off : nullptr; off += r64(floats); return p; }
};

constexpr int RING = 4;

}  // namespace

struct tg_dyg {
    tg_dyg_cfg c;
    int d = 0, Kx = 0, Kp = 0, wmax = 0, nten = 0;
    int64_t nmax = 0, parts = 0;
    // device regions
    double* times = nullptr; int32_t* ids = nullptr;
    int32_t *w_nbr = nullptr, *w_eid = nullptr, *w_len = nullptr, *mask = nullptr;
    float *w_t = nullptr, *cnt_s = nullptr, *cnt_d = nullptr, *X = nullptr, *hs = nullptr, *dtv = nullptr, *cnt = nullptr;
    float *Wbd = nullptr, *bbd = nullptr, *b2x2 = nullptr, *x0 = nullptr, *means = nullptr, *emb = nullptr, *pk_wbd = nullptr;
    std::vector<BlockBuf> blk;
    float *d_means = nullptr, *dxa = nullptr, *dxb = nullptr, *d_f = nullptr, *d_hgd = nullptr, *d_y2 = nullptr, *d_o1 = nullptr, *d_ao = nullptr,
          *d_att = nullptr, *dqkv = nullptr, *d_y1 = nullptr, *part = nullptr, *d_tf = nullptr, *d_cf = nullptr, *d_hs = nullptr, *te_part = nullptr;
    float* gblock = nullptr; int64_t g_wbd = 0, g_bbd = 0, g_w2s = 0, g_b2s = 0, g_floats = 0;
    int Cp = 0;
    // pinned staging ring
    void* pinned = nullptr; int64_t stage_bytes = 0; hipEvent_t copied[RING] = {}; bool copy_pending[RING] = {}; int ring = 0;
    // the forward in flight
    bool fwd_pending = false; int64_t B = 0; int ws = 0, wd = 0; float p = 0.f; uint64_t seeds[32] = {};
    bool use_pk = false;                  // this step's d-deep products run against the pre-split weights
};

namespace {

int check_cfg(const tg_dyg_cfg* c) {
    TG_REQUIRE(c && c->graph && c->d_node && c->d_edge && c->d_param, "tg_dyg: null pointer in the configuration");
    TG_REQUIRE(c->layers >= 1 && c->layers <= 4, "tg_dyg: 1..4 transformer blocks");
    TG_REQUIRE(c->dn > 0 && c->de > 0 && c->dt_dim > 0 && c->channel > 0 && c->channel <= 256 && c->max_edges > 0 && c->num_edge_rows > 0, "tg_dyg: dimensions");
    TG_REQUIRE(c->max_len - 1 > 0, "Maximal number of neighbors for each node should be greater than 1!");
    const int d = 4 * c->channel;
    TG_REQUIRE(c->heads >= 1 && d % c->heads == 0, "tg_dyg: the token width 4 C must divide by the heads");
    if (2 * c->max_len > 64 || (d / c->heads) % 4 != 0 || d / c->heads > 100) {
        tg::set_error("tg_dyg: the native step covers two sides of at most 32 positions and heads of at most 100 columns (multiple of 4): "
                      "take the autograd path for other shapes");
        return TG_ESHAPE;
    }
    const int nten = P_BLOCK0 + B_COUNT * c->layers + 2;
    for (int i = 0; i < nten; ++i)
        TG_REQUIRE(c->poff[i] >= 0 && c->poff[i] % 4 == 0 && c->poff[i] < c->param_floats, "tg_dyg: parameter offsets must be multiples of 4 floats inside the flat parameter");
    TG_REQUIRE((reinterpret_cast<uintptr_t>(c->d_param) & 15) == 0, "tg_dyg: the flat parameter must be 16-byte aligned");
    return TG_OK;
}

void layout(tg_dyg* st, float* base, int64_t* total) {
    const tg_dyg_cfg& c = st->c;
    const int d = st->d, C = c.channel, T = c.dt_dim, H = c.heads;
    const int64_t B = c.max_edges, wmax = st->wmax, n = st->nmax, S = 2 * wmax;
    Arena A(base);
    st->times = reinterpret_cast<double*>(A.take(4 * B + 2 * B));               // [times (2 B doubles) | ids (2 B int32)]: ONE copy
    st->ids = st->times ? reinterpret_cast<int32_t*>(st->times + 2 * B) : nullptr;
    st->w_nbr = reinterpret_cast<int32_t*>(A.take(2 * B * wmax)); st->w_eid = reinterpret_cast<int32_t*>(A.take(2 * B * wmax));
    st->w_t = A.take(2 * B * wmax); st->w_len = reinterpret_cast<int32_t*>(A.take(2 * B));
    st->cnt_s = A.take(B * wmax * 2); st->cnt_d = A.take(B * wmax * 2);
    st->X = A.take(n * st->Kp); st->hs = A.take(n * st->Cp); st->dtv = A.take(n); st->cnt = A.take(2 * n); st->mask = reinterpret_cast<int32_t*>(A.take(n));
    st->Wbd = A.take((int64_t)d * st->Kp); st->bbd = A.take(d); st->b2x2 = A.take(C);
    st->x0 = A.take(n * d);
    st->blk.resize((size_t)c.layers);
    for (BlockBuf& b : st->blk) {
        b.y1 = A.take(n * d); b.m1 = A.take(n); b.r1 = A.take(n); b.qkv = A.take(n * 3 * d); b.prob = A.take(B * H * S * S); b.att = A.take(n * d);
        b.ao = A.take(n * d); b.o1 = A.take(n * d); b.y2 = A.take(n * d); b.m2 = A.take(n); b.r2 = A.take(n); b.h = A.take(n * 4 * d);
        b.hgd = A.take(n * 4 * d); b.out = A.take(n * d);
        const bool pk = tg::packed32_floats(d, d) > 0;
        b.pk_in = pk ? A.take(tg::packed32_floats(3 * d, d)) : nullptr; b.pk_out = pk ? A.take(tg::packed32_floats(d, d)) : nullptr;
        b.pk_fc1 = pk ? A.take(tg::packed32_floats(4 * d, d)) : nullptr; b.pk_fc2t = pk ? A.take(tg::packed32_floats(4 * d, d)) : nullptr;
        b.pk_outt = pk ? A.take(tg::packed32_floats(d, d)) : nullptr;
        const bool pkl = pk && tg::packed32_floats(d, 4 * d) > 0;
        b.pk_fc2 = pkl ? A.take(tg::packed32_floats(d, 4 * d)) : nullptr; b.pk_fc1t = pkl ? A.take(tg::packed32_floats(d, 4 * d)) : nullptr;
        b.pk_int = pkl ? A.take(tg::packed32_floats(d, 3 * d)) : nullptr;
    }
    st->pk_wbd = tg::packed32_floats(d, st->Kp) > 0 ? A.take(tg::packed32_floats(d, st->Kp)) : nullptr;
    st->means = A.take(2 * B * d); st->emb = A.take(2 * B * c.dn);
    st->d_means = A.take(2 * B * d); st->dxa = A.take(n * d); st->dxb = A.take(n * d); st->d_f = A.take(n * d); st->d_hgd = A.take(n * 4 * d);
    st->d_y2 = A.take(n * d); st->d_o1 = A.take(n * d); st->d_ao = A.take(n * d); st->d_att = A.take(n * d); st->dqkv = A.take(n * 3 * d);
    st->d_y1 = A.take(n * d);
    st->parts = tg_rowop_parts(n);
    st->part = A.take(st->parts * 4 * d);                       // [LN1: dgamma | dbeta][LN2: dgamma | dbeta] per row-op workgroup
    st->d_tf = A.take(n * T); st->d_cf = A.take(n * st->Cp); st->d_hs = A.take(n * C);      // (d_cf: pad columns zeroed once at creation)
    st->te_part = A.take(st->parts * 2 * T);
    // gradient block, zero-filled once per backward: [parameter gradients (flat layout) | d Wbd | d bbd | d W2 (padded) | its column sums]
    st->g_wbd = r4(c.param_floats);
    st->g_bbd = st->g_wbd + r4((int64_t)d * st->Kp);
    st->g_w2s = st->g_bbd + r4(d);
    st->g_b2s = st->g_w2s + r4((int64_t)st->Cp * st->Cp);
    st->g_floats = st->g_b2s + r4(st->Cp);
    st->gblock = A.take(st->g_floats);
    *total = A.off;
}

void derive(tg_dyg* st) {
    const tg_dyg_cfg& c = st->c;
    st->d = 4 * c.channel;
    st->Kx = c.dn + c.de + c.dt_dim + c.channel;
    st->Kp = (int)r4(st->Kx);
    st->Cp = (int)r4(c.channel);
    st->wmax = c.max_len;                                     // patch size 1: a side is as wide as its longest sequence, at most max_len
    st->nmax = (int64_t)c.max_edges * 2 * st->wmax;
    st->nten = P_BLOCK0 + B_COUNT * c.layers + 2;
}

inline float* P(const tg_dyg* st, int i) { return st->c.d_param + st->c.poff[i]; }
inline float* G(const tg_dyg* st, int i) { return st->gblock + st->c.poff[i]; }
inline int blk_i(int l, int j) { return P_BLOCK0 + B_COUNT * l + j; }

// C = A W^T (+ bias) (tb) or A W (W given K x N): against the pre-split copy of the weight where this step has one, else the general product
int prod(const tg_dyg* st, const float* packed, int tb, int64_t M, int N, int K, const float* A, int64_t lda, const float* W, int64_t ldw, float* C,
         int64_t ldc, const float* bias, void* stream) {
    if (st->use_pk && packed && tg::gemm_pk_nt(M, N, K, A, lda, packed, C, ldc, bias, (hipStream_t)stream)) return tg::launch_status("gemm_pk_s_kernel");
    return tg_gemm_f32(0, tb, M, N, K, 1.f, A, lda, W, ldw, C, ldc, bias, 0, 0, stream);
}

// weight (and bias) gradients: the grouped split-bf16 launch where it covers the shapes, else one exact product + one column sum per job
int wgrad(int n, const tg_wgrad_job* jobs, int64_t rows, void* stream) {
    if (rows >= 256) {
        const int rc = tg_wgrad_group(n, jobs, rows, stream);
        if (rc != TG_ESHAPE) return rc;
    }
    for (int i = 0; i < n; ++i) {
        const tg_wgrad_job& q = jobs[i];
        TG_TRY(tg_gemm_f32(1, 0, q.M, q.N, rows, 1.f, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, nullptr, 0, 1, stream));
        if (q.colsum_A) TG_TRY(tg_colsum(q.A, q.lda, rows, q.M, q.colsum_A, 1, stream));
    }
    return TG_OK;
}

}  // namespace

extern "C" int64_t tg_dyg_arena_floats(const tg_dyg_cfg* cfg) {
    if (check_cfg(cfg) != TG_OK) return -1;
    tg_dyg st{};
    st.c = *cfg;
    derive(&st);
    int64_t total = 0;
    layout(&st, nullptr, &total);
    return total;
}

extern "C" void tg_dyg_destroy(tg_dyg* st) {
    if (!st) return;
    for (hipEvent_t e : st->copied) if (e) (void)hipEventDestroy(e);
    if (st->pinned) (void)hipHostFree(st->pinned);
    delete st;
}

extern "C" int tg_dyg_create(const tg_dyg_cfg* cfg, float* d_arena, int64_t arena_floats, tg_dyg** out) {
    TG_TRY(check_cfg(cfg));
    TG_REQUIRE(d_arena && out && (reinterpret_cast<uintptr_t>(d_arena) & 255) == 0, "tg_dyg_create: the arena must be 256-byte aligned");
    tg_dyg* st = new tg_dyg();
    st->c = *cfg;
    derive(st);
    int64_t total = 0;
    layout(st, d_arena, &total);
    if (total > arena_floats) { delete st; TG_REQUIRE(false, "tg_dyg_create: arena smaller than tg_dyg_arena_floats()"); }
    st->stage_bytes = (2 * cfg->max_edges) * (int64_t)(sizeof(double) + sizeof(int32_t));
    if (hipHostMalloc(&st->pinned, (size_t)(RING * st->stage_bytes), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); delete st; tg::set_error("tg_dyg_create: pinned staging allocation failed"); return TG_ENOMEM;
    }
    for (hipEvent_t& e : st->copied)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); tg_dyg_destroy(st); tg::set_error("tg_dyg_create: event"); return TG_EHIP; }
    if (hipMemset(st->d_cf, 0, sizeof(float) * (size_t)(st->nmax * st->Cp)) != hipSuccess) { (void)hipGetLastError(); tg_dyg_destroy(st); tg::set_error("tg_dyg_create: memset"); return TG_EHIP; }
    *out = st;
    return TG_OK;
}

/* offsets (floats from d_arena): [0] gradient block (flat-parameter layout), [1] embeddings (2 B, dn): source rows then destination rows */
extern "C" int tg_dyg_regions(const tg_dyg* st, const float* d_arena, int64_t* off2) {
    TG_REQUIRE(st && d_arena && off2, "tg_dyg_regions: null pointer");
    off2[0] = st->gblock - d_arena;
    off2[1] = st->emb - d_arena;
    return TG_OK;
}

// as tg_stepper_set_graph: the model's sampler was swapped (models/DyGFormer.py:308-317 as called from PTCL/M_step.py:34, :200)
extern "C" int tg_dyg_set_graph(tg_dyg* st, const tg_graph* graph) {
    TG_REQUIRE(st && graph, "tg_dyg_set_graph: null pointer");
    st->c.graph = graph;
    return TG_OK;
}

extern "C" int tg_dyg_forward(tg_dyg* st, const int64_t* h_src, const int64_t* h_dst, const double* h_t, int64_t B, int ws, int wd,
                              float dropout_p, const uint64_t* seeds, void* stream, float** d_emb) {
    TG_REQUIRE(st && h_src && h_dst && h_t, "tg_dyg_forward: null pointer");
    const tg_dyg_cfg& c = st->c;
    TG_REQUIRE(B > 0 && B <= c.max_edges, "tg_dyg_forward: batch larger than the stepper was sized for");
    TG_REQUIRE(ws >= 1 && wd >= 1 && ws <= st->wmax && wd <= st->wmax, "tg_dyg_forward: side widths must be in 1..max_input_sequence_length");
    TG_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || seeds), "tg_dyg_forward: dropout");
    hipStream_t s = (hipStream_t)stream;
    const int d = st->d, C = c.channel, T = c.dt_dim, H = c.heads, S = ws + wd, Kp = st->Kp, wmax = st->wmax;
    const int64_t n = B * S, rows_g = tg_graph_num_rows(c.graph);
    st->fwd_pending = false;
    // ---- ids / times to the device: one copy out of a pinned ring slot --------------------------------------------------------
    const int r = st->ring;
    if (st->copy_pending[r]) { if (hipEventSynchronize(st->copied[r]) != hipSuccess) { (void)hipGetLastError(); return TG_EHIP; } st->copy_pending[r] = false; }
    char* hb = static_cast<char*>(st->pinned) + (int64_t)r * st->stage_bytes;
    double* ht = reinterpret_cast<double*>(hb);
    int32_t* hi = reinterpret_cast<int32_t*>(hb + 2 * c.max_edges * sizeof(double));
    for (int64_t i = 0; i < B; ++i) {
        if (h_src[i] < 0 || h_src[i] >= rows_g || h_dst[i] < 0 || h_dst[i] >= rows_g) { tg::set_error("list index out of range"); return TG_ERANGE; }
        hi[i] = (int32_t)h_src[i]; hi[B + i] = (int32_t)h_dst[i];
        ht[i] = h_t[i]; ht[B + i] = h_t[i];
    }
    // (device layout = staging layout for max_edges; a smaller batch copies the two pieces)
    if (B == c.max_edges) {
        if (hipMemcpyAsync(st->times, hb, (size_t)st->stage_bytes, hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); return TG_EHIP; }
    } else {
        if (hipMemcpyAsync(st->times, ht, (size_t)(2 * B) * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(st->ids, hi, (size_t)(2 * B) * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); return TG_EHIP; }
    }
    if (hipEventRecord(st->copied[r], s) != hipSuccess) { (void)hipGetLastError(); return TG_EHIP; }
    st->copy_pending[r] = true;
    st->ring = (r + 1) % RING;
    // ---- sequences, counts, the assembled operand -------------------------------------------------------------------------------
    TG_TRY(tg_first_hop_window(c.graph, st->ids, st->times, 2 * B, c.max_len, wmax, st->w_nbr, st->w_eid, st->w_t, st->w_len, stream));
    TG_TRY(tg_cooccurrence(st->w_nbr, wmax, ws, st->w_nbr + B * wmax, wmax, wd, B, st->cnt_s, st->cnt_d, stream));
    const int ko[4] = {c.dn, c.de, T, C}, oo[4] = {0, c.dn, c.dn + c.de, c.dn + c.de + T};
    {
        WbdArgs w{};
        for (int i = 0; i < 4; ++i) { w.W[i] = P(st, P_PN_W + 2 * i); w.b[i] = P(st, P_PN_B + 2 * i); w.k[i] = ko[i]; w.off[i] = oo[i]; }
        w.C = C; w.Kp = Kp; w.co_b2 = P(st, P_CO_B2); w.Wbd = st->Wbd; w.bbd = st->bbd; w.b2x2 = st->b2x2;
        const int64_t tot = (int64_t)d * Kp;
        dyg_wbd_kernel<<<(unsigned)std::min<int64_t>((tot + 255) / 256, tg::kMaxGridBlocks), 256, 0, s>>>(w);
        TG_TRY(tg::launch_status("dyg_wbd_kernel"));
        static const bool no_pk = getenv("FLID_GEMM_TUNE") && getenv("FLID_NO_PK") && atoi(getenv("FLID_NO_PK")) != 0;
        st->use_pk = !no_pk && tg_get_gemm_mode() != 0 && st->blk[0].pk_in != nullptr && d % 4 == 0 && c.layers <= 3;
        if (st->use_pk) {                               // the weights moved with the last update: split them again, one launch
            tg_pack32_job jobs[32];
            int nj = 0;
            if (st->pk_wbd) jobs[nj++] = tg_pack32_job{st->Wbd, Kp, d, Kp, 0, st->pk_wbd};
            for (int l = 0; l < c.layers; ++l) {
                const BlockBuf& b = st->blk[(size_t)l];
                jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_IN_W)), d, 3 * d, d, 0, b.pk_in};
                jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_OUT_W)), d, d, d, 0, b.pk_out};
                jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_FC1_W)), d, 4 * d, d, 0, b.pk_fc1};
                jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_FC2_W)), 4 * (int64_t)d, 4 * d, d, 1, b.pk_fc2t};
                jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_OUT_W)), d, d, d, 1, b.pk_outt};
                if (b.pk_fc2) {
                    jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_FC2_W)), 4 * (int64_t)d, d, 4 * d, 0, b.pk_fc2};
                    jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_FC1_W)), d, d, 4 * d, 1, b.pk_fc1t};
                    jobs[nj++] = tg_pack32_job{P(st, blk_i(l, B_IN_W)), d, d, 3 * d, 1, b.pk_int};
                }
            }
            TG_TRY(tg::pack32_weights(nj, jobs, s));
        }
        AssembleArgs a{};
        a.nbr = st->w_nbr; a.eid = st->w_eid; a.tt = st->w_t; a.cnt_s = st->cnt_s; a.cnt_d = st->cnt_d; a.t64 = st->times;
        a.node = c.d_node; a.node_ld = c.node_ld; a.edge = c.d_edge; a.edge_ld = c.edge_ld; a.num_edge_rows = c.num_edge_rows;
        a.te_w = P(st, P_TE_W); a.te_b = P(st, P_TE_B); a.co_w0 = P(st, P_CO_W0); a.co_b0 = P(st, P_CO_B0);
        a.B = B; a.ws = ws; a.wd = wd; a.wmax = wmax; a.dn = c.dn; a.de = c.de; a.T = T; a.C = C; a.Kp = Kp; a.Cp = st->Cp;
        a.X = st->X; a.hs = st->hs; a.dtv = st->dtv; a.cnt = st->cnt; a.mask = st->mask;
        dyg_assemble_kernel<<<(unsigned)std::min<int64_t>((n + 3) / 4, 4 * tg::kMaxGridBlocks), 256, 0, s>>>(a);
        TG_TRY(tg::launch_status("dyg_assemble_kernel"));
    }
    // co-occurrence encoding = (h(c0) + h(c1)) W2^T + 2 b2, straight into its columns of X; then every channel's projection in one product
    TG_TRY(tg_gemm_f32(0, 1, n, C, C, 1.f, st->hs, st->Cp, P(st, P_CO_W2), C, st->X + oo[3], Kp, st->b2x2, 0, 0, stream));
    TG_TRY(prod(st, st->pk_wbd, 1, n, d, Kp, st->X, Kp, st->Wbd, Kp, st->x0, d, st->bbd, stream));
    // ---- transformer blocks (DyGFormer.py:418-461, pre-LN) ------------------------------------------------------------------------
    const float p = dropout_p;
    const float* x = st->x0;
    for (int l = 0; l < c.layers; ++l) {
        BlockBuf& b = st->blk[(size_t)l];
        const uint64_t* sd = seeds ? seeds + 4 * l : nullptr;
        const uint64_t s0 = p > 0.f ? sd[0] : 0, s1 = p > 0.f ? sd[1] : 0, s2 = p > 0.f ? sd[2] : 0, s3 = p > 0.f ? sd[3] : 0;
        // (block l > 0: y1 came out of the pass that closed block l - 1)
        if (l == 0) TG_TRY(tg_add_layernorm_fwd(x, nullptr, n, d, P(st, blk_i(l, B_LN1_G)), P(st, blk_i(l, B_LN1_B)), b.y1, b.m1, b.r1, stream));
        TG_TRY(prod(st, b.pk_in, 1, n, 3 * d, d, b.y1, d, P(st, blk_i(l, B_IN_W)), d, b.qkv, 3 * d, P(st, blk_i(l, B_IN_B)), stream));
        TG_TRY(tg_seq_attn_fwd(b.qkv, B, S, d, H, p, s0, b.att, b.prob, stream));
        TG_TRY(prod(st, b.pk_out, 1, n, d, d, b.att, d, P(st, blk_i(l, B_OUT_W)), d, b.ao, d, P(st, blk_i(l, B_OUT_B)), stream));
        // o1 = x + dropout(ao) and y2 = LayerNorm(o1) in one pass
        TG_TRY(tg_add_layernorm_fwd_res(x, b.ao, n, d, P(st, blk_i(l, B_LN2_G)), P(st, blk_i(l, B_LN2_B)), p, s1, b.o1, b.y2, b.m2, b.r2, stream));
        // (the element-wise passes stay launches of their own: folded into the products' epilogues they cost those exactly what they
        // cost alone -- measured, 3.028 vs 3.006 ms per step -- because the epilogue of a short-K product is on its critical path)
        TG_TRY(prod(st, b.pk_fc1, 1, n, 4 * d, d, b.y2, d, P(st, blk_i(l, B_FC1_W)), d, b.h, 4 * d, P(st, blk_i(l, B_FC1_B)), stream));
        TG_TRY(tg_gelu_dropout_fwd(b.h, n * 4 * d, p, s2, b.hgd, stream));
        TG_TRY(prod(st, b.pk_fc2, 1, n, d, 4 * d, b.hgd, 4 * d, P(st, blk_i(l, B_FC2_W)), 4 * d, b.ao, d, P(st, blk_i(l, B_FC2_B)), stream));
        if (l + 1 < c.layers) {                          // out = o1 + dropout(f) and the next block's y1 = LayerNorm(out) in one pass
            BlockBuf& nb = st->blk[(size_t)l + 1];
            TG_TRY(tg_add_layernorm_fwd_res(b.o1, b.ao, n, d, P(st, blk_i(l + 1, B_LN1_G)), P(st, blk_i(l + 1, B_LN1_B)), p, s3, b.out, nb.y1, nb.m1, nb.r1, stream));
        } else {
            TG_TRY(tg_dropout_add(b.ao, b.o1, n * d, p, s3, b.out, stream));
        }
        x = b.out;
    }
    // ---- per-side means over the patches and the output layer (:185-194) ----------------------------------------------------------
    TG_TRY(tg_segment_mean_fwd(x, B, S, d, 0, ws, st->means, stream));
    TG_TRY(tg_segment_mean_fwd(x, B, S, d, ws, S, st->means + B * d, stream));
    const int io = P_BLOCK0 + B_COUNT * c.layers;
    TG_TRY(tg_gemm_f32(0, 1, 2 * B, c.dn, d, 1.f, st->means, d, P(st, io), d, st->emb, c.dn, P(st, io + 1), 0, 0, stream));
    st->B = B; st->ws = ws; st->wd = wd; st->p = p;
    if (p > 0.f) memcpy(st->seeds, seeds, sizeof(uint64_t) * 4 * (size_t)c.layers);
    st->fwd_pending = true;
    if (d_emb) *d_emb = st->emb;
    return TG_OK;
}

extern "C" int tg_dyg_backward(tg_dyg* st, const float* d_demb, void* stream, const tg_adam_args* adam, float** d_grad) {
    TG_REQUIRE(st && d_demb, "tg_dyg_backward: null pointer");
    TG_REQUIRE(st->fwd_pending, "tg_dyg_backward: no forward in flight");
    st->fwd_pending = false;
    const tg_dyg_cfg& c = st->c;
    hipStream_t s = (hipStream_t)stream;
    const int d = st->d, C = c.channel, T = c.dt_dim, H = c.heads, ws = st->ws, wd = st->wd, S = ws + wd, Kp = st->Kp;
    const int64_t B = st->B, n = B * S;
    const float p = st->p;
    const int64_t parts = tg_rowop_parts(n);           // slab rows THIS batch's row-wise launches write (the regions are sized for the largest batch)
    if (hipMemsetAsync(st->gblock, 0, sizeof(float) * (size_t)st->g_floats, s) != hipSuccess) { (void)hipGetLastError(); return TG_EHIP; }
    // ---- output layer, per-side means ------------------------------------------------------------------------------------------------
    const int io = P_BLOCK0 + B_COUNT * c.layers;
    TG_TRY(tg_gemm_f32(0, 0, 2 * B, d, c.dn, 1.f, d_demb, c.dn, P(st, io), d, st->d_means, d, nullptr, 0, 0, stream));
    {
        const tg_wgrad_job j{d_demb, c.dn, c.dn, st->means, d, d, G(st, io), d, G(st, io + 1)};
        TG_TRY(wgrad(1, &j, 2 * B, stream));
    }
    float *dcur = st->dxa, *dnext = st->dxb;
    {
        const int64_t tot = n * d;
        dyg_segmean_bwd_kernel<<<(unsigned)std::min<int64_t>((tot + 255) / 256, tg::kMaxGridBlocks), 256, 0, s>>>(
            st->d_means, B, S, d, ws, p, p > 0.f ? st->seeds[4 * (c.layers - 1) + 3] : 0, dcur, p > 0.f ? st->d_f : nullptr,
            (d % 4 == 0 && ((reinterpret_cast<uintptr_t>(st->d_means) | reinterpret_cast<uintptr_t>(dcur) | reinterpret_cast<uintptr_t>(st->d_f)) & 15) == 0) ? 1 : 0);
        TG_TRY(tg::launch_status("dyg_segmean_bwd_kernel"));
    }
    // ---- transformer blocks, last to first -------------------------------------------------------------------------------------------
    for (int l = c.layers - 1; l >= 0; --l) {
        BlockBuf& b = st->blk[(size_t)l];
        const float* xin = l == 0 ? st->x0 : st->blk[(size_t)l - 1].out;
        const uint64_t* sd = st->seeds + 4 * l;
        const float* d_f = dcur;
        if (p > 0.f) d_f = st->d_f;      // the dropped copy came with dcur: from the means' backward (top block) or the block above's LayerNorm backward
        TG_TRY(prod(st, b.pk_fc2t, 0, n, 4 * d, d, d_f, d, P(st, blk_i(l, B_FC2_W)), 4 * d, st->d_hgd, 4 * d, nullptr, stream));
        float* d_h = st->d_hgd;                                                         // element-wise, in place
        TG_TRY(tg_gelu_dropout_bwd(b.h, st->d_hgd, n * 4 * d, p, p > 0.f ? sd[2] : 0, d_h, stream));
        TG_TRY(prod(st, b.pk_fc1t, 0, n, d, 4 * d, d_h, 4 * d, P(st, blk_i(l, B_FC1_W)), d, st->d_y2, d, nullptr, stream));
        // d o1 = d out + dLN2(d y2); the gradient entering the attention branch is its dropout
        const float* d_ao = st->d_o1;
        // (both LayerNorms' partial sums of dgamma / dbeta side by side: one column-sum launch per block, below)
        TG_TRY(tg_add_layernorm_bwd_res(b.o1, nullptr, st->d_y2, n, d, P(st, blk_i(l, B_LN2_G)), b.m2, b.r2, dcur, st->d_o1, st->part + 2 * d,
                                        p, p > 0.f ? sd[1] : 0, p > 0.f ? st->d_ao : nullptr, 4 * (int64_t)d, stream));
        if (p > 0.f) d_ao = st->d_ao;
        TG_TRY(prod(st, b.pk_outt, 0, n, d, d, d_ao, d, P(st, blk_i(l, B_OUT_W)), d, st->d_att, d, nullptr, stream));
        TG_TRY(tg_seq_attn_bwd(b.qkv, b.prob, st->d_att, B, S, d, H, p, p > 0.f ? sd[0] : 0, st->dqkv, stream));
        TG_TRY(prod(st, b.pk_int, 0, n, d, 3 * d, st->dqkv, 3 * d, P(st, blk_i(l, B_IN_W)), d, st->d_y1, d, nullptr, stream));
        const tg_wgrad_job jobs[4] = {
            {d_f, d, d, b.hgd, 4 * (int64_t)d, 4 * d, G(st, blk_i(l, B_FC2_W)), 4 * (int64_t)d, G(st, blk_i(l, B_FC2_B))},
            {d_h, 4 * (int64_t)d, 4 * d, b.y2, d, d, G(st, blk_i(l, B_FC1_W)), d, G(st, blk_i(l, B_FC1_B))},
            {d_ao, d, d, b.att, d, d, G(st, blk_i(l, B_OUT_W)), d, G(st, blk_i(l, B_OUT_B))},
            {st->dqkv, 3 * (int64_t)d, 3 * d, b.y1, d, d, G(st, blk_i(l, B_IN_W)), d, G(st, blk_i(l, B_IN_B))}};
        TG_TRY(wgrad(4, jobs, n, stream));
        // d x = d o1 + dLN1(d y1); the block below wants dropout(d x) under ITS last seed (written into d_f: the weight gradients above were
        // the last readers of this block's)
        const bool below = p > 0.f && l > 0;
        TG_TRY(tg_add_layernorm_bwd_res(xin, nullptr, st->d_y1, n, d, P(st, blk_i(l, B_LN1_G)), b.m1, b.r1, st->d_o1, dnext, st->part,
                                        below ? p : 0.f, below ? st->seeds[4 * (l - 1) + 3] : 0, below ? st->d_f : nullptr, 4 * (int64_t)d, stream));
        {
            float* g4[4] = {G(st, blk_i(l, B_LN1_G)), G(st, blk_i(l, B_LN1_B)), G(st, blk_i(l, B_LN2_G)), G(st, blk_i(l, B_LN2_B))};
            if (g4[1] == g4[0] + d && g4[2] == g4[1] + d && g4[3] == g4[2] + d) {
                TG_TRY(tg_colsum(st->part, 4 * (int64_t)d, parts, 4 * d, g4[0], 1, stream));
            } else {
                for (int i = 0; i < 4; ++i) TG_TRY(tg_colsum(st->part + i * d, 4 * (int64_t)d, parts, d, g4[i], 1, stream));
            }
        }
        std::swap(dcur, dnext);
    }
    // ---- patch projection, time encoder, co-occurrence encoder ------------------------------------------------------------------------
    const float* dY = dcur;
    const int oo[4] = {0, c.dn, c.dn + c.de, c.dn + c.de + T}, ko[4] = {c.dn, c.de, T, C};
    const int Cp = st->Cp;
    TG_TRY(tg_gemm_f32(0, 0, n, T, C, 1.f, dY + 2 * C, d, P(st, P_PT_W), T, st->d_tf, T, nullptr, 0, 0, stream));
    TG_TRY(tg_gemm_f32(0, 0, n, C, C, 1.f, dY + 3 * C, d, P(st, P_PC_W), C, st->d_cf, Cp, nullptr, 0, 0, stream));
    {
        // the block-diagonal projection weight's gradient and the co-occurrence encoder's second layer (operands padded to a multiple of 4
        // columns, zero pads: 51 us as a split-K tile product + fold + column sums of its own) in one grouped launch
        const tg_wgrad_job jobs[2] = {{dY, d, d, st->X, Kp, Kp, st->gblock + st->g_wbd, Kp, st->gblock + st->g_bbd},
                                      {st->d_cf, Cp, Cp, st->hs, Cp, Cp, st->gblock + st->g_w2s, Cp, st->gblock + st->g_b2s}};
        TG_TRY(wgrad(2, jobs, n, stream));
    }
    TG_TRY(tg_time_encode_bwd(st->dtv, st->mask, n, P(st, P_TE_W), P(st, P_TE_B), T, st->d_tf, st->te_part, stream));
    if (G(st, P_TE_B) == G(st, P_TE_W) + T) {
        TG_TRY(tg_colsum(st->te_part, 2 * (int64_t)T, parts, 2 * T, G(st, P_TE_W), 1, stream));
    } else {
        TG_TRY(tg_colsum(st->te_part, 2 * (int64_t)T, parts, T, G(st, P_TE_W), 1, stream));
        TG_TRY(tg_colsum(st->te_part + T, 2 * (int64_t)T, parts, T, G(st, P_TE_B), 1, stream));
    }
    TG_TRY(tg_gemm_f32(0, 0, n, C, C, 1.f, st->d_cf, Cp, P(st, P_CO_W2), C, st->d_hs, C, nullptr, 0, 0, stream));
    dyg_cooc_bwd_kernel<<<(unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 63) / 64, 512)), 256, 0, s>>>(st->d_hs, st->cnt, n, C, P(st, P_CO_W0), P(st, P_CO_B0),
                                                                                                         G(st, P_CO_W0), G(st, P_CO_B0));
    TG_TRY(tg::launch_status("dyg_cooc_bwd_kernel"));
    {
        ProjGradArgs a{};
        for (int i = 0; i < 4; ++i) { a.G[i] = G(st, P_PN_W + 2 * i); a.gb[i] = G(st, P_PN_B + 2 * i); a.k[i] = ko[i]; a.off[i] = oo[i]; }
        a.C = C; a.Kp = Kp; a.Cp = Cp; a.dWbd = st->gblock + st->g_wbd; a.dbbd = st->gblock + st->g_bbd; a.g_co_b2 = G(st, P_CO_B2);
        a.w2s = st->gblock + st->g_w2s; a.b2s = st->gblock + st->g_b2s; a.g_co_w2 = G(st, P_CO_W2);
        const int64_t tot = (int64_t)d * Kp;
        dyg_proj_grad_kernel<<<(unsigned)std::min<int64_t>((tot + 255) / 256, tg::kMaxGridBlocks), 256, 0, s>>>(a);
        TG_TRY(tg::launch_status("dyg_proj_grad_kernel"));
    }
    if (adam) {
        const int64_t na = adam->n > 0 ? adam->n : c.param_floats;
        TG_REQUIRE(na <= c.param_floats && adam->d_exp_avg && adam->d_exp_avg_sq, "tg_dyg_backward: optimizer state");
        TG_TRY(tg_adam_f32(c.d_param, st->gblock, adam->d_exp_avg, adam->d_exp_avg_sq, na, adam->lr, adam->beta1, adam->beta2, adam->eps,
                           adam->weight_decay, adam->step, stream));
    }
    if (d_grad) *d_grad = st->gblock;
    return TG_OK;
}
