// LDS-ring form of the gather-fused single-query temporal attention (tg_attn_fwd / tg_attn_bwd): same math, same C entry points and
// the same dropout stream as tg_attn_fast.hip / tg_attn.hip, which stay as the fall-backs for the shapes this file does not cover.
//
// replaces: models/modules.py:190-228 (neighbor side of MultiHeadAttention.forward) + the gathers of models/TGAT.py:110-129 /
//           models/MemoryModel.py:679-700, and their autograd.
//
// What bounded the register-staged kernels (tg_attn_fast.hip) was bytes in flight: a wave kept 2-4 neighbor rows (2.7-5.5 KB) on their
// way in two register sets, 3-4 waves per SIMD fitted (125 / 141 VGPRs), every instance began with two dependent round trips (slot
// lists, then rows) that nothing overlapped, and 13 k instances over 256 x 12 wave slots leave a fifth round a quarter full.  Here
//   * every byte a wave reads arrives by LDS-DMA (global_load_lds_dwordx4: per-lane SOURCE addresses make one instruction a row
//     gather, no VGPR is held while the row is on its way) into a ring of NS row slots per wave: with NS = 8 a wave keeps 6-8 rows
//     (8-11 KB) in flight, 12-13 waves per CU;
//   * the ring runs ACROSS instances: a wave owns a fixed list of instances and the rows (and the u / dagg / agg rows, which pass
//     through the same ring as `header` units, and the slot lists, which have a small buffer of their own) of the next instance are
//     on their way while the current one is reduced -- the two dependent round trips are paid once per wave, not once per instance;
//   * one workgroup per CU whose wave count S is chosen per launch so that the CU's share of the instances is a whole number of
//     rounds (50.75 instances per CU: 13 waves x 4 rounds, not 12 x 5);
//   * wave-uniform arithmetic is scalar: the wave number comes from v_readfirstlane, and the dropout decisions of an instance are
//     hashed once, slot s on lane s, instead of once per slot on every lane (64-bit multiplies on the vector ALU: a quarter of the
//     old kernels' issue time).
//
// Ordering.  LDS-DMA completes in issue order among loads (s_waitcnt vmcnt counts loads, stores and atomics together; stores and
// atomics may only make a wait longer): a unit has landed once at most as many LOADS as were issued after it are outstanding, so
// wait_loads(n) with n = the loads issued after the unit is exact for loads and safe whatever the stores do.  Every unit is exactly
// ZO load instructions (a lane that would be idle in an otherwise empty instruction re-reads chunk 0 into its own, unused, place).
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "tg_common.h"

#ifndef FLID_RING_NS
#define FLID_RING_NS 8
#endif

namespace {

using tg::kWave;

constexpr int kLdsBytes = 163840;      // one workgroup may take the CU's whole LDS
constexpr int kMaxS = 16;              // waves per workgroup (1024 threads)
constexpr int kMaxNS = 24;

struct RingCfg {
    int S, NS;          // waves per workgroup, ring slots per wave
    int nch;            // 16-byte chunks of a gathered row [node | edge]
    int hu;             // ring units of one header row (heads * dk floats)
    int wsz;            // LDS bytes per wave: ring + list buffer (+ transpose scratch)
    int tr_off;         // offset of the transpose scratch inside a wave's region (backward with feature gradients)
    int64_t per_wg;     // instances per workgroup
};

// One LDS-DMA instruction: lane l copies 16 bytes from its own global address g to LDS byte (dst + 16 l), dst wave-uniform.  Inline
// assembly on purpose: hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows a __builtin_amdgcn_global_load_lds it cannot
// disambiguate -- the whole ring would drain at every step.  This way the compiler does not count these loads; wait_loads() does.
// (M0 carries the LDS address and is compiler-reserved: saved and restored inside the statement.)
__device__ __forceinline__ void glds16(const void* g, char* lds_dst) {
    const uint32_t dst = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

// at most n vector-memory operations of this wave still outstanding (n wave-uniform; the immediate tops out at 63)
__device__ __forceinline__ void wait_loads(int n) {
#define W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    switch (n) {
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23)
        W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31)
        default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    }
#undef W
}
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void ld4s(const char* p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void st4(float* __restrict__ p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void zero4(float (&v)[4]) { v[0] = v[1] = v[2] = v[3] = 0.f; }
__device__ __forceinline__ int rl(int v, int s) { return __builtin_amdgcn_readlane(v, s); }
__device__ __forceinline__ float rlf(float v, int s) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), s)); }

// A wave-uniform pointer as an opaque SSA value.  Without it hipcc folds `cond ? p : q` over pointers that were LOADED from the kernel's
// arguments into a load from a selected ADDRESS, and then keeps the whole argument block (and the Ring object) in scratch memory.
template <class T>
__device__ __forceinline__ T* uni(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

// The per-wave machine shared by both directions: cursors of the unit stream, the issue side, the waits.
//   unit stream of one instance: [HG * hu header units][k row units]; header group g, unit j carries chunks [j nch, (j + 1) nch) of
//   row `inst` of header source g (u; backward: u, dagg, agg)
template <int H, int HG, int LO>
struct Ring {
    // (everything by value: a reference to the kernel's argument structs kept in here made hipcc spill both structs to scratch)
    const float *d_feat, *d_edge;
    const char* lbase;          // this lane's source of a slot-list fetch, instance 0: chunk (lane & 15) of array (lane >> 4)
    int64_t feat_ld, edge_ld;
    const float* hsrc0;         // header sources: group 0, and the byte distances of groups 1, 2 from it (a select between three POINTER
    int64_t hd1, hd2;           // members became a dynamically indexed load from this object, which then lived in scratch)
    const float* probs;         // backward: (m, H, k) probabilities, fetched with the slot lists (LO == 2)
    char* ring;
    char* lbuf;
    int lane;
    int nch, NS, hu, SEG, k, zo, uch, hrow;      // uch: chunks of a header row; hrow: its floats
    bool has0, has1, node0, node1;
    int col0, col1;
    int64_t wg_hi;
    int S;
    // consume side
    int64_t row;                // instance being reduced (-1 before the first)
    int cpos;                   // ring slot of the next unit to consume
    int p;                      // units consumed
    // issue side
    int64_t irow;
    int iu, ipos, q;            // unit inside the instance, ring slot, units issued
    int lists_mark;             // q when the newest slot-list fetch was issued
    bool ln_ready;              // that fetch has landed in lbuf (the issue side reads the NEXT instance's row numbers from there)
    int Lc_f, Lc_e, Lc_n;       // the CURRENT instance's lists, slot s on lane s
    float Lc_dt, Lc_p[H];

    __device__ __forceinline__ Ring(const tg_attn_desc& a, const RingCfg& g, char* lds, int wave, int lane_) : lane(lane_) {
        d_feat = uni(a.d_feat); d_edge = uni(a.d_edge);
        feat_ld = a.feat_ld; edge_ld = a.edge_ld;
        {
            const int arr = lane >> 4;
            const char *l0 = reinterpret_cast<const char*>(uni(a.d_feat_idx)), *l1 = reinterpret_cast<const char*>(uni(a.d_edge_idx));
            const char *l2 = reinterpret_cast<const char*>(uni(a.d_nbr)), *l3 = reinterpret_cast<const char*>(uni(a.d_dt));
            const char* b = arr == 0 ? l0 : arr == 1 ? l1 : arr == 2 ? l2 : l3;
            lbase = b + (lane & 15) * 16;
        }
        hsrc0 = nullptr; hd1 = hd2 = 0; probs = nullptr;
        nch = g.nch; NS = g.NS; hu = g.hu; k = a.k; SEG = HG * hu + k; zo = nch > kWave ? 2 : 1;
        hrow = H * (a.dn + a.de + a.dt_dim); uch = hrow >> 2;
        ring = lds + wave * g.wsz;
        lbuf = ring + NS * nch * 16;
        const int ndn = a.dn >> 2;
        has0 = lane < nch; has1 = lane + kWave < nch;
        node0 = lane < ndn; node1 = lane + kWave < ndn;
        col0 = (node0 ? lane : lane - ndn) * 4;
        col1 = (node1 ? lane + kWave : lane + kWave - ndn) * 4;
        S = g.S;
        row = -1; cpos = 0; p = 0; iu = 0; ipos = 0; q = 0; lists_mark = 0; ln_ready = false;
        Lc_f = Lc_e = Lc_n = 0; Lc_dt = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) Lc_p[h] = 0.f;
    }
    __device__ __forceinline__ char* slot(int pos) const { return ring + pos * (nch * 16); }

    // slot lists of instance r: one instruction, lane l fetches chunk (l & 15) of array (l >> 4) -> lbuf[(l >> 4) 256 + (l & 15) 16]
    __device__ __forceinline__ void issue_lists(int64_t r) {
        const int c = lane & 15;
        const bool v = 4 * c < k;
        if (v) glds16(lbase + r * k * 4, lbuf);
        if (LO == 2) {
            const bool vp = 4 * lane < H * k;
            if (vp) glds16(reinterpret_cast<const char*>(probs) + (r * H * k + 4 * lane) * 4, lbuf + 1024);
        }
        lists_mark = q;
        ln_ready = false;
    }
    // wait for the newest list fetch
    __device__ __forceinline__ void take_lists() {
        wait_loads((q - lists_mark) * zo);
        ln_ready = true;
    }
    __device__ __forceinline__ void issue_unit() {
        if (irow >= wg_hi) return;
        char* dst = slot(ipos);
        if (iu < HG * hu) {
            const int gi = HG == 1 ? 0 : (iu >= hu ? 1 : 0) + (iu >= 2 * hu ? 1 : 0), j = iu - gi * hu;
            int64_t hoff = 0;
            if (HG > 1 && gi == 1) hoff = hd1;
            if (HG > 2 && gi == 2) hoff = hd2;
            const float* base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(hsrc0) + hoff) + irow * hrow;
            const int c0 = j * nch + lane, c1 = c0 + kWave;
            const bool v0 = has0 && c0 < uch, v1 = has1 && c1 < uch;
            if (v0) glds16(base + 4 * c0, dst);
            if (zo == 2) {
                if (v1 || lane == 0) glds16(base + (v1 ? 4 * c1 : 0), dst + 1024);     // never an empty instruction: the count of loads per unit is fixed
            }
        } else {
            if (!ln_ready && irow != row) take_lists();                                   // (uniform) first row unit of the next instance
            const int s = iu - HG * hu;
            int64_t fi, ei;
            if (irow != row) {                        // (uniform) the next instance's numbers: still in the list buffer, same word for every lane
                const int* lb = reinterpret_cast<const int*>(lbuf);
                fi = lb[s]; ei = lb[64 + s];
            } else {
                fi = rl(Lc_f, s); ei = rl(Lc_e, s);
            }
            const float* p0 = node0 ? d_feat + fi * feat_ld + col0 : d_edge + ei * edge_ld + col0;
            if (has0) glds16(p0, dst);
            if (zo == 2) {
                const float* p1 = node1 ? d_feat + fi * feat_ld + col1 : d_edge + ei * edge_ld + col1;
                if (has1) glds16(p1, dst + 1024);
            }
        }
        ++iu; ++q;
        if (++ipos == NS) ipos = 0;
        if (iu == SEG) { iu = 0; irow += S; }
    }
    // the n units starting at the consume cursor have landed
    __device__ __forceinline__ void wait_units(int n) {
        const int last = p + n - 1;
        wait_loads((q - last - 1) * zo + (last < lists_mark ? LO : 0));
    }
    // ... have been read (the caller waited for its LDS reads): free their slots, refill them
    __device__ __forceinline__ void release(int n) {
        p += n;
        cpos += n;
        if (cpos >= NS) cpos -= NS;
        for (int i = 0; i < n; ++i) issue_unit();
    }
    __device__ __forceinline__ void begin(int64_t first, int64_t hi) {
        wg_hi = hi;
        irow = first;
        if (first >= hi) return;
        issue_lists(first);
        for (int i = 0; i < NS; ++i) issue_unit();
    }
    // top of an instance: its lists go to registers (slot s on lane s), the next instance's are fetched into the buffer
    __device__ __forceinline__ void begin_instance(int64_t r) {
        if (!ln_ready) take_lists();
        row = r;
        const bool sl = lane < k;
        const int* lb = reinterpret_cast<const int*>(lbuf);
        Lc_f = sl ? lb[lane] : 0;
        Lc_e = sl ? lb[64 + lane] : 0;
        Lc_n = sl ? lb[128 + lane] : 0;
        Lc_dt = sl ? __builtin_bit_cast(float, lb[192 + lane]) : 0.f;
        if (LO == 2) {
#pragma unroll
            for (int h = 0; h < H; ++h) Lc_p[h] = sl ? __builtin_bit_cast(float, lb[256 + h * k + lane]) : 0.f;
        }
        lds_reads_done();
        ln_ready = false;
        if (r + S < wg_hi) issue_lists(r + S);
        else ln_ready = true;                     // nothing more to fetch (the issue cursor never asks again)
    }
    // address of float f of a header row whose first unit sits at ring slot cpos (units may wrap around the ring)
    __device__ __forceinline__ const char* hdr(int f) const {
        const int c = f >> 2;
        int j = 0;
        for (int t = 1; t < hu; ++t) j += c >= t * nch ? 1 : 0;
        int pos = cpos + j;
        if (pos >= NS) pos -= NS;
        return slot(pos) + (c - j * nch) * 16 + (f & 3) * 4;
    }
    // a row unit at ring slot pos into registers
    __device__ __forceinline__ void read_row(int pos, float (&z)[2][4]) const {
        const char* s = slot(pos);
        if (has0) ld4s(s + lane * 16, z[0]); else zero4(z[0]);
        if (has1) ld4s(s + (lane + kWave) * 16, z[1]); else zero4(z[1]);
    }
};

// ------------------------------------------------------------------------------------------------ forward
template <int H, int MAXW>
__global__ void __launch_bounds__(64 * MAXW) attn_fwd_ring_kernel(tg_attn_desc a, const float* __restrict__ u, float* __restrict__ agg,
                                                                  float* __restrict__ prob, RingCfg g) {
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int nfe = a.dn + a.de, dk = nfe + a.dt_dim, k = a.k, T = a.dt_dim;
    const bool ht0 = lane < T, ht1 = lane + kWave < T;
    const float w0 = ht0 ? a.d_te_w[lane] : 0.f, b0 = ht0 ? a.d_te_b[lane] : 0.f;
    const float w1 = ht1 ? a.d_te_w[lane + kWave] : 0.f, b1 = ht1 ? a.d_te_b[lane + kWave] : 0.f;
    const bool two_t = T > kWave;
    const bool sl = lane < k;
    // the four loads above must have landed before the first LDS-DMA is issued: hipcc waits for them with vmcnt(0) at their first use,
    // which would otherwise sit inside the row loop and drain the ring in every trip
    asm volatile("" :: "v"(w0), "v"(b0), "v"(w1), "v"(b1));

    Ring<H, 1, 1> R(a, g, lds, wave, lane);
    R.hsrc0 = uni(u);
    const int64_t wg_lo = (int64_t)blockIdx.x * g.per_wg;
    const int64_t wg_hi = wg_lo + g.per_wg < a.m ? wg_lo + g.per_wg : a.m;
    R.begin(wg_lo + wave, wg_hi);

    for (int64_t row = wg_lo + wave; row < wg_hi; row += g.S) {
        R.begin_instance(row);
        // ONE loop over the instance's steps (header group, then pairs of row units), so that the wait / release machinery -- a switch
        // over immediates and the issue side -- exists once in the code
        float uh[H][2][4], ut[H][2], acc[H][2][4], at[H][2], mx[H], den[H], keep_score[H], keep[H];
        const int nsteps = 1 + ((k + 1) >> 1);
        for (int step = 0; step < nsteps; ++step) {
            const bool hdr = step == 0;
            const int sb = 2 * (step - 1);
            const int nv = hdr ? R.hu : (k - sb < 2 ? k - sb : 2);
            R.wait_units(nv);
            float zb[2][2][4];
            if (hdr) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    if (R.has0) ld4s(R.hdr(h * dk + lane * 4), uh[h][0]); else zero4(uh[h][0]);
                    if (R.has1) ld4s(R.hdr(h * dk + (lane + kWave) * 4), uh[h][1]); else zero4(uh[h][1]);
                    ut[h][0] = ht0 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane)) : 0.f;
                    ut[h][1] = ht1 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane + kWave)) : 0.f;
                    zero4(acc[h][0]); zero4(acc[h][1]);
                    at[h][0] = at[h][1] = 0.f;
                    mx[h] = -INFINITY; den[h] = 0.f; keep_score[h] = 0.f;
                    keep[h] = sl ? tg::dropout_keep_scale(a.seed, a.row0 + row, h, lane, a.dropout_p) : 0.f;     // slot s on lane s
                }
            } else {
                R.read_row(R.cpos, zb[0]);
                if (nv == 2) R.read_row(R.cpos + 1 >= R.NS ? 0 : R.cpos + 1, zb[1]);
                else { zero4(zb[1][0]); zero4(zb[1][1]); }
            }
            lds_reads_done();
            R.release(nv);
            if (hdr) continue;

            float zt[2][2], part[2 * H];
            int nb[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int s = sb + r;
                const bool live = s < k;
                const int ss = live ? s : 0;
                nb[r] = rl(R.Lc_n, ss);
                const float dt = rlf(R.Lc_dt, ss);
                zt[r][0] = (live && ht0) ? tg::cos_phase(fmaf(dt, w0, b0)) : 0.f;
                zt[r][1] = 0.f;
                if (two_t) zt[r][1] = (live && ht1) ? tg::cos_phase(fmaf(dt, w1, b1)) : 0.f;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float pp = ut[h][0] * zt[r][0];
                    pp = fmaf(ut[h][1], zt[r][1], pp);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) pp = fmaf(uh[h][i][e], zb[r][i][e], pp);
                    part[r * H + h] = pp;
                }
            }
            tg::wave_sum_n<2 * H>(part);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int s = sb + r;
                if (s >= k) break;                     // wave-uniform
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float sc = part[r * H + h] * a.scale;
                    if (nb[r] == 0) sc = -1e10f;                                        // modules.py:221
                    if (lane == s) keep_score[h] = sc;
                    if (sc > mx[h]) {                  // wave-uniform: the running maximum moves, rescale what was gathered so far
                        const float corr = __expf(mx[h] - sc);
                        den[h] *= corr;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[h][i][e] *= corr;
                        at[h][0] *= corr; at[h][1] *= corr;
                        mx[h] = sc;
                    }
                    const float pe = __expf(sc - mx[h]);
                    den[h] += pe;
                    const float wgt = pe * rlf(keep[h], s);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[h][i][e] = fmaf(wgt, zb[r][i][e], acc[h][i][e]);
                    at[h][0] = fmaf(wgt, zt[r][0], at[h][0]);
                    at[h][1] = fmaf(wgt, zt[r][1], at[h][1]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float inv = 1.f / den[h];
            if (sl) prob[(row * H + h) * k + lane] = __expf(keep_score[h] - mx[h]) * inv;
            float* ar = agg + (row * H + h) * dk;
            if (R.has0) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[h][0][e] * inv;
                st4(ar + lane * 4, o);
            }
            if (R.has1) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[h][1][e] * inv;
                st4(ar + (lane + kWave) * 4, o);
            }
            if (ht0) ar[nfe + lane] = at[h][0] * inv;
            if (ht1) ar[nfe + lane + kWave] = at[h][1] * inv;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the wave's LDS
}

// ------------------------------------------------------------------------------------------------ backward
// d score_{h,n} = a'_{h,n} (dagg_h . z_n) - a_{h,n} (dagg_h . agg_h)      a' = dropped/scaled prob, a = softmax prob
// masked slots get no score gradient (masked_fill), but still pass d z through a'.  DF: gradient w.r.t. the gathered node rows.
template <int H, bool DF, int MAXW>
__global__ void __launch_bounds__(64 * MAXW) attn_bwd_ring_kernel(tg_attn_desc a, const float* __restrict__ u, const float* __restrict__ agg,
        const float* __restrict__ prob, const float* __restrict__ dagg, float* __restrict__ du, float* __restrict__ dfeat, int64_t dfeat_ld,
        int64_t pad_row, float* __restrict__ dte_part, int nparts, RingCfg g) {
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int nfe = a.dn + a.de, dk = nfe + a.dt_dim, k = a.k, T = a.dt_dim;
    const bool ht0 = lane < T, ht1 = lane + kWave < T;
    const float w0 = ht0 ? a.d_te_w[lane] : 0.f, b0 = ht0 ? a.d_te_b[lane] : 0.f;
    const float w1 = ht1 ? a.d_te_w[lane + kWave] : 0.f, b1 = ht1 ? a.d_te_b[lane + kWave] : 0.f;
    const bool two_t = T > kWave;
    const bool sl = lane < k;
    // the four loads above must have landed before the first LDS-DMA is issued: hipcc waits for them with vmcnt(0) at their first use,
    // which would otherwise sit inside the row loop and drain the ring in every trip
    asm volatile("" :: "v"(w0), "v"(b0), "v"(w1), "v"(b1));

    Ring<H, 3, 2> R(a, g, lds, wave, lane);
    R.hsrc0 = uni(u);
    R.hd1 = reinterpret_cast<const char*>(uni(dagg)) - reinterpret_cast<const char*>(R.hsrc0);
    R.hd2 = reinterpret_cast<const char*>(uni(agg)) - reinterpret_cast<const char*>(R.hsrc0);
    R.probs = uni(prob);
    float* tr = reinterpret_cast<float*>(R.ring + g.tr_off);
    const int64_t wg_lo = (int64_t)blockIdx.x * g.per_wg;
    const int64_t wg_hi = wg_lo + g.per_wg < a.m ? wg_lo + g.per_wg : a.m;
    R.begin(wg_lo + wave, wg_hi);

    float gw[2] = {0.f, 0.f}, gb[2] = {0.f, 0.f};
    float dpad[2][4];
    zero4(dpad[0]); zero4(dpad[1]);

    for (int64_t row = wg_lo + wave; row < wg_hi; row += g.S) {
        R.begin_instance(row);
        float uh[H][2][4], ut[H][2], dg[H][2][4], dgt[H][2], dacc[H][2][4], dat[H][2], cterm[H], keep[H];
        // ---- header: u, dagg, agg of this instance, one group of hu units each.  (Header groups and row pairs as ONE loop over steps, as
        // the forward has it, halves the code but costs ~30 registers here: every array is then live around the whole loop.)
        R.wait_units(R.hu);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            if (DF) {
                if (R.has0) ld4s(R.hdr(h * dk + lane * 4), uh[h][0]); else zero4(uh[h][0]);
                if (R.has1) ld4s(R.hdr(h * dk + (lane + kWave) * 4), uh[h][1]); else zero4(uh[h][1]);
            }
            ut[h][0] = ht0 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane)) : 0.f;
            ut[h][1] = ht1 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane + kWave)) : 0.f;
        }
        lds_reads_done();
        R.release(R.hu);
        R.wait_units(R.hu);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            if (R.has0) ld4s(R.hdr(h * dk + lane * 4), dg[h][0]); else zero4(dg[h][0]);
            if (R.has1) ld4s(R.hdr(h * dk + (lane + kWave) * 4), dg[h][1]); else zero4(dg[h][1]);
            dgt[h][0] = ht0 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane)) : 0.f;
            dgt[h][1] = ht1 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane + kWave)) : 0.f;
        }
        lds_reads_done();
        R.release(R.hu);
        R.wait_units(R.hu);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float ag[2][4];
            if (R.has0) ld4s(R.hdr(h * dk + lane * 4), ag[0]); else zero4(ag[0]);
            if (R.has1) ld4s(R.hdr(h * dk + (lane + kWave) * 4), ag[1]); else zero4(ag[1]);
            float pp = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) pp = fmaf(dg[h][i][e], ag[i][e], pp);
            pp = fmaf(dgt[h][0], ht0 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane)) : 0.f, pp);
            pp = fmaf(dgt[h][1], ht1 ? *reinterpret_cast<const float*>(R.hdr(h * dk + nfe + lane + kWave)) : 0.f, pp);
            cterm[h] = pp;
            zero4(dacc[h][0]); zero4(dacc[h][1]);
            dat[h][0] = dat[h][1] = 0.f;
            keep[h] = sl ? tg::dropout_keep_scale(a.seed, a.row0 + row, h, lane, a.dropout_p) : 0.f;
        }
        lds_reads_done();
        R.release(R.hu);
        tg::wave_sum_n<H>(cterm);

        for (int sb = 0; sb < k; sb += 2) {
            const int nv = k - sb < 2 ? k - sb : 2;
            R.wait_units(nv);
            float zb[2][2][4];
            R.read_row(R.cpos, zb[0]);
            if (nv == 2) R.read_row(R.cpos + 1 >= R.NS ? 0 : R.cpos + 1, zb[1]);
            else { zero4(zb[1][0]); zero4(zb[1][1]); }
            lds_reads_done();
            R.release(nv);

            float zt[2][2], sn[2][2], dts[2], part[2 * H];
            int nb[2];
            int64_t fis[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int s = sb + r;
                const bool live = s < k;
                const int ss = live ? s : 0;
                nb[r] = rl(R.Lc_n, ss);
                fis[r] = rl(R.Lc_f, ss);
                dts[r] = rlf(R.Lc_dt, ss);
                float s0, c0, s1 = 0.f, c1 = 0.f;
                tg::sincos_phase(fmaf(dts[r], w0, b0), &s0, &c0);
                if (two_t) tg::sincos_phase(fmaf(dts[r], w1, b1), &s1, &c1);
                zt[r][0] = (live && ht0) ? c0 : 0.f;  sn[r][0] = (live && ht0) ? s0 : 0.f;
                zt[r][1] = (live && ht1) ? c1 : 0.f;  sn[r][1] = (live && ht1) ? s1 : 0.f;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float pp = dgt[h][0] * zt[r][0];
                    pp = fmaf(dgt[h][1], zt[r][1], pp);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) pp = fmaf(dg[h][i][e], zb[r][i][e], pp);
                    part[r * H + h] = pp;
                }
            }
            tg::wave_sum_n<2 * H>(part);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int s = sb + r;
                if (s >= k) break;                     // wave-uniform
                float pd[H], dsc[H];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float pr = rlf(R.Lc_p[h], s);
                    pd[h] = pr * rlf(keep[h], s);
                    dsc[h] = nb[r] == 0 ? 0.f : (pd[h] * part[r * H + h] - pr * cterm[h]) * a.scale;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) dacc[h][i][e] = fmaf(dsc[h], zb[r][i][e], dacc[h][i][e]);
                    dat[h][0] = fmaf(dsc[h], zt[r][0], dat[h][0]);
                    dat[h][1] = fmaf(dsc[h], zt[r][1], dat[h][1]);
                }
                // time encoder: d phase = -sin(phase) dz_time
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float dz = 0.f;
#pragma unroll
                    for (int h = 0; h < H; ++h) dz = fmaf(pd[h], dgt[h][j], fmaf(dsc[h], ut[h][j], dz));
                    const float dph = -sn[r][j] * dz;
                    gw[j] = fmaf(dts[r], dph, gw[j]);
                    gb[j] += dph;
                }
                if constexpr (DF) {
                    float dz[2][4];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = 0.f;
#pragma unroll
                            for (int h = 0; h < H; ++h) v = fmaf(pd[h], dg[h][i][e], fmaf(dsc[h], uh[h][i][e], v));
                            dz[i][e] = v;
                        }
                    if (nb[r] == 0 && pad_row >= 0) {               // wave-uniform: every padded slot gathers the same row
                        if (R.has0 && R.node0)
#pragma unroll
                            for (int e = 0; e < 4; ++e) dpad[0][e] += dz[0][e];
                        if (R.has1 && R.node1)
#pragma unroll
                            for (int e = 0; e < 4; ++e) dpad[1][e] += dz[1][e];
                    } else {
                        // transpose through LDS: lane l then owns columns l, l+64, ... and one atomic instruction covers 256
                        // contiguous bytes of the destination row
                        if (R.has0 && R.node0) st4(tr + R.col0, dz[0]);
                        if (R.has1 && R.node1) st4(tr + R.col1, dz[1]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        float* dst = dfeat + fis[r] * dfeat_ld;
                        for (int c = lane; c < a.dn; c += kWave) atomicAdd(dst + c, tr[c]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float* dr = du + (row * H + h) * dk;
            if (R.has0) st4(dr + lane * 4, dacc[h][0]);
            if (R.has1) st4(dr + (lane + kWave) * 4, dacc[h][1]);
            if (ht0) dr[nfe + lane] = dat[h][0];
            if (ht1) dr[nfe + lane + kWave] = dat[h][1];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every LDS-DMA of this wave has landed: the ring becomes reduction scratch
    __syncthreads();

    // block partial of (dw | db): waves -> LDS -> one slab row per workgroup (no atomics, deterministic); slab rows nobody owns are zeroed
    float* red = reinterpret_cast<float*>(lds);
    const int S = g.S;
    if (ht0) { red[wave * 2 * T + lane] = gw[0]; red[wave * 2 * T + T + lane] = gb[0]; }
    if (ht1) { red[wave * 2 * T + lane + kWave] = gw[1]; red[wave * 2 * T + T + lane + kWave] = gb[1]; }
    float* redp = red + S * 2 * T;
    if (DF && pad_row >= 0) {
        if (R.has0 && R.node0) st4(redp + wave * a.dn + R.col0, dpad[0]);
        if (R.has1 && R.node1) st4(redp + wave * a.dn + R.col1, dpad[1]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * T; j += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < S; ++w) s += red[w * 2 * T + j];
        dte_part[(int64_t)blockIdx.x * 2 * T + j] = s;
        for (int64_t pr = gridDim.x + blockIdx.x; pr < nparts; pr += gridDim.x) dte_part[pr * 2 * T + j] = 0.f;
    }
    if (DF && pad_row >= 0) {
        for (int j = threadIdx.x; j < a.dn; j += blockDim.x) {
            float s = 0.f;
            for (int w = 0; w < S; ++w) s += redp[w * a.dn + j];
            if (s != 0.f) atomicAdd(dfeat + pad_row * dfeat_ld + j, s);
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int g_ncu = 0;
int ncu() {
    if (!g_ncu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        g_ncu = n;
    }
    return g_ncu;
}

// HG header groups per instance, extra LDS bytes per wave besides ring + list buffer; false = shape not covered
bool make_cfg(const tg_attn_desc& a, int HG, int lbuf_bytes, int tr_bytes, int red_bytes_per_wave, int smax_kernel, RingCfg* c) {
    const int dk = a.dn + a.de + a.dt_dim;
    if (a.dn % 4 || a.de % 4 || a.dt_dim % 4 || a.k % 4 || a.k > kWave || a.k < 1 || (a.heads != 1 && a.heads != 2)) return false;
    const int nch = (a.dn + a.de) / 4;
    if (nch < 1 || nch > 2 * kWave || a.dt_dim > 2 * kWave || a.feat_ld % 4 || a.edge_ld % 4) return false;
    if (!aligned16(a.d_feat) || !aligned16(a.d_edge) || !aligned16(a.d_feat_idx) || !aligned16(a.d_edge_idx) || !aligned16(a.d_nbr) || !aligned16(a.d_dt)) return false;
    const int usz = nch * 16, uch = a.heads * dk / 4, hu = (uch + nch - 1) / nch;
    const int seg = HG * hu + a.k;
    const int fixed = lbuf_bytes + tr_bytes;
    const int64_t per_cu = (a.m + ncu() - 1) / ncu();
    int S, NS;
    auto ns_for = [&](int s) { return std::min({seg, kMaxNS, (kLdsBytes / s - fixed) / usz}); };
    static const int env_ns = getenv("FLID_RING_NS") ? atoi(getenv("FLID_RING_NS")) : FLID_RING_NS;     // least ring depth of a many-rounds launch
    static const int env_s = getenv("FLID_RING_S") ? atoi(getenv("FLID_RING_S")) : 0;                   // A/B: fixed wave count
    static const int env_smax = getenv("FLID_RING_SMAX") ? atoi(getenv("FLID_RING_SMAX")) : kMaxS;     // A/B: most waves per workgroup
    const int smax = std::max(1, std::min({env_smax, kMaxS, smax_kernel}));
    if (per_cu <= smax) {
        S = (int)per_cu;                                   // one instance per wave, the deepest ring that fits
    } else if (env_s > 0) {
        S = std::min(env_s, kMaxS);
    } else {
        S = 0;
        double best = 0.;
        for (int s = std::min(8, smax); s <= smax; ++s) {
            if (ns_for(s) < std::min(env_ns, seg)) continue;
            const int64_t rounds = (per_cu + s - 1) / s;
            const double eff = (double)per_cu / (double)(rounds * s);
            if (eff > best + 1e-9 || (eff > best - 1e-9 && s > S)) { best = eff; S = s; }
        }
        if (!S) return false;
    }
    NS = ns_for(S);
    if (NS < hu || NS < 2) return false;
    c->S = S; c->NS = NS; c->nch = nch; c->hu = hu;
    c->tr_off = NS * usz + lbuf_bytes;
    c->wsz = NS * usz + fixed;
    c->per_wg = per_cu;
    if ((int64_t)S * red_bytes_per_wave > (int64_t)S * c->wsz) return false;
    return true;
}

}  // namespace

namespace tg {

// returns TG_OK when launched, 1 when the shape is not covered (the caller falls back)
// A scratch reload is a vector-memory load: hipcc waits for it with vmcnt(0), which drains the ring.  So every variant that runs must
// be spill-free, and the wave count is capped by what its register budget allows: two heads need ~137 (forward) / ~135 (backward)
// registers = at most 12 waves (168 registers each); with feature gradients ~200 = at most 8 waves.  FLID_RING_WIDE=1 (A/B only) lets
// two-head launches take 13-16 waves on the 128-register builds, which spill.
static const bool g_wide = getenv("FLID_RING_WIDE") && atoi(getenv("FLID_RING_WIDE")) != 0;

int attn_fwd_ring(const tg_attn_desc& a, const float* u, float* agg, float* prob, hipStream_t s) {
    RingCfg c;
    const int smax = a.heads == 1 || g_wide ? 16 : 12;
    if (!aligned16(u) || !aligned16(agg) || !make_cfg(a, 1, 1024, 0, 0, smax, &c)) return 1;
    const unsigned grid = (unsigned)((a.m + c.per_wg - 1) / c.per_wg);
    const size_t lds = (size_t)c.S * c.wsz;
    static bool attr[4] = {false, false, false, false};
#define FLID_LAUNCH(i, HH, MW)                                                                                                                 \
    do {                                                                                                                                       \
        if (!attr[i]) { TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_ring_kernel<HH, MW>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes)); attr[i] = true; } \
        attn_fwd_ring_kernel<HH, MW><<<dim3(grid), dim3(c.S * kWave), lds, s>>>(a, u, agg, prob, c);                                          \
    } while (0)
    if (a.heads == 1) FLID_LAUNCH(0, 1, 16);
    else if (c.S <= 12) FLID_LAUNCH(1, 2, 12);
    else FLID_LAUNCH(2, 2, 16);
#undef FLID_LAUNCH
    return launch_status("attn_fwd_ring_kernel");
}

int attn_bwd_ring(const tg_attn_desc& a, const float* u, const float* agg, const float* prob, const float* dagg, float* du, float* dfeat,
                  int64_t dfeat_ld, int64_t pad_row, float* dte, int nparts, hipStream_t s) {
    RingCfg c;
    const int tr_bytes = dfeat ? ((a.dn * 4 + 15) / 16) * 16 : 0;
    const int red_per_wave = (2 * a.dt_dim + a.dn) * 4;
    const int smax = dfeat ? (a.heads == 1 ? 12 : 8) : (a.heads == 1 || g_wide ? 16 : 12);
    if (!aligned16(u) || !aligned16(agg) || !aligned16(dagg) || !aligned16(du) || !aligned16(prob) || !make_cfg(a, 3, 1024 + 512, tr_bytes, red_per_wave, smax, &c)) return 1;
    if (a.heads * a.k * 4 > 512) return 1;
    const unsigned grid = (unsigned)((a.m + c.per_wg - 1) / c.per_wg);
    if ((int)grid > nparts) return 1;
    const size_t lds = (size_t)c.S * c.wsz;
    static bool attr[8] = {false, false, false, false, false, false, false, false};
#define FLID_LAUNCH(i, HH, DFF, MW)                                                                                                            \
    do {                                                                                                                                       \
        if (!attr[i]) { TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_ring_kernel<HH, DFF, MW>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes)); attr[i] = true; } \
        attn_bwd_ring_kernel<HH, DFF, MW><<<dim3(grid), dim3(c.S * kWave), lds, s>>>(a, u, agg, prob, dagg, du, dfeat, dfeat_ld, pad_row, dte, nparts, c); \
    } while (0)
    if (dfeat) {
        if (a.heads == 1) FLID_LAUNCH(0, 1, true, 12); else FLID_LAUNCH(1, 2, true, 8);
    } else {
        if (a.heads == 1) FLID_LAUNCH(2, 1, false, 16);
        else if (c.S <= 12) FLID_LAUNCH(3, 2, false, 12);
        else FLID_LAUNCH(4, 2, false, 16);
    }
#undef FLID_LAUNCH
    return launch_status("attn_bwd_ring_kernel");
}

}  // namespace tg
