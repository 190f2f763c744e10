// Row-block CHAINS of a temporal-attention layer: the products that follow one another on the same rows run in ONE launch, a
// 64-row block per workgroup, the intermediate rows handed from product to product through LDS (bf16 hi / lo images) instead of HBM.
//
// replaces, per layer (models/modules.py:229-238 + :58-69 as called from models/TGAT.py:132-142, and their autograd):
//   forward  (tg_chain_fwd):  ctx_h = agg_h Wv_h^T  ->  res = ctx Wr^T + br  ->  y = LayerNorm(dropout(res) + [own | cos b])
//                             ->  f1 = relu([y | raw] W1^T + b1)  ->  out = f1 W2^T + b2
//   backward (tg_chain_bwd):  df1 = (dout W2) * (f1 > 0)  ->  dy = df1 W1[:, :dq]  ->  LayerNorm backward (dsum, dres, column sums)
//                             ->  dctx = dres Wr  ->  dagg_h = dctx_h Wv_h
// Every intermediate the backward / the weight gradients need is still written to HBM once (ctx, res, y, f1; df1, dres, dctx,
// dagg) -- what disappears is reading them back, and above all the launch + first-load + drain latency of five dependent launches:
// at 13.6 k rows a product is 213 workgroups that each wait ~2 us for their first rows, compute for 1-3 us and drain
// (round-3 ablation builds: 11-17 us per launch whatever is ablated), so a layer's ten products + two LayerNorm passes cost ~200 us for
// ~25 us of MFMA work.
//
// Machine.  4 waves; wave w owns a block of 16-column tiles of the current product's output, all 64 rows (four 16-row blocks):
//   * weights come PACKED (tg_pack_weights: bf16 hi / lo in MFMA fragment order) straight from L2 into registers, two register sets
//     of two 32-deep steps each, the set for steps s+2, s+3 in flight while s, s+1 multiply;
//   * the MFMA is issued with the WEIGHT fragment as operand A: the accumulator then holds, per lane, 4 consecutive output columns
//     of one row -- a float4 for the HBM store and an 8-byte hi / lo pair for the LDS image of the next product;
//   * the first product of a chain streams its rows from HBM through a two-buffer ring (as gemm_rows_kernel) that lives in the
//     still empty panel; later products read the whole K from the panel the previous epilogue wrote.  LDS: 15 chunks x (64 rows x
//     32 k x hi|lo) + reductions = 126 848 B -- one workgroup per CU.  Few rows (the 1 200-row root layer) take 16-row blocks:
//     the weights are streamed per workgroup whatever its height (~1.3 MB at ~65 GB/s per CU), so more, shorter blocks only add CUs.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "tg_common.h"
#include "tg_split.h"

#ifndef FLID_CHAIN_STAMPS
#define FLID_CHAIN_STAMPS 0
#endif
#ifndef FLID_CHAIN_SCHED
#define FLID_CHAIN_SCHED 1
#endif
#ifndef FLID_CHAIN_X4
#define FLID_CHAIN_X4 1   // keep the lo * lo term (see step()); 0: three terms, as the launch-per-product kernels
#endif
#ifndef FLID_CHAIN_ROT
#define FLID_CHAIN_ROT 1   // every workgroup walks a product's contraction from its own starting group (see Wave::rot_seed): chain_fwd
                           // 71.8 -> 67.3 us, chain_bwd 74.7 -> 71.6 us at 13.6 k rows (round 5, 4-wave form).  A row's rounding then depends
                           // on the row block it sits in -- reproducible run to run because the shared rows of a call are numbered in the
                           // order of their first occurrence (tg_dedupe_pairs), not in the hash set's arrival order as before.  0: all from k = 0
#endif
#ifndef FLID_CHAIN_EXP
#define FLID_CHAIN_EXP 0   // timing experiments only (results wrong): 1 no steady-state weight loads, 2 no MFMAs, 3 all fragment reads from one chunk
#endif

namespace {

using namespace tgs;

constexpr int NCH = 15;                      // panel chunks: [y (9) | raw (6)] of the merge layer is the widest operand
// RB row blocks of 16; NW waves (4, or 8 = two per SIMD).  Measured on the 13.6 k-row layer: 8 waves run the chain in the same time as
// 4 (112.8 k vs 115 k cycles per workgroup) -- every product moves its weights at ~20-27 B / cycle / CU whatever the wave count: the
// launch is bound by the CU's path to L2, not by issue slots -- and spill registers; the launchers instantiate 4.
template <int RB, int NW>
struct Geo {
    static constexpr int ROWS = 16 * RB;
    static constexpr int NTH = 64 * NW;
    static constexpr int CHS = ROWS * 64 + 64;       // one plane of one 32-k chunk (+64: the chunk stores of one row spread over banks)
    static constexpr int CHUNK = 2 * CHS;            // hi plane | lo plane
    static constexpr int PANEL = NCH * CHUNK;
    static constexpr int RED_OFF = PANEL;
    static constexpr int PAR_OFF = RED_OFF + 2 * NW * ROWS * 4;   // bias / LayerNorm vectors of the layer (forward chain): 3 x 320 + 2 x 192 floats
    static constexpr int LDS_BYTES = PAR_OFF + (3 * 320 + 2 * 192) * 4;
    static constexpr int RPP = 4 * NW;               // rows per pass when a product streams its rows (16 float4 per row and group)
    static constexpr int PER = ROWS / RPP;           // float4 per thread, operand and group
    static constexpr int NTW = NW == 16 ? 2 : NW == 8 ? 3 : 5;      // most column tiles a wave owns in any product of a chain (<= 20 / NW ... 18 tiles)
    static_assert(ROWS % RPP == 0, "row passes must tile the block");
};
static_assert(Geo<4, 16>::LDS_BYTES <= 163840 && Geo<4, 8>::LDS_BYTES <= 163840, "one workgroup must fit the CU's LDS");

// state of one wave inside a chain
template <int RB, int NW, int NTW_ = Geo<RB, NW>::NTW>
struct Wave {
    using G = Geo<RB, NW>;
    static constexpr int NTW = NTW_;
    char* lds;
    int lane, wave;
    int64_t row0, R;
    f32x4 acc[RB][NTW];
    bf16x8 b0h[2][NTW], b0l[2][NTW], b1h[2][NTW], b1l[2][NTW];
    const uint4* bptr[NTW];
    int S, t0, tcnt;          // current product: steps, first tile of this wave (inside its operand), tiles it owns
    // Every workgroup of a launch streams the SAME packed weights, and they all start together: walking the contraction in the same
    // order they ask the XCD's L2 for the same lines at the same time (one or two of its 16 channels busy, the rest idle).  So workgroup
    // b starts a product's contraction at group (b / 8) mod ngroups and wraps around: the 32 CUs of an XCD spread over the whole operand.
    // (The order of a row's partial sums then depends on its row block -- fixed per row, so results stay reproducible run to run.)
    int rot_seed = 0, rot = 0, ng = 1;
    __device__ __forceinline__ int Gr(int i) const { const int x = i + rot; return x >= ng ? x - ng : x; }
    __device__ __forceinline__ int Gx(int i) const { return i < ng ? Gr(i) : ng; }     // past the end: a group of zero rows
    bool pre = false;         // the first weight group of the product about to begin() is already on its way into b0 (prefetch())
#if FLID_CHAIN_STAMPS >= 2
    unsigned long long* fine = nullptr;     // diagnostic: stamps inside run_panel (one product only)
    int fine_i = 0;
    __device__ __forceinline__ void fstamp() {
        __builtin_amdgcn_sched_barrier(0);
        if (fine && threadIdx.x == 0 && fine_i < 16) fine[fine_i] = __builtin_amdgcn_s_memtime();
        ++fine_i;
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    __device__ __forceinline__ void fstamp() {}
#endif

    // the product's nt column tiles are dealt in blocks to waves [w0, w0 + nw); `packed` = that operand
    __device__ __forceinline__ void begin(const void* packed, int nt, int steps, int w0 = 0, int nw = NW) {
        S = steps;
        ng = (steps + 1) >> 1;
        rot = FLID_CHAIN_ROT ? rot_seed % ng : 0;
        const int cpw = (nt + nw - 1) / nw;
        t0 = (wave - w0) * cpw;
        tcnt = nt - t0 < cpw ? nt - t0 : cpw;
        if (tcnt < 0 || wave < w0 || wave >= w0 + nw) tcnt = 0;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            int t = t0 + j;
            if (t > nt - 1) t = nt - 1;
            if (t < 0) t = 0;
            bptr[j] = reinterpret_cast<const uint4*>(packed) + (int64_t)t * S * 128 + lane;
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The NEXT product's first weight group, issued while the current product's epilogue runs (its b0 set is dead by then): a product
    // otherwise starts with one exposed round trip to L2 (~3 k cycles; five products per chain, one wave per SIMD).  Same arguments as the
    // begin() that follows; the current product's t0 / tcnt / S stay valid for its epilogue.
    template <int NT>
    __device__ __forceinline__ void prefetch(const void* packed, int nt, int steps, int w0 = 0, int nw = NW) {
        const int cpw = (nt + nw - 1) / nw;
        const int pt0 = (wave - w0) * cpw;
        const int g0 = FLID_CHAIN_ROT ? rot_seed % ((steps + 1) >> 1) : 0;      // the group that product starts with (begin() computes the same)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int sc = 2 * g0 + sl < steps ? 2 * g0 + sl : steps - 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int t = pt0 + j;
                if (t > nt - 1) t = nt - 1;
                if (t < 0) t = 0;
                const uint4* q = reinterpret_cast<const uint4*>(packed) + (int64_t)t * steps * 128 + lane + sc * 128;
                b0h[sl][j] = __builtin_bit_cast(bf16x8, q[0]);
                b0l[sl][j] = __builtin_bit_cast(bf16x8, q[64]);
            }
        }
        pre = true;
    }
    // NT = tiles a wave owns in the current product (<= NTW): the loops below run over NT, not over the register arrays' NTW
    template <int NT>
    __device__ __forceinline__ void loadB(bf16x8 (&bh)[2][NTW], bf16x8 (&bl)[2][NTW], int g) {
        if (FLID_CHAIN_EXP == 1 && g > 0) return;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int s = 2 * g + sl;
            const int sc = s < S ? s : S - 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[sl][j] = __builtin_bit_cast(bf16x8, bptr[j][sc * 128]);
                bl[sl][j] = __builtin_bit_cast(bf16x8, bptr[j][sc * 128 + 64]);
            }
        }
    }
    // one 32-deep step: A fragments of the row blocks from the chunk at `chunk`, against one step of a B set
    template <int NT>
    __device__ __forceinline__ void step(const char* chunk, const bf16x8 (&bh)[NTW], const bf16x8 (&bl)[NTW]) {
        const char* p = (FLID_CHAIN_EXP == 3 ? lds : chunk) + frag_off(lane);
        bf16x8 ah[RB], al[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            ah[rb] = *reinterpret_cast<const bf16x8*>(p + rb * 1024);
            al[rb] = *reinterpret_cast<const bf16x8*>(p + G::CHS + rb * 1024);
        }
        if (FLID_CHAIN_EXP == 2) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float4 x = __builtin_bit_cast(float4, ah[rb]), y = __builtin_bit_cast(float4, bh[j]);
                    const float4 z = __builtin_bit_cast(float4, al[rb]), w = __builtin_bit_cast(float4, bl[j]);
                    acc[rb][j][0] += x.x + y.y + z.z + w.w;
                }
            return;
        }
        // weight fragment as operand A: acc[rb][j][r] = C[row 16 rb + (lane & 15)][column 16 (t0 + j) + 4 (lane >> 4) + r]
        // The chain keeps the lo * lo term too (4 MFMAs per fragment pair): its launches are bound by the weight stream and by latency,
        // not by the matrix pipe (PMC: 23 % busy), and with three terms the realistic full-size fixture showed hundreds of gradient
        // entries off by 1e-3 of the tensor's largest (ReLU units of the merge layer flipping; see tools/fullsize_err.py) where four show none.
        if (RB == 1 || FLID_CHAIN_X4) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], al[rb], acc[rb][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[rb], acc[rb][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[rb], acc[rb][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[rb], acc[rb][j], 0, 0, 0);
    }

    // Scheduling hint for a block that issues NV vector loads and then one step's 4 RB NT MFMAs: the fragment reads first, then one
    // load behind every RB MFMAs.  Issued back to back, a group's 4 NT weight loads (1 KiB each: 16 cycles of the CU's address
    // path apiece, times four waves) held the wave's in-order issue for 400-1 450 cycles before its first MFMA (stamps inside the res
    // product: steps 1 100 cycles, load blocks 400-1 450) -- spread out, they cost no issue time at all.
    template <int NT, int NV>
    __device__ __forceinline__ void spread_loads() {
#if FLID_CHAIN_SCHED
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * RB, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, (4 * RB * NT) / NV > 0 ? (4 * RB * NT) / NV : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#endif
    }

    // ---- product whose rows come from the LDS panel: chunks [chunk0, chunk0 + S).  No barrier inside.
    template <int NT>
    __device__ __forceinline__ void run_panel(int chunk0) {
        const char* base = lds + chunk0 * G::CHUNK;
        fstamp();
        if (!pre) loadB<NT>(b0h, b0l, Gr(0));                   // (uniform: set by prefetch() at compile-time-known call sites)
        pre = false;
        fstamp();
        for (int g = 0; g < ng; g += 2) {
            const int g0 = Gr(g), g1 = g + 1 < ng ? Gr(g + 1) : g0, g2 = g + 2 < ng ? Gr(g + 2) : g0;
            loadB<NT>(b1h, b1l, g1);
            fstamp();
            step<NT>(base + (2 * g0) * G::CHUNK, b0h[0], b0l[0]);
            spread_loads<NT, 4 * NT>();
            fstamp();
            if (2 * g0 + 1 < S) step<NT>(base + (2 * g0 + 1) * G::CHUNK, b0h[1], b0l[1]);
            fstamp();
            fstamp();
            if (g + 1 < ng) {                                  // (uniform; the loads of group g + 2 inside the block whose MFMAs hide them)
                loadB<NT>(b0h, b0l, g2);
                step<NT>(base + (2 * g1) * G::CHUNK, b1h[0], b1l[0]);
                spread_loads<NT, 4 * NT>();
            }
            fstamp();
            if (g + 1 < ng && 2 * g1 + 1 < S) step<NT>(base + (2 * g1 + 1) * G::CHUNK, b1h[1], b1l[1]);
            fstamp();
        }
    }

    // ---- product whose rows stream from HBM (fp32, split here): HH operands side by side (the heads of the value projection: operand
    // hh = columns [hh a_stride, hh a_stride + K) of A), this wave multiplies operand `mine`.  Ring: operand hh, buffer b, step sl at
    // panel chunk (2 hh + b) * 2 + sl -- the panel is still empty when a chain's first product runs.  Half h multiplies group h
    // (buffer h & 1, B set h & 1) while group h + 1 goes registers -> LDS and the rows of group h + 3 start their trip from HBM; loads
    // are unconditional (past the end of K they re-read k = 0 and are stored as zeros).  Ends with a barrier.
    template <int NT, int HH>
    __device__ __forceinline__ void run_stream(const float* __restrict__ A, int64_t lda, int64_t a_stride, int K, int mine) {
        const int tid = wave * 64 + lane;
        const int c4 = tid & 15, r16 = tid >> 4;           // 16 float4 per row, operand and group; RPP rows per pass
        constexpr int PER = G::PER;
        const float* aptr[PER];
        bool rok[PER];
        int aoff[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = r16 + G::RPP * i;
            int64_t rg = row0 + r;
            rok[i] = rg < R;
            if (rg > R - 1) rg = R - 1;
            aptr[i] = A + rg * lda;
            aoff[i] = (c4 >> 3) * G::CHUNK + chunk_off(r, (c4 & 7) * 4);
        }
        auto loadA = [&](float4 (&ra)[HH][PER], int g) {
            const int k = 64 * g + 4 * c4;
            const int ko = k < K ? k : 0;
#pragma unroll
            for (int hh = 0; hh < HH; ++hh)
#pragma unroll
                for (int i = 0; i < PER; ++i) ra[hh][i] = *reinterpret_cast<const float4*>(aptr[i] + hh * a_stride + ko);
        };
        auto writeA = [&](int buf, const float4 (&ra)[HH][PER], int g) {
            const bool kok = 64 * g + 4 * c4 < K;
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) {
                char* base = lds + ((2 * hh + buf) * 2) * G::CHUNK;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const bool ok = kok && rok[i];
                    const float4 v = make_float4(ok ? ra[hh][i].x : 0.f, ok ? ra[hh][i].y : 0.f, ok ? ra[hh][i].z : 0.f, ok ? ra[hh][i].w : 0.f);
                    uint2 hi, lo;
                    split4(v, hi, lo);
                    *reinterpret_cast<uint2*>(base + aoff[i]) = hi;
                    *reinterpret_cast<uint2*>(base + G::CHS + aoff[i]) = lo;
                }
            }
        };
        const int ngroups = ng;
        float4 ra0[HH][PER], ra1[HH][PER];
        fstamp();
        loadA(ra0, Gx(0));
        loadB<NT>(b0h, b0l, Gr(0));
        loadA(ra1, Gx(1));
        writeA(0, ra0, Gx(0));
        loadA(ra0, Gx(2));
        __syncthreads();
        fstamp();
        const char* r0 = lds + (2 * mine) * 2 * G::CHUNK, *r1 = r0 + 2 * G::CHUNK;
        for (int g = 0; g < ngroups; g += 2) {
            writeA(1, ra1, Gx(g + 1));
            fstamp();
            loadA(ra1, Gx(g + 3));
            loadB<NT>(b1h, b1l, g + 1 < ngroups ? Gr(g + 1) : Gr(g));
            step<NT>(r0, b0h[0], b0l[0]);
            spread_loads<NT, 4 * NT + HH * PER>();
            step<NT>(r0 + G::CHUNK, b0h[1], b0l[1]);        // (a step past the end of K multiplies zero rows)
            fstamp();
            __syncthreads();
            fstamp();
            writeA(0, ra0, Gx(g + 2));
            fstamp();
            if (g + 1 < ngroups) {                              // (uniform; past the last group nothing more is needed)
                loadA(ra0, Gx(g + 4));
                loadB<NT>(b0h, b0l, g + 2 < ngroups ? Gr(g + 2) : Gr(g));
                step<NT>(r1, b1h[0], b1l[0]);
                spread_loads<NT, 4 * NT + HH * PER>();
                step<NT>(r1 + G::CHUNK, b1h[1], b1l[1]);
            }
            fstamp();
            __syncthreads();
            fstamp();
        }
    }
    // coordinates of acc[rb][j]: row (inside the block) and first of its 4 columns (inside the wave's operand)
    __device__ __forceinline__ int out_row(int rb) const { return rb * 16 + (lane & 15); }
    __device__ __forceinline__ int out_col(int j) const { return (t0 + j) * 16 + 4 * (lane >> 4); }
    // 4 columns [col, col + 4) of `row` into the panel image whose column 0 sits at chunk `chunk0`
    __device__ __forceinline__ void panel_store(int chunk0, int row, int col, const float4& v) {
        uint2 hi, lo;
        split4(v, hi, lo);
        char* p = lds + (chunk0 + (col >> 5)) * G::CHUNK + chunk_off(row, col & 31);
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + G::CHS) = lo;
    }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(const f32x4& a) { return make_float4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (they hold the other column groups of the same row)
__device__ __forceinline__ float quad_rows_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// sum over the 16 lanes that share l >> 4 (they hold the same columns of the block's 16 rows): DPP inside the VALU, every lane of the
// row ends up with the row's sum (the first four steps of tg::wave_sum)
__device__ __forceinline__ float rows16_sum(float v) {
    v = tg::dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    v = tg::dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    v = tg::dpp_add<0x141>(v);         // row_half_mirror
    v = tg::dpp_add<0x140>(v);         // row_mirror
    return v;
}

struct ChainFwdArgs {
    int64_t R;
    int H, dn, T, de;                 // dq = dn + T, hd = dq / H, dk = dn + de + T
    int hp;                           // per-head block of ctx inside the panel (hd rounded up to 16)
    const float* agg;                 // (R, H dk)
    int64_t packed_bytes;             // pWv .. end of pW2: one contiguous block (L2 warm-up)
    const void *pWv, *pWr, *pW1, *pW2;   // packed: Wv heads back to back (tg_packed_floats(hd, dk) apart), Wr with K = H hp, W1 with K = [y | raw] padded
    const float *br, *b1, *b2, *ln_g, *ln_b, *cosb;
    const float* own; int64_t own_ld;
    float p_res; uint64_t seed;
    float* ctx;                       // (R, dq)
    float* res;                       // (R, dq)
    float* y; int64_t y_ld;           // (R, dq) inside the [y | raw] buffer
    const float* raw; int64_t raw_ld; // (R, dn)
    float *mean, *rstd, *f1, *out;
    unsigned long long* dbg;          // diagnostic builds only: 16 stamps per workgroup
};

// HH = heads (1 or 2); NW = waves.  Tiles per wave (compile-time loop bounds): the widest product (res, dq <= 16 NTW NW columns)
// takes NTW, the merge layer's (dn columns) NTF.
template <int RB, int HH, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) chain_fwd_kernel(ChainFwdArgs a) {
    using G = Geo<RB, NW>;
    constexpr int ROWS = G::ROWS, NTH = G::NTH, NTW = G::NTW, NTF = NW == 16 ? 1 : NW == 8 ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    Wave<RB, NW> w;
    w.lds = lds;
    w.lane = threadIdx.x & 63;
    w.wave = threadIdx.x >> 6;
    w.row0 = (int64_t)blockIdx.x * ROWS;
    w.R = a.R;
    w.rot_seed = blockIdx.x >> 3;
    const int lane = w.lane, tid = threadIdx.x;
    const int dq = a.dn + a.T, hd = dq / HH, dk = a.dn + a.de + a.T;
    const int ychunks = (dq + 31) >> 5, rchunks = (a.dn + 31) >> 5;
    float* red = reinterpret_cast<float*>(lds + G::RED_OFF);
#if FLID_CHAIN_STAMPS == 1
    int stamp_i = 0;
#define STAMP() do { if (a.dbg && tid == 0) a.dbg[blockIdx.x * 16 + stamp_i] = __builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define STAMP() do {} while (0)
#endif
    STAMP();
    // the layer's bias and LayerNorm vectors into LDS (visible behind the first product's barriers): read from memory inside an epilogue
    // each of them was an exposed round trip to L2 with nothing else for the wave to do
    float* par = reinterpret_cast<float*>(lds + G::PAR_OFF);
    float *p_br = par, *p_g = par + 320, *p_b = par + 640, *p_b1 = par + 960, *p_b2 = par + 1152;
    for (int i = tid; i < dq; i += NTH) { p_br[i] = a.br[i]; p_g[i] = a.ln_g[i]; p_b[i] = a.ln_b[i]; }
    for (int i = tid; i < a.dn; i += NTH) { p_b1[i] = a.b1[i]; p_b2[i] = a.b2[i]; }

    // ---- the merge layer's raw rows: in flight now, into the panel (behind y) after the first product
    constexpr int RAWN = ROWS * 48 / NTH;                     // float4 per thread: ROWS x 192 columns
    float4 rawv[RAWN];
    const int rcols4 = a.dn >> 2;
#pragma unroll
    for (int i = 0; i < RAWN; ++i) {
        const int f = tid + NTH * i, r = f / 48, c = f % 48;
        int64_t rg = w.row0 + r;
        if (rg > a.R - 1) rg = a.R - 1;
        rawv[i] = ld4(a.raw + rg * a.raw_ld + (c < rcols4 ? 4 * c : 0));
    }

    // ---- ctx_h = agg_h Wv_h^T: both heads at once, waves [0, 2) on head 0 and [2, 4) on head 1 (one head: all four waves)
    {
        const int ht = (hd + 15) >> 4;                        // tiles per head
        const int64_t wv_stride = (int64_t)((hd + 15) / 16) * ((dk + 31) / 32) * 512;   // floats per packed head (tg_packed_floats(hd, dk))
        const int mine = HH == 2 ? w.wave / (NW / 2) : 0;
        w.begin(reinterpret_cast<const float*>(a.pWv) + mine * wv_stride, ht, (dk + 31) >> 5, HH == 2 ? (NW / 2) * mine : 0, NW / HH);
#if FLID_CHAIN_STAMPS == 3
        w.fine = a.dbg ? a.dbg + blockIdx.x * 16 : nullptr;
        w.fine_i = 0;
#endif
        w.template run_stream<NTW, HH>(a.agg, (int64_t)HH * dk, dk, dk, mine);
#if FLID_CHAIN_STAMPS == 3
        w.fine = nullptr;
#endif
        STAMP();
        w.template prefetch<NTW>(a.pWr, (dq + 15) >> 4, (HH * a.hp + 31) >> 5);
#pragma unroll
        for (int i = 0; i < RAWN; ++i) {
            const int f = tid + NTH * i, r = f / 48, c = f % 48;
            const bool ok = c < rcols4 && w.row0 + r < a.R;
            if (4 * c < 32 * rchunks) w.panel_store(ychunks, r, 4 * c, ok ? rawv[i] : zero4());
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);                     // inside the head's block
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = w.out_row(rb);
                const float4 v = f4(w.acc[rb][j]);
                w.panel_store(0, r, mine * a.hp + col, v);    // columns >= hd of the block are zero (zero rows of the packed weight)
                if (col < hd && w.row0 + r < a.R) st4(a.ctx + (w.row0 + r) * dq + mine * hd + col, v);
            }
        }
    }
    if ((HH * a.hp) & 31) {                                    // tail of the ctx panel's last chunk: multiplies zero columns of Wr, must be finite
        const int c0 = HH * a.hp, per = (32 - (c0 & 31)) >> 2;
        for (int f = tid; f < ROWS * per; f += NTH) w.panel_store(0, f / per, c0 + 4 * (f % per), zero4());
    }
    __syncthreads();

    // ---- res = ctx Wr^T + br ;  y = LayerNorm(dropout(res) + [own | cos b]) * g + b
    {
        const int kc = (HH * a.hp + 31) >> 5;
        STAMP();
        w.begin(a.pWr, (dq + 15) >> 4, kc);
        // the residual's rows [own | cos b] for this wave's columns: in flight under the product
        f32x4 x[RB][NTW];
        float s1[RB];
        bool rok[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) { s1[rb] = 0.f; rok[rb] = w.row0 + w.out_row(rb) < a.R; }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int col = w.out_col(j);
            const bool cok = j < w.tcnt && col < dq;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                int64_t rg = w.row0 + w.out_row(rb);
                if (rg > a.R - 1) rg = a.R - 1;
                const float* src = !cok ? a.cosb : (col < a.dn ? a.own + rg * a.own_ld + col : a.cosb + (col - a.dn));
                const float4 o = ld4(src);
                x[rb][j] = f32x4{o.x, o.y, o.z, o.w};
            }
        }
#if FLID_CHAIN_STAMPS == 2
        w.fine = a.dbg ? a.dbg + blockIdx.x * 16 : nullptr;
        w.fine_i = 0;
#endif
        w.template run_panel<NTW>(0);
#if FLID_CHAIN_STAMPS == 2
        w.fine = nullptr;
#endif
        STAMP();
        w.template prefetch<NTF>(a.pW1, (a.dn + 15) >> 4, ychunks + rchunks);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);
            if (col >= dq) {                                   // columns past dq in the last tile (lane-wise): not part of the row
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) x[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const float4 b4 = ld4(p_br + col);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                int64_t rg = w.row0 + w.out_row(rb);
                if (rg > a.R - 1) rg = a.R - 1;
                float4 v = add4(f4(w.acc[rb][j]), b4);
                float ks[4];
                tg::res_keep_scale4(a.seed, rg * dq + col, a.p_res, ks);
                v.x = v.x * ks[0] + x[rb][j][0];
                v.y = v.y * ks[1] + x[rb][j][1];
                v.z = v.z * ks[2] + x[rb][j][2];
                v.w = v.w * ks[3] + x[rb][j][3];
                x[rb][j] = f32x4{v.x, v.y, v.z, v.w};
                s1[rb] += (v.x + v.y) + (v.z + v.w);
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            s1[rb] = quad_rows_sum(s1[rb]);
            if (lane < 16) red[w.wave * ROWS + rb * 16 + lane] = s1[rb];
        }
        __syncthreads();                                       // (also: every wave is done reading the ctx panel)
        float mu[RB], s2[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NW; ++q) t += red[q * ROWS + r];
            mu[rb] = t / dq;
            s2[rb] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= w.tcnt) break;
            if (w.out_col(j) >= dq) continue;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = x[rb][j][e] - mu[rb]; s2[rb] = fmaf(d, d, s2[rb]); }
        }
        float* red2 = red + NW * ROWS;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            s2[rb] = quad_rows_sum(s2[rb]);
            if (lane < 16) red2[w.wave * ROWS + rb * 16 + lane] = s2[rb];
        }
        __syncthreads();
        float rs[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NW; ++q) t += red2[q * ROWS + r];
            rs[rb] = rsqrtf(t / dq + 1e-5f);
            if (w.wave == 0 && lane < 16 && rok[rb]) { a.mean[w.row0 + r] = mu[rb]; a.rstd[w.row0 + r] = rs[rb]; }
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);
            if (col >= dq) continue;                           // (the zero fill below covers the panel's tail)
            const float4 g4 = ld4(p_g + col), be4 = ld4(p_b + col);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = w.out_row(rb);
                // the normalised input is what the backward needs of this stage: it is saved in the `res` buffer (with the chain, `res`
                // holds xhat = (dropout(res) + [own | cos b] - mean) * rstd, not the projection's raw output)
                const float4 xn = make_float4((x[rb][j][0] - mu[rb]) * rs[rb], (x[rb][j][1] - mu[rb]) * rs[rb],
                                              (x[rb][j][2] - mu[rb]) * rs[rb], (x[rb][j][3] - mu[rb]) * rs[rb]);
                float4 v = make_float4(xn.x * g4.x + be4.x, xn.y * g4.y + be4.y, xn.z * g4.z + be4.z, xn.w * g4.w + be4.w);
                if (!rok[rb]) v = zero4();
                w.panel_store(0, r, col, v);
                if (rok[rb]) { st4(a.y + (w.row0 + r) * a.y_ld + col, v); st4(a.res + (w.row0 + r) * dq + col, xn); }
            }
        }
        // the tail of y's last chunk (columns dq .. 32 ychunks) multiplies zero columns of the packed W1 but must be finite
        if (32 * ychunks > dq) {
            const int per = (32 * ychunks - dq) >> 2;
            for (int f = tid; f < ROWS * per; f += NTH) w.panel_store(0, f / per, dq + 4 * (f % per), zero4());
        }
    }
    __syncthreads();

    // ---- f1 = relu([y | raw] W1^T + b1)
    STAMP();
    w.begin(a.pW1, (a.dn + 15) >> 4, ychunks + rchunks);
    w.template run_panel<NTF>(0);
    STAMP();
    w.template prefetch<NTF>(a.pW2, (a.dn + 15) >> 4, rchunks);
    __syncthreads();                                           // every wave is done with [y | raw]: f1 takes its place
#pragma unroll
    for (int j = 0; j < NTF; ++j) {
        if (j >= w.tcnt) break;
        const int col = w.out_col(j);
        if (col >= a.dn) {                                     // padding columns of the last tile: zeros for the next product
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) w.panel_store(0, w.out_row(rb), col, zero4());
            continue;
        }
        const float4 b4 = ld4(p_b1 + col);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            float4 v = add4(f4(w.acc[rb][j]), b4);
            v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            w.panel_store(0, r, col, v);
            if (w.row0 + r < a.R) st4(a.f1 + (w.row0 + r) * a.dn + col, v);
        }
    }
    __syncthreads();

    // ---- out = f1 W2^T + b2
    STAMP();
    w.begin(a.pW2, (a.dn + 15) >> 4, rchunks);
    w.template run_panel<NTF>(0);
    STAMP();
#pragma unroll
    for (int j = 0; j < NTF; ++j) {
        if (j >= w.tcnt) break;
        const int col = w.out_col(j);
        if (col >= a.dn) continue;
        const float4 b4 = ld4(p_b2 + col);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            if (w.row0 + r < a.R) st4(a.out + (w.row0 + r) * a.dn + col, add4(f4(w.acc[rb][j]), b4));
        }
    }
    STAMP();
}

struct ChainBwdArgs {
    int64_t R;
    int H, dn, T, de;
    int hpb;                          // per-head block of dctx inside the panel (hd rounded up to 32: a head's block starts on a chunk)
    const float* dout;                // (R, dn)
    const float* f1;                  // (R, dn) forward activations (ReLU mask)
    const void *pW2T, *pW1aT, *pWrT, *pWvT;   // packed transposed weights; pWvT: heads back to back (tg_packed_floats(dk, hd) apart)
    const void* pW1bT;                // optional (with d_raw): W1[:, dq:]^T packed
    float* d_raw;                     // optional: (R, dn) gradient w.r.t. the merge layer's raw rows = df1 W1[:, dq:]
    const float *ln_g, *cosb;
    const float* own; int64_t own_ld;
    const float* res;                 // (R, dq)
    const float *mean, *rstd;
    float p_res; uint64_t seed;
    float* df1;                       // (R, dn)
    float* dres;                      // (R, dq)   dropout-masked LayerNorm input gradient
    float* dctx;                      // (R, dq)
    float* dagg;                      // (R, H dk)
    float* d_own; int64_t d_own_ld; int d_own_acc;      // optional: (+)= the residual's share dsum[:, :dn]
    float* part;                      // (workgroups, 4 dq): [sum dy xhat | sum dy | sum dsum | sum dres] per workgroup
    unsigned long long* dbg;
};

template <int RB, int HH, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) chain_bwd_kernel(ChainBwdArgs a) {
    using G = Geo<RB, NW>;
    constexpr int ROWS = G::ROWS, NTH = G::NTH, NTW = G::NTW, NTF = NW == 16 ? 1 : NW == 8 ? 2 : 3;
    constexpr int NTD = NW == 16 ? 2 : NW == 8 ? 4 : 7;                       // tiles per wave of the widest product (dagg_h: dk columns in one pass)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    Wave<RB, NW, NTD> w;
    w.lds = lds;
    w.lane = threadIdx.x & 63;
    w.wave = threadIdx.x >> 6;
    w.row0 = (int64_t)blockIdx.x * ROWS;
    w.R = a.R;
    w.rot_seed = blockIdx.x >> 3;
    const int lane = w.lane, tid = threadIdx.x;
    const int dq = a.dn + a.T, hd = dq / HH, dk = a.dn + a.de + a.T;
    const int nchunks_dn = (a.dn + 31) >> 5, nchunks_dq = (dq + 31) >> 5;
    float* red = reinterpret_cast<float*>(lds + G::RED_OFF);
    bool rok[RB];
    int64_t rgc[RB];                                           // this lane's global rows, clamped
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t rg = w.row0 + w.out_row(rb);
        rok[rb] = rg < a.R;
        rgc[rb] = rg < a.R ? rg : a.R - 1;
    }

#if FLID_CHAIN_STAMPS == 1
    int stamp_i = 0;
#endif
    STAMP();
    // ---- df1 = (dout W2) * (f1 > 0)
    {
        w.begin(a.pW2T, (a.dn + 15) >> 4, nchunks_dn);
        f32x4 m[RB][NTF];                                        // the forward activations of this wave's columns: in flight under the product
#pragma unroll
        for (int j = 0; j < NTF; ++j) {
            const int col = w.out_col(j);
            const bool cok = j < w.tcnt && col < a.dn;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float4 v = ld4(a.f1 + (cok ? rgc[rb] * a.dn + col : 0));
                m[rb][j] = f32x4{v.x, v.y, v.z, v.w};
            }
        }
        // dout is narrow (dn <= 192 columns): the whole block goes into the panel at once (one trip to HBM, no ring, no barriers in
        // the product) -- streamed in three 64-k groups it took 18 k cycles for 4.6 k cycles of MFMA work
        {
            constexpr int DN4 = 48;                            // float4 per row (192 columns)
            constexpr int NL = ROWS * DN4 / NTH;
            float4 dv[NL];
            const int c4n = a.dn >> 2;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int f = tid + NTH * i, r = f / DN4, c = f % DN4;
                int64_t rg = w.row0 + r;
                if (rg > a.R - 1) rg = a.R - 1;
                dv[i] = ld4(a.dout + rg * a.dn + (c < c4n ? 4 * c : 0));
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int f = tid + NTH * i, r = f / DN4, c = f % DN4;
                const bool ok = c < c4n && w.row0 + r < a.R;
                if (4 * c < 32 * nchunks_dn) w.panel_store(0, r, 4 * c, ok ? dv[i] : zero4());
            }
        }
        __syncthreads();
        w.template run_panel<NTF>(0);
        STAMP();
        __syncthreads();                                       // every wave is done reading dout: df1 takes its place
#pragma unroll
        for (int j = 0; j < NTF; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = w.out_row(rb);
                float4 v = f4(w.acc[rb][j]);
                if (col >= a.dn) v = zero4();                  // padding columns of the last tile: zeros for the next product
                else {
                    v.x = m[rb][j][0] > 0.f ? v.x : 0.f; v.y = m[rb][j][1] > 0.f ? v.y : 0.f;
                    v.z = m[rb][j][2] > 0.f ? v.z : 0.f; v.w = m[rb][j][3] > 0.f ? v.w : 0.f;
                    if (rok[rb]) st4(a.df1 + rgc[rb] * a.dn + col, v);
                    else v = zero4();
                }
                w.panel_store(0, r, col, v);
            }
        }
        if ((16 * ((a.dn + 15) >> 4)) & 31) {                  // second half of the last chunk when dn's tiles end mid-chunk
            const int c0 = 16 * ((a.dn + 15) >> 4);
            for (int f = tid; f < ROWS * 4; f += NTH) w.panel_store(0, f >> 2, c0 + 4 * (f & 3), zero4());
        }
    }
    __syncthreads();
    STAMP();

    // ---- d raw = df1 W1[:, dq:] (layers whose raw rows carry a gradient): the panel holds df1, nothing else is needed
    if (a.d_raw) {                                             // (uniform)
        w.begin(a.pW1bT, (a.dn + 15) >> 4, nchunks_dn);
        w.template run_panel<NTF>(0);
#pragma unroll
        for (int j = 0; j < NTF; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);
            if (col >= a.dn) continue;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                if (rok[rb]) st4(a.d_raw + rgc[rb] * a.dn + col, f4(w.acc[rb][j]));
        }
    }

    // ---- dy = df1 W1[:, :dq] ;  LayerNorm backward: dsum, dres = dsum * dropout mask, the workgroup's column sums
    {
        w.begin(a.pW1aT, (dq + 15) >> 4, nchunks_dn);
        f32x4 xh[RB][NTW];                                     // LayerNorm's normalised input, saved by the forward chain in `res`
        unsigned km[RB][NTW];                                  // dropout keep bits of each float4 (bit e = element e kept)
        const float kscale = a.p_res > 0.f ? 1.f / (1.f - a.p_res) : 1.f;
        float rs[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) rs[rb] = a.rstd[rgc[rb]];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int col = w.out_col(j);
            const bool cok = j < w.tcnt && col < dq;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float4 xv = ld4(a.res + (cok ? rgc[rb] * dq + col : 0));
                xh[rb][j] = f32x4{xv.x, xv.y, xv.z, xv.w};
                float ks[4];
                tg::res_keep_scale4(a.seed, rgc[rb] * dq + (cok ? col : 0), a.p_res, ks);
                km[rb][j] = (ks[0] > 0.f ? 1u : 0u) | (ks[1] > 0.f ? 2u : 0u) | (ks[2] > 0.f ? 4u : 0u) | (ks[3] > 0.f ? 8u : 0u);
            }
        }
        w.template run_panel<NTW>(0);
        STAMP();
        // g = dy * gamma (in place in the accumulators); row sums of g and g * xhat
        float s1[RB], s2[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) s1[rb] = s2[rb] = 0.f;
        float4 gam[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int col = w.out_col(j);
            const bool cok = j < w.tcnt && col < dq;
            gam[j] = cok ? ld4(a.ln_g + col) : zero4();
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (!cok) { xh[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                const f32x4 d = w.acc[rb][j];
                s1[rb] += (d[0] * gam[j].x + d[1] * gam[j].y) + (d[2] * gam[j].z + d[3] * gam[j].w);
                s2[rb] = fmaf(d[0] * gam[j].x, xh[rb][j][0], fmaf(d[1] * gam[j].y, xh[rb][j][1],
                         fmaf(d[2] * gam[j].z, xh[rb][j][2], fmaf(d[3] * gam[j].w, xh[rb][j][3], s2[rb]))));
            }
        }
        float* red2 = red + NW * ROWS;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            s1[rb] = quad_rows_sum(s1[rb]);
            s2[rb] = quad_rows_sum(s2[rb]);
            if (lane < 16) { red[w.wave * ROWS + rb * 16 + lane] = s1[rb]; red2[w.wave * ROWS + rb * 16 + lane] = s2[rb]; }
        }
        __syncthreads();                                       // (also: every wave is done reading the df1 panel)
        float m1[RB], m2[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = w.out_row(rb);
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int q = 0; q < NW; ++q) { t1 += red[q * ROWS + r]; t2 += red2[q * ROWS + r]; }
            m1[rb] = t1 / dq;
            m2[rb] = t2 / dq;
        }
        float* part = a.part + (int64_t)blockIdx.x * 4 * dq;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);
            if (col >= dq) continue;
            const float gm[4] = {gam[j].x, gam[j].y, gam[j].z, gam[j].w};
            float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f}, c2[4] = {0.f, 0.f, 0.f, 0.f}, c3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = w.out_row(rb);
                float dx[4], dr[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = w.acc[rb][j][e];
                    dx[e] = rs[rb] * (d * gm[e] - m1[rb] - xh[rb][j][e] * m2[rb]);
                    dr[e] = ((km[rb][j] >> e) & 1u) ? dx[e] * kscale : 0.f;
                    if (!rok[rb]) { dx[e] = 0.f; dr[e] = 0.f; }
                    else { c0[e] = fmaf(d, xh[rb][j][e], c0[e]); c1[e] += d; c2[e] += dx[e]; c3[e] += dr[e]; }
                }
                const float4 drv = make_float4(dr[0], dr[1], dr[2], dr[3]);
                w.panel_store(0, r, col, drv);
                if (rok[rb]) {
                    st4(a.dres + rgc[rb] * dq + col, drv);
                    if (a.d_own && col < a.dn) {
                        float* o = a.d_own + rgc[rb] * a.d_own_ld + col;
                        float4 dv = make_float4(dx[0], dx[1], dx[2], dx[3]);
                        if (a.d_own_acc) dv = add4(dv, ld4(o));
                        st4(o, dv);
                    }
                }
            }
            // column sums over the workgroup's rows: the 16 lanes with the same l >> 4 hold the same columns
#pragma unroll
            for (int e = 0; e < 4; ++e) { c0[e] = rows16_sum(c0[e]); c1[e] = rows16_sum(c1[e]); c2[e] = rows16_sum(c2[e]); c3[e] = rows16_sum(c3[e]); }
            if ((lane & 15) == 0) {
                st4(part + col, make_float4(c0[0], c0[1], c0[2], c0[3]));
                st4(part + dq + col, make_float4(c1[0], c1[1], c1[2], c1[3]));
                st4(part + 2 * dq + col, make_float4(c2[0], c2[1], c2[2], c2[3]));
                st4(part + 3 * dq + col, make_float4(c3[0], c3[1], c3[2], c3[3]));
            }
        }
        if (32 * nchunks_dq > dq) {                            // tail of dres's last chunk
            const int per = (32 * nchunks_dq - dq) >> 2;
            for (int f = tid; f < ROWS * per; f += NTH) w.panel_store(0, f / per, dq + 4 * (f % per), zero4());
        }
    }
    __syncthreads();
    STAMP();

    // ---- dctx = dres Wr   (columns laid out in per-head blocks of hpb for the next product)
    {
        const int ntile = (HH * a.hpb) >> 4;
        w.begin(a.pWrT, ntile, nchunks_dq);
        w.template run_panel<NTW>(0);
        STAMP();
        __syncthreads();                                       // every wave is done with the dres panel
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);                      // padded column: head col / hpb, column col % hpb inside the head
            const int h = col / a.hpb, ch = col - h * a.hpb;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float4 v = f4(w.acc[rb][j]);
                if (!rok[rb]) v = zero4();
                w.panel_store(0, w.out_row(rb), col, v);       // (columns >= hd of a block are zero: zero rows of the packed weight)
                if (rok[rb] && ch < hd) st4(a.dctx + rgc[rb] * dq + h * hd + ch, v);
            }
        }
    }
    __syncthreads();
    STAMP();

    // ---- dagg_h = dctx_h Wv_h : dk columns per head, NTW NW tiles per pass
    {
        const int ht = (dk + 15) >> 4, hs = a.hpb >> 5;
        const int64_t wv_stride = (int64_t)ht * hs * 512;      // floats per packed head (tg_packed_floats(dk, hd), hd padded to hpb)
        const int npass = (ht + NTD * NW - 1) / (NTD * NW);   // passes per head (dk = 444: 28 tiles = one pass of 7 per wave)
        const int tpp = (ht + npass - 1) / npass;
        for (int h = 0; h < HH; ++h)
            for (int tb = 0; tb < ht; tb += tpp) {
                const int nt = ht - tb < tpp ? ht - tb : tpp;
                w.begin(reinterpret_cast<const float*>(a.pWvT) + h * wv_stride + (int64_t)tb * hs * 512, nt, hs);
                w.template run_panel<NTD>(h * hs);
#pragma unroll
                for (int j = 0; j < NTD; ++j) {
                    if (j >= w.tcnt) break;
                    const int col = 16 * tb + w.out_col(j);
                    if (col >= dk) continue;
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        if (rok[rb]) st4(a.dagg + rgc[rb] * ((int64_t)HH * dk) + h * dk + col, f4(w.acc[rb][j]));
                }
            }
    }
    STAMP();
}


// ---- the query side of a SHORT layer (the 1 200-row root layer, TGN's layer; the long layers take the merged projection) -------------
// forward:   q = own Wq[:, :dn]^T + qb   ->   u_h = q_h Wk_h            (models/modules.py:199-210 reassociated)
// backward:  dq_h = du_h Wk_h^T          ->   d_own (+)= dq Wq[:, :dn]
// Two products each, a launch apiece before (9 + 11 us forward, 20 + 8 us backward at 1 200 rows: single latency chains); here a 16-row
// block runs both, q / dq handed over through the LDS panel, against packed weights (the layer's prelude launch packs them).
struct QuFwdArgs {
    int64_t R;
    int H, dn, T, de, hpb;            // hpb: per-head block of q inside the panel (hd rounded up to 32)
    const float* own; int64_t own_ld;
    const void *pWq, *pWkT;           // packed Wq[:, :dn] (N = dq, K = dn); per head Wk_h^T (N = dk, K = hpb), heads back to back
    const float* qbias;
    float *q, *u;                     // (R, dq), (R, H dk)
};
template <int HH, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) qu_fwd_kernel(QuFwdArgs a) {
    using G = Geo<1, NW>;
    constexpr int NTH = G::NTH, NTD = 7, NTQ = 5;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    Wave<1, NW, NTD> w;
    w.lds = lds; w.lane = threadIdx.x & 63; w.wave = threadIdx.x >> 6;
    w.row0 = (int64_t)blockIdx.x * 16; w.R = a.R; w.rot_seed = blockIdx.x >> 3;
    const int tid = threadIdx.x;
    const int dq = a.dn + a.T, hd = dq / HH, dk = a.dn + a.de + a.T;
    const int r = w.out_row(0);
    const bool rok = w.row0 + r < a.R;
    const int64_t rg = rok ? w.row0 + r : a.R - 1;
    // q = own Wq[:, :dn]^T + qb: rows streamed from memory, all dq columns (NTQ tiles per wave)
    w.begin(a.pWq, (dq + 15) >> 4, (a.dn + 31) >> 5);
    w.template run_stream<NTQ, 1>(a.own, a.own_ld, 0, a.dn, 0);
#pragma unroll
    for (int j = 0; j < NTQ; ++j) {
        if (j >= w.tcnt) break;
        const int col = w.out_col(j);
        if (col >= dq) continue;
        const float4 v = rok ? add4(f4(w.acc[0][j]), ld4(a.qbias + col)) : zero4();
        if (rok) st4(a.q + rg * dq + col, v);
        const int h = col / hd, ch = col - h * hd;               // (hd % 4 == 0: a float4 never straddles two heads)
        w.panel_store(0, r, h * a.hpb + ch, v);
    }
    // the tails of the heads' blocks ([hd, hpb)) multiply zero rows of the packed weights but must be finite
    {
        const int per = (a.hpb - hd) >> 2;
        for (int f = tid; f < 16 * HH * per; f += NTH) {
            const int rr = f / (HH * per), x = f % (HH * per), h = x / per;
            w.panel_store(0, rr, h * a.hpb + hd + 4 * (x % per), zero4());
        }
    }
    __syncthreads();
    // u_h = q_h Wk_h: dk columns per head, NTD tiles per wave and pass
    {
        const int ht = (dk + 15) >> 4, hs = a.hpb >> 5;
        const int64_t stride = (int64_t)ht * hs * 512;
        const int npass = (ht + NTD * NW - 1) / (NTD * NW);
        const int tpp = (ht + npass - 1) / npass;
        for (int h = 0; h < HH; ++h)
            for (int tb = 0; tb < ht; tb += tpp) {
                const int nt = ht - tb < tpp ? ht - tb : tpp;
                w.begin(reinterpret_cast<const float*>(a.pWkT) + h * stride + (int64_t)tb * hs * 512, nt, hs);
                w.template run_panel<NTD>(h * hs);
#pragma unroll
                for (int j = 0; j < NTD; ++j) {
                    if (j >= w.tcnt) break;
                    const int col = 16 * tb + w.out_col(j);
                    if (col < dk && rok) st4(a.u + rg * ((int64_t)HH * dk) + h * dk + col, f4(w.acc[0][j]));
                }
            }
    }
}

struct DqBwdArgs {
    int64_t R;
    int H, dn, T, de, hp;             // hp: per-head block of dq inside the panel (hd rounded up to 16)
    const float* du;                  // (R, H dk)
    const void *pWk, *pWqT;           // packed per head Wk_h (N = hd, K = dk), heads back to back; Wq[:, :dn]^T (N = dn, K = H hp)
    float* dq;                        // (R, dq)
    float* d_own; int64_t d_own_ld; int d_own_acc;     // optional
    float* dq_sum;                    // (16-row blocks, dq) slab: row b = column sums of block b's dq rows (plain stores): what the time half of dWq needs
};
template <int HH, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) dq_bwd_kernel(DqBwdArgs a) {
    using G = Geo<1, NW>;
    constexpr int NTH = G::NTH, NTW = G::NTW, NTF = 3;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    Wave<1, NW> w;
    w.lds = lds; w.lane = threadIdx.x & 63; w.wave = threadIdx.x >> 6;
    w.row0 = (int64_t)blockIdx.x * 16; w.R = a.R; w.rot_seed = blockIdx.x >> 3;
    const int tid = threadIdx.x;
    const int dq = a.dn + a.T, hd = dq / HH, dk = a.dn + a.de + a.T;
    const int r = w.out_row(0);
    const bool rok = w.row0 + r < a.R;
    const int64_t rg = rok ? w.row0 + r : a.R - 1;
    // dq_h = du_h Wk_h^T: both heads at once (waves [0, NW / 2) on head 0, the rest on head 1), rows streamed from memory
    {
        const int ht = (hd + 15) >> 4;
        const int64_t stride = (int64_t)ht * ((dk + 31) / 32) * 512;
        const int mine = HH == 2 ? w.wave / (NW / 2) : 0;
        w.begin(reinterpret_cast<const float*>(a.pWk) + mine * stride, ht, (dk + 31) >> 5, HH == 2 ? (NW / 2) * mine : 0, NW / HH);
        w.template run_stream<NTW, HH>(a.du, (int64_t)HH * dk, dk, dk, mine);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= w.tcnt) break;
            const int col = w.out_col(j);                        // inside the head's block
            const float4 v = rok ? f4(w.acc[0][j]) : zero4();
            w.panel_store(0, r, mine * a.hp + col, v);           // columns >= hd of the block are zero (zero rows of the packed weight)
            if (col < hd && rok) st4(a.dq + rg * dq + mine * hd + col, v);
            if (a.dq_sum) {                                      // this block's 16 rows summed (rows past the end hold zeros)
                const float s0 = rows16_sum(v.x), s1 = rows16_sum(v.y), s2 = rows16_sum(v.z), s3 = rows16_sum(v.w);
                if ((w.lane & 15) == 0 && col < hd) st4(a.dq_sum + (int64_t)blockIdx.x * dq + mine * hd + col, make_float4(s0, s1, s2, s3));
            }
        }
    }
    if ((HH * a.hp) & 31) {                                      // tail of the panel's last chunk
        const int c0 = HH * a.hp, per = (32 - (c0 & 31)) >> 2;
        for (int f = tid; f < 16 * per; f += NTH) w.panel_store(0, f / per, c0 + 4 * (f % per), zero4());
    }
    __syncthreads();
    if (!a.d_own) return;                                        // (uniform)
    // d_own (+)= dq Wq[:, :dn]
    w.begin(a.pWqT, (a.dn + 15) >> 4, (HH * a.hp + 31) >> 5);
    w.template run_panel<NTF>(0);
#pragma unroll
    for (int j = 0; j < NTF; ++j) {
        if (j >= w.tcnt) break;
        const int col = w.out_col(j);
        if (col >= a.dn || !rok) continue;
        float* p = a.d_own + rg * a.d_own_ld + col;
        float4 v = f4(w.acc[0][j]);
        if (a.d_own_acc) v = add4(v, ld4(p));
        st4(p, v);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
unsigned long long* g_chain_dbg = nullptr;
unsigned long long* g_chain_dbg_bwd = nullptr;

template <int RB, int HH, int NW>
int launch_fwd(const ChainFwdArgs& a, hipStream_t s) {
    using G = Geo<RB, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd_kernel<RB, HH, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        attr_set = true;
    }
    chain_fwd_kernel<RB, HH, NW><<<(unsigned)((a.R + G::ROWS - 1) / G::ROWS), G::NTH, G::LDS_BYTES, s>>>(a);
    return tg::launch_status("chain_fwd_kernel");
}

template <int RB, int HH, int NW>
int launch_bwd(const ChainBwdArgs& a, hipStream_t s) {
    using G = Geo<RB, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd_kernel<RB, HH, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        attr_set = true;
    }
    chain_bwd_kernel<RB, HH, NW><<<(unsigned)((a.R + G::ROWS - 1) / G::ROWS), G::NTH, G::LDS_BYTES, s>>>(a);
    return tg::launch_status("chain_bwd_kernel");
}

}  // namespace

namespace tg {

// geometry the chain kernels cover (everything else takes the launch-per-product path of tg_layer.hip)
bool chain_shape_ok(int H, int dn, int T, int de) {
    if (H < 1 || H > 2 || dn % 4 || T % 4 || de % 4) return false;
    const int dq = dn + T, dk = dn + de + T;
    if (dq % H || (dq / H) % 4) return false;
    const int hd = dq / H, hp = (hd + 15) / 16 * 16;
    // tiles per wave: res <= 5, ctx (4 / H waves per head) <= 5, fc1 / fc2 <= 3; panel: ctx <= 8 chunks beside nothing, [y | raw] <= 15
    if (dq > 320 || dn > 192 || hp / 16 > 5 * (4 / H) || H * hp > 32 * NCH) return false;
    if ((dq + 31) / 32 + (dn + 31) / 32 > NCH || dk > 32 * 64) return false;
    const int hpb = (hd + 31) / 32 * 32;
    if (H * hpb > 320) return false;                                          // dctx: 20 tiles, 10 chunks
    return true;
}
int chain_hp(int H, int dn, int T) { const int hd = (dn + T) / H; return (hd + 15) / 16 * 16; }

int chain_fwd(const tg_layer_desc* L, const void* pWv, const void* pWr, const void* pW1, const void* pW2, int64_t packed_bytes, hipStream_t s) {
    const tg_attn_desc& at = L->attn;
    ChainFwdArgs a;
    a.R = at.m; a.H = at.heads; a.dn = at.dn; a.T = at.dt_dim; a.de = at.de;
    a.hp = chain_hp(at.heads, at.dn, at.dt_dim);
    a.agg = L->agg;
    a.pWv = pWv; a.pWr = pWr; a.pW1 = pW1; a.pW2 = pW2;
    a.packed_bytes = packed_bytes;
    const tg_layer_params& P = L->params;
    a.br = P.br; a.b1 = P.b1; a.b2 = P.b2; a.ln_g = P.ln_g; a.ln_b = P.ln_b; a.cosb = L->cosb;
    a.own = L->own; a.own_ld = L->own_ld;
    a.p_res = L->res_dropout_p; a.seed = L->res_seed;
    a.ctx = L->ctx; a.res = L->res; a.y = L->y; a.y_ld = L->y_ld ? L->y_ld : at.dn + at.dt_dim;
    a.raw = L->raw; a.raw_ld = L->raw_ld;
    a.mean = L->mean; a.rstd = L->rstd; a.f1 = L->f1; a.out = L->out;
    a.dbg = g_chain_dbg;
    const int dq = at.dn + at.dt_dim, dk = at.dn + at.de + at.dt_dim;
    const double macs = (double)dk * dq + (double)dq * dq + (double)(dq + at.dn) * at.dn + (double)at.dn * at.dn;
    ProfScope prof("gemm", 2.0 * at.m * macs, s);
    // 64-row blocks once they fill the chip; fewer rows take 16-row blocks (a workgroup streams all the weights whatever its height)
    static const bool force_rb1 = getenv("FLID_GEMM_TUNE") != nullptr && getenv("FLID_CHAIN_RB1") != nullptr;
    const bool tall = at.m >= 64 * 128 && !force_rb1;
    // tall launches: 8 waves (two per SIMD).  The products are bound by the CU's weight stream whatever the wave count, but the epilogues
    // between them (LayerNorm, dropout, split + panel stores, the global stores) are vector-ALU work that one wave per SIMD issues at
    // half the rate of two: round 5, same box, 78.1 -> 69.0 us forward and 76.0 -> 68.5 us backward at 13.6 k rows (round 3 measured the
    // 8-wave form equal: the epilogues were a smaller share of a longer launch then).  FLID_CHAIN_NW4=1 (tuning mode): 4 waves.
    static const bool nw4 = getenv("FLID_GEMM_TUNE") != nullptr && getenv("FLID_CHAIN_NW4") != nullptr;
    if (at.heads == 2) return tall ? (nw4 ? launch_fwd<4, 2, 4>(a, s) : launch_fwd<4, 2, 8>(a, s)) : launch_fwd<1, 2, 4>(a, s);
    return tall ? (nw4 ? launch_fwd<4, 1, 4>(a, s) : launch_fwd<4, 1, 8>(a, s)) : launch_fwd<1, 1, 4>(a, s);
}

int chain_hpb(int H, int dn, int T) { const int hd = (dn + T) / H; return (hd + 31) / 32 * 32; }

// the query side of a short layer (see qu_fwd_kernel / dq_bwd_kernel); shapes: chain_shape_ok plus dk <= 16 * 7 * 4 columns per head
bool qu_shape_ok(int H, int dn, int T, int de) {
    if (!chain_shape_ok(H, dn, T, de)) return false;
    const int dq = dn + T, dk = dn + de + T, hd = dq / H;
    return hd % 4 == 0 && (dq + 15) / 16 <= 5 * 4 && (dk + 15) / 16 <= 2 * 7 * 4 && H * chain_hpb(H, dn, T) <= 32 * NCH && (dn + 15) / 16 <= 3 * 4;
}
int qu_fwd(const tg_layer_desc* L, const void* pWq, const void* pWkT, hipStream_t s) {
    const tg_attn_desc& at = L->attn;
    QuFwdArgs a{at.m, at.heads, at.dn, at.dt_dim, at.de, chain_hpb(at.heads, at.dn, at.dt_dim), L->own, L->own_ld, pWq, pWkT, L->qbias, L->q, L->u};
    const int dq = at.dn + at.dt_dim, dk = at.dn + at.de + at.dt_dim;
    ProfScope prof("gemm", 2.0 * at.m * ((double)dq * at.dn + (double)dq * dk), s);
    using G = Geo<1, 4>;
    const unsigned grid = (unsigned)((at.m + 15) / 16);
    static bool attr[2] = {false, false};
    if (at.heads == 2) {
        if (!attr[1]) { TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qu_fwd_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)); attr[1] = true; }
        qu_fwd_kernel<2, 4><<<grid, G::NTH, G::LDS_BYTES, s>>>(a);
    } else {
        if (!attr[0]) { TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qu_fwd_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)); attr[0] = true; }
        qu_fwd_kernel<1, 4><<<grid, G::NTH, G::LDS_BYTES, s>>>(a);
    }
    return tg::launch_status("qu_fwd_kernel");
}
int dq_bwd(const tg_layer_desc* L, const tg_layer_bwd_desc* Bw, const void* pWk, const void* pWqT, float* dq_slab, hipStream_t s) {
    const tg_attn_desc& at = L->attn;
    DqBwdArgs a{at.m, at.heads, at.dn, at.dt_dim, at.de, chain_hp(at.heads, at.dn, at.dt_dim), Bw->du, pWk, pWqT, Bw->dq,
                Bw->d_own, Bw->d_own_ld, 1, dq_slab};      // (d_own: the residual's share is already there, chain_bwd wrote or added it)
    const int dq = at.dn + at.dt_dim, dk = at.dn + at.de + at.dt_dim;
    ProfScope prof("gemm", 2.0 * at.m * ((double)dq * dk + (Bw->d_own ? (double)dq * at.dn : 0.0)), s);
    using G = Geo<1, 4>;
    const unsigned grid = (unsigned)((at.m + 15) / 16);
    static bool attr[2] = {false, false};
    if (at.heads == 2) {
        if (!attr[1]) { TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dq_bwd_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)); attr[1] = true; }
        dq_bwd_kernel<2, 4><<<grid, G::NTH, G::LDS_BYTES, s>>>(a);
    } else {
        if (!attr[0]) { TG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dq_bwd_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)); attr[0] = true; }
        dq_bwd_kernel<1, 4><<<grid, G::NTH, G::LDS_BYTES, s>>>(a);
    }
    return tg::launch_status("dq_bwd_kernel");
}
// workgroups of a chain launch over `rows` rows (= slabs of column sums the backward chain leaves in `part`)
int64_t chain_blocks(int64_t rows) { return rows >= 64 * 128 ? (rows + 63) / 64 : (rows + 15) / 16; }

int chain_bwd(const tg_layer_desc* L, const tg_layer_bwd_desc* Bw, float* dres, float* part, const void* pW2T, const void* pW1aT,
              const void* pWrT, const void* pWvT, hipStream_t s, const void* pW1bT) {
    const tg_attn_desc& at = L->attn;
    ChainBwdArgs a;
    a.R = at.m; a.H = at.heads; a.dn = at.dn; a.T = at.dt_dim; a.de = at.de;
    a.hpb = chain_hpb(at.heads, at.dn, at.dt_dim);
    a.dout = Bw->dout; a.f1 = L->f1;
    a.pW2T = pW2T; a.pW1aT = pW1aT; a.pWrT = pWrT; a.pWvT = pWvT;
    a.pW1bT = pW1bT; a.d_raw = pW1bT ? Bw->d_raw : nullptr;
    a.ln_g = L->params.ln_g; a.cosb = L->cosb;
    a.own = L->own; a.own_ld = L->own_ld;
    a.res = L->res; a.mean = L->mean; a.rstd = L->rstd;
    a.p_res = L->res_dropout_p; a.seed = L->res_seed;
    a.df1 = Bw->df1; a.dres = dres; a.dctx = Bw->dctx; a.dagg = Bw->dagg;
    a.d_own = Bw->d_own; a.d_own_ld = Bw->d_own_ld; a.d_own_acc = Bw->d_own_accumulate;
    a.part = part;
    a.dbg = g_chain_dbg_bwd;
    const int dq = at.dn + at.dt_dim, dk = at.dn + at.de + at.dt_dim;
    const double macs = (double)at.dn * at.dn + (double)dq * at.dn + (double)dq * dq + (double)dk * dq;
    ProfScope prof("gemm", 2.0 * at.m * macs, s);
    const bool tall = at.m >= 64 * 128;
    static const bool nw4 = getenv("FLID_GEMM_TUNE") != nullptr && getenv("FLID_CHAIN_NW4") != nullptr;
    if (at.heads == 2) return tall ? (nw4 ? launch_bwd<4, 2, 4>(a, s) : launch_bwd<4, 2, 8>(a, s)) : launch_bwd<1, 2, 4>(a, s);
    return tall ? (nw4 ? launch_bwd<4, 1, 4>(a, s) : launch_bwd<4, 1, 8>(a, s)) : launch_bwd<1, 1, 4>(a, s);
}

}  // namespace tg

// diagnostic builds (-DFLID_CHAIN_STAMPS=1): the chain kernels write 16 s_memtime stamps per workgroup here (null = off)
extern "C" void tg_chain_debug_buffer(void* p) { g_chain_dbg = reinterpret_cast<unsigned long long*>(p); }
extern "C" void tg_chain_debug_buffer_bwd(void* p) { g_chain_dbg_bwd = reinterpret_cast<unsigned long long*>(p); }
