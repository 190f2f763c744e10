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

}  // namespace

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
