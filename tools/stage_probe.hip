// which part of a GEMM K-stage costs the time?  Builds the stage up piece by piece (no global loads).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int STRIDE = 36;
template <int MODE>   // 0: mfma only, 1: + ds_read frags, 2: + ds_write, 3: + barrier, 4: + if(live) wrapper
__global__ void __launch_bounds__(256) k(float* out, int stages, int live) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * STRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rl = lane & 31, kh = lane >> 5;
    for (int i = threadIdx.x; i < 2 * 128 * STRIDE; i += 256) lds[i] = i * 1e-5f;
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 reg[4];
    for (int j = 0; j < 4; ++j) reg[j] = make_float4(lane, j, 1.f, 2.f);
    for (int st = 0; st < stages; ++st) {
        const int cur = st & 1;
        float fa[16], fb[16];
        if (MODE >= 1) {
            const float* pa = lds + cur * 128 * STRIDE + ((wave >> 1) * 32 + rl) * STRIDE + kh * 16;
            const float* pb = lds + cur * 128 * STRIDE + (64 + (wave & 1) * 32 + rl) * STRIDE + kh * 16;
            for (int g = 0; g < 4; ++g) {
                float4 v = *reinterpret_cast<const float4*>(pa + 4 * g), w = *reinterpret_cast<const float4*>(pb + 4 * g);
                fa[4*g]=v.x; fa[4*g+1]=v.y; fa[4*g+2]=v.z; fa[4*g+3]=v.w; fb[4*g]=w.x; fb[4*g+1]=w.y; fb[4*g+2]=w.z; fb[4*g+3]=w.w;
            }
        } else {
            for (int q = 0; q < 16; ++q) { fa[q] = lane * 0.01f + q; fb[q] = 1.f + q; }
        }
        if (MODE >= 2) {
            for (int j = 0; j < 4; ++j) {
                const int idx = threadIdx.x + j * 256;
                *reinterpret_cast<float4*>(lds + (cur ^ 1) * 128 * STRIDE + (idx >> 3) * STRIDE + (idx & 7) * 4) = reg[j];
            }
        }
        if (MODE >= 4) {
            if (live) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[q], acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[q], acc, 0, 0, 0);
        }
        if (MODE >= 3) __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float* out, int blocks, int stages) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, stages, 1); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) k<MODE><<<blocks, 256>>>(out, stages, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("mode %d blocks %5d stages %4d: %8.1f us  -> %6.0f cycles/stage @2.4GHz (mfma alone = 1024)  %.1f TF\n", MODE, blocks, stages, ms * 1e3,
           ms * 1e-3 * 2.4e9 / stages / ((blocks + 255) / 256 > 1 ? (blocks / 256.0 / 1.0) : 1.0), blocks * 4.0 * stages * 16 * 4096.0 / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    for (int blocks : {256, 1024}) for (int stages : {9, 900}) {
        run<0>(out, blocks, stages); run<1>(out, blocks, stages); run<2>(out, blocks, stages); run<3>(out, blocks, stages); run<4>(out, blocks, stages);
    }
    return 0;
}
