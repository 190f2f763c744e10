// micro-benchmark: fp32 MFMA chain rate and shader clock (s_memtime vs s_memrealtime @100 MHz)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES_CHAINS>
__global__ void k(float* out, unsigned long long* clk, int iters) {
    f32x16 acc[WAVES_CHAINS];
    for (int c = 0; c < WAVES_CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < WAVES_CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < WAVES_CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 4096 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 1024}) for (int threads : {64, 256, 512}) for (int iters : {144, 1000, 20000}) {
        k<1><<<blocks, threads>>>(out, clk, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) k<1><<<blocks, threads>>>(out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double waves = blocks * (threads / 64.0);
        double flops = waves * iters * 4096.0;
        printf("blocks %4d threads %3d iters %5d: %8.1f us  %6.1f TFLOP/s  shader clk %.2f GHz (cyc %llu, real %llu)  cycles/mfma/wave %.1f\n", blocks, threads,
               iters, ms * 1e3, flops / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0) , h[0], h[1], (double)h[0] / iters);
    }
    return 0;
}
