// Bare MFMA issue rate: NW waves per workgroup, one workgroup per CU, each wave N x v_mfma_f32_32x32x16_bf16 on NACC accumulators in turn.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(512) k(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const uint4 u = make_uint4(threadIdx.x, 1, 2, 3);
    bf16x8 x = __builtin_bit_cast(bf16x8, u), y = __builtin_bit_cast(bf16x8, make_uint4(3, 2, 1, threadIdx.x));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int nw, int iters) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC><<<256, 64 * nw>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<NACC><<<256, 64 * nw>>>(out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC;
    printf("waves/WG %d  accumulators %d: %.1f ns per MFMA per wave (wall), %.1f s_memtime ticks per MFMA, %.0f TFLOP/s chip\n", nw, NACC, ms * 1e6 / n,
           (double)c / n, 256.0 * nw * n * 32768 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int nw : {4, 5, 8}) { run<1>(nw, 20000); run<7>(nw, 3000); }
    return 0;
}
