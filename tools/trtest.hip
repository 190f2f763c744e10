#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, q = i >> 2, p = i & 3, g = l >> 4;
    // block: rows 4*... test: group g reads rows q of block g, cols 4p
    const short* addr = lds + (g * 4 + q) * 64 + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short h[4096], *d, *o, r[256];
    for (int i = 0; i < 4096; ++i) h[i] = i;   // value = row*64 + col
    hipMalloc(&d, 8192); hipMalloc(&o, 512); hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", r[l*4+e] / 64, r[l*4+e] % 64); printf("\n"); }
}
