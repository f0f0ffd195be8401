// Does gfx950's v_cvt_pk_bf16_f32 agree with the library's integer round-to-nearest-even (NaN kept NaN with the quiet bit set) on EVERY
// float32 bit pattern?  Walks all 2^32 inputs; prints the number of mismatches per class and the first few.
// Build: hipcc --offload-arch=gfx950 -O3 cvt_bf16_check.hip -o cvt_bf16_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ inline uint16_t sw(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__global__ void k(unsigned long long* counts, uint32_t* first) {
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t it = 0; it < (uint32_t)((1ull << 32) / stride); ++it, u += stride) {
        const float f = __uint_as_float(u);
        const uint16_t a = sw(f);
        const uint16_t b = __builtin_bit_cast(uint16_t, (__bf16)f);
        if (a != b) {
            const uint32_t e = (u >> 23) & 0xff, m = u & 0x7fffff;
            const int cls = e == 0xff ? (m ? 0 : 1) : (e == 0 ? 2 : 3);      // nan, inf, denormal/zero, normal
            const unsigned long long n = atomicAdd(&counts[cls], 1ull);
            if (n < 4) { first[(cls * 4 + n) * 3] = u; first[(cls * 4 + n) * 3 + 1] = a; first[(cls * 4 + n) * 3 + 2] = b; }
        }
    }
}
int main() {
    unsigned long long* c; uint32_t* f;
    hipMalloc(&c, 32); hipMalloc(&f, 4 * 4 * 3 * 4); hipMemset(c, 0, 32); hipMemset(f, 0, 192);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, c, f);
    unsigned long long hc[4]; uint32_t hf[48];
    hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost); hipMemcpy(hf, f, 192, hipMemcpyDeviceToHost);
    const char* names[4] = {"nan", "inf", "denormal/zero", "normal"};
    for (int i = 0; i < 4; ++i) {
        printf("%-14s mismatches %llu", names[i], hc[i]);
        for (int j = 0; j < 4 && j < (int)hc[i]; ++j) printf("  [in %08x sw %04x hw %04x]", hf[(i * 4 + j) * 3], hf[(i * 4 + j) * 3 + 1], hf[(i * 4 + j) * 3 + 2]);
        printf("\n");
    }
    return 0;
}
