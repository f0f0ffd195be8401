// Probe: gfx950 v_permlane16_swap / v_permlane32_swap as the cross-row steps of a 64-lane reduction (no ds_bpermute).
// build: hipcc --offload-arch=gfx950 -O3 dev/exp/permlane_probe.hip -o gpurun_out/permlane_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ float swap_sum16(float v) { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
__device__ float swap_sum32(float v) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
__global__ void k(const float* x, float* y16, float* y32, float* yall) {
    const float v = x[threadIdx.x];
    y16[threadIdx.x] = swap_sum16(v);
    y32[threadIdx.x] = swap_sum32(v);
    float s = v;
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
    yall[threadIdx.x] = swap_sum32(swap_sum16(s));
}
int main() {
    float hx[64], h16[64], h32[64], hall[64];
    for (int i = 0; i < 64; ++i) hx[i] = (float)(1 << (i % 20)) + i * 0.5f;
    float *x, *a, *b, *c;
    hipMalloc(&x, 256); hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 256);
    hipMemcpy(x, hx, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, a, b, c);
    hipMemcpy(h16, a, 256, hipMemcpyDeviceToHost); hipMemcpy(h32, b, 256, hipMemcpyDeviceToHost); hipMemcpy(hall, c, 256, hipMemcpyDeviceToHost);
    int bad = 0; double tot = 0; for (int i = 0; i < 64; ++i) tot += hx[i];
    for (int i = 0; i < 64; ++i) {
        if (h16[i] != hx[i] + hx[i ^ 16]) ++bad;
        if (h32[i] != hx[i] + hx[i ^ 32]) ++bad;
        if (hall[i] != (float)tot) ++bad;
    }
    printf("permlane probe: %d mismatches (sum %.1f, lane0 all=%.1f)\n", bad, tot, hall[0]);
    return bad != 0;
}
