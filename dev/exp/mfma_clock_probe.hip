// What the MFMA pipes sustain on this card, and at what clock: a kernel of nothing but v_mfma_f32_16x16x32_bf16 on register operands
// (16 independent accumulator tiles per wavefront, no memory traffic), over a sweep of workgroups (one per CU at most) and wavefronts
// per SIMD, run for a few milliseconds.  Per configuration: TFLOP/s from the launch's wall time (events), the shader clock from
// s_memtime ticks of workgroup 0 against that wall time, and the MFMA issue rate (cycles per MFMA per SIMD; 16 = the pipe never waits).
// The dense-peak figure of the guide (2.5 PFLOP/s) is 256 CUs x 4 SIMDs x 1 024 FLOP per cycle x 2.4 GHz: whatever clock the card
// holds under this load scales it.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_clock_probe.hip -o mfma_clock_probe ; run: ./mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void mfma_only(int iters, float* sink, unsigned long long* ticks) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3c00 + lane + e); b[e] = (short)(0x3c00 + 3 * lane + e); }
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.678f) sink[0] = s;                          // keeps the loop alive
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    float* sink;
    unsigned long long* ticks;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&ticks, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int wgs_list[] = {1, 64, 128, 180, 256};
    const int waves_list[] = {4, 8};
    printf("workgroups  waves/WG  ms       TFLOP/s   clock GHz   cycles per MFMA per SIMD\n");
    for (int wgs : wgs_list)
        for (int waves : waves_list) {
            const int iters = 60000 / (waves / 4);             // ~ the same wall time per configuration (a few ms)
            hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(64 * waves), 0, 0, iters / 10, sink, ticks);      // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(64 * waves), 0, 0, iters, sink, ticks);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long tk;
            CHECK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
            const double mfma = (double)iters * 16 * waves * wgs;                        // MFMA instructions in the launch
            const double tflops = mfma * 16384.0 / (ms * 1e-3) / 1e12;
            const double ghz = (double)tk / (ms * 1e-3) / 1e9;                           // ticks of workgroup 0 ~ the launch (one round of workgroups)
            const double cyc = (double)tk / ((double)iters * 16 * (waves / 4));          // per SIMD: waves / 4 wavefronts share one pipe
            printf("%10d  %8d  %7.3f  %8.1f  %9.3f  %10.2f\n", wgs, waves, ms, tflops, ghz, cyc);
        }
    return 0;
}
