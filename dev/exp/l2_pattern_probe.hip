// Probe: does the ADDRESS PATTERN of an LDS-DMA piece (global_load_lds_dwordx4, 64 lanes x 16 B) change the rate at which a CU
// pulls L2-resident bytes?  The GEMM kernels load a 64-deep K slab as pieces of 8 rows x 128 B (row pitch = K * 2 bytes) with the
// 16-byte chunks of each row XOR-permuted among the row's 8 lanes (the bank-conflict swizzle lives on the source side).
// Patterns: 0 = 1 KiB contiguous, lane-linear (the reference); 1 = contiguous, chunks XOR-permuted inside each 128-B line;
// 2 = 8 rows at 1 KiB pitch, chunks in lane order; 3 = 8 rows at 1 KiB pitch, XOR-permuted (the GEMM's pattern).
// One workgroup per CU, each cycling over its own 64 KiB (L2-resident after the first pass).
// Build: hipcc --offload-arch=gfx950 -O3 l2_pattern_probe.hip -o l2_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH>
__global__ __launch_bounds__(512) void pat_kernel(const unsigned char* src, unsigned* sink, int iters, int pattern) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * (64 << 10);
    const int row = lane >> 3, chunk = (pattern & 1) ? ((lane & 7) ^ (lane >> 3)) : (lane & 7);
    // piece q of this wave: contiguous: bytes q*1024 ..; strided: rows (q % 8) * 8 .. + 7 of a 64 x 1 KiB image, column slab q / 8
    int q = wave;
    unsigned char* dst = smem + (wave * DEPTH) * 1024;
    auto addr = [&](int qq) {
        qq &= 63;
        if (pattern & 2) return base + (size_t)((qq & 7) * 8 + row) * 1024 + (qq >> 3) * 128 + chunk * 16;
        return base + (size_t)qq * 1024 + row * 128 + chunk * 16;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr(q), (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, 0, 0);
        q += nw;
    }
    for (int i = DEPTH; i < iters; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            wait_vmcnt<DEPTH - 1>();
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr(q), (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, 0, 0);
            q += nw;
        }
    }
    wait_vmcnt<0>();
    if (*reinterpret_cast<unsigned*>(smem + threadIdx.x * 4) == 0x12345678u) sink[0] = 1;
}

int main() {
    unsigned char* src; unsigned* sink;
    if (hipMalloc(&src, 256ull << 20) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(src, 1, 256ull << 20); hipMemset(sink, 0, 64);
    for (int waves : {4, 8}) for (int pattern = 0; pattern < 4; ++pattern) {
        const int iters = 4096 / waves * 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((pat_kernel<8>), dim3(256), dim3(64 * waves), 0, 0, src, sink, iters, pattern);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((pat_kernel<8>), dim3(256), dim3(64 * waves), 0, 0, src, sink, iters, pattern);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)iters * waves * 1024;
        printf("waves %d depth 8 pattern %d: %6.1f GB/s per CU, %5.2f TB/s total\n", waves, pattern, bytes / (ms * 1e-3) / 1e9, bytes * 256 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
