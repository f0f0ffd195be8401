// Probe: how many bytes per second ONE compute unit can pull from L2 (a) with LDS-DMA (global_load_lds_dwordx4, the GEMM's loader)
// and (b) with plain 16-byte global loads into registers, as a function of the wavefronts per workgroup and of the 1-KiB wave
// loads each keeps in flight.  One workgroup per CU (grid = 256), every workgroup cycles over its own 64 KiB region (16 MiB in
// all: L2-resident after the warm-up pass), or over a region SHARED by all workgroups (mode bit 2: the broadcast case).
// Build: hipcc --offload-arch=gfx950 -O3 l2_stream_probe.hip -o l2_stream_probe ; run: ./l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DEPTH wave loads (1 KiB each) in flight per wavefront; iters = wave loads per wavefront in total
template <int DEPTH, bool LDSDMA>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned char* src, unsigned* sink, int iters, size_t region, int shared_region) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = src + (shared_region ? 0 : (size_t)blockIdx.x * region);
    const size_t mask = region - 1;
    size_t off = (size_t)wave * 1024 + lane * 16;
    const size_t step = (size_t)nw * 1024;
    u32x4 acc = u32x4{0u, 0u, 0u, 0u};
    if constexpr (LDSDMA) {
        unsigned char* dst = smem + (wave * DEPTH) * 1024;      // each in-flight load has its own KiB of LDS (8 waves x 8 = 64 KiB)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off & mask)),
                                             (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, 0, 0);
            off += step;
        }
        for (int i = DEPTH; i < iters; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                wait_vmcnt<DEPTH - 1>();
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off & mask)),
                                                 (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, 0, 0);
                off += step;
            }
        }
        wait_vmcnt<0>();
        acc[0] = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    } else {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { v[d] = *reinterpret_cast<const u32x4*>(base + (off & mask)); off += step; }
        for (int i = DEPTH; i < iters; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                acc ^= v[d];
                v[d] = *reinterpret_cast<const u32x4*>(base + (off & mask));
                off += step;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;      // keep the loads alive
}

template <int DEPTH, bool LDSDMA>
static void run(const unsigned char* src, unsigned* sink, int waves, size_t region, int shared_region, int grid) {
    const int iters = 4096 / waves * 4;                       // 16 MiB per workgroup in all, whatever the wave count
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<DEPTH, LDSDMA>), dim3(grid), dim3(64 * waves), 0, 0, src, sink, iters, region, shared_region);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<DEPTH, LDSDMA>), dim3(grid), dim3(64 * waves), 0, 0, src, sink, iters, region, shared_region);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_wg = (double)iters * waves * 1024;
    printf("%s grid %3d waves %d depth %2d region %4zu KiB%s: %7.1f GB/s per CU, %6.2f TB/s total\n", LDSDMA ? "lds-dma" : "vgpr   ", grid, waves, DEPTH,
           region >> 10, shared_region ? " shared" : "", bytes_per_wg / (ms * 1e-3) / 1e9, bytes_per_wg * grid / (ms * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    const size_t total = 256ull << 20;
    unsigned char* src; unsigned* sink;
    if (hipMalloc(&src, total) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, total); hipMemset(sink, 0, 64);
    for (int grid : {256, 64}) {
        for (int waves : {1, 2, 4, 8}) {
            run<1, true>(src, sink, waves, 64 << 10, 0, grid);
            run<2, true>(src, sink, waves, 64 << 10, 0, grid);
            run<4, true>(src, sink, waves, 64 << 10, 0, grid);
            run<8, true>(src, sink, waves, 64 << 10, 0, grid);
            run<4, false>(src, sink, waves, 64 << 10, 0, grid);
            run<8, false>(src, sink, waves, 64 << 10, 0, grid);
            run<16, false>(src, sink, waves, 64 << 10, 0, grid);
        }
    }
    // shared region (every workgroup reads the same 64 KiB / 2 MiB): the weight-panel case; and a 1 MiB private region (Infinity Cache)
    for (int waves : {4, 8}) {
        run<8, true>(src, sink, waves, 64 << 10, 1, 256);
        run<8, true>(src, sink, waves, 2 << 20, 1, 256);
        run<8, true>(src, sink, waves, 1 << 20, 0, 256);
        run<16, false>(src, sink, waves, 1 << 20, 0, 256);
    }
    return 0;
}
