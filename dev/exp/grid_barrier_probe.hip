// Probe: cost and correctness of an in-kernel barrier among G co-resident workgroups that exchange a
// small payload through global memory with write-through (sc1) stores and sc1 loads (no fences).
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier_probe.hip -o probe ; run: ./probe [G] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one barrier: every storing wave drains its stores, workgroup barrier, one lane arrives, one lane polls
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!good) atomicExch(timeout, 1u);
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

typedef __attribute__((ext_vector_type(4))) float f4;
// mode 0: barrier only; mode 1: 16-byte sc1 buffer stores + 16-byte sc1 buffer loads of the whole exchange
__global__ __launch_bounds__(256) void probe2(float* buf, unsigned* counter, unsigned* timeout, unsigned* errors, int G, int rounds, int n, int mode) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    const size_t bytes = sizeof(float) * 2 * (size_t)G * n;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    unsigned err = 0;
    for (int r = 0; r < rounds; ++r) {
        const int base = (r & 1) * G * n;
        if (mode == 1)
            for (int i = tid * 4; i < n; i += 1024) {
                f4 v = {(float)(r * 1000 + wg), (float)i, (float)(i + 1), (float)(r + wg)};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs, (base + wg * n + i) * 4, 0, 16);
            }
        if (!grid_barrier(counter, (unsigned)(G * (r + 1)), timeout)) return;
        if (mode == 1) {
            // each thread reads G*n/4/256 vectors; all issued before any is checked
            const int total4 = G * n / 4;
            for (int j0 = tid; j0 < total4; j0 += 256 * 8) {
                f4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * 256;
                    v[u] = j < total4 ? __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (base + j * 4) * 4, 0, 16)) : f4{0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * 256;
                    if (j < total4) { const int w = (j * 4) / n, i = (j * 4) % n; if (v[u][0] != (float)(r * 1000 + w) || v[u][1] != (float)i || v[u][3] != (float)(r + w)) ++err; }
                }
            }
        }
    }
    if (err) atomicAdd(errors, err);
}

__global__ __launch_bounds__(256) void probe(float* buf, unsigned* counter, unsigned* timeout, unsigned* errors, int G, int rounds, int n) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    float* bufs[2] = {buf, buf + (size_t)G * n};
    unsigned err = 0;
    for (int r = 0; r < rounds; ++r) {
        float* wr = bufs[r & 1];
        // phase A: every workgroup publishes n floats that encode (round, wg, index)
        for (int i = tid; i < n; i += 256) st_sc1(wr + (size_t)wg * n + i, (float)(r * 1000 + wg) + i * 1e-3f);
        if (!grid_barrier(counter, (unsigned)(G * (r + 1)), timeout)) return;
        // phase B: every workgroup reads what ALL others wrote this round
        for (int w = 0; w < G; ++w)
            for (int i = tid; i < n; i += 256) {
                const float v = ld_sc1(wr + (size_t)w * n + i);
                if (v != (float)(r * 1000 + w) + i * 1e-3f) ++err;
            }
    }
    if (err) atomicAdd(errors, err);
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 64, rounds = argc > 2 ? atoi(argv[2]) : 200, n = 512;
    float* buf; unsigned *counter, *timeout, *errors;
    hipMalloc(&buf, sizeof(float) * 2 * G * n); hipMalloc(&counter, 4); hipMalloc(&timeout, 4); hipMalloc(&errors, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 4); hipMemset(timeout, 0, 4); hipMemset(errors, 0, 4); hipMemset(buf, 0, sizeof(float) * 2 * G * n);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe2, dim3(G), dim3(256), 0, 0, buf, counter, timeout, errors, G, rounds, n, mode);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned t, e; hipMemcpy(&t, timeout, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost);
        printf("probe2 mode=%d G=%d: %.2f us per round, timeout=%u errors=%u\n", mode, G, ms * 1e3 / rounds, t, e);
    }
    for (int rep = 0; rep < 1; ++rep) {
        hipMemset(counter, 0, 4); hipMemset(timeout, 0, 4); hipMemset(errors, 0, 4); hipMemset(buf, 0, sizeof(float) * 2 * G * n);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, buf, counter, timeout, errors, G, rounds, n);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned t, e; hipMemcpy(&t, timeout, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost);
        printf("G=%d rounds=%d: %.2f us per (publish %d floats + barrier + read all), timeout=%u errors=%u\n", G, rounds, ms * 1e3 / rounds, n, t, e);
    }
    return 0;
}
