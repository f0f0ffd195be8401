// Probe for the round-3 kernel family ("row block resident in LDS, weights streamed from L2 straight into MFMA fragments"):
//   y[M][N] = x[M][512] . W[N][512]^T (+ bias), bf16 in, bf16 out, f32 accumulate.
// One workgroup of 8 wavefronts per 64-row block: the block's x rows sit in LDS (64 KiB, XOR-swizzled 16-byte chunks), every wavefront
// owns N / 8 output columns and walks them in chunks of 64 (4 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16).  The weights are
// stored FRAGMENT-MAJOR -- [column chunk of 64][k step of 32][16-column tile][lane][8 bf16] -- so that a wavefront's whole weight stream
// is one linear run of 1-KiB wave loads, each landing directly in the MFMA operand layout: no LDS staging of weights, no workgroup
// barrier in the K loop, every weight byte read once per workgroup.
// What the probe answers: the time per launch against the weight-stream bound (N * 1 KiB per workgroup at the ~120 GB/s a CU takes
// from a shared L2-resident region, profiles/r02_l2_stream_probe.txt) and against the MFMA bound.
// Build: hipcc --offload-arch=gfx950 -O3 dw_gemm_probe.hip -o dw_gemm_probe ; run: ./dw_gemm_probe [M]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

static uint16_t host_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float host_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    const f2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
}

constexpr int K = 512, KB = K / 32, ROWS = 64, PF = 3;      // (the runtime-loop instance uses 4 slots: 16 % 4 == 0)

// LDS image of the row block: row r, 16-byte chunk c (64 per row) at chunk (c ^ (r & 15)) of the row: the 16 lanes of a ds_read_b128
// service group then hit 16 different bank quads
__device__ __forceinline__ int lds_off(int r, int c) { return r * (K * 2) + ((c ^ (r & 15)) << 4); }

template <int MODE, int CH>      // MODE 0: full; 1: no epilogue stores; 2: no MFMA (loads only).  CH > 0: column chunks per wavefront, unrolled
__global__ __launch_bounds__(512) void dw_gemm_kernel(const bf16_t* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                       bf16_t* __restrict__ y, int M, int N, int wrap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * ROWS;
    const int chunks = CH > 0 ? CH : N / 64 / 8;                    // column chunks per wavefront
    // the wavefront's weight stream: chunk c0 .. c0 + chunks - 1, KB steps each, 4 fragments per step, linear in memory
    const u32x4* ws = wp + ((size_t)(wave * chunks) * KB * 4) * 64 + lane;
    constexpr int NSLOT = CH > 0 ? PF : 4;
    bf16x8 wb[NSLOT][4];
#pragma unroll
    for (int p = 0; p < NSLOT; ++p)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wb[p][nt] = __builtin_bit_cast(bf16x8, ws[(size_t)(p * 4 + nt) * 64]);
    // ---- stage the row block
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + i * 512, r = c >> 6, ch = c & 63;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (m0 + r < M) v = *reinterpret_cast<const u32x4*>(x + (size_t)(m0 + r) * K + ch * 8);
        *reinterpret_cast<u32x4*>(smem + lds_off(r, ch)) = v;
    }
    __syncthreads();

#pragma unroll
    for (int c = 0; c < chunks; ++c) {
        asm volatile("" ::: "memory");                // the LDS image is re-read per chunk (hoisted out of the loop it would be 256 registers)
        f32x4 acc[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            bf16x8 xa[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xa[mt] = *reinterpret_cast<const bf16x8*>(smem + lds_off(mt * 16 + l15, kb * 4 + g));
            const int slot = (CH > 0 ? c * KB + kb : kb) % (CH > 0 ? PF : 4);
            if (MODE != 2) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[slot][nt], xa[mt], acc[nt][mt], 0, 0, 0);
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt][0][0] += __builtin_bit_cast(float, (int)wb[slot][nt][0] | ((int)xa[nt][1] << 16));
            }
            // (the last PF steps of the last wavefront read past the matrix: the allocation is padded)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wb[slot][nt] = __builtin_bit_cast(bf16x8, ws[(size_t)((NSLOT * 4 + nt) * 64)]);
            ws += 4 * 64;
            __builtin_amdgcn_sched_barrier(0);        // the prefetch stays PF steps ahead of its use (the scheduler would sink the loads next to it)
        }
        if (wrap) ws -= KB * 4 * 64;                  // re-stream the same 64 KiB: the weights then stay in L2 whatever N is
        // ---- epilogue: lane holds y[m = mt*16 + l15][n = n0 + nt*16 + g*4 + 0..3]
        const int n0 = (wave * chunks + c) * 64;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + nt * 16 + g * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m0 + mt * 16 + l15;
                uint2 o;
                o.x = pack_bf16(acc[nt][mt][0] + bv.x, acc[nt][mt][1] + bv.y);
                o.y = pack_bf16(acc[nt][mt][2] + bv.z, acc[nt][mt][3] + bv.w);
                if (MODE == 0) { if (m < M) *reinterpret_cast<uint2*>(y + (size_t)m * N + n0 + nt * 16 + g * 4) = o; }
                else if (o.x == 0x12345678u && o.y == 0x9abcdef0u) y[0] = 1;
            }
        }
    }
}

static void pack_w(const std::vector<uint16_t>& w, int N, std::vector<uint16_t>& out) {
    out.resize((size_t)N * K);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const int ch = n / 64, nt = (n % 64) / 16, r = n % 16, kb = k / 32, gq = (k % 32) / 8, e = k % 8;
            const size_t idx = ((((size_t)ch * KB + kb) * 4 + nt) * 64 + gq * 16 + r) * 8 + e;
            out[idx] = w[(size_t)n * K + k];
        }
}

template <int MODE, int CH>
static float time_kernel(const bf16_t* x, const u32x4* wp, const float* bias, bf16_t* y, int M, int N, int iters, int wrap = 0) {
    const int grid = (M + ROWS - 1) / ROWS;
    hipFuncSetAttribute((const void*)&dw_gemm_kernel<MODE, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, ROWS * K * 2);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dw_gemm_kernel<MODE, CH>), dim3(grid), dim3(512), ROWS * K * 2, 0, x, wp, bias, y, M, N, wrap);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((dw_gemm_kernel<MODE, CH>), dim3(grid), dim3(512), ROWS * K * 2, 0, x, wp, bias, y, M, N, wrap);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 11520;
    const int Ns[] = {512, 1536, 2048, 7168};
    for (int N : Ns) {
        std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hwp;
        std::vector<float> hb(N);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hx) v = host_bf16(rnd());
        for (auto& v : hw) v = host_bf16(rnd() * 0.1f);
        for (auto& v : hb) v = rnd();
        pack_w(hw, N, hwp);
        bf16_t *dx, *dy; u32x4* dw; float* db;
        hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hwp.size() * 2 + 65536); hipMalloc(&dy, (size_t)M * N * 2); hipMalloc(&db, N * 4);
        hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw, hwp.data(), hwp.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
        hipMemset(dy, 0, (size_t)M * N * 2);
        const int ch = N / 512;
        auto tk = [&](int mode) {
            if (ch == 1) return mode == 0 ? time_kernel<0, 1>(dx, dw, db, dy, M, N, 50) : mode == 1 ? time_kernel<1, 1>(dx, dw, db, dy, M, N, 50) : time_kernel<2, 1>(dx, dw, db, dy, M, N, 50);
            if (ch == 3) return mode == 0 ? time_kernel<0, 3>(dx, dw, db, dy, M, N, 50) : mode == 1 ? time_kernel<1, 3>(dx, dw, db, dy, M, N, 50) : time_kernel<2, 3>(dx, dw, db, dy, M, N, 50);
            if (ch == 4) return mode == 0 ? time_kernel<0, 4>(dx, dw, db, dy, M, N, 50) : mode == 1 ? time_kernel<1, 4>(dx, dw, db, dy, M, N, 50) : time_kernel<2, 4>(dx, dw, db, dy, M, N, 50);
            return mode == 0 ? time_kernel<0, 0>(dx, dw, db, dy, M, N, 50) : mode == 1 ? time_kernel<1, 0>(dx, dw, db, dy, M, N, 50) : time_kernel<2, 0>(dx, dw, db, dy, M, N, 50);
        };
        const float t0 = tk(0);
        std::vector<uint16_t> hy((size_t)M * N);
        hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
        double maxerr = 0.0;
        for (int t = 0; t < 4000; ++t) {
            const int m = (int)(((uint64_t)t * 2654435761u) % M), n = (int)(((uint64_t)t * 40503u + 17) % N);
            double ref = hb[n];
            for (int k = 0; k < K; ++k) ref += (double)host_f32(hx[(size_t)m * K + k]) * host_f32(hw[(size_t)n * K + k]);
            const double err = fabs(ref - host_f32(hy[(size_t)m * N + n])) / (fabs(ref) + 1.0);
            if (err > maxerr) maxerr = err;
        }
        const float t1 = tk(1);
        const float t1w = ch == 1 ? time_kernel<1, 1>(dx, dw, db, dy, M, N, 50, 1) : ch == 3 ? time_kernel<1, 3>(dx, dw, db, dy, M, N, 50, 1) : ch == 4 ? time_kernel<1, 4>(dx, dw, db, dy, M, N, 50, 1) : time_kernel<1, 0>(dx, dw, db, dy, M, N, 50, 1);
        const float t2w = ch == 1 ? time_kernel<2, 1>(dx, dw, db, dy, M, N, 50, 1) : ch == 3 ? time_kernel<2, 3>(dx, dw, db, dy, M, N, 50, 1) : ch == 4 ? time_kernel<2, 4>(dx, dw, db, dy, M, N, 50, 1) : time_kernel<2, 0>(dx, dw, db, dy, M, N, 50, 1);
        const float t2 = tk(2);
        const double flop = 2.0 * M * N * K;
        const int wgs = (M + ROWS - 1) / ROWS;
        printf("M %5d N %4d: full %7.2f us (%6.1f TFLOP/s, weight stream %5.1f GB/s per workgroup, %d workgroups)  no-store %7.2f  loads-only %7.2f | same 512 KiB re-streamed: no-store %7.2f  loads-only %7.2f | max rel err %.2e\n",
               M, N, t0, flop / t0 * 1e-6, (double)N * K * 2 / t0 * 1e-3, wgs, t1, t2, t1w, t2w, maxerr);
        hipFree(dx); hipFree(dw); hipFree(dy); hipFree(db);
    }
    return 0;
}
