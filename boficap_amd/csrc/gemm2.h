// Shared by the LDS-DMA GEMM kernels (gemm_glds.hip: one tile per workgroup; gemm_pers.hip: persistent workgroups, loader wavefronts):
// the MFMA wrappers, the kernel parameter block and the counted-wait helper.
#pragma once
#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

template <typename T> struct GMma;
template <> struct GMma<bf16_t> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct GMma<float> {
    typedef float4 Frag;     // lane quarter q holds k = 16g + 4q + s for step s, in A and in B alike
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
        return c;
    }
};

struct Gemm2Params {
    const void* x; int ldx;
    const void* w;
    const float* bias;
    const float* residual; int ldr;
    void* y; int ldy; int y_is_f32;
    int M, N, K;
    int relu;
    const int* row_len; int rows_per_group;
    const int* skip_if_ge; int skip_threshold;
    // folded pre-norm LayerNorm of the consumer: x is the RAW residual stream, w = W * gain, bias = c,
    // y = rstd[m] * (acc - mean[m] * colsum[n]) + c[n]; mean/rstd come from per-32-column partial
    // (sum, sum of squares) pairs written by the producer's epilogue
    const float* ln_stats; const float* ln_colsum; int ln_groups;
    float* stats_out;         // this GEMM is a producer: partial (sum, sumsq) of its OUTPUT rows, [M][N/32][2]
    void* y2; int ldy2;       // optional second copy of the output in the compute dtype
    int splitk;               // > 1: blockIdx.y walks K slices; slice s writes its partial tile to y + s*M*ldy (f32),
                              // bias / residual are added by slice 0 only, the consumer sums the slabs
    int vec_ok;               // N, ldy, ldr multiples of 4 and y/bias/residual 16-byte aligned
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed;     // training dropout on act(..) before the residual (0: off)
    const uint64_t* drop_step;
    float mask_scale;         // != 0: residual is a mask (see LinearArgs)
    int dbg;                  // developer ablation (BOFI_GEMM_DBG): 1 = no loads, 2 = no MFMA/ds_read
    // row list (FEAT bit 6): the GEMM runs over rows row_idx[0 .. *m_dev) of x and writes the same rows of y / y2 / the statistics
    // (p.M is the capacity the grid was sized for; tiles past *m_dev return at once)
    const int* row_idx; const int* m_dev;
    int row_bands;            // XCD tile order: 8 (bands of A rows per XCD), 4, 2 or 1 (bands of weight columns per XCD)
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

extern int g_env_generation;
int launch_gemm_pers(const Gemm2Params& p, int feat, hipStream_t st);      // gemm_pers.hip: -1 = shape / feature set not covered

}  // namespace bofi
