// y = act(x . w^T + bias) [+ residual]  --  the nn.Linear layers of the bound+fill path
// (reference captioning/models/TransformerModel.py:1454-1456,1467 attention projections,
//  :1477-1478 FFN, :1642-1645 att_embed, :1316-1319 generator.proj).
//
// MFMA kernel for gfx950.  One workgroup = 4 wavefronts (2x2) computes a BM x BN output tile;
// every wave owns (BM/2) x (BN/2) as 16x16 MFMA tiles accumulated in float32.
//   bf16 operands : v_mfma_f32_16x16x32_bf16, K tile 64
//   f32 operands  : v_mfma_f32_16x16x4_f32 (exact f32 fma chain), K tile 32
// A ([M,K] activations) and B ([N,K] weights, K contiguous = the nn.Linear layout, so the MFMA B
// fragment is a plain 16-byte row read) are staged global -> registers -> LDS, double buffered:
// the loads of tile t+1 are in flight while tile t is multiplied, one barrier per K tile.
// LDS rows are 128 B of data + 16 B pad (row stride 144 B spreads the 16 rows of a fragment read
// over all 64 banks).  The A loader can convert float32 -> bf16 on the fly and can apply the
// BoFiCap LayerNorm to the row (fused pre-norm of SublayerConnection, TransformerModel.py:1361-1363).
// Epilogue: + bias, ReLU, padded-row zeroing, + float32 residual, store float32 or compute dtype.
#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KGROUP = 32;                 // K elements one fragment covers
    typedef bf16x8 Frag;
    static __device__ __forceinline__ Frag load(const bf16_t* row, int g, int lane) {
        return *reinterpret_cast<const bf16x8*>(row + g * 32 + (lane >> 4) * 8);
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // one 16-byte fragment feeds four 16x16x4 MFMAs: lane quarter q holds k = 16g + 4q + s for
    // step s; A and B use the same k assignment, so the sum over k is complete and exact f32.
    static constexpr int KGROUP = 16;
    typedef float4 Frag;
    static __device__ __forceinline__ Frag load(const float* row, int g, int lane) {
        return *reinterpret_cast<const float4*>(row + g * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
        return c;
    }
};

constexpr int ROWB = 144;   // LDS bytes per tile row: 128 data + 16 pad

struct GemmParams {
    const void* x; int ldx;
    const void* w;
    const float* bias;
    const float* residual; int ldr;
    void* y; int ldy; int y_is_f32;
    int M, N, K;
    int relu;
    const int* row_len; int rows_per_group;
    const float* ln_gain; const float* ln_bias;
    const int* skip_if_ge; int skip_threshold;
};

// 16 bytes of compute-dtype operand, as loaded for one staging slot
template <typename T, bool AF32> struct StageRegs;
template <> struct StageRegs<bf16_t, false> { u32x4 v; };
template <> struct StageRegs<bf16_t, true> { float4 lo, hi; };
template <> struct StageRegs<float, false> { float4 v; };
template <> struct StageRegs<float, true> { float4 v; };

template <typename T, int BM, int BN, bool AF32, bool LN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    static_assert(!LN || AF32, "fused LayerNorm reads float32 rows");
    constexpr int EPC = 16 / sizeof(T);              // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;                      // 64 (bf16) or 32 (f32): 128 bytes per row
    constexpr int TM = BM / 32, TN = BN / 32;        // 16x16 tiles per wave
    constexpr int PA = BM / 32, PB = BN / 32;        // staging passes (32 rows per pass)
    constexpr int STAGE = (BM + BN) * ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + (LN ? BM * 8 : 0)];

    if (p.skip_if_ge && *p.skip_if_ge >= p.skip_threshold) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int srow = tid >> 3, schunk = tid & 7;     // staging slot: row within pass, 16-B chunk

    float* ln_mean = reinterpret_cast<float*>(smem + 2 * STAGE);
    float* ln_rden = ln_mean + BM;
    if constexpr (LN) {
        // row statistics of this block's BM rows (wave per row, K <= 2048)
        const float* xf = static_cast<const float*>(p.x);
        for (int r = wave; r < BM; r += 4) {
            const int m = m0 + r;
            float s = 0.f, q = 0.f, mean = 0.f;
            if (m < p.M) {
                const float* row = xf + (size_t)m * p.ldx;
                for (int k = lane * 4; k < p.K; k += 256) {
                    const float4 t = *reinterpret_cast<const float4*>(row + k);
                    s += (t.x + t.y) + (t.z + t.w);
                }
                mean = wave_sum(s) / (float)p.K;
                for (int k = lane * 4; k < p.K; k += 256) {
                    const float4 t = *reinterpret_cast<const float4*>(row + k);
                    const float a = t.x - mean, b = t.y - mean, c = t.z - mean, d = t.w - mean;
                    q += (a * a + b * b) + (c * c + d * d);
                }
                q = wave_sum(q);
            }
            if (lane == 0) {
                ln_mean[r] = mean;
                ln_rden[r] = sqrtf(q / (float)(p.K - 1)) + 1e-6f;
            }
        }
        __syncthreads();
    }

    StageRegs<T, AF32> ra[PA];
    StageRegs<T, false> rb[PB];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK + schunk * EPC;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m0 + i * 32 + srow;
            if constexpr (AF32 && sizeof(T) == 2) {
                if (m < p.M) {
                    const float* src = static_cast<const float*>(p.x) + (size_t)m * p.ldx + k0;
                    ra[i].lo = *reinterpret_cast<const float4*>(src);
                    ra[i].hi = *reinterpret_cast<const float4*>(src + 4);
                } else {
                    ra[i].lo = make_float4(0.f, 0.f, 0.f, 0.f); ra[i].hi = ra[i].lo;
                }
            } else if constexpr (sizeof(T) == 2) {
                if (m < p.M) ra[i].v = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(p.x) + (size_t)m * p.ldx + k0);
                else ra[i].v = u32x4{0u, 0u, 0u, 0u};
            } else {
                if (m < p.M) ra[i].v = *reinterpret_cast<const float4*>(static_cast<const float*>(p.x) + (size_t)m * p.ldx + k0);
                else ra[i].v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int n = n0 + i * 32 + srow;
            if constexpr (sizeof(T) == 2) {
                if (n < p.N) rb[i].v = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(p.w) + (size_t)n * p.K + k0);
                else rb[i].v = u32x4{0u, 0u, 0u, 0u};
            } else {
                if (n < p.N) rb[i].v = *reinterpret_cast<const float4*>(static_cast<const float*>(p.w) + (size_t)n * p.K + k0);
                else rb[i].v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    auto store_tile = [&](int buf, int kt) {
        unsigned char* sa = smem + buf * STAGE;
        unsigned char* sb = sa + BM * ROWB;
        [[maybe_unused]] const int k0 = kt * BK + schunk * EPC;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = i * 32 + srow;
            unsigned char* dst = sa + r * ROWB + schunk * 16;
            if constexpr (AF32 && sizeof(T) == 2) {
                float4 lo = ra[i].lo, hi = ra[i].hi;
                if constexpr (LN) {
                    const float mean = ln_mean[r], den = ln_rden[r];
                    const float4 g0 = *reinterpret_cast<const float4*>(p.ln_gain + k0), g1 = *reinterpret_cast<const float4*>(p.ln_gain + k0 + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(p.ln_bias + k0), b1 = *reinterpret_cast<const float4*>(p.ln_bias + k0 + 4);
                    lo.x = g0.x * (lo.x - mean) / den + b0.x; lo.y = g0.y * (lo.y - mean) / den + b0.y;
                    lo.z = g0.z * (lo.z - mean) / den + b0.z; lo.w = g0.w * (lo.w - mean) / den + b0.w;
                    hi.x = g1.x * (hi.x - mean) / den + b1.x; hi.y = g1.y * (hi.y - mean) / den + b1.y;
                    hi.z = g1.z * (hi.z - mean) / den + b1.z; hi.w = g1.w * (hi.w - mean) / den + b1.w;
                }
                u32x4 o;
                o.x = pack_bf16(lo.x, lo.y);
                o.y = pack_bf16(lo.z, lo.w);
                o.z = pack_bf16(hi.x, hi.y);
                o.w = pack_bf16(hi.z, hi.w);
                *reinterpret_cast<u32x4*>(dst) = o;
            } else if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<u32x4*>(dst) = ra[i].v;
            } else {
                float4 v = ra[i].v;
                if constexpr (LN) {
                    const float mean = ln_mean[r], den = ln_rden[r];
                    const float4 g0 = *reinterpret_cast<const float4*>(p.ln_gain + k0);
                    const float4 b0 = *reinterpret_cast<const float4*>(p.ln_bias + k0);
                    v.x = g0.x * (v.x - mean) / den + b0.x; v.y = g0.y * (v.y - mean) / den + b0.y;
                    v.z = g0.z * (v.z - mean) / den + b0.z; v.w = g0.w * (v.w - mean) / den + b0.w;
                }
                *reinterpret_cast<float4*>(dst) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            unsigned char* dst = sb + (i * 32 + srow) * ROWB + schunk * 16;
            if constexpr (sizeof(T) == 2) *reinterpret_cast<u32x4*>(dst) = rb[i].v;
            else *reinterpret_cast<float4*>(dst) = rb[i].v;
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* sa = smem + buf * STAGE + (wr * (BM / 2) + (lane & 15)) * ROWB;
        const unsigned char* sb = smem + buf * STAGE + BM * ROWB + (wc * (BN / 2) + (lane & 15)) * ROWB;
#pragma unroll
        for (int g = 0; g < BK / Mma<T>::KGROUP; ++g) {
            typename Mma<T>::Frag fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = Mma<T>::load(reinterpret_cast<const T*>(sa + i * 16 * ROWB), g, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Mma<T>::load(reinterpret_cast<const T*>(sb + j * 16 * ROWB), g, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1, kt + 1);
        __syncthreads();
    }

    // epilogue.  C/D fragment: column = lane & 15, rows = (lane >> 4) * 4 + r
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wc * (BN / 2) + j * 16 + (lane & 15);
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.row_len) {
                    const int grp = m / p.rows_per_group;
                    if (m - grp * p.rows_per_group >= p.row_len[grp]) v = 0.f;
                }
                if (p.residual) v = p.residual[(size_t)m * p.ldr + n] + v;
                if (p.y_is_f32) static_cast<float*>(p.y)[(size_t)m * p.ldy + n] = v;
                else ElemOps<T>::store(static_cast<T*>(p.y) + (size_t)m * p.ldy + n, v);
            }
        }
    }
}

template <typename T, bool AF32, bool LN>
static int launch_tiles(const GemmParams& p, hipStream_t st) {
    // tile choice: fill the 256 CUs first, then prefer the larger tile
    const long w128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const dim3 block(256);
    if (p.M > 64 && w128 >= 224) {
        const dim3 grid((p.N + 127) / 128, (p.M + 127) / 128);
        hipLaunchKernelGGL((gemm_kernel<T, 128, 128, AF32, LN>), grid, block, 0, st, p);
    } else if (p.M > 64 || p.N >= 2048) {
        const dim3 grid((p.N + 63) / 64, (p.M + 63) / 64);
        hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AF32, LN>), grid, block, 0, st, p);
    } else {
        const dim3 grid((p.N + 31) / 32, (p.M + 63) / 64);
        hipLaunchKernelGGL((gemm_kernel<T, 64, 32, AF32, LN>), grid, block, 0, st, p);
    }
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// host-side tally of the GEMM work handed to the device (bofi_gemm_flops): 2 M N K per launch; launches that carry an early-out
// word (iterations of the bounding / semi-autoregressive loops that return at once when every image is finished) separately
double g_gemm_flops = 0.0, g_gemm_flops_skippable = 0.0;

int launch_linear(const LinearArgs& a, hipStream_t st) {
    if (!a.x || !a.w || !a.y || a.M < 0 || a.N <= 0 || a.K <= 0) return BOFI_ERR_ARG;
    if (a.M == 0) return BOFI_OK;
    // (a row-list launch covers *m_dev <= M rows, a number the host does not know: it is left out of the tally rather than counted at capacity)
    if (!a.row_idx) (a.skip_if_ge ? g_gemm_flops_skippable : g_gemm_flops) += 2.0 * a.M * a.N * a.K;
    const bool bf = a.w_dtype == BOFI_DT_BF16;
    if (!bf && a.w_dtype != BOFI_DT_F32) return BOFI_ERR_ARG;
    if (a.x_dtype != a.w_dtype && a.x_dtype != BOFI_DT_F32) return BOFI_ERR_ARG;
    if (a.y_dtype != a.w_dtype && a.y_dtype != BOFI_DT_F32) return BOFI_ERR_ARG;
    const int bk = bf ? 64 : 32;
    if (a.K % bk) return BOFI_ERR_ARG;
    const int xel = a.x_dtype == BOFI_DT_F32 ? 4 : 2;
    if (a.ldx < a.K || (a.ldx * xel) % 16 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return BOFI_ERR_ARG;
    if (a.ldy < a.N || (a.residual && a.ldr != 0 && a.ldr < a.N)) return BOFI_ERR_ARG;   // ldr == 0 broadcasts one row
    if (a.row_len && a.rows_per_group <= 0) return BOFI_ERR_ARG;
    {
        const int rc = launch_linear_glds(a, st);          // operands already in the compute dtype: LDS-DMA kernel
        if (rc >= 0) return rc;
    }
    if (a.ln_stats || a.stats_out || a.y2 || a.splitk > 1 || a.drop_thresh || a.mask_scale != 0.f || a.row_idx) return BOFI_ERR_ARG;     // only the LDS-DMA kernel implements these
    const bool ln = a.ln_gain != nullptr;
    if (ln && (a.x_dtype != BOFI_DT_F32 || !a.ln_bias || a.K % 8)) return BOFI_ERR_ARG;
    GemmParams p;
    p.x = a.x; p.ldx = a.ldx; p.w = a.w; p.bias = a.bias; p.residual = a.residual; p.ldr = a.ldr;
    p.y = a.y; p.ldy = a.ldy; p.y_is_f32 = a.y_dtype == BOFI_DT_F32; p.M = a.M; p.N = a.N; p.K = a.K;
    p.relu = a.relu; p.row_len = a.row_len; p.rows_per_group = a.rows_per_group;
    p.ln_gain = a.ln_gain; p.ln_bias = a.ln_bias; p.skip_if_ge = a.skip_if_ge; p.skip_threshold = a.skip_threshold;
    if (bf) {
        if (a.x_dtype == BOFI_DT_BF16) return launch_tiles<bf16_t, false, false>(p, st);
        return ln ? launch_tiles<bf16_t, true, true>(p, st) : launch_tiles<bf16_t, true, false>(p, st);
    }
    return ln ? launch_tiles<float, true, true>(p, st) : launch_tiles<float, true, false>(p, st);
}

}  // namespace bofi

extern "C" int bofi_linear(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* bias,
                           const float* residual, int ldr, void* y, int y_dtype, int ldy, int M, int N, int K, int relu,
                           const int* row_len, int rows_per_group, void* stream) {
    bofi::LinearArgs a{};
    a.x = x; a.x_dtype = x_dtype; a.ldx = ldx; a.w = w; a.w_dtype = w_dtype; a.bias = bias;
    a.residual = residual; a.ldr = ldr; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy; a.M = M; a.N = N; a.K = K;
    a.relu = relu; a.row_len = row_len; a.rows_per_group = rows_per_group;
    return bofi::launch_linear(a, (hipStream_t)stream);
}

extern "C" int bofi_linear_rows(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* bias, const float* residual, int ldr,
                                void* y, int y_dtype, int ldy, int M, int N, int K, int relu, const int* row_idx, const int* n_rows, void* stream) {
    if (!row_idx || !n_rows) return BOFI_ERR_ARG;
    bofi::LinearArgs a{};
    a.x = x; a.x_dtype = x_dtype; a.ldx = ldx; a.w = w; a.w_dtype = w_dtype; a.bias = bias;
    a.residual = residual; a.ldr = ldr; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.relu = relu;
    a.row_idx = row_idx; a.m_dev = n_rows;
    return bofi::launch_linear(a, (hipStream_t)stream);
}

extern "C" int bofi_linear_masked(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* mask, int ldm, float scale, void* y,
                                  int y_dtype, int ldy, int M, int N, int K, void* stream) {
    if (!mask || !(scale > 0.f)) return BOFI_ERR_ARG;
    bofi::LinearArgs a{};
    a.x = x; a.x_dtype = x_dtype; a.ldx = ldx; a.w = w; a.w_dtype = w_dtype;
    a.residual = mask; a.ldr = ldm; a.mask_scale = scale; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy; a.M = M; a.N = N; a.K = K;
    return bofi::launch_linear(a, (hipStream_t)stream);
}

extern "C" int bofi_linear_ex(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* bias,
                              const float* residual, int ldr, void* y, int y_dtype, int ldy, int M, int N, int K, int relu,
                              const int* row_len, int rows_per_group, float drop_p, uint64_t drop_seed, const uint64_t* drop_step, void* y2, int ldy2,
                              void* stream) {
    if (!(drop_p >= 0.f && drop_p < 1.f)) return BOFI_ERR_ARG;
    bofi::LinearArgs a{};
    a.x = x; a.x_dtype = x_dtype; a.ldx = ldx; a.w = w; a.w_dtype = w_dtype; a.bias = bias;
    a.residual = residual; a.ldr = ldr; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy; a.M = M; a.N = N; a.K = K;
    a.relu = relu; a.row_len = row_len; a.rows_per_group = rows_per_group;
    if (drop_p > 0.f) { a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_scale = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.drop_step = drop_step; }
    a.y2 = y2; a.ldy2 = ldy2;
    return bofi::launch_linear(a, (hipStream_t)stream);
}

extern "C" int bofi_linear_fused(const void* x, int ldx, const void* w, const float* bias, const float* residual, int ldr, void* y, int y_dtype, int ldy,
                                 void* y2, int ldy2, const float* ln_stats, const float* ln_colsum, int ln_groups, float* stats_out, int M, int N, int K,
                                 int relu, void* stream) {
    bofi::LinearArgs a{};
    a.x = x; a.x_dtype = BOFI_DT_BF16; a.ldx = ldx; a.w = w; a.w_dtype = BOFI_DT_BF16; a.bias = bias;
    a.residual = residual; a.ldr = ldr; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.relu = relu;
    a.y2 = y2; a.ldy2 = ldy2; a.ln_stats = ln_stats; a.ln_colsum = ln_colsum; a.ln_groups = ln_groups; a.stats_out = stats_out;
    return bofi::launch_linear(a, (hipStream_t)stream);
}

namespace bofi { extern int g_env_generation; }
extern "C" void bofi_reload_env(void) { ++bofi::g_env_generation; }

extern "C" double bofi_gemm_flops(int reset, double* skippable) {
    const double v = bofi::g_gemm_flops;
    if (skippable) *skippable = bofi::g_gemm_flops_skippable;
    if (reset) bofi::g_gemm_flops = bofi::g_gemm_flops_skippable = 0.0;
    return v;
}
