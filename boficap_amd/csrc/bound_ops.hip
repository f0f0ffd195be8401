// Row-0 stages of one bounding iteration (LengthPredictor_UIC: DecoderLayer_UIC.sublayer[1], sublayer[2] of the reference,
// TransformerModel.py:283-297): one activation row per image, bf16 engine, d_model = 512.
//
// A few dozen to a few hundred activation rows: the LDS-DMA GEMM of gemm_glds.hip spends its time in ring fill, barriers and the staged
// epilogue (≈5.3 us per launch inside a graph against a ≈3.8 us floor of any load-compute-store kernel).  Here nothing is
// staged: both MFMA operands of v_mfma_f32_16x16x32_bf16 are loaded straight from global memory into the fragment layout
// (A row / B column = lane & 15, eight consecutive k per lane quarter = one 16-byte load), every load of a wavefront is in
// flight before the first MFMA, the four wavefronts of a workgroup split K and meet once in LDS, and the epilogue runs from
// the accumulator layout (lane = image, four consecutive output columns).
//
//   bound_qattn_kernel   q = Wq_src . LN(y1) for one head and 8 images, then that head's cross-attention of the 8 query rows
//                        over the image's regions: scores by lane = key, softmax in the wavefront, P.V by lane = (8-column
//                        chunk, key subset).  Replaces one GEMM launch and one attention launch.
//   rowgemm_kernel       y[M][N] = epilogue(x . W^T): LayerNorm fold on the input rows, bias, ReLU, float32 residual,
//                        bf16 copy, per-16-column row statistics for the next fold, split-K partial slabs.
#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

typedef __attribute__((ext_vector_type(4))) uint32_t rg_u32x4;

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// row mean and 1 / (std + eps) (unbiased std, eps = 1e-6: LayerNorm of TransformerModel.py:223-233) from `ng` <= 32 partial (sum, sumsq)
// pairs: every pair is requested before the first is summed (a loop over a run-time count waits for each load in turn: 8-16 L2 round trips)
__device__ __forceinline__ void row_norm(const float* stats, int ng, int d, float& mean, float& rstd) {
    const float4* sp = reinterpret_cast<const float4*>(stats);
    float4 t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = (2 * i < ng) ? sp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sm += t[i].x + t[i].z; sq += t[i].y + t[i].w; }
    mean = sm / (float)d;
    const float var = fmaxf((sq - sm * mean) / (float)(d - 1), 0.f);
    rstd = 1.0f / (sqrtf(var) + 1e-6f);
}

__global__ __launch_bounds__(256) void bound_qattn_kernel(BoundQAttnArgs a) {
    constexpr int D = 512, DK = 64, G = 8;
    __shared__ float4 red[4][4][64];          // [k quarter][16-column tile of the head][lane]
    __shared__ float qs[G][DK];
    __shared__ float ps[4][2][64];
    __shared__ float s_mean[G], s_rstd[G];
    if (a.skip_if_ge && *a.skip_if_ge >= a.skip_threshold) return;      // (before any load: an idle launch must not pull the kernel's operands)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int h = blockIdx.x, b0 = blockIdx.y * G, R = a.R;
    // query rows of this workgroup: b0 .. b0 + 7 of the batch, or of a device-side row list (rows of several per image: the image,
    // whose regions are the keys, is row / rows_per_image)
    const int nq = a.n_rows ? min(a.B, *a.n_rows) : a.B;
    if (b0 >= nq) return;
    auto mem_row = [&](int i) { i = min(i, nq - 1); return a.row_idx ? a.row_idx[i] : i; };
    auto image_of = [&](int row) { return a.rows_per_image > 0 ? row / a.rows_per_image : row; };

    // ---- every load of the kernel is issued here, in the order of use (vmcnt retires in order): GEMM operands, K rows
    // (lane = key), V chunks (lane = chunk, key subset)
    const int cch = (lane & 8) ? 7 - (lane & 7) : (lane & 7);          // row_mirror partners (l, 15 - l) hold the same chunk
    const int js = ((lane >> 4) << 1) | ((lane >> 3) & 1);
    bf16x8 fw[4][4], fx[4];
    {
        const int kq = wave * 128 + q * 8;
        const bf16_t* xp = a.x + (size_t)mem_row(b0 + (r & 7)) * D + kq;
#pragma unroll
        for (int s = 0; s < 4; ++s) fx[s] = ld_frag(xp + s * 32);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const bf16_t* wp = a.wq + (size_t)(h * DK + nt * 16 + r) * D + kq;
#pragma unroll
            for (int s = 0; s < 4; ++s) fw[nt][s] = ld_frag(wp + s * 32);
        }
    }
    if (tid < G) {
        float mean, rstd;
        const int sg = a.stats_groups > 0 ? a.stats_groups : D / 32;
        row_norm(a.stats + (size_t)mem_row(b0 + tid) * sg * 2, sg, D, mean, rstd);
        s_mean[tid] = mean; s_rstd[tid] = rstd;
    }
    const int idx0 = tid * 2;                       // the two q values this thread finalises: image idx >> 6, column idx & 63
    const float2 cs2 = *reinterpret_cast<const float2*>(a.colsum + h * DK + (idx0 & 63));
    const float2 bs2 = *reinterpret_cast<const float2*>(a.bias + h * DK + (idx0 & 63));
    int kl[2], bimg[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { bimg[u] = image_of(mem_row(b0 + 2 * wave + u)); kl[u] = a.att_len ? max(0, min(a.att_len[bimg[u]], R)) : R; }

    rg_u32x4 kk[2][8], vv[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const bf16_t* kp = a.k + ((size_t)bimg[u] * R + min(lane, R - 1)) * a.ldkv + h * DK;
#pragma unroll
        for (int c = 0; c < 8; ++c) kk[u][c] = reinterpret_cast<const rg_u32x4*>(kp)[c];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = min(js + 8 * t, R - 1);         // keys past R: the last row again, weighted 0 below
            vv[u][t] = *reinterpret_cast<const rg_u32x4*>(a.v + ((size_t)bimg[u] * R + j) * a.ldkv + h * DK + cch * 8);
        }
    // ---- q = Wq . x over this wavefront's K quarter
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt][s], fx[s], acc[nt], 0, 0, 0);
        red[wave][nt][lane] = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    }
    __syncthreads();
    {
        const int i = idx0 >> 6, n = idx0 & 63, nt = n >> 4, rr = n & 15, src = (rr >> 2) * 16 + i, comp = rr & 3;   // comp is 0 or 2
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* p = reinterpret_cast<const float*>(&red[w][nt][src]) + comp;
            v0 += p[0]; v1 += p[1];
        }
        const float mu = s_mean[i], rs = s_rstd[i];
        qs[i][n] = rs * (v0 - mu * cs2.x) + bs2.x;
        qs[i][n + 1] = rs * (v1 - mu * cs2.y) + bs2.y;
    }
    __syncthreads();

    // ---- attention of this wavefront's two images
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = 2 * wave + u;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 qa = *reinterpret_cast<const float4*>(&qs[i][c * 8]);
            const float4 qb = *reinterpret_cast<const float4*>(&qs[i][c * 8 + 4]);
            const rg_u32x4 kv = kk[u][c];
            s += qa.x * bf_lo(kv[0]) + qa.y * bf_hi(kv[0]) + qa.z * bf_lo(kv[1]) + qa.w * bf_hi(kv[1]);
            s += qb.x * bf_lo(kv[2]) + qb.y * bf_hi(kv[2]) + qb.z * bf_lo(kv[3]) + qb.w * bf_hi(kv[3]);
        }
        s *= 0.125f;                                      // / sqrt(d_k), d_k = 64
        const bool live = lane < kl[u];
        const float m = wave_max(live ? s : -INFINITY);
        const float e = live ? expf(s - m) : 0.f;
        const float sum = wave_sum(e);
        float p = e / sum;                                // no visible region: NaN, as softmax over an all-masked row of -inf
        ps[wave][u][lane] = p;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float p = ps[wave][u][js + 8 * t];       // 0 from the key count on (NaN everywhere for an image without regions)
            const rg_u32x4 v = vv[u][t];
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[2 * e] += p * bf_lo(v[e]); o[2 * e + 1] += p * bf_hi(v[e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = o[e];
            v += dpp_f32<DPP_MIRROR>(v);
            o[e] = xor32_sum(xor16_sum(v));
        }
        if (lane < 8 && b0 + 2 * wave + u < nq) {
            rg_u32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack_bf16(o[2 * e], o[2 * e + 1]);
            *reinterpret_cast<rg_u32x4*>(a.out + (size_t)mem_row(b0 + 2 * wave + u) * D + h * DK + cch * 8) = w;
        }
    }
}

int launch_bound_qattn(const BoundQAttnArgs& a, hipStream_t st) {
    if ((a.row_idx != nullptr) != (a.n_rows != nullptr)) return BOFI_ERR_ARG;
    if (a.d != 512 || a.H != 8 || a.R < 1 || a.R > 64 || a.B < 1 || a.ldkv % 8 || !a.x || !a.stats || !a.wq || !a.bias || !a.colsum ||
        !a.k || !a.v || !a.out)
        return BOFI_ERR_ARG;
    hipLaunchKernelGGL(bound_qattn_kernel, dim3(a.H, (a.B + 7) / 8), dim3(256), 0, st, a);
    if (!a.row_idx) (a.skip_if_ge ? g_gemm_flops_skippable : g_gemm_flops) += 2.0 * a.B * a.d * a.d;
    return hipGetLastError() == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------------------------------------
// y = epilogue(x . W^T) for a few rows (the bounding loop: one row per image); one workgroup per NT * 16 output columns, K slice of 512
// (blockIdx.y) and block of 64 rows (blockIdx.z), its four wavefronts take 128 k each.
// NT: the chip starts about one workgroup per 8 ns (tools/exp/mb_rowgemm.py: 4.8 us up to 256 workgroups, 7.1 at 320, 9.7 at 640, 14.8 at
// 1 280), so a launch of these latency-bound kernels wants at most one workgroup per CU: the launcher picks the narrowest tile that gets there.
template <int NT>
__global__ __launch_bounds__(256) void rowgemm_kernel(RowGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rg_smem[];
    float4 (*red)[4][NT][64] = reinterpret_cast<float4 (*)[4][NT][64]>(rg_smem);          // [k quarter][16-row group][column tile][lane]
    if (a.skip_if_ge && *a.skip_if_ge >= a.skip_threshold) return;      // (before any load: an idle launch must not pull the kernel's operands)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT, ks = blockIdx.y, mbase = blockIdx.z * 64;
    const int M = a.m_dev ? min(a.M, *a.m_dev) : a.M;        // row list: the count lives on the device (a.M sized the grid)
    if (mbase >= M) return;
    const int ng = min(4, (M - mbase + 15) >> 4);
    auto mem_row = [&](int m) { m = min(m, M - 1); return a.row_idx ? a.row_idx[m] : m; };     // GEMM row -> row of x / y / residual / statistics

    // epilogue operands of the row group this wavefront finalises (wave = group): row m, columns n .. n + 3 of every column tile
    const int m = mbase + wave * 16 + r, n = n0 + q * 4;
    const bool mine = wave < ng, rowok = m < M;
    const int mr = mem_row(m);
    int xrow[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xrow[g] = mem_row(mbase + g * 16 + r);

    f32x4 acc[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[g][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 sv[16];                             // the row's partial (sum, sum of squares) pairs, two per float4 (stats_groups <= 32)
    for (int c = 0; c < a.kchunks; ++c) {
        const int kq = (ks * a.kchunks + c) * 512 + wave * 128 + q * 8;
        bf16x8 fw[NT][4], fx[4][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16_t* wp = a.w + (size_t)(n0 + nt * 16 + r) * a.K + kq;
#pragma unroll
            for (int s = 0; s < 4; ++s) fw[nt][s] = ld_frag(wp + s * 32);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < ng) {
                const bf16_t* xp = a.x + (size_t)xrow[g] * a.ldx + kq;
#pragma unroll
                for (int s = 0; s < 4; ++s) fx[g][s] = ld_frag(xp + s * 32);
            }
        if (c == 0 && mine && a.stats) {       // requested behind the first chunk's operands, all at once: one round trip under the MFMAs
            const float4* sp = reinterpret_cast<const float4*>(a.stats + (size_t)mr * a.stats_groups * 2);
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = (2 * i < a.stats_groups) ? sp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < ng) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt][s], fx[g][s], acc[g][nt], 0, 0, 0);
            }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (g < ng) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) red[wave][g][nt][lane] = make_float4(acc[g][nt][0], acc[g][nt][1], acc[g][nt][2], acc[g][nt][3]);
        }
    // the epilogue's column constants and the residual: requested before the barrier, used behind it
    float4 cs[NT], bv[NT], rv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        cs[nt] = bv[nt] = rv[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!mine) continue;
        if (a.stats) cs[nt] = *reinterpret_cast<const float4*>(a.colsum + n + nt * 16);
        if (ks == 0) {
            bv[nt] = *reinterpret_cast<const float4*>(a.bias + n + nt * 16);
            if (a.residual) rv[nt] = *reinterpret_cast<const float4*>(a.residual + (size_t)mr * a.ldr + n + nt * 16);
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (mine && a.stats) {                     // (sums in the order of row_norm: pairs ascending)
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sm += sv[i].x + sv[i].z; sq += sv[i].y + sv[i].w; }
        mean = sm / (float)a.K;
        const float var = fmaxf((sq - sm * mean) / (float)(a.K - 1), 0.f);
        rstd = 1.0f / (sqrtf(var) + 1e-6f);
    }
    __syncthreads();
    if (!mine) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float4 v = red[0][wave][nt][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 t = red[w][wave][nt][lane]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (a.stats) {
            v.x = rstd * (v.x - mean * cs[nt].x); v.y = rstd * (v.y - mean * cs[nt].y);
            v.z = rstd * (v.z - mean * cs[nt].z); v.w = rstd * (v.w - mean * cs[nt].w);
        }
        v.x += bv[nt].x; v.y += bv[nt].y; v.z += bv[nt].z; v.w += bv[nt].w;
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        v.x += rv[nt].x; v.y += rv[nt].y; v.z += rv[nt].z; v.w += rv[nt].w;
        const int nn = n + nt * 16;
        if (a.stats_out) {                    // partial sums over 16 columns: lanes l, l ^ 16, l ^ 32, l ^ 48
            const float psum = xor32_sum(xor16_sum((v.x + v.y) + (v.z + v.w)));
            const float psq = xor32_sum(xor16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)));
            if (q == 0 && rowok) reinterpret_cast<float2*>(a.stats_out)[(size_t)mr * (a.N >> 4) + (n0 >> 4) + nt] = make_float2(psum, psq);
        }
        if (!rowok) continue;
        if (a.y) *reinterpret_cast<float4*>(a.y + ((size_t)ks * a.M + mr) * a.ldy + nn) = v;
        if (a.yb) {
            uint2 o;
            o.x = pack_bf16(v.x, v.y);
            o.y = pack_bf16(v.z, v.w);
            *reinterpret_cast<uint2*>(a.yb + (size_t)mr * a.ldyb + nn) = o;
        }
    }
}

template <int NT>
static void launch_rowgemm_t(const RowGemmArgs& b, int splitk, hipStream_t st) {
    hipLaunchKernelGGL((rowgemm_kernel<NT>), dim3(b.N / (16 * NT), splitk, (b.M + 63) / 64), dim3(256), (size_t)NT * 16384, st, b);
}

int launch_rowgemm(const RowGemmArgs& a, hipStream_t st) {
    const int splitk = a.splitk > 1 ? a.splitk : 1;
    RowGemmArgs b = a;
    b.kchunks = a.K / (512 * splitk);                   // without split-K a workgroup walks the whole K in chunks of 512
    if ((a.row_idx != nullptr) != (a.m_dev != nullptr)) return BOFI_ERR_ARG;
    if (a.M < 1 || a.N % 16 || b.kchunks < 1 || a.K != 512 * splitk * b.kchunks || a.ldx % 8 || !a.x || !a.w || !a.bias || (!a.y && !a.yb)) return BOFI_ERR_ARG;
    if (a.stats && (!a.colsum || a.stats_groups % 2 || a.stats_groups > 32 || splitk > 1)) return BOFI_ERR_ARG;
    if (splitk > 1 && (a.relu || a.stats_out || a.yb || !a.y)) return BOFI_ERR_ARG;
    if ((a.y && a.ldy % 4) || (a.yb && a.ldyb % 4) || (a.residual && a.ldr % 4)) return BOFI_ERR_ARG;
    // column tile: the narrowest of 16 / 32 / 64 columns that keeps the launch within one workgroup per CU (a tile's sums do not depend on it:
    // every output element is the same four k-quarter sums whatever the tile)
    const int forced = BOFI_ENV_INT("BOFI_ROWGEMM_NT", 0);      // developer knob: 1, 2 or 4
    const int per_tile = splitk * ((a.M + 63) / 64);
    int nt = 1;
    while (nt < 4 && a.N % (32 * nt) == 0 && (a.N / (16 * nt)) * per_tile > 256) nt *= 2;
    if (forced && a.N % (16 * forced) == 0) nt = forced;
    if (nt == 4) launch_rowgemm_t<4>(b, splitk, st);
    else if (nt == 2) launch_rowgemm_t<2>(b, splitk, st);
    else launch_rowgemm_t<1>(b, splitk, st);
    if (!a.row_idx) (a.skip_if_ge ? g_gemm_flops_skippable : g_gemm_flops) += 2.0 * a.M * a.N * a.K;
    return hipGetLastError() == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}

}  // namespace bofi

extern "C" int bofi_rowgemm(const void* x, int ldx, const void* w, const float* bias, const float* stats, int stats_groups, const float* colsum,
                            const float* residual, int ldr, float* y, int ldy, void* yb, int ldyb, float* stats_out, int M, int N, int K,
                            int splitk, int relu, const int* skip, int skip_threshold, const int* row_idx, const int* n_rows, void* stream) {
    bofi::RowGemmArgs a{};
    a.row_idx = row_idx; a.m_dev = n_rows;
    a.x = (const uint16_t*)x; a.ldx = ldx; a.w = (const uint16_t*)w; a.bias = bias; a.stats = stats; a.stats_groups = stats_groups; a.colsum = colsum;
    a.residual = residual; a.ldr = ldr; a.y = y; a.ldy = ldy; a.yb = (uint16_t*)yb; a.ldyb = ldyb; a.stats_out = stats_out;
    a.M = M; a.N = N; a.K = K; a.splitk = splitk; a.relu = relu; a.skip_if_ge = skip; a.skip_threshold = skip_threshold;
    return bofi::launch_rowgemm(a, (hipStream_t)stream);
}

extern "C" int bofi_bound_qattn(const void* x, const float* stats, const void* wq, const float* bias, const float* colsum, const void* k,
                                const void* v, int ldkv, const int* att_len, void* out, int B, int R, const int* skip, int skip_threshold,
                                const int* row_idx, const int* n_rows, int rows_per_image, void* stream) {
    bofi::BoundQAttnArgs a{};
    a.row_idx = row_idx; a.n_rows = n_rows; a.rows_per_image = rows_per_image;
    a.x = (const uint16_t*)x; a.stats = stats; a.wq = (const uint16_t*)wq; a.bias = bias; a.colsum = colsum; a.k = (const uint16_t*)k;
    a.v = (const uint16_t*)v; a.ldkv = ldkv; a.att_len = att_len; a.out = (uint16_t*)out; a.B = B; a.R = R; a.d = 512; a.H = 8;
    a.skip_if_ge = skip; a.skip_threshold = skip_threshold;
    return bofi::launch_bound_qattn(a, (hipStream_t)stream);
}

