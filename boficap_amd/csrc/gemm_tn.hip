// Weight-gradient GEMM of the training path:  C[i][j] += sum_m A[m][i] * B[m][j]   (dW = dz^T x)
//
// Both operands are row-major with the CONTRACTION index as the row -- the layout the activations already have -- so
// no transposed copies are made: tiles of 64 rows go to LDS as they lie in HBM and the MFMA operands are gathered with
// the transposing LDS read of gfx950 (ds_read_b64_tr_b16), the same idiom attn_bf16.hip uses for P.V.  The contraction
// runs over the rows of the batch (thousands) while the output is a weight matrix (often only 64 tiles), so the rows are
// split across workgroups (blockIdx.z) and every workgroup adds its partial tile into C with float atomics -- C is the
// trainer's gradient buffer, which accumulates by definition.
//
// bf16 operands, fp32 accumulation.  Workgroup = 4 wavefronts, tile 64 x 64, each wavefront 32 x 32 (2 x 2 MFMA
// 16x16x32 tiles), 64 contraction rows per barrier.  LDS: two stages x (A tile + B tile) x 64 rows x 128 B = 32 KB.
// LDS rows are swizzled per 16-byte chunk so that the transposing reads are bank-conflict free (see key() below).
#include "bofi_common.h"
#include "bofi_kernels.h"
#include "gemm2.h"

namespace bofi {

extern int g_env_generation;

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct GemmTnParams {
    const bf16_t* a; int lda; int a_cols;        // a_cols: readable columns of A (multiple of 8, >= NI)
    const bf16_t* b; int ldb; int b_cols;
    float* c; int ldc;
    int M, NI, NJ, m_per_block;
    float* colsum;                               // optional: colsum[i] += sum_m A[m][i] (the bias gradient when A is dz)
};

// one 64 x 64 output tile at (i0, j0), contraction rows m_begin .. m_end - 1, added into C (and, when ``first_col``, the column
// sums of the A tiles into colsum)
// WT: 16-column MFMA tiles per wavefront side.  2 -> workgroup tile 64 x 64; 4 -> 128 x 128, which halves the LDS bytes read per
// MFMA (WT + WT operand fragments feed WT * WT MFMAs): the 64 x 64 tile asks the LDS for 256 B/clk per CU, twice what it delivers.
template <int WT>
__device__ __forceinline__ void gemm_tn_tile(const GemmTnParams& p, int i0, int j0, int m_begin, int m_end, bool first_col) {
    constexpr int T = 32 * WT, TM = 64;                          // 64 contraction rows per barrier (two MFMA k-steps)
    constexpr int CPR = T / 8, NL = TM * CPR / 256;              // 16-byte chunks per tile row; chunks per thread and operand
    __shared__ __attribute__((aligned(16))) bf16_t sa[2][TM * T];
    __shared__ __attribute__((aligned(16))) bf16_t sb[2][TM * T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3;
    if (m_begin >= m_end) return;
    const int wi = (wave >> 1) * 16 * WT, wj = (wave & 1) * 16 * WT;
    const int lr = tid / CPR, lc = tid % CPR;                    // loader: rows lr, lr + 256/CPR, ...; 16-byte chunk lc
    constexpr int RSTEP = 256 / CPR;                             // 32 (WT 2) or 16 (WT 4): a multiple of 16, so row + RSTEP swizzles like row
    const bool a_ok = i0 + lc * 8 < p.a_cols, b_ok = j0 + lc * 8 < p.b_cols;
    // Swizzle: 16-byte chunk c of LDS row r holds source chunk c ^ key(r).  A transposing read is banked per 32-lane half
    // (MI355X_MICROARCH.md, LDS): 8 rows x 32 bytes, each row's 32 bytes being one EVEN-ALIGNED PAIR of chunks -- so the key must
    // move whole pairs (bit 0 clear) and give the 8 rows 8 different bank ranges: with 256-byte rows (T = 128) every row starts at
    // bank 0 and the 8 rows take 8 different pairs, (r & 7) << 1; with 128-byte rows (T = 64) rows r, r + 1 sit in opposite halves
    // of the bank row and the 4 row pairs take 4 different chunk pairs, ((r >> 1) & 3) << 1.
    auto key = [](int r) { return T == 128 ? ((r & 7) << 1) : (((r >> 1) & 3) << 1); };
    const int sw = (lc ^ key(lr)) * 8;

    u32x4 va[NL], vb[NL];
    auto load = [&](int m0) {
#pragma unroll
        for (int hh = 0; hh < NL; ++hh) {
            const int m = m0 + lr + RSTEP * hh;
            va[hh] = u32x4{0u, 0u, 0u, 0u};
            vb[hh] = va[hh];
            if (m < m_end) {
                if (a_ok) va[hh] = *reinterpret_cast<const u32x4*>(p.a + (size_t)m * p.lda + i0 + lc * 8);
                if (b_ok) vb[hh] = *reinterpret_cast<const u32x4*>(p.b + (size_t)m * p.ldb + j0 + lc * 8);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int hh = 0; hh < NL; ++hh) {
            *reinterpret_cast<u32x4*>(&sa[buf][(lr + RSTEP * hh) * T + sw]) = va[hh];
            *reinterpret_cast<u32x4*>(&sb[buf][(lr + RSTEP * hh) * T + sw]) = vb[hh];
        }
    };

    f32x4 acc[WT][WT];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand gather: lane (g, tq, tp) addresses row 4g + tq (and + 16), columns 4tp .. 4tp + 3 of a 16-column block
    const int row0 = 4 * g + tq, rsw = key(row0);                 // (row0 + 16) swizzles like row0
    auto frag = [&](const bf16_t* tile, int col, int sub) -> bf16x8 {
        const bf16_t* p0 = tile + (row0 + 32 * sub) * T + (((col >> 3) ^ rsw) << 3) + (col & 7);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * T));
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };

    // column sums of the A tiles (workgroups of the first tile column only): thread -> column tid & 63, rows (tid >> 6) * 16 .. + 15
    const bool do_colsum = p.colsum != nullptr && first_col;
    float csum[WT / 2];
#pragma unroll
    for (int q = 0; q < WT / 2; ++q) csum[q] = 0.f;
    const int cc = tid & 63, crg = tid >> 6;

    load(m_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += TM) {
        const bool more = m0 + TM < m_end;
        if (more) load(m0 + TM);                                  // next tile in flight while this one is multiplied
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            bf16x8 fa[WT], fb[WT];
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                fa[t] = frag(sa[buf], wi + t * 16 + 4 * tp, sub);
                fb[t] = frag(sb[buf], wj + t * 16 + 4 * tp, sub);
            }
#pragma unroll
            for (int a = 0; a < WT; ++a)
#pragma unroll
                for (int b = 0; b < WT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        if (do_colsum) {
#pragma unroll
            for (int q = 0; q < WT / 2; ++q)
#pragma unroll
                for (int hh = 0; hh < 16; ++hh) {
                    const int r = crg * 16 + hh, c = cc + 64 * q;
                    csum[q] += bf16_to_f32(sa[buf][r * T + ((((c >> 3) ^ key(r))) << 3) + (c & 7)]);
                }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (do_colsum) {                                              // combine the four row groups, one atomic per column
        float* red = reinterpret_cast<float*>(&sa[0][0]);
#pragma unroll
        for (int q = 0; q < WT / 2; ++q) red[crg * T + cc + 64 * q] = csum[q];
        __syncthreads();
        if (tid < T && i0 + tid < p.NI) atomicAdd(&p.colsum[i0 + tid], (red[tid] + red[T + tid]) + (red[2 * T + tid] + red[3 * T + tid]));
    }
    // lane holds C[i = .. + 4g + r][j = .. + l15]
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b) {
            const int j = j0 + wj + b * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi + a * 16 + 4 * g + r;
                if (i < p.NI && j < p.NJ) atomicAdd(&p.c[(size_t)i * p.ldc + j], acc[a][b][r]);
            }
        }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnParams p) {
    const int m_begin = blockIdx.z * p.m_per_block;
    gemm_tn_tile<2>(p, blockIdx.x * 64, blockIdx.y * 64, m_begin, min(p.M, m_begin + p.m_per_block), blockIdx.y == 0);
}

// Many independent weight-gradient GEMMs in one launch.  A training step has ~126 of them (one per Linear use), each a
// small output with a long contraction; launched one by one every GEMM is a latency chain of its own with two workgroups
// per CU at best.  Grouped, the step's weight gradients are ~15 000 tiles in a few launches: no row split (no extra atomics),
// five workgroups per CU hiding each other's loads.  The problems travel BY VALUE in the kernel arguments, so a captured step
// graph holds them without a device-side table.
constexpr int TN_GROUP_MAX = 40;
struct GemmTnGroup {
    GemmTnParams prob[TN_GROUP_MAX];
    int tile_first[TN_GROUP_MAX + 1];                            // tiles of the problems before e
    int n, splits;
};

template <int WT>
__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(const GemmTnGroup grp) {
    constexpr int T = 32 * WT;
    // XCD-aware order (workgroup i runs on XCD i % 8, each with its own 4 MiB L2): XCD x walks a CONTIGUOUS run of tiles, so the
    // tiles that share an operand column block -- neighbours in the list -- meet in one L2 instead of being dealt round the chip.
    // The kernel is bound by operand re-fetch from L2 / Infinity Cache (64 FLOP per byte loaded at 128 x 128 x 64-row steps:
    // ~9 TB/s at the measured 530 TFLOP/s), not by LDS or MFMA issue: an LDS-DMA loader (no ds_write_b128) measured no faster.
    int t = blockIdx.x;
    {
        const int nt = gridDim.x, q8 = nt >> 3, r8 = nt & 7, xcd = t & 7, idx = t >> 3;
        t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    int lo = 0, hi = grp.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (grp.tile_first[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const GemmTnParams& q = grp.prob[lo];
    const int local = t - grp.tile_first[lo], tj = (q.NJ + T - 1) / T;
    const int m_begin = blockIdx.y * q.m_per_block;
    gemm_tn_tile<WT>(q, (local / tj) * T, (local % tj) * T, m_begin, min(q.M, m_begin + q.m_per_block), local % tj == 0);
}

static int check_tn(const void* a, int lda, int a_cols, const void* b, int ldb, int b_cols, const float* c, int ldc, int M, int NI, int NJ) {
    if (!a || !b || !c || M < 0 || NI <= 0 || NJ <= 0 || ldc < NJ) return BOFI_ERR_ARG;
    if (a_cols < NI || b_cols < NJ || a_cols % 8 || b_cols % 8 || lda < a_cols || ldb < b_cols || lda % 8 || ldb % 8) return BOFI_ERR_ARG;
    if (((uintptr_t)a % 16) || ((uintptr_t)b % 16)) return BOFI_ERR_ARG;
    return BOFI_OK;
}

int launch_gemm_tn_grouped(int n, const void* const* a, const int* lda, const int* a_cols, const void* const* b, const int* ldb,
                           const int* b_cols, float* const* c, const int* ldc, const int* M, const int* NI, const int* NJ,
                           float* const* colsum, hipStream_t st) {
    if (n < 0 || (n && (!a || !lda || !a_cols || !b || !ldb || !b_cols || !c || !ldc || !M || !NI || !NJ))) return BOFI_ERR_ARG;
    for (int e = 0; e < n; ++e)
        if (int rc = check_tn(a[e], lda[e], a_cols[e], b[e], ldb[e], b_cols[e], c[e], ldc[e], M[e], NI[e], NJ[e])) return rc;
    for (int e = 0; e < n; ++e) g_gemm_flops += 2.0 * M[e] * NI[e] * NJ[e];
    const int forced_wt = BOFI_ENV_INT("BOFI_TN_WT", 0);   // developer knob: 2 or 4 = one register-staged tile class for everything
    // two classes of problems: outputs of at least 128 x 128 take the register-staged 128 x 128 tile, the small ones (classifier heads) 64 x 64.
    // (Round 3's third class -- 256 x 256 outputs fed by an LDS-DMA ring -- measured no faster, 292 against 262 us on 24 decoder-sized problems,
    // and left the library in round 4: profiles/r03_tn_tile_256.txt, docs/history/r03.md.)
    auto cls_of = [&](int e) {
        if (forced_wt) return forced_wt == 4 ? 1 : 0;
        return (NI[e] >= 128 && NJ[e] >= 128) ? 1 : 0;
    };
    for (int cls = 1; cls >= 0; --cls) {
        const int T = cls == 1 ? 128 : 64;
        int e = 0;
        while (e < n) {
            GemmTnGroup g;
            g.n = 0;
            int tiles = 0, max_m = 0;
            for (; e < n && g.n < TN_GROUP_MAX; ++e) {
                if (M[e] == 0 || cls_of(e) != cls) continue;
                g.prob[g.n] = GemmTnParams{static_cast<const bf16_t*>(a[e]), lda[e], a_cols[e], static_cast<const bf16_t*>(b[e]), ldb[e], b_cols[e],
                                           c[e], ldc[e], M[e], NI[e], NJ[e], 0, colsum ? colsum[e] : nullptr};
                g.tile_first[g.n] = tiles;
                tiles += ((NI[e] + T - 1) / T) * ((NJ[e] + T - 1) / T);
                max_m = max(max_m, M[e]);
                ++g.n;
            }
            if (!g.n) break;
            g.tile_first[g.n] = tiles;
            // enough workgroups to fill the CUs (LDS: 1 / 2 / 5 workgroups per CU); a split costs one more tile of atomics per output tile
            const int want = cls == 1 ? 512 : 1280;
            g.splits = max(1, min(min(8, (max_m + 255) / 256), (want + tiles - 1) / tiles));
            for (int k = 0; k < g.n; ++k) g.prob[k].m_per_block = ((g.prob[k].M + g.splits - 1) / g.splits + 63) / 64 * 64;
            if (cls == 1) hipLaunchKernelGGL(gemm_tn_grouped_kernel<4>, dim3(tiles, g.splits), dim3(256), 0, st, g);
            else hipLaunchKernelGGL(gemm_tn_grouped_kernel<2>, dim3(tiles, g.splits), dim3(256), 0, st, g);
            BOFI_CHECK_LAUNCH();
        }
    }
    return BOFI_OK;
}

int launch_gemm_tn(const void* a, int lda, int a_cols, const void* b, int ldb, int b_cols, float* c, int ldc, int M, int NI, int NJ,
                   float* colsum, hipStream_t st) {
    if (!a || !b || !c || M < 0 || NI <= 0 || NJ <= 0 || ldc < NJ) return BOFI_ERR_ARG;
    if (a_cols < NI || b_cols < NJ || a_cols % 8 || b_cols % 8 || lda < a_cols || ldb < b_cols || lda % 8 || ldb % 8) return BOFI_ERR_ARG;
    if (((uintptr_t)a % 16) || ((uintptr_t)b % 16)) return BOFI_ERR_ARG;
    if (M == 0) return BOFI_OK;
    g_gemm_flops += 2.0 * M * NI * NJ;
    const int ti = (NI + 63) / 64, tj = (NJ + 63) / 64;
    // workgroups to aim for: every extra row split adds a tile of atomics, so small outputs (<= 128 tiles) take half as many
    // (measured, tools/mb_tn.py: 512x512 20.0 -> 17.1 us, 1024x512 over 2304 rows 17.1 -> 12.8 us; larger outputs prefer 1024)
    const int forced_wgs = BOFI_ENV_INT("BOFI_TN_WGS", 0);   // developer knob
    const int target_wgs = forced_wgs ? forced_wgs : (ti * tj <= 128 ? 512 : 1024);
    int splits = target_wgs / (ti * tj);
    splits = max(1, min(splits, (M + 127) / 128));                 // at least 4 tiles of rows per workgroup
    int mpb = ((M + splits - 1) / splits + 63) / 64 * 64;
    splits = (M + mpb - 1) / mpb;
    GemmTnParams p{static_cast<const bf16_t*>(a), lda, a_cols, static_cast<const bf16_t*>(b), ldb, b_cols, c, ldc, M, NI, NJ, mpb, colsum};
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(ti, tj, splits), dim3(256), 0, st, p);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi

extern "C" int bofi_gemm_tn_grouped(int n, const void* const* a, const int* lda, const int* a_cols, const void* const* b, const int* ldb,
                                    const int* b_cols, float* const* c, const int* ldc, const int* M, const int* NI, const int* NJ,
                                    float* const* colsum, void* stream) {
    return bofi::launch_gemm_tn_grouped(n, a, lda, a_cols, b, ldb, b_cols, c, ldc, M, NI, NJ, colsum, (hipStream_t)stream);
}

extern "C" int bofi_gemm_tn_acc(const void* a, int lda, int a_cols, const void* b, int ldb, int b_cols, float* c, int ldc, int M, int NI,
                                int NJ, float* colsum, void* stream) {
    return bofi::launch_gemm_tn(a, lda, a_cols, b, ldb, b_cols, c, ldc, M, NI, NJ, colsum, (hipStream_t)stream);
}
