// y = act(x . w^T + bias) [+ residual] with both operands already in the compute dtype.
// This is the hot GEMM of the bound+fill path (reference nn.Linear layers, see gemm.hip).
//
// gfx950 structure:
//   * 4 wavefronts (2x2) per workgroup, BM x BN output tile, 16x16 MFMA tiles, f32 accumulate
//     (v_mfma_f32_16x16x32_bf16, or four exact v_mfma_f32_16x16x4_f32 per 16-byte fragment).
//   * K is walked in 128-byte slabs (64 bf16 / 32 f32).  Slabs are copied global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4: no VGPR staging) into an NS-deep ring; the loads of the
//     next NS-1 slabs stay in flight across the single barrier of each K step, retired by a
//     COUNTED s_waitcnt vmcnt (never 0 inside the loop).
//   * LDS-DMA writes lane-linearly (1 KiB = 8 rows of 128 B per wave instruction), so rows cannot
//     be padded; bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle
//     applied to the per-lane SOURCE address (chunk c of row r is stored at chunk c ^ (r & 7))
//     and undone in the fragment read.
//   * Epilogue: accumulators -> LDS (reusing the ring) -> bias / ReLU / padded-row zeroing /
//     float32 residual -> global, one full 256-byte row segment per wave instruction.
#include <cstdio>
#include <cstdlib>

#include "bofi_common.h"
#include "bofi_kernels.h"
#include "gemm2.h"

namespace bofi {

int g_env_generation = 0;      // bumped by bofi_reload_env(): cached developer knobs are read again

// counted wait that leaves `younger` slabs (LPS LDS-DMA instructions each) in flight, younger in [0, MAXY] (wave-uniform)
template <int MAXY, int LPS> __device__ __forceinline__ void wait_slabs(int younger) {
    if constexpr (MAXY <= 0) {
        wait_vmcnt<0>();
    } else {
        if (younger >= MAXY) wait_vmcnt<(MAXY * LPS < 63 ? MAXY * LPS : 63)>();
        else wait_slabs<MAXY - 1, LPS>(younger);
    }
}

// FEAT: the optional parts of the epilogue / control this instantiation carries (bit 0: folded LayerNorm in, 1: row statistics
// and compute-dtype copy out, 2: padded-row zeroing, 3: dropout, 4: early-out word and developer ablations, 5: residual-as-mask, 6: row list).  A specialisation
// drops the kernel arguments of the parts it does not carry, which is what matters: the full kernel spills scalar registers.
constexpr int FEAT_ALL = 127;
template <typename T, int BM, int BN, int NS, int WM = 2, int WN = 2, int FEAT = FEAT_ALL>     // WM x WN wavefronts, each owns (BM/WM) x (BN/WN)
__global__ __launch_bounds__(64 * WM * WN) void gemm_glds_kernel(Gemm2Params p) {
    if constexpr (!(FEAT & 1)) { p.ln_stats = nullptr; p.ln_colsum = nullptr; }
    if constexpr (!(FEAT & 2)) { p.stats_out = nullptr; p.y2 = nullptr; }
    if constexpr (!(FEAT & 4)) { p.row_len = nullptr; }
    if constexpr (!(FEAT & 8)) { p.drop_thresh = 0; p.drop_step = nullptr; }
    if constexpr (!(FEAT & 16)) { p.skip_if_ge = nullptr; p.dbg = 0; }
    if constexpr (!(FEAT & 32)) { p.mask_scale = 0.f; }
    if constexpr (!(FEAT & 64)) { p.row_idx = nullptr; p.m_dev = nullptr; }
    constexpr int NW = WM * WN;
    constexpr int EPC = 16 / sizeof(T);                 // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;                         // one 128-byte slab row
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int STAGE = (BM + BN) * 128;              // bytes per ring slot
    constexpr int LA = BM / 8 / NW, LB = BN / 8 / NW;   // LDS-DMA instructions per wave per slab (A, B)
    static_assert(LA >= 1 && LB >= 1 && LA * 8 * NW == BM && LB * 8 * NW == BN, "tile rows must split evenly over the waves");
    constexpr int LPS = LA + LB;
    constexpr int EPI = BM * (BN + 4) * 4;              // epilogue staging, float32, +4 columns pad
    constexpr int SMEM_MAIN = (NS * STAGE > EPI) ? NS * STAGE : EPI;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_MAIN + BM * 8];

    if (p.skip_if_ge && *p.skip_if_ge >= p.skip_threshold) return;
    if (p.dbg & 4) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    // XCD-aware tile order (workgroup i runs on XCD i % 8, each XCD has its own 4 MiB L2 and fetches what it touches from the
    // Infinity Cache itself): XCD x gets a CONTIGUOUS run of tiles of a linear order that walks `row_bands` bands of m-tiles one
    // after the other, and inside a band n-tile major / m-tile minor.  With 8 bands an XCD owns a band of A rows and streams the
    // whole weight panel (fetch = 8 W + A); with 1 band it owns a band of weight columns and streams all rows (W + 8 A); 2 and 4 are
    // the 2-D splits in between (rb W + (8 / rb) A).  The host picks the cheapest; bijective for any tile count.
    const int ntn = (p.N + BN - 1) / BN, ntiles = gridDim.x;
    int tile = blockIdx.x;
    int mt, nt;
    if (!(p.dbg & 16) && !p.m_dev) {                   // (a row list fills the first row tiles only: plain order spreads them over the XCDs)
        const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        const int ntm = ntiles / ntn, mb = (ntm + p.row_bands - 1) / p.row_bands;     // m-tiles per band (the last band may be shorter)
        const int band = tile / (mb * ntn), local = tile - band * mb * ntn;
        const int rows_here = min(mb, ntm - band * mb);
        nt = local / rows_here;
        mt = band * mb + (local - nt * rows_here);
    } else {
        mt = tile / ntn; nt = tile - mt * ntn;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    if (p.m_dev) {
        p.M = min(p.M, *p.m_dev);
        if (m0 >= p.M) return;
    }
    auto row_of = [&](int m) { return p.row_idx ? p.row_idx[m < p.M ? m : p.M - 1] : m; };      // memory row of GEMM row m

    // LDS-DMA source pointers: wave-instruction j of this wave covers tile rows (wave*L + j)*8 .. +7;
    // lane i -> row +(i>>3), LDS chunk (i&7) <- global chunk (i&7) ^ (i>>3)   (XOR swizzle on the source)
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
    const T* asrc[LA];
    const T* bsrc[LB];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        int m = m0 + (wave * LA + j) * 8 + lrow;
        m = m < p.M ? m : p.M - 1;                      // clamp: rows past M are computed and dropped
        m = row_of(m);
        asrc[j] = static_cast<const T*>(p.x) + (size_t)m * p.ldx + lchunk * EPC + (size_t)blockIdx.y * (p.K / p.splitk);
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        int n = n0 + (wave * LB + j) * 8 + lrow;
        n = n < p.N ? n : p.N - 1;
        bsrc[j] = static_cast<const T*>(p.w) + (size_t)n * p.K + lchunk * EPC + (size_t)blockIdx.y * (p.K / p.splitk);
    }
    auto issue = [&](int slot, int kt) {
        if (p.dbg & 1) return;
        unsigned char* sa = smem + slot * STAGE + wave * LA * 1024;
        unsigned char* sb = smem + slot * STAGE + BM * 128 + wave * LB * 1024;
#pragma unroll
        for (int j = 0; j < LA; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + (size_t)kt * BK),
                                             (__attribute__((address_space(3))) void*)(sa + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < LB; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[j] + (size_t)kt * BK),
                                             (__attribute__((address_space(3))) void*)(sb + j * 1024), 16, 0, 0);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / p.splitk / BK;
    if (blockIdx.y > 0) { p.bias = nullptr; p.residual = nullptr; }
    p.y = static_cast<char*>(p.y) + (size_t)blockIdx.y * p.M * p.ldy * sizeof(float);      // split-K slabs are float32
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, s);

    // Epilogue operands of the vector path (bias, colsum, residual, region counts) are requested NOW, before
    // the K loop: their L2 round trip then overlaps the loop instead of following it (they are older than
    // every LDS-DMA, so the counted vmcnt waits of the loop cover them).  vmcnt counts stores too on gfx950,
    // so all epilogue loads must precede the first store anyway.
    constexpr int LPR = BN / 4;                        // lanes per output row (4 columns each)
    constexpr int RPI = 64 / LPR;                      // rows per wave instruction
    constexpr int NR = BM / (NW * RPI);                // rows per lane
    const int lr = lane / LPR, lc = (lane % LPR) * 4;
    const int n = n0 + lc;
    const bool ncol_ok = n < p.N;                      // N % 4 == 0: the four columns are in or out together
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs = bv;
    float4 rv[NR];
    bool live[NR], zero[NR];
    int mrow_out[NR];                                  // memory row of each of this lane's output rows
    if (p.vec_ok) {
        if (p.bias && ncol_ok) bv = *reinterpret_cast<const float4*>(p.bias + n);
        if (p.ln_stats && ncol_ok) cs = *reinterpret_cast<const float4*>(p.ln_colsum + n);
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int r = wave * RPI + lr + u * NW * RPI, m = m0 + r;
            live[u] = m < p.M && ncol_ok;
            mrow_out[u] = row_of(m);
            rv[u] = (p.residual && live[u]) ? *reinterpret_cast<const float4*>(p.residual + (size_t)mrow_out[u] * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            zero[u] = false;
            if (p.row_len && live[u]) {
                const int grp = m / p.rows_per_group;
                zero[u] = (m - grp * p.rows_per_group) >= p.row_len[grp];
            }
        }
    }

    // folded LayerNorm: row mean / 1/(std+eps) from the producer's partial sums.  Done first so that the
    // L2 round trip overlaps the K loop; the two small arrays live past the ring and the epilogue tile.
    float* s_mean = reinterpret_cast<float*>(smem + SMEM_MAIN);
    float* s_rstd = s_mean + BM;
    if (p.ln_stats && tid < BM) {
        const int m = m0 + tid;
        float sm = 0.f, sq = 0.f;
        if (m < p.M) {
            const int np4 = (p.ln_groups > 0 ? p.ln_groups : p.K >> 5) >> 1;      // two (sum, sumsq) pairs per 16-byte load
            const float4* sp = reinterpret_cast<const float4*>(p.ln_stats) + (size_t)row_of(m) * np4;
            if (np4 == 8) {
                // 16 pairs (d_model 512 in 32-column groups): summed as the balanced tree ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7)) over the
                // 8 loads -- the order an 8-lane DPP reduction produces, which is how gemm_pers.hip's loaders sum the same rows
                float a[8], q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float4 t = sp[i]; a[i] = t.x + t.z; q[i] = t.y + t.w; }
                sm = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[7] + a[6]) + (a[5] + a[4]));
                sq = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[7] + q[6]) + (q[5] + q[4]));
            } else
            for (int i = 0; i < np4; ++i) { const float4 t = sp[i]; sm += t.x + t.z; sq += t.y + t.w; }
        }
        const float mean = sm / (float)p.K;
        const float var = fmaxf((sq - sm * mean) / (float)(p.K - 1), 0.f);
        s_mean[tid] = mean;
        s_rstd[tid] = 1.0f / (sqrtf(var) + 1e-6f);
    }


    // fragment read offsets: tile row R = base + (lane & 15), wanted chunk g = 4*grp + (lane >> 4),
    // stored at chunk g ^ (R & 7); (base is a multiple of 16, so R & 7 == lane & 7)
    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    for (int kt = 0; kt < nk; ++kt) {
        // retire slab kt: at most min(NS-2, nk-1-kt) younger slabs may stay in flight
        const int younger = nk - 1 - kt;
        if constexpr (NS > 4) {
            wait_slabs<NS - 2, LPS>(younger);            // deep ring (whole K in flight): every step waits for exactly its own slab
        } else {
            if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LPS>();
            else if (NS > 3 && younger == 1) wait_vmcnt<LPS>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, kt + NS - 1);
        if (p.dbg & 2) continue;
        const unsigned char* sa = smem + (kt % NS) * STAGE + (wr * (BM / WM) + frow) * 128;
        const unsigned char* sb = smem + (kt % NS) * STAGE + BM * 128 + (wc * (BN / WN) + frow) * 128;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int coff = (((g * 4 + fq) ^ fx) << 4);
            typename GMma<T>::Frag fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const typename GMma<T>::Frag*>(sa + i * 16 * 128 + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const typename GMma<T>::Frag*>(sb + j * 16 * 128 + coff);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = GMma<T>::mma(fb[j], fa[i], acc[i][j]);   // D[n][m]: W is the MFMA "A"
        }
    }

    // ---- epilogue.  With W as the MFMA row operand the C/D fragment of tile (i, j) holds, per lane,
    // output row m = i*16 + (lane & 15) and the FOUR CONSECUTIVE columns n = j*16 + (lane >> 4)*4 + r.
    // Stores straight from that layout would be 32..64-byte pieces (measured: ~1 TB/s); the tile is
    // staged in LDS (reusing the ring) and written back as whole row segments, 16 B per lane.
    const int mrow = wr * (BM / WM) + (lane & 15), ncol = wc * (BN / WN) + (lane >> 4) * 4;
    if (p.dbg & 8) { if (acc[0][0][0] == 123.456f) static_cast<float*>(p.y)[0] = 1.f; return; }
    __syncthreads();
    float* es = reinterpret_cast<float*>(smem);
    constexpr int ES = BN + 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            *reinterpret_cast<float4*>(&es[(mrow + i * 16) * ES + ncol + j * 16]) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __syncthreads();
    if (p.vec_ok) {
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int r = wave * RPI + lr + u * NW * RPI, m = m0 + r;
            float4 v = *reinterpret_cast<const float4*>(&es[r * ES + lc]);
            if (p.ln_stats) {
                const float mu = s_mean[r], rs = s_rstd[r];
                v.x = rs * (v.x - mu * cs.x); v.y = rs * (v.y - mu * cs.y); v.z = rs * (v.z - mu * cs.z); v.w = rs * (v.w - mu * cs.w);
            }
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (zero[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.drop_thresh) {
                const uint64_t e = (uint64_t)m * p.N + n, dseed = p.drop_seed + (p.drop_step ? *p.drop_step : 0ull);
                v.x = drop_hash(dseed, e) >= p.drop_thresh ? v.x * p.drop_scale : 0.f;
                v.y = drop_hash(dseed, e + 1) >= p.drop_thresh ? v.y * p.drop_scale : 0.f;
                v.z = drop_hash(dseed, e + 2) >= p.drop_thresh ? v.z * p.drop_scale : 0.f;
                v.w = drop_hash(dseed, e + 3) >= p.drop_thresh ? v.w * p.drop_scale : 0.f;
            }
            if (p.mask_scale != 0.f) {
                v.x = rv[u].x > 0.f ? v.x * p.mask_scale : 0.f; v.y = rv[u].y > 0.f ? v.y * p.mask_scale : 0.f;
                v.z = rv[u].z > 0.f ? v.z * p.mask_scale : 0.f; v.w = rv[u].w > 0.f ? v.w * p.mask_scale : 0.f;
            } else {
                v.x = rv[u].x + v.x; v.y = rv[u].y + v.y; v.z = rv[u].z + v.z; v.w = rv[u].w + v.w;
            }
            if (p.stats_out) {                         // partial sums over this lane's aligned 32-column group (8 lanes)
                float ps = live[u] ? (v.x + v.y) + (v.z + v.w) : 0.f;
                float pq = live[u] ? __fadd_rn(__fmaf_rn(v.x, v.x, __fmul_rn(v.y, v.y)), __fmaf_rn(v.z, v.z, __fmul_rn(v.w, v.w))) : 0.f;      // (contraction spelled out: the two GEMM kernels must agree bit for bit)
                ps = oct_sum(ps); pq = oct_sum(pq);
                if (live[u] && (lane & 7) == 0)
                    reinterpret_cast<float2*>(p.stats_out)[(size_t)mrow_out[u] * (p.N >> 5) + (n >> 5)] = make_float2(ps, pq);
            }
            if (!live[u]) continue;
            if (p.y2) {
                if constexpr (sizeof(T) == 2) {
                    uint2 o;
                    o.x = pack_bf16(v.x, v.y);
                    o.y = pack_bf16(v.z, v.w);
                    *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.y2) + (size_t)mrow_out[u] * p.ldy2 + n) = o;
                } else {
                    *reinterpret_cast<float4*>(static_cast<float*>(p.y2) + (size_t)mrow_out[u] * p.ldy2 + n) = v;
                }
            }
            if (p.y_is_f32) {
                *reinterpret_cast<float4*>(static_cast<float*>(p.y) + (size_t)mrow_out[u] * p.ldy + n) = v;
            } else if constexpr (sizeof(T) == 2) {
                uint2 o;
                o.x = pack_bf16(v.x, v.y);
                o.y = pack_bf16(v.z, v.w);
                *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.y) + (size_t)mrow_out[u] * p.ldy + n) = o;
            }
        }
        return;
    }
    // general case (e.g. the vocabulary projection, N = 9491, rows not 16-byte aligned): one float per lane
    constexpr int NC = (BN + 63) / 64;
    constexpr int NRS = BM / NW;
    float sbv[NC], scs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = n0 + c * 64 + lane;
        const bool ok = c * 64 + lane < BN && n < p.N;
        sbv[c] = (p.bias && ok) ? p.bias[n] : 0.f;
        scs[c] = (p.ln_stats && ok) ? p.ln_colsum[n] : 0.f;
    }
    float srv[NRS][NC];
    bool szero[NRS];
    int smrow[NRS];
#pragma unroll
    for (int u = 0; u < NRS; ++u) {
        const int m = m0 + wave + NW * u;
        smrow[u] = row_of(m);
        szero[u] = false;
        if (p.row_len && m < p.M) {
            const int grp = m / p.rows_per_group;
            szero[u] = (m - grp * p.rows_per_group) >= p.row_len[grp];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int n = n0 + c * 64 + lane;
            srv[u][c] = (p.residual && m < p.M && c * 64 + lane < BN && n < p.N) ? p.residual[(size_t)smrow[u] * p.ldr + n] : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < NRS; ++u) {
        const int r = wave + NW * u, m = m0 + r;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int n = n0 + c * 64 + lane;
            float v = es[r * ES + c * 64 + lane];
            if (p.ln_stats) v = s_rstd[r] * (v - s_mean[r] * scs[c]);
            v += sbv[c];
            if (p.relu) v = fmaxf(v, 0.f);
            if (szero[u]) v = 0.f;
            if (p.drop_thresh) v = drop_hash(p.drop_seed + (p.drop_step ? *p.drop_step : 0ull), (uint64_t)m * p.N + n) >= p.drop_thresh ? v * p.drop_scale : 0.f;
            v = srv[u][c] + v;
            if (m >= p.M || c * 64 + lane >= BN || n >= p.N) continue;
            if (p.y_is_f32) static_cast<float*>(p.y)[(size_t)smrow[u] * p.ldy + n] = v;
            else ElemOps<T>::store(static_cast<T*>(p.y) + (size_t)smrow[u] * p.ldy + n, v);
        }
    }
}

template <typename T, int BM, int BN, int NS, int WM = 2, int WN = 2, int FEAT = FEAT_ALL>
static void launch_one(const Gemm2Params& p, hipStream_t st) {
    hipLaunchKernelGGL((gemm_glds_kernel<T, BM, BN, NS, WM, WN, FEAT>), dim3(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), p.splitk), dim3(64 * WM * WN), 0, st, p);
}

// the three tile shapes the heuristic picks, specialised for the feature sets the decode and training paths actually use
template <typename T, int FEAT>
static bool launch_specialised(const Gemm2Params& p, int bm, int bn, int ns, int nw, hipStream_t st) {
    if (bm == 128 && bn == 64 && ns == 2 && nw == 8) { launch_one<T, 128, 64, 2, 4, 2, FEAT>(p, st); return true; }
    if (bm == 64 && bn == 64 && ns == 2 && nw == 8) { launch_one<T, 64, 64, 2, 4, 2, FEAT>(p, st); return true; }
    if (bm == 64 && bn == 32 && ns == 4 && nw == 4) { launch_one<T, 64, 32, 4, 2, 2, FEAT>(p, st); return true; }
    if constexpr (sizeof(T) == 2 && (FEAT == 18 || FEAT == 82)) {     // the K >= 1024 rule inside the bounding / semi-autoregressive loops
        if (nw == 8 && bm == 64 && bn == 64 && ns == 4) { launch_one<T, 64, 64, 4, 4, 2, FEAT>(p, st); return true; }
    }
    if constexpr (sizeof(T) == 2 && FEAT < 3) {      // deeper rings / larger tiles for the encoder and fill GEMMs (one workgroup per CU)
        if (nw == 8) {
            if (bm == 64 && bn == 64 && ns == 4) { launch_one<T, 64, 64, 4, 4, 2, FEAT>(p, st); return true; }
            if (bm == 64 && bn == 64 && ns == 6) { launch_one<T, 64, 64, 6, 4, 2, FEAT>(p, st); return true; }
            if (bm == 64 && bn == 64 && ns == 9) { launch_one<T, 64, 64, 9, 4, 2, FEAT>(p, st); return true; }
            if (bm == 128 && bn == 64 && ns == 3) { launch_one<T, 128, 64, 3, 4, 2, FEAT>(p, st); return true; }
            if (bm == 64 && bn == 64 && ns == 3) { launch_one<T, 64, 64, 3, 4, 2, FEAT>(p, st); return true; }
            if (bm == 128 && bn == 64 && ns == 4) { launch_one<T, 128, 64, 4, 4, 2, FEAT>(p, st); return true; }
            if (bm == 128 && bn == 64 && ns == 6) { launch_one<T, 128, 64, 6, 4, 2, FEAT>(p, st); return true; }
            if (bm == 128 && bn == 128 && ns == 3) { launch_one<T, 128, 128, 3, 4, 2, FEAT>(p, st); return true; }
            if (bm == 128 && bn == 128 && ns == 4) { launch_one<T, 128, 128, 4, 4, 2, FEAT>(p, st); return true; }
            if (bm == 64 && bn == 128 && ns == 4) { launch_one<T, 64, 128, 4, 2, 4, FEAT>(p, st); return true; }
            if (bm == 64 && bn == 128 && ns == 6) { launch_one<T, 64, 128, 6, 2, 4, FEAT>(p, st); return true; }
            if (bm == 256 && bn == 128 && ns == 3) { launch_one<T, 256, 128, 3, 4, 2, FEAT>(p, st); return true; }
        }
    }
    if constexpr (sizeof(T) == 2) {      // M <= 64, K <= 512 per slice: all 8 slabs of the K extent in flight at once (108 / 90 KB of LDS)
        if (bm == 64 && bn == 32 && ns == 9 && nw == 4) { launch_one<T, 64, 32, 9, 2, 2, FEAT>(p, st); return true; }
        if (bm == 64 && bn == 16 && ns == 9 && nw == 2) { launch_one<T, 64, 16, 9, 2, 1, FEAT>(p, st); return true; }
    }
    return false;
}

template <typename T>
static int launch_glds_t(const Gemm2Params& p, hipStream_t st) {
    // developer override: BOFI_GEMM_TILE=<BM>x<BN>x<NS>
    int bm = 0, bn = 0, ns = 0;
    int nw = 4;
    const int feat = (p.ln_stats ? 1 : 0) | ((p.stats_out || p.y2) ? 2 : 0) | (p.row_len ? 4 : 0) | (p.drop_thresh ? 8 : 0) |
                     ((p.skip_if_ge || p.dbg) ? 16 : 0) | (p.mask_scale != 0.f ? 32 : 0) | (p.row_idx ? 64 : 0);
    if (const char* t = getenv("BOFI_GEMM_TILE")) { if (p.M > 64) sscanf(t, "%dx%dx%dx%d", &bm, &bn, &ns, &nw); }
    if constexpr (sizeof(T) == 2) {
        // GEMMs of >= 90 tiles of 256 x 128 (N % 128 == 0): persistent workgroups with loader wavefronts (gemm_pers.hip; same bits).
        // Alone such a launch is about as fast as this kernel (1.0-1.2x at >= 700 tiles, 0.8-0.95x below); with several decodes in
        // flight it is what keeps their big GEMMs from interleaving thousands of workgroups on every CU: 115 -> 135 k img/s on the
        // default bench (tools/exp/ab_bench2.sh, thresholds 1000 / 500 / 250 / 150 / 90: +3 / +8 / +9 / +14 / +17 %).
        // BOFI_GEMM_PERS=0 turns it off, BOFI_GEMM_PERS_MIN=<tiles> moves the threshold (developer knobs)
        if (!bm && (feat & ~16) <= 3 && !p.skip_if_ge) {       // (feature bit 4 alone = developer ablations)
            static int env_seen = -1, pers_on = 1;            // the two knobs are read once, and again after bofi_reload_env() (tests flip them)
            static long pers_min = 90;
            if (env_seen != g_env_generation) {
                const char* e = getenv("BOFI_GEMM_PERS");
                const char* m = getenv("BOFI_GEMM_PERS_MIN");
                pers_on = !e || atoi(e); pers_min = m ? atol(m) : 90; env_seen = g_env_generation;
            }
            const long t256 = (long)((p.M + 255) / 256) * (p.N / 128);
            if (pers_on && t256 >= pers_min) {
                const int r = launch_gemm_pers(p, feat & 3, st);
                if (r != -1) return r;
            }
        }
    }
    if (!bm) {
        // measured on MI355X (tools/microbench_ops.py, round 1): occupancy beats ring depth at K = 512;
        // 2 stages keep 3-5 workgroups per CU so that one's prologue/epilogue hides under another's loop
        const long t = (long)((p.M + 127) / 128) * ((p.N + 63) / 64);
        const int heur2 = BOFI_ENV_INT("BOFI_GEMM_HEUR2", 1);
        // 128-row tiles from 200 tiles on (round 2: the fill pass's qkv and w_1 at M = 1280 gain 15-25 % with four decodes in flight)
        const long thr = heur2 ? 200 : 400;
        if (p.M <= 64) {
            bm = 64; bn = 32; ns = 4;
            // the bounding loop's GEMMs (64 rows, K = 512 per slice) are latency chains: one L2 round trip per slab with a 4-deep
            // ring.  With the whole K extent (<= 8 slabs) issued up front the chain is one round trip; consumers that emit no row
            // statistics take 16-column tiles (twice the workgroups, half the weight bytes per workgroup)
            const int deep = BOFI_ENV_INT("BOFI_GEMM_DEEP", 1);
            if (deep && sizeof(T) == 2 && p.K / p.splitk <= 8 * 64) {
                ns = 9;
                if (!(p.stats_out || p.y2) && p.vec_ok && p.N % 16 == 0) { bn = 16; nw = 2; }
            }
        }
        else if (t >= thr) { bm = 128; bn = 64; ns = 2; nw = 8; }
        else { bm = 64; bn = 64; ns = 2; nw = 8; }          // 8 waves of 16x32 / 32x32: more waves per CU hide the slab latency
        // long K on few tiles (FFN w_2, att_embed: N = 512, K = 2048): the loop is a chain of slab round trips, three slabs in
        // flight instead of one (measured round 2, tools/mb_tiles2.py / mb_tiles3.py: 17.4 -> 11.6 us alone, 10.0 -> 7.6 us with
        // four in flight at M = 1280; 18.3 -> 15.2 us alone at M = 2304)
        if (heur2 && sizeof(T) == 2 && bm == 64 && bn == 64 && p.M > 64 && p.splitk == 1 && p.K >= 1024 && (feat < 3 || feat == 6 || feat == 18 || feat == 82)) ns = 4;
    }
    {
        bool done = false;
        switch (feat) {
            case 0: done = launch_specialised<T, 0>(p, bm, bn, ns, nw, st); break;          // plain: bias / ReLU / residual
            case 1: done = launch_specialised<T, 1>(p, bm, bn, ns, nw, st); break;          // consumer of a folded LayerNorm
            case 2: done = launch_specialised<T, 2>(p, bm, bn, ns, nw, st); break;          // producer of a residual stream
            case 16: done = launch_specialised<T, 16>(p, bm, bn, ns, nw, st); break;        // the same three inside the bound loop
            case 17: done = launch_specialised<T, 17>(p, bm, bn, ns, nw, st); break;
            case 18: done = launch_specialised<T, 18>(p, bm, bn, ns, nw, st); break;
            case 8: done = launch_specialised<T, 8>(p, bm, bn, ns, nw, st); break;          // training: dropout (+ compute-dtype copy)
            case 10: done = launch_specialised<T, 10>(p, bm, bn, ns, nw, st); break;
            case 32: done = launch_specialised<T, 32>(p, bm, bn, ns, nw, st); break;        // training: dX through relu (+ dropout)
            case 81: done = launch_specialised<T, 81>(p, bm, bn, ns, nw, st); break;        // SAIC decoder pass over a row list (+ halt word)
            case 82: done = launch_specialised<T, 82>(p, bm, bn, ns, nw, st); break;
            default: break;
        }
        if (done) { BOFI_CHECK_LAUNCH(); return BOFI_OK; }
    }
    const int key = bm * 10000 + bn * 10 + ns + (nw == 8 ? 100000000 : nw == 16 ? 200000000 : 0);
    switch (key) {
        case 100640642: launch_one<T, 64, 64, 2, 4, 2>(p, st); break;
        case 100640643: launch_one<T, 64, 64, 3, 4, 2>(p, st); break;
        case 100640644: launch_one<T, 64, 64, 4, 4, 2>(p, st); break;
        case 101280643: launch_one<T, 128, 64, 3, 4, 2>(p, st); break;
        case 201281282: launch_one<T, 128, 128, 2, 4, 4>(p, st); break;    // 16 waves
        case 101281282: launch_one<T, 128, 128, 2, 4, 2>(p, st); break;     // 8 waves
        case 101281283: launch_one<T, 128, 128, 3, 4, 2>(p, st); break;
        case 102561282: launch_one<T, 256, 128, 2, 4, 2>(p, st); break;
        case 101282562: launch_one<T, 128, 256, 2, 2, 4>(p, st); break;
        case 101280642: launch_one<T, 128, 64, 2, 4, 2>(p, st); break;
        case 1281282: launch_one<T, 128, 128, 2>(p, st); break;
        case 1281283: launch_one<T, 128, 128, 3>(p, st); break;
        case 1281284: launch_one<T, 128, 128, 4>(p, st); break;
        case 1280642: launch_one<T, 128, 64, 2>(p, st); break;
        case 1280643: launch_one<T, 128, 64, 3>(p, st); break;
        case 640642: launch_one<T, 64, 64, 2>(p, st); break;
        case 640643: launch_one<T, 64, 64, 3>(p, st); break;
        case 640644: launch_one<T, 64, 64, 4>(p, st); break;
        case 640322: launch_one<T, 64, 32, 2>(p, st); break;
        case 640324: launch_one<T, 64, 32, 4>(p, st); break;
        default: return BOFI_ERR_ARG;
    }
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// Operands in the compute dtype, no fused LayerNorm: the LDS-DMA kernel.  Returns -1 if the call
// is not eligible (the caller then uses the register-staged kernel of gemm.hip).
int launch_linear_glds(const LinearArgs& a, hipStream_t st) {
    if (a.x_dtype != a.w_dtype || a.ln_gain) return -1;
    if (a.ln_stats && (!a.ln_colsum || a.K % 64 || (uintptr_t)a.ln_stats % 16)) return BOFI_ERR_ARG;
    const int el = a.w_dtype == BOFI_DT_F32 ? 4 : 2;
    const int bk = 128 / el;
    if (a.K % bk || (a.ldx * el) % 16 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return -1;
    Gemm2Params p;
    p.x = a.x; p.ldx = a.ldx; p.w = a.w; p.bias = a.bias; p.residual = a.residual; p.ldr = a.ldr;
    p.y = a.y; p.ldy = a.ldy; p.y_is_f32 = a.y_dtype == BOFI_DT_F32; p.M = a.M; p.N = a.N; p.K = a.K;
    p.relu = a.relu; p.row_len = a.row_len; p.rows_per_group = a.rows_per_group;
    p.skip_if_ge = a.skip_if_ge; p.skip_threshold = a.skip_threshold;
    p.ln_stats = a.ln_stats; p.ln_colsum = a.ln_colsum; p.ln_groups = a.ln_groups; p.stats_out = a.stats_out; p.y2 = a.y2; p.ldy2 = a.ldy2;
    p.splitk = a.splitk > 1 ? a.splitk : 1;
    p.drop_thresh = a.drop_thresh; p.drop_scale = a.drop_scale; p.drop_seed = a.drop_seed; p.drop_step = a.drop_step;
    p.mask_scale = a.mask_scale;
    p.row_idx = a.row_idx; p.m_dev = a.m_dev;
    if (p.row_idx && (!p.m_dev || a.row_len || a.drop_thresh || a.splitk > 1 || a.mask_scale != 0.f)) return BOFI_ERR_ARG;
    if (p.drop_thresh && (a.splitk > 1 || a.stats_out)) return BOFI_ERR_ARG;
    if (p.mask_scale != 0.f && (!a.residual || a.splitk > 1 || a.stats_out || a.ln_stats)) return BOFI_ERR_ARG;
    if (p.splitk > 1 && (a.y_dtype != BOFI_DT_F32 || a.relu || a.ln_stats || a.stats_out || a.y2 || a.row_len || (a.K / p.splitk) % bk || a.K % p.splitk))
        return BOFI_ERR_ARG;
    { const char* dv = getenv("BOFI_GEMM_DBG"); p.dbg = dv ? atoi(dv) : 0; }
    {   // bytes an XCD layout makes the eight L2s fetch: rb * W + (8 / rb) * A
        const int forced = BOFI_ENV_INT("BOFI_GEMM_BANDS", 0);
        const double wb = (double)a.N * a.K, ab = (double)a.M * a.K;
        int best = 8; double cost = 8 * wb + ab;
        for (int rb : {4, 2, 1}) { const double c = rb * wb + (8 / rb) * ab; if (c < 0.9 * cost) { best = rb; cost = c; } }
        p.row_bands = (forced == 1 || forced == 2 || forced == 4 || forced == 8) ? forced : best;
    }
    const int yel = p.y_is_f32 ? 4 : el;
    p.vec_ok = (a.N % 4 == 0) && (a.ldy % 4 == 0) && ((uintptr_t)a.y % 16 == 0) && ((uintptr_t)a.y * 0 + (size_t)a.ldy * yel) % 8 == 0 &&
               (!a.bias || (uintptr_t)a.bias % 16 == 0) &&
               (!a.residual || (((uintptr_t)a.residual % 16 == 0) && (a.ldr % 4 == 0))) &&
               (!a.ln_colsum || (uintptr_t)a.ln_colsum % 16 == 0) && (!a.y2 || (((uintptr_t)a.y2 % 16 == 0) && a.ldy2 % 4 == 0));
    if ((a.stats_out || a.y2) && (!p.vec_ok || a.N % 32)) return BOFI_ERR_ARG;
    if (p.mask_scale != 0.f && !p.vec_ok) return BOFI_ERR_ARG;
    return el == 4 ? launch_glds_t<float>(p, st) : launch_glds_t<bf16_t>(p, st);
}

}  // namespace bofi
