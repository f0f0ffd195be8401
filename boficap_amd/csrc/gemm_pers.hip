// y = act(x . w^T + bias) [+ residual] for the large bf16 GEMMs of the encoder / fill stacks and the generator (same MFMA K order, same
// epilogue expressions in the same order and so the same bits as gemm_glds.hip's kernel), as PERSISTENT workgroups with dedicated loader
// wavefronts.  DESIGN.md section 12.12 has the measurements behind every choice below.
//
// Why: at 128 x 64 tiles the one-tile-per-workgroup kernel moves (128 + 64) * 128 B from L2 per 256 MFMA cycles of a SIMD = 96 B/clk per
// CU, against the 53 B/clk a CU takes in (profiles/r02_l2_stream_probe.txt); its per-workgroup start-up and epilogue overlap nothing
// inside the workgroup; and with several decodes in flight thousands of workgroups of different GEMMs interleave on every CU.  Here:
//   * 256 x 128 tiles (48 B/clk per CU at the full MFMA rate; 128 x 128 for shapes of at most 128 such tiles), 8 consumer wavefronts
//     of (BM / 4) x 64 each (16 x 16 x 32 MFMA tiles, TM x 4 accumulators);
//   * one workgroup per CU for the whole launch (grid = tiles / rounds), walking tiles blockIdx.x, blockIdx.x + grid, ... of the same
//     XCD-aware order;
//   * 4 loader wavefronts that only issue LDS-DMA (global_load_lds_dwordx4, 12 pieces of 1 KiB each per 64-deep K slab at 256 rows)
//     into a 3-slot ring and wait on their OWN vmcnt: the slab stream runs across tile boundaries, so the first two slabs of the next
//     tile land while the consumers run the epilogue of this one, and a consumer's vmcnt carries only its own epilogue operands and
//     stores (a wavefront's vmcnt retires in order: in a one-role kernel stores would sit in front of the next slabs).  The loaders
//     also turn the folded LayerNorm's partial sums into row mean / rstd, a tile ahead of their use;
//   * one workgroup barrier per K step (slab s landed; slot of slab s-1 free);
//   * epilogue: in registers where the output is bf16 without residual / copy / statistics (FAST: arithmetic in the MFMA C/D layout,
//     bf16 pairs transposed by permlane swaps, 16-byte stores; no LDS, no extra barrier), else staged through LDS as in
//     gemm_glds.hip -- 16 rows x 64 columns of float32 per wavefront at a time, private to the wavefront, in the slot of the tile's
//     last slab (one more workgroup barrier per tile: every consumer has read that slab) or, at 128 rows, in an area of its own.
#include <cstdio>
#include <cstdlib>

#include "gemm2.h"

namespace bofi {

constexpr int PBN = 128, PNS = 3, PBK = 64;
constexpr int PCONS = 8, PLOAD = 4;                    // consumer / loader wavefronts (2 and 8 loaders measure the same)
constexpr int PES = 68;                                // staging row stride in floats (64 + 4 pad)

// FEAT as in gemm_glds.hip: bit 0 = folded LayerNorm in, bit 1 = row statistics / compute-dtype copy out
// FAST: bf16 output, no residual / second copy / statistics -- the epilogue stays in registers (see below).  (Measured: with float32
// output and a float32 residual the register form is SLOWER than the staged one, 52 against 44 us at 11520 x 512 x 2048 -- its loads and
// stores are 64-byte row pieces, 16 lines per instruction, where the staged rows are 256 bytes -- so those GEMMs keep the staging.)
// BM: tile rows, 256 or 128 (128: shapes of at most 128 tiles of 256 rows -- twice the workgroups, half the K loop each)
template <int FEAT, bool RES, bool FAST, int BM>
__global__ __launch_bounds__(64 * (PCONS + PLOAD)) void gemm_pers_kernel(Gemm2Params p) {
    typedef bf16_t T;
    constexpr int TM = BM / 64;                          // 16-row MFMA tiles per consumer wavefront (its rows: BM / 4)
    constexpr int WROWS = BM / 4;
    constexpr int NLS = BM / 32;                         // statistics loads per loader (8 rows each)
    constexpr int PSTAGE = (BM + PBN) * 128;             // bytes per ring slot
    constexpr int PLA = BM / 8 / PLOAD, PLB = PBN / 8 / PLOAD;      // LDS-DMA pieces per loader per slab (4 loaders: 8 of A, 4 of W)
    constexpr int PLPS = PLA + PLB;
    if constexpr (!RES) p.residual = nullptr;
    if constexpr (!(FEAT & 1)) { p.ln_stats = nullptr; p.ln_colsum = nullptr; }
    if constexpr (!(FEAT & 2)) { p.stats_out = nullptr; p.y2 = nullptr; }
    // (128-row tiles: a ring slot is smaller than the staged epilogue's 8 x 16 x 68 floats -- they get an area of their own behind the statistics)
    constexpr int ESTAGE = BM == 128 ? PCONS * 16 * PES * 4 : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PNS * PSTAGE + BM * 16 + ESTAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = p.N / PBN, ntm = (p.M + BM - 1) / BM, ntiles = ntn * ntm;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int nk = p.K / PBK;
    // tile v of the XCD-aware order of gemm_glds.hip (workgroup i runs on XCD i % 8; grid is a multiple of 8, so tile v is on XCD v % 8)
    auto tile_coords = [&](int v, int& mt, int& nt) {
        const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = v & 7, idx = v >> 3;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        const int mb = (ntm + p.row_bands - 1) / p.row_bands;
        const int band = t / (mb * ntn), local = t - band * mb * ntn;
        const int rows_here = min(mb, ntm - band * mb);
        nt = local / rows_here;
        mt = band * mb + (local - nt * rows_here);
    };

    // developer aid (BOFI_GEMM_DBG & 64 with BOFI_GEMM_DBG_BUF=<device address>): 100 MHz stamps of wavefront 0 of each role in
    // workgroups 0..7: [workgroup][role][256] 64-bit words
    unsigned long long* stamps = ((p.dbg & 64) && blockIdx.x < 8 && (wave == 0 || wave == PCONS) && lane == 0)
                                     ? reinterpret_cast<unsigned long long*>(const_cast<int*>(p.skip_if_ge)) + (blockIdx.x * 2 + (wave ? 1 : 0)) * 256 : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() { if (stamps && n_stamp < 256) stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); };
    if (wave >= PCONS) {
        // ------------------------------------------------------------------ loader wavefronts
        const int lw = wave - PCONS;
        const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);       // XOR swizzle on the source, as in gemm_glds.hip
        const T* asrc[PLA];
        const T* bsrc[PLB];
        const T* nasrc[PLA];                                // the NEXT tile's sources: worked out two slabs into the current tile, while the
        const T* nbsrc[PLB];                                // loader would otherwise sit at the barrier (tile_coords is several integer divisions)
        int nsrow = 0;                                      // folded LayerNorm: first row of this loader's 64 rows of partial sums, of the tile setup() was last called for
        auto setup = [&](int t) {                           // -> nasrc / nbsrc / nsp of this workgroup's t-th tile
            if (t >= my_tiles) return;
            int mt, nt;
            tile_coords((int)blockIdx.x + t * G, mt, nt);
#pragma unroll
            for (int j = 0; j < PLA; ++j) {
                int m = mt * BM + (lw * PLA + j) * 8 + lrow;
                m = m < p.M ? m : p.M - 1;                  // rows past M are computed and dropped
                nasrc[j] = static_cast<const T*>(p.x) + (size_t)m * p.ldx + lchunk * 8;
            }
#pragma unroll
            for (int j = 0; j < PLB; ++j) {
                const int n = nt * PBN + (lw * PLB + j) * 8 + lrow;
                nbsrc[j] = static_cast<const T*>(p.w) + (size_t)n * p.K + lchunk * 8;
            }
            nsrow = mt * BM + lw * WROWS;
        };
        int i_tile = 0, i_kt = 0, i_slot = 0;
        setup(0);
        auto issue_next = [&]() {
            if (i_tile >= my_tiles) return;
            if (i_kt == 0) {
#pragma unroll
                for (int j = 0; j < PLA; ++j) asrc[j] = nasrc[j];
#pragma unroll
                for (int j = 0; j < PLB; ++j) bsrc[j] = nbsrc[j];
            }
            unsigned char* sa = smem + i_slot * PSTAGE + lw * PLA * 1024;
            unsigned char* sb = smem + i_slot * PSTAGE + BM * 128 + lw * PLB * 1024;
            if (!(p.dbg & 1)) {                             // (developer ablation BOFI_GEMM_DBG: 1 = no loads, 2 = no LDS reads / MFMA, 8 = no epilogue)
#pragma unroll
            for (int j = 0; j < PLA; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + (size_t)i_kt * PBK),
                                                 (__attribute__((address_space(3))) void*)(sa + j * 1024), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < PLB; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[j] + (size_t)i_kt * PBK),
                                                 (__attribute__((address_space(3))) void*)(sb + j * 1024), 16, 0, 0);
            }
            i_slot = i_slot + 1 == PNS ? 0 : i_slot + 1;
            if (++i_kt == nk) { i_kt = 0; ++i_tile; }
        };
        const int total = my_tiles * nk;
        // folded LayerNorm (16 partial-sum pairs per row): the loaders also turn the producer's partial (sum, sumsq) pairs into the row
        // mean / 1/(std+eps) the consumers' epilogue needs (they have the registers and, at a tile's first step, the time: the
        // consumers are still in the previous tile's epilogue).  A row's pairs are 128 contiguous bytes: 8 loads of 8 rows each per
        // loader (8 lanes per row, one line each; one load per lane and ROW would touch 64 lines per instruction -- measured 1.3 us of
        // issue per tile).  The loads of tile c+1 go out at the first step of tile c, in front of the slab that step issues -- the
        // counted wait of the second step leaves them in flight with that slab, every later wait covers them -- and are summed a tile
        // later, in the idle time before tile c+1's first barrier: the 8 pieces of a row as the balanced tree an 8-lane DPP reduction
        // gives (gemm_glds.hip sums its 8 loads in the same order); the results go to the half of s_mean / s_rstd the consumers are
        // not reading.
        // (the 8 loads are inline assembly: compiler-tracked loads would make it drain vmcnt to 0 -- the newest slab included -- at
        // their first use and again before the next tile's loads overwrite the registers)
        float* s_mean = reinterpret_cast<float*>(smem + PNS * PSTAGE);
        f32x4 st[8];
        auto load_stats = [&]() {                           // rows of the tile setup() was last called for
#pragma unroll
            for (int c = 0; c < NLS; ++c) {                 // always NLS loads: the waits count them
                const int m = nsrow + (lane >> 3) * NLS + c;      // load c: rows c, NLS + c, .. 7*NLS + c of this loader's BM / 4 (8 lanes per row)
                const float4* sp = reinterpret_cast<const float4*>(p.ln_stats) + (size_t)(m < p.M ? m : p.M - 1) * 8 + (lane & 7);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(st[c]) : "v"(sp) : "memory");
            }
        };
        if (p.ln_stats) load_stats();                       // tile 0's: in front of the first slabs
        issue_next();
        issue_next();
        int kt = 0, c_tile = 0;
        for (int s = 0; s < total; ++s) {
            if (kt == 0) {
                // first step of a tile: the consumers are still in the previous tile's epilogue
                if (p.ln_stats) {
                    if (s == 0) wait_vmcnt<2 * PLPS>();     // tile 0's statistics (the two slabs behind them may be in flight); later tiles': covered long ago
                    if constexpr (NLS == 8) asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4]), "+v"(st[5]), "+v"(st[6]), "+v"(st[7]) : : "memory");
                    else asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]) : : "memory");
                    float row_sm = 0.f, row_sq = 0.f;
#define BOFI_STAT_ROWS(c)                                                                                                   \
                    {   /* row NLS*j + c: its 8 pieces sit in lanes 8*j .. 8*j+7; every one of them gets the sum, lane 8*j + c keeps it */ \
                        const float sm = oct_sum(st[c][0] + st[c][2]), sq = oct_sum(st[c][1] + st[c][3]);                   \
                        if ((lane & 7) == (c)) { row_sm = sm; row_sq = sq; }                                                \
                    }
                    BOFI_STAT_ROWS(0) BOFI_STAT_ROWS(1) BOFI_STAT_ROWS(2) BOFI_STAT_ROWS(3)
                    if constexpr (NLS == 8) { BOFI_STAT_ROWS(4) BOFI_STAT_ROWS(5) BOFI_STAT_ROWS(6) BOFI_STAT_ROWS(7) }
#undef BOFI_STAT_ROWS
                    const float mean = row_sm / (float)p.K;
                    const float var = fmaxf((row_sq - row_sm * mean) / (float)(p.K - 1), 0.f);
                    if ((lane & 7) < NLS) {                 // lane 8*j + c holds row NLS*j + c
                        float* dst = s_mean + (c_tile & 1) * (2 * BM) + lw * WROWS + (lane >> 3) * NLS + (lane & 7);
                        dst[0] = mean;
                        dst[BM] = 1.0f / (sqrtf(var) + 1e-6f);
                    }
                }
                setup(c_tile + 1);                          // the next tile's sources (tile_coords is several integer divisions) ...
            }
            const bool stats_next = p.ln_stats && c_tile + 1 < my_tiles;
            if (s + 1 >= total) wait_vmcnt<0>();
            else if (stats_next && kt == 1) wait_vmcnt<PLPS + NLS>();
            else wait_vmcnt<PLPS>();                        // slab s has landed; slab s+1 may be in flight
            stamp();
            if (kt == 0 && stats_next) load_stats();        // ... and its statistics, behind slab s+1, in front of slab s+2
            __builtin_amdgcn_s_barrier();                   // step barrier: the consumers are past slab s-1
            stamp();
            issue_next();                                   // slab s+2 -> the slot of slab s-1
            stamp();
            if (++kt == nk) { kt = 0; ++c_tile; if constexpr (!FAST) __builtin_amdgcn_s_barrier(); }      // tile barrier (the staged epilogue's)
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer wavefronts
    const int wr = wave >> 1, wc = wave & 1;                // 4 x 2 wavefronts, 64 x 64 each
    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    const int lc = (lane & 15) * 4, lr = lane >> 4;         // epilogue: 16 lanes per row (4 columns each), 4 rows per wave instruction
    const float* s_stat = reinterpret_cast<const float*>(smem + PNS * PSTAGE) + wr * WROWS;      // this wavefront's rows of [tile parity][mean | rstd][BM] (written by the loaders)
    int slot = 0;
    for (int jt = 0; jt < my_tiles; ++jt) {
        int mt, nt;
        tile_coords((int)blockIdx.x + jt * G, mt, nt);
        const float* s_mean = s_stat + (jt & 1) * (2 * BM);
        const float* s_rstd = s_mean + BM;
        const int m0 = mt * BM + wr * WROWS, n = nt * PBN + wc * 64 + lc;
        f32x4 acc[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs = bv;
        float4 rv[TM][4];
        float4 bvf[4], csf[4];                              // FAST: bias / column sums of this lane's columns j*16 + (lane >> 4)*4 .. +3 in the C/D layout
        const int nwv = nt * PBN + wc * 64;                 // first column of this wavefront
        for (int kt = 0; kt < nk; ++kt) {
            stamp();
            __builtin_amdgcn_s_barrier();
            stamp();
            if (kt == nk - 1) {                             // the small epilogue vectors ride under the last step
                if constexpr (FAST) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bvf[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nwv + j * 16 + fq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        csf[j] = p.ln_stats ? *reinterpret_cast<const float4*>(p.ln_colsum + nwv + j * 16 + fq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else {
                    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
                    if (p.ln_stats) cs = *reinterpret_cast<const float4*>(p.ln_colsum + n);
                }
            }
            const unsigned char* sa = smem + slot * PSTAGE + (wr * WROWS + frow) * 128;
            const unsigned char* sb = smem + slot * PSTAGE + BM * 128 + (wc * 64 + frow) * 128;
            if (!(p.dbg & 2)) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int coff = (((g * 4 + fq) ^ fx) << 4);
                bf16x8 fa[TM], fb[4];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 128 + coff);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 128 + coff);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = GMma<bf16_t>::mma(fb[j], fa[i], acc[i][j]);   // D[n][m]: W is the MFMA "A"
            }
            }
            if (kt + 1 < nk) slot = slot + 1 == PNS ? 0 : slot + 1;
        }
        if constexpr (RES) {                                // the residual rows: requested once the fragments' registers are free, all before the first store
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int m = m0 + i * 16 + u * 4 + lr;
                    rv[i][u] = *reinterpret_cast<const float4*>(p.residual + (size_t)(m < p.M ? m : p.M - 1) * p.ldr + n);
                }
        }
        stamp();
        if constexpr (FAST) {
            // Register epilogue.  The C/D fragment of tile (i, j) holds row i*16 + (lane & 15), columns j*16 + (lane >> 4)*4 .. +3: the
            // arithmetic (same expressions, same order as the staged path: the same bits) runs right there, the results are packed to
            // bf16 pairs, and a 4 x 4 transpose over (tile j, lane row lane >> 4) -- two v_permlane32_swap + two v_permlane16_swap per
            // dword -- leaves every lane with 16 CONSECUTIVE columns of its row: two 16-byte stores per lane and 16-row group.  No LDS
            // staging, no tile barrier: the epilogue is ~340 vector instructions per wavefront instead of ~540 plus 32 dependent LDS trips.
            slot = slot + 1 == PNS ? 0 : slot + 1;
            if (p.dbg & 8) { if (acc[0][0][0] == 123.456f) static_cast<float*>(p.y)[0] = acc[1][1][1] + acc[TM - 1][2][2]; continue; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float mu = 0.f, rs = 1.f;
                if (p.ln_stats) { mu = s_mean[i * 16 + frow]; rs = s_rstd[i * 16 + frow]; }
                uint32_t d[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    if (p.ln_stats) {
                        v.x = rs * (v.x - mu * csf[j].x); v.y = rs * (v.y - mu * csf[j].y); v.z = rs * (v.z - mu * csf[j].z); v.w = rs * (v.w - mu * csf[j].w);
                    }
                    v.x += bvf[j].x; v.y += bvf[j].y; v.z += bvf[j].z; v.w += bvf[j].w;
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    d[j][0] = pack_bf16(v.x, v.y);
                    d[j][1] = pack_bf16(v.z, v.w);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {               // d[j][c] in lane row q  ->  d[q][c] of lane row j
                    auto s02 = __builtin_amdgcn_permlane32_swap(d[0][c], d[2][c], false, false);
                    auto s13 = __builtin_amdgcn_permlane32_swap(d[1][c], d[3][c], false, false);
                    auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                    auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                    d[0][c] = t01[0]; d[1][c] = t01[1]; d[2][c] = t23[0]; d[3][c] = t23[1];
                }
                const int m = m0 + i * 16 + frow;           // this lane now holds columns nwv + (lane >> 4)*16 .. +15 of row m
                if (m < p.M) {
                    bf16_t* dst = static_cast<bf16_t*>(p.y) + (size_t)m * p.ldy + nwv + fq * 16;
                    *reinterpret_cast<uint4*>(dst) = make_uint4(d[0][0], d[0][1], d[1][0], d[1][1]);
                    *reinterpret_cast<uint4*>(dst + 8) = make_uint4(d[2][0], d[2][1], d[3][0], d[3][1]);
                }
            }
            stamp();
            continue;
        }
        __builtin_amdgcn_s_barrier();                       // tile barrier: every consumer has read the last slab; its slot is free until the next step barrier
        float* es = reinterpret_cast<float*>(smem + (BM == 128 ? PNS * PSTAGE + BM * 16 : slot * PSTAGE)) + wave * (16 * PES);
        slot = slot + 1 == PNS ? 0 : slot + 1;
        if (p.dbg & 8) { if (acc[0][0][0] == 123.456f) static_cast<float*>(p.y)[0] = acc[1][1][1] + acc[TM - 1][2][2]; continue; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // C/D fragment of tile (i, j): row i*16 + (lane & 15), columns j*16 + (lane >> 4)*4 .. +3
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(&es[frow * PES + j * 16 + fq * 4]) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = u * 4 + lr, r = i * 16 + rr, m = m0 + r;      // r: row inside the wavefront's 64
                const bool live = m < p.M;
                float4 v = *reinterpret_cast<const float4*>(&es[rr * PES + lc]);
                if (p.ln_stats) {
                    const float mu = s_mean[r], rs = s_rstd[r];
                    v.x = rs * (v.x - mu * cs.x); v.y = rs * (v.y - mu * cs.y); v.z = rs * (v.z - mu * cs.z); v.w = rs * (v.w - mu * cs.w);
                }
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if constexpr (RES) { v.x = rv[i][u].x + v.x; v.y = rv[i][u].y + v.y; v.z = rv[i][u].z + v.z; v.w = rv[i][u].w + v.w; }
                if (p.stats_out) {                         // partial sums over this lane's aligned 32-column group (8 lanes)
                    float ps = live ? (v.x + v.y) + (v.z + v.w) : 0.f;
                    float pq = live ? __fadd_rn(__fmaf_rn(v.x, v.x, __fmul_rn(v.y, v.y)), __fmaf_rn(v.z, v.z, __fmul_rn(v.w, v.w))) : 0.f;      // (contraction spelled out: the two GEMM kernels must agree bit for bit)
                    ps = oct_sum(ps); pq = oct_sum(pq);
                    if (live && (lane & 7) == 0)
                        reinterpret_cast<float2*>(p.stats_out)[(size_t)m * (p.N >> 5) + (n >> 5)] = make_float2(ps, pq);
                }
                uint2 o;
                o.x = pack_bf16(v.x, v.y);
                o.y = pack_bf16(v.z, v.w);
                if (!live) continue;
                if (p.y2) *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.y2) + (size_t)m * p.ldy2 + n) = o;
                if (p.y_is_f32) *reinterpret_cast<float4*>(static_cast<float*>(p.y) + (size_t)m * p.ldy + n) = v;
                else *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.y) + (size_t)m * p.ldy + n) = o;
            }
        }
        stamp();
    }
}

int launch_gemm_pers(const Gemm2Params& p_in, int feat, hipStream_t st) {
    Gemm2Params p = p_in;
    if (p.dbg & 64) { const char* b = getenv("BOFI_GEMM_DBG_BUF"); p.skip_if_ge = b ? reinterpret_cast<const int*>(strtoull(b, nullptr, 0)) : nullptr; if (!p.skip_if_ge) p.dbg &= ~64; }
    if (feat > 3 || p.splitk != 1 || !p.vec_ok || p.N % PBN || p.K % PBK || p.K < 3 * PBK || p.row_len || p.row_idx || p.drop_thresh || p.mask_scale != 0.f ||
        (p.skip_if_ge && !(p.dbg & 64)))
        return -1;
    if (p.ln_stats) { const int g = p.ln_groups > 0 ? p.ln_groups : p.K >> 5; if (g != 16) return -1; }      // (other group counts: gemm_glds.hip)
    // 128-row tiles where 256-row tiles would leave half of the chip idle (at most 128 of them: the fill pass's N = 512 GEMMs are 92)
    const int bm128_max = BOFI_ENV_INT("BOFI_GEMM_PERS_BM128", 128);      // developer knob: 0 = 256-row tiles only
    const int t256 = (p.N / PBN) * ((p.M + 255) / 256);
    const int bm = t256 <= bm128_max ? 128 : 256;
    const int ntiles = (p.N / PBN) * ((p.M + bm - 1) / bm);
    const int cus = BOFI_ENV_INT("BOFI_GEMM_PERS_GRID", 256);
    // every workgroup walks `rounds` tiles: the grid is the smallest multiple of 8 (tile v stays on XCD v % 8) that covers the tiles in
    // as many rounds as all CUs would need -- 540 tiles run on 184 CUs in 3 rounds, not on 256 in 3, and the rest stay free for the
    // other decodes in flight
    const int min_rounds = BOFI_ENV_INT("BOFI_GEMM_PERS_ROUNDS", 1);      // developer knob: tiles per workgroup at least
    int rounds = (ntiles + cus - 1) / cus;
    if (rounds < min_rounds) rounds = min_rounds;
    int grid = ntiles;
    if (ntiles > cus || rounds > 1) { grid = (((ntiles + rounds - 1) / rounds + 7) / 8) * 8; if (grid > cus) grid = cus; if (grid > ntiles) grid = ntiles; }
    const dim3 g(grid), b(64 * (PCONS + PLOAD));
    const int fast_ok = BOFI_ENV_INT("BOFI_GEMM_PERS_FAST", 1);      // developer knob: 0 = staged epilogue everywhere
    const bool fast = fast_ok && !(feat & 2) && !p.residual && !p.y_is_f32 && p.ldy % 8 == 0 && (uintptr_t)p.y % 16 == 0;
#define PERS_LAUNCH(F, R, Q)                                                                       \
    { if (bm == 128) hipLaunchKernelGGL((gemm_pers_kernel<F, R, Q, 128>), g, b, 0, st, p);          \
      else hipLaunchKernelGGL((gemm_pers_kernel<F, R, Q, 256>), g, b, 0, st, p); }
    switch (feat * 2 + (p.residual ? 1 : 0)) {
        case 0: if (fast) PERS_LAUNCH(0, false, true) else PERS_LAUNCH(0, false, false) break;
        case 1: PERS_LAUNCH(0, true, false) break;
        case 2: if (fast) PERS_LAUNCH(1, false, true) else PERS_LAUNCH(1, false, false) break;
        case 3: PERS_LAUNCH(1, true, false) break;
        case 4: PERS_LAUNCH(2, false, false) break;
        case 5: PERS_LAUNCH(2, true, false) break;
        case 6: PERS_LAUNCH(3, false, false) break;
        default: PERS_LAUNCH(3, true, false) break;
    }
#undef PERS_LAUNCH
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi
