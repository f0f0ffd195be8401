// Row-block kernels of the bf16 decode path (rounds 3 and 4): a workgroup owns a block of activation rows for a whole SUBLAYER, keeps
// the block in LDS, and streams every weight it needs from L2 straight into MFMA operand registers.
//
//   rb_ffn5_kernel       x <- x + w_2 . relu(w_1 . LN(x) + b_1) + b_2      (PositionwiseFeedForward behind SublayerConnection,
//   (rb_ffn2_kernel)     reference TransformerModel.py:1477-1478, 1361-1377): 80 (64) rows per workgroup; the 2 048-wide hidden rows
//                        never leave the CU (a ring in LDS between producer and consumer wavefronts), the row statistics of the LayerNorm
//                        are computed while the block is staged; one launch instead of two GEMMs, no hidden tensor in HBM.  With the projection
//                        tail (rb_ffn5_kernel<PROJ>, bofi_ffn_linear_block) the same launch also computes the LayerNorm-folded projection that
//                        reads the sublayer's output next -- the next layer's q|k|v (:1454-1456), the stacked cross K|V -- from each closed block.
//   rb_attn_kernel       x <- x + W_o . attention(q, k, v) + b_o             (MultiHeadedAttention.forward :1454-1467 behind the
//                        sublayer's residual): a workgroup owns G images, wavefront = (image, head); the heads' outputs meet in LDS and
//                        the output projection runs on them in place: one launch instead of attention + GEMM, no ctx tensor.
//   rb_gemm_kernel       y <- act(W' . LN(x) + c): the LayerNorm-folded projections reading the float32 stream (96- / 64-row blocks).
//   rb_pack_frag_kernel  the weight layout all of them read.
//
// Why this shape.  d_model = 512: a sublayer's weights are 0.5 - 4 MB -- they live in L2 / Infinity Cache, and at K = 512 a tiled
// GEMM is all prologue and epilogue (docs/history/r02.md 12.9: 8 K-steps between a cold first slab and a staged epilogue; the residual
// GEMMs ran at 6 % of the MFMA peak).  Here the activation block is the resident operand and the weights are the stream:
//   * weights are stored FRAGMENT-MAJOR (rb_pack_frag_kernel): [64-column chunk][k step of 32][16-column tile][lane][8 bf16], so a
//     wavefront's weight stream is a linear run of 1-KiB wave loads, each landing in the v_mfma_f32_16x16x32_bf16 operand layout:
//     no LDS staging of weights, no barrier in the K loop, every weight byte read once per workgroup;
//   * the 8 wavefronts split the OUTPUT columns (64 each, 4 x MT accumulator tiles), so no two wavefronts load the same weight;
//   * RB_PF steps (4 KiB each) of the stream are in flight per wavefront at any time, across segment and phase boundaries
//     (the compiler's own vmcnt counting; a scheduling barrier per step keeps the loads where they are written);
//   * the activation block sits in LDS with its 16-byte chunks XOR-swizzled by row, so the ds_read_b128 of an MFMA operand
//     (16 rows x 64 B) is conflict-free.
// Measured on MI355X: dev/exp/dw_gemm_probe.hip (the bare stream + MFMA loop), profiles/r03_*, profiles/r04_*.
#include <cstdlib>

#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

extern int g_env_generation;                   // bumped by bofi_reload_env (gemm_glds.hip)

// developer aid (BOFI_RB_DBG & 16): s_memtime stamps of workgroup 0, [wave][slot], read back by bofi_rb_stamps
__device__ unsigned long long g_rb_stamps[16 * 16];
#define RB_STAMP(dbg, wave, lane, slot) do { if (((dbg) & 16) && blockIdx.x == 0 && (lane) == 0) g_rb_stamps[(wave) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)

constexpr int RB_PF = 4;             // weight-stream steps in flight per wavefront (divides 16: a segment starts at slot 0)

// byte offset of 16-byte chunk c of row r in a block of 1 024-byte rows
__device__ __forceinline__ int rb_off(int r, int c) { return r * 1024 + ((c ^ (r & 15)) << 4); }

__device__ __forceinline__ bf16x8 rb_ldw(const u32x4* p) { return __builtin_bit_cast(bf16x8, *p); }

// the first RB_PF steps of a wavefront's first segment
template <int NT, int PF = RB_PF>
__device__ __forceinline__ void rb_prime(const u32x4* seg, bf16x8 (&wb)[PF * NT]) {      // (flat: slot p, fragment nt at p*NT + nt)
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wb[p * NT + nt] = rb_ldw(seg + (p * 4 + nt) * 64);
}

// One SEGMENT of a wavefront's weight stream: 16 k-steps (K = 512) of 4 fragments = 64 output columns; acc[nt][mt] += W-tile nt .
// block rows mt*16 .. +15.  `cur` is the segment (lane offset included), `nxt` the one that follows (its first RB_PF steps are
// requested during this segment's last ones).  D[n][m]: the weight is the MFMA row operand, so lane (l15, g) ends with row
// mt*16 + l15, columns nt*16 + g*4 .. +3.
// A lane's MFMA-operand address in a block: row mt*16 + l15, chunk kb*4 + g swizzled by the row -> (rb_lane_base ^ (kb << 6)) + mt*16384:
// the swizzle only touches bits 4..9, so a k-step costs one v_xor with an inline constant and the tile index rides in the offset field.
__device__ __forceinline__ int rb_lane_base(int l15, int g) { return l15 * 1024 + (((l15 >> 2) << 6) | ((g ^ (l15 & 3)) << 4)); }

template <int MT, int NT = 4, int PF = RB_PF, int MH = MT, bool PIPE = false>      // NT < 4: the wavefront takes NT of a step's four 16-column tiles (cur / nxt point at its first one); PF divides 16;
                                                                  // MH: row tiles per operand batch (MH < MT: fewer operand registers live at a time);
                                                                  // PIPE: the NEXT k-step's block operands are requested before this step's MFMAs (MT * 4 more registers): the LDS
                                                                  // latency of a step no longer sits between its reads and its MFMAs
__device__ __forceinline__ void rb_segment(const u32x4* cur, const u32x4* nxt, bf16x8 (&wb)[PF * NT], const unsigned char* smem, int lbase,
                                           f32x4 (&acc)[NT][MT], int kstride = 256) {      // kstride: u32x4 per k-step of the stream (256; 0 = a diagnostic that re-reads step 0)
    static_assert(MT % MH == 0, "operand batches divide the row tiles");
    static_assert(!PIPE || MH == MT, "the pipelined form reads a whole step's operands at once");
    asm volatile("" : "+v"(lbase));                   // (keeps the sixteen k-step addresses from being hoisted out of the caller's loops and spilled)
    if constexpr (PIPE) {
        bf16x8 xa[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xa[0][mt] = *reinterpret_cast<const bf16x8*>(smem + lbase + mt * 16384);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            if (kb + 1 < 16) {
                const unsigned char* xn = smem + (lbase ^ ((kb + 1) << 6));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xa[(kb + 1) & 1][mt] = *reinterpret_cast<const bf16x8*>(xn + mt * 16384);
                __builtin_amdgcn_sched_barrier(0);            // (the scheduler sinks these reads behind this step's MFMAs otherwise: the point is that they fly meanwhile)
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[(kb % PF) * NT + nt], xa[kb & 1][mt], acc[nt][mt], 0, 0, 0);
            const u32x4* src = kb + PF < 16 ? cur + (kb + PF) * kstride : nxt + (kb + PF - 16) * kstride;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wb[(kb % PF) * NT + nt] = rb_ldw(src + nt * 64);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        int ad = lbase ^ (kb << 6);
        if constexpr (MT > 6) asm volatile("" : "+v"(ad));     // (eight row tiles: offsets past the 16-bit field -- computed here, per step, not sixteen steps ahead)
        const unsigned char* xp = smem + ad;
#pragma unroll
        for (int mb = 0; mb < MT; mb += MH) {
            bf16x8 xa[MH];
#pragma unroll
            for (int mt = 0; mt < MH; ++mt) xa[mt] = *reinterpret_cast<const bf16x8*>(xp + (mb + mt) * 16384);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MH; ++mt) acc[nt][mb + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[(kb % PF) * NT + nt], xa[mt], acc[nt][mb + mt], 0, 0, 0);
        }
        const u32x4* src = kb + PF < 16 ? cur + (kb + PF) * kstride : nxt + (kb + PF - 16) * kstride;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wb[(kb % PF) * NT + nt] = rb_ldw(src + nt * 64);
        __builtin_amdgcn_sched_barrier(0);            // the scheduler would sink the loads next to their use: the prefetch distance is the point
    }
}

// Closing a sublayer: out = residual + (accumulators + bias), written as WHOLE ROWS.  The accumulator layout (lane = row, 4 columns)
// makes 64-byte row pieces at best and the chip's store path is issue-bound on pieces (measured: the bf16 copy written 8 bytes per
// lane cost 4 us per workgroup, the float32 rows 16 bytes per lane another 5).  So the tiles go through LDS once: 32 block rows per
// pass, float32, row pitch 2 064 B (the 16-byte pad spreads the 16 rows of a tile over all banks), and every wavefront then owns
// whole rows of the pass: a lane holds columns lane*4 .. +3 and 256 + lane*4 .. +3 of its row -- the residual is loaded and the
// float32 row stored as 1-KiB wave accesses, the bf16 copy as 512-byte ones, and a 32-column statistics group is 8 adjacent lanes.
constexpr int RB_SPITCH = 2064;
struct RbOut {
    const float* x; int ldx;                  // residual stream in
    float* y; int ldy;                        // out (may be x: the residual of a pass is in registers before its rows are stored)
    bf16_t* yb; float* stats;                 // optional: bf16 copy [M][512], partial (sum, sum of squares) per 32 columns [M][16][2]
};

// residual values of this wavefront's RPW rows of the pass whose first block row is `prow0` (issued BEFORE the staging barrier)
template <int RPW>
__device__ __forceinline__ void rb_pass_residual(const RbOut& o, int prow0, int m0, int rows_live, int wave, int lane, float4 (&res)[RPW][2]) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = prow0 + wave * RPW + j;
        const float* xr = o.x + (size_t)(m0 + min(r, rows_live - 1)) * o.ldx + lane * 4;
        res[j][0] = *reinterpret_cast<const float4*>(xr);
        res[j][1] = *reinterpret_cast<const float4*>(xr + 256);
    }
}
// one accumulator tile (+ bias) into the staging area: tile row tr (0 / 1 of the pass), columns col .. col + 3 of row tr*16 + l15
__device__ __forceinline__ void rb_stage_tile(unsigned char* stage, int tr, int l15, int col, const f32x4& acc, const float4& bias) {
    *reinterpret_cast<float4*>(stage + (tr * 16 + l15) * RB_SPITCH + col * 4) = make_float4(acc[0] + bias.x, acc[1] + bias.y, acc[2] + bias.z, acc[3] + bias.w);
}
// after the barrier behind the staging writes: this wavefront's rows of the pass -> memory
template <int RPW>
__device__ __forceinline__ void rb_pass_store(const unsigned char* stage, const RbOut& o, int prow0, int pass_rows, int m0, int rows_live, int wave, int lane,
                                              const float4 (&res)[RPW][2]) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int lr = wave * RPW + j, r = prow0 + lr;             // row of the pass / of the block
        const bool live = lr < pass_rows && r < rows_live;
        const float4 s0 = *reinterpret_cast<const float4*>(stage + lr * RB_SPITCH + lane * 16);
        const float4 s1 = *reinterpret_cast<const float4*>(stage + lr * RB_SPITCH + 1024 + lane * 16);
        const float4 o0 = make_float4(res[j][0].x + s0.x, res[j][0].y + s0.y, res[j][0].z + s0.z, res[j][0].w + s0.w);
        const float4 o1 = make_float4(res[j][1].x + s1.x, res[j][1].y + s1.y, res[j][1].z + s1.z, res[j][1].w + s1.w);
        const size_t m = live ? (size_t)(m0 + r) : 0;
        if (live) {
            float* yr = o.y + m * o.ldy + lane * 4;
            *reinterpret_cast<float4*>(yr) = o0;
            *reinterpret_cast<float4*>(yr + 256) = o1;
            if (o.yb) {
                *reinterpret_cast<uint2*>(o.yb + m * 512 + lane * 4) = make_uint2(pack_bf16(o0.x, o0.y), pack_bf16(o0.z, o0.w));
                *reinterpret_cast<uint2*>(o.yb + m * 512 + 256 + lane * 4) = make_uint2(pack_bf16(o1.x, o1.y), pack_bf16(o1.z, o1.w));
            }
        }
        if (o.stats) {                        // groups of 32 columns = 8 adjacent lanes: groups lane / 8 and 8 + lane / 8
            const float a0 = oct_sum((o0.x + o0.y) + (o0.z + o0.w)), q0 = oct_sum((o0.x * o0.x + o0.y * o0.y) + (o0.z * o0.z + o0.w * o0.w));
            const float a1 = oct_sum((o1.x + o1.y) + (o1.z + o1.w)), q1 = oct_sum((o1.x * o1.x + o1.y * o1.y) + (o1.z * o1.z + o1.w * o1.w));
            if (live && !(lane & 7)) {
                float2* sp = reinterpret_cast<float2*>(o.stats + m * 32);
                sp[lane >> 3] = make_float2(a0, q0);
                sp[8 + (lane >> 3)] = make_float2(a1, q1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same sublayer with the two GEMMs on different wavefronts (the default).  In rb_ffn_kernel every wavefront runs w_1, then the ReLU
// epilogue, then w_2 of the same 512 hidden columns, two workgroup barriers per round: all eight wavefronts do their vector work at
// the same time (the matrix pipes idle) and every weight stream pauses at every phase change (measured: 75 us per launch where the
// 4 MB weight stream alone is ~35).  Here wavefronts 0-3 PRODUCE hidden columns (w_1, 256 per chunk, 64 each) and wavefronts 4-7 CONSUME
// them (w_2, 128 output columns each, accumulators resident for the whole block); a SIMD holds one of each, so the producer's ReLU /
// rounding epilogue runs beside the consumer's MFMAs, and neither stream waits for the other's phase.  The hidden chunks go through a
// two-slot ring in LDS, handed over with two counters per slot (LDS atomics, polled with s_sleep): no workgroup barrier in the loop.
constexpr int RB_HC = 256;                    // hidden columns per chunk
__device__ __forceinline__ void rb_wait_ge(unsigned* flag, unsigned target) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void rb_signal(unsigned* flag, int lane) {      // behind this wavefront's LDS traffic (LDS serves a wavefront in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(512) void rb_ffn2_kernel(RbFfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xt = smem;                                  // [64][512] bf16, swizzled, row pitch 1 024 B
    unsigned char* hr = smem + 65536;                          // 2 slots x [64][256] bf16, swizzled, row pitch 512 B
    float* c1s = reinterpret_cast<float*>(smem + 131072);      // [dff]
    float* cs1s = c1s + a.dff;                                 // [dff]
    float* b2s = cs1s + a.dff;                                 // [512]
    float* s_mean = b2s + 512;                                 // [64]
    float* s_rstd = s_mean + 64;                               // [64]
    unsigned* flags = reinterpret_cast<unsigned*>(s_rstd + 64);      // full[2], empty[2]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 64, nch = a.dff / RB_HC;
    const bool producer = wave < 4;
    const int w4 = wave & 3;
    if ((a.dbg & 16) && blockIdx.x == 0 && lane == 0) g_rb_stamps[(8 + wave) * 16] = __builtin_amdgcn_s_memtime();      // entry (rows 8-15: this kernel has 8 wavefronts)

    // ---- the first steps of this wavefront's weight stream, then constants and the block (as rb_ffn_kernel)
    constexpr int PPF = 4;                                     // producer: 4 steps x 4 fragments in flight (8 measured SLOWER: 66 against 53 us per workgroup)
    bf16x8 wbuf[PPF * 4];                                      // ONE ring for both roles (two arrays would both be live across the shared staging code);
                                                               // the consumer uses its first 16 entries as 2 steps x 8 fragments
    auto w1seg = [&](int c) { return a.w1p + (size_t)(c * 4 + w4) * (16 * 256) + lane; };       // 64-column chunk c*4 + w4 of w_1: 16 steps
    const u32x4* w2s = a.w2p + (size_t)(2 * w4) * (a.dff >> 5) * 256 + lane;                    // 64-column chunks 2*w4, 2*w4 + 1 of w_2: step s at + s*256
    const size_t w2j = (size_t)(a.dff >> 5) * 256;
    if (producer) rb_prime<4, PPF>(w1seg(0), wbuf);
    else {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int f = 0; f < 8; ++f) wbuf[p * 8 + f] = rb_ldw(w2s + (f >> 2) * w2j + p * 256 + (f & 3) * 64);
    }
    for (int i = tid; i < a.dff; i += 512) { c1s[i] = a.c1[i]; cs1s[i] = a.cs1[i]; }
    b2s[tid] = a.b2[tid];
    if (tid < 4) flags[tid] = 0u;
    {
        const int r = wave * 8 + (lane >> 3), sub = lane & 7, m = m0 + r;
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {                  // (two batches of eight loads: the weight ring already holds 128 registers)
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = m < a.M ? *reinterpret_cast<const float4*>(a.x + (size_t)m * a.ldx + (half * 8 + j) * 32 + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
                sq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
                *reinterpret_cast<uint2*>(xt + rb_off(r, (half * 8 + j) * 4 + (sub >> 1)) + (sub & 1) * 8) = make_uint2(pack_bf16(v[j].x, v[j].y), pack_bf16(v[j].z, v[j].w));
            }
        }
        sm = oct_sum(sm); sq = oct_sum(sq);
        if (sub == 0) {
            const float mean = sm * (1.0f / 512.0f);
            const float var = fmaxf((sq - sm * mean) * (1.0f / 511.0f), 0.f);
            s_mean[r] = mean;
            s_rstd[r] = 1.0f / (sqrtf(var) + 1e-6f);
        }
    }
    __syncthreads();

    f32x4 acc2[8][4];
    if (producer) {
        // ================= hidden chunk c: columns c*256 + w4*64 .. +63 of the block =================
        const int lbase = rb_lane_base(l15, g);
        float mu[4], rs[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { mu[mt] = s_mean[mt * 16 + l15]; rs[mt] = s_rstd[mt * 16 + l15]; }
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            f32x4 acc1[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc1[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            rb_segment<4, 4, PPF>(w1seg(c), w1seg(c + 1 < nch ? c + 1 : c), wbuf, smem, lbase, acc1);
            RB_STAMP(a.dbg, wave, lane, 2 * c);
            if (c >= 2) rb_wait_ge(flags + 2 + (c & 1), 4u * (unsigned)(c >> 1));        // the consumers are through with chunk c - 2
            unsigned char* hs = hr + (c & 1) * 32768;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int hc = c * RB_HC + w4 * 64 + nt * 16 + g * 4;
                const float4 cc = *reinterpret_cast<const float4*>(c1s + hc);
                const float4 cs = *reinterpret_cast<const float4*>(cs1s + hc);
                const int ch = w4 * 8 + nt * 2 + (g >> 1);                               // 16-byte chunk of the slot's 512-byte rows
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 t = acc1[nt][mt];
                    const float h0 = fmaxf(rs[mt] * (t[0] - mu[mt] * cs.x) + cc.x, 0.f), h1 = fmaxf(rs[mt] * (t[1] - mu[mt] * cs.y) + cc.y, 0.f);
                    const float h2 = fmaxf(rs[mt] * (t[2] - mu[mt] * cs.z) + cc.z, 0.f), h3 = fmaxf(rs[mt] * (t[3] - mu[mt] * cs.w) + cc.w, 0.f);
                    const int row = mt * 16 + l15;
                    *reinterpret_cast<uint2*>(hs + row * 512 + ((ch ^ l15) << 4) + (g & 1) * 8) = make_uint2(pack_bf16(h0, h1), pack_bf16(h2, h3));
                }
            }
            rb_signal(flags + (c & 1), lane);
            RB_STAMP(a.dbg, wave, lane, 2 * c + 1);
        }
    } else {
        // ================= output columns w4*128 .. +127, K = the hidden chunks as they arrive =================
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc2[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int hbase = 65536 + l15 * 512 + (((l15 >> 2) << 6) | ((g ^ (l15 & 3)) << 4));
        const int nsteps = nch * 8;
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            rb_wait_ge(flags + (c & 1), 4u * (unsigned)((c >> 1) + 1));                  // chunk c is in its slot
            RB_STAMP(a.dbg, wave, lane, 2 * c);
            int hb = hbase + (c & 1) * 32768;
            asm volatile("" : "+v"(hb));
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                bf16x8 xa[4];
                const unsigned char* xp = smem + (hb ^ (kb << 6));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) xa[mt] = *reinterpret_cast<const bf16x8*>(xp + mt * 8192);
#pragma unroll
                for (int f = 0; f < 8; ++f)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc2[f][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[(kb & 1) * 8 + f], xa[mt], acc2[f][mt], 0, 0, 0);
                const int sn = min(c * 8 + kb + 2, nsteps - 1);                          // (the last two steps re-read the last one: never consumed)
#pragma unroll
                for (int f = 0; f < 8; ++f) wbuf[(kb & 1) * 8 + f] = rb_ldw(w2s + (f >> 2) * w2j + (size_t)sn * 256 + (f & 3) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
            rb_signal(flags + 2 + (c & 1), lane);
            RB_STAMP(a.dbg, wave, lane, 2 * c + 1);
        }
    }
    __syncthreads();                                           // every hidden chunk consumed: x block and ring are dead
    RB_STAMP(a.dbg, 8 + wave, lane, 4);

    // ---- closing epilogue: the consumers stage their tiles, all eight wavefronts store whole rows
    const RbOut out{a.x, a.ldx, a.y, a.ldy, a.yb, a.stats_out};
    const int rows_live = min(64, a.M - m0);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        float4 res[4][2];
        rb_pass_residual<4>(out, ps * 32, m0, rows_live, wave, lane, res);
        if (!producer) {
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const int col = w4 * 128 + f * 16 + g * 4;
                const float4 bb = *reinterpret_cast<const float4*>(b2s + col);
#pragma unroll
                for (int tr = 0; tr < 2; ++tr) rb_stage_tile(smem, tr, l15, col, acc2[f][ps * 2 + tr], bb);
            }
        }
        __syncthreads();
        RB_STAMP(a.dbg, 8 + wave, lane, 5 + 2 * ps);            // (5, 7: tiles staged; 6: first pass stored)
        rb_pass_store<4>(smem, out, ps * 32, 32, m0, rows_live, wave, lane, res);
        if (ps == 0) { RB_STAMP(a.dbg, 8 + wave, lane, 6); __syncthreads(); }
    }
    RB_STAMP(a.dbg, 8 + wave, lane, 2);
    RB_STAMP(a.dbg, 8 + wave, lane, 1);                        // exit
}

// developer aid (BOFI_RB_DBG & 16): wavefronts 0 (producer) and 4 (consumer) of workgroup 0 append s_memtime stamps: [0..127] / [128..255]
#define RB3_STAMP(on, n) do { if ((on) && (n) < 126) g_rb_stamps[stamp_base + (n)++] = __builtin_amdgcn_s_memtime(); } while (0)

// ------------------------------------------------------------------------------------------------------------------------------
// The chunk loop of the LayerNorm-folded projections (rb_gemm_kernel below; also the projection tail of rb_ffn5_kernel): a block of bf16 rows
// in LDS, its row statistics, and a wavefront's 64-column chunks of the fragment-major weight.
template <bool F32OUT, int MT>
struct RbGemmCfg {
    static constexpr int BR = MT * 16;                          // rows per block
    static constexpr int SP = F32OUT ? 272 : 144;               // staging row pitch (bytes): 64 columns + 16 B
    static constexpr int TPS = (MT == 8 || MT == 5) ? 1 : MT == 6 ? (F32OUT ? 1 : 2) : (F32OUT ? 2 : 4);      // row tiles staged at a time (divides MT)
    static constexpr int STG = TPS * 16 * SP;                   // staging bytes per wavefront
    static constexpr int XT = BR * 1024;
    static constexpr int STAT = XT + 8 * STG;                   // s_mean[BR], s_rstd[BR]
    static constexpr int CST = STAT + BR * 8;                   // per wavefront [2][64]: c | cs of the current chunk
    static constexpr int LDS = CST + 8 * 512;
};

template <bool F32OUT, int MT>
__device__ __forceinline__ void rb_gemm_chunks(const RbGemmArgs& a, const unsigned char* blk, unsigned char* stage_all, float* cst, const float* s_mean,
                                               const float* s_rstd, int m0, int wave, int lane, int ch0, int chstep, bf16x8 (&wb)[RB_PF * 4]) {
    using Cfg = RbGemmCfg<F32OUT, MT>;
    constexpr int SP = Cfg::SP, TPS = Cfg::TPS;
    const int l15 = lane & 15, g = lane >> 4, nchunks = a.N >> 6;
    auto seg = [&](int ch) { return a.wp + (size_t)ch * (16 * 256) + lane; };
    constexpr bool STATS_IN_REGS = MT <= 6;                    // (128-row blocks: the 16 statistics registers are what the accumulators need -- read per pass from LDS)
    float mu[STATS_IN_REGS ? MT : 1], rs[STATS_IN_REGS ? MT : 1];
    if constexpr (STATS_IN_REGS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { mu[mt] = s_mean[mt * 16 + l15]; rs[mt] = s_rstd[mt * 16 + l15]; }
    }
    const int lbase = rb_lane_base(l15, g);
    unsigned char* stage = stage_all + wave * Cfg::STG;
    int nstamp = 0;
    float* mycst = cst + wave * 128;
    const int rows_live = min(Cfg::BR, a.M - m0);

#pragma unroll 1
    for (int ch = ch0; ch < nchunks; ch += chstep) {
        // this chunk's column constants: requested now (older than the weight prefetch), parked in LDS at the epilogue
        const float cv = a.c[ch * 64 + lane], csv = a.cs[ch * 64 + lane];
        f32x4 acc[4][MT];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        rb_segment<MT, 4, RB_PF, (MT == 5 ? 5 : MT > 4 ? MT / 2 : MT)>(seg(ch), seg(ch + chstep < nchunks ? ch + chstep : ch), wb, blk, lbase, acc);
        if (nstamp < 8) RB_STAMP(a.dbg, wave, lane, 2 * nstamp);            // chunk's MFMAs issued
        mycst[lane] = cv; mycst[64 + lane] = csv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (the row pointers of the stores hang on a lane value the compiler cannot see through and advance by additions: as invariants of the chunk
        // loop they are hoisted above it -- two registers per store -- and the accumulators spill)
        constexpr int LPR = F32OUT ? 16 : 8, RPI = 64 / LPR;
        int lnv = lane;
        asm volatile("" : "+v"(lnv));
        int rrow = lnv / LPR;                                  // row of the block this lane stores next
        unsigned char* yp = static_cast<unsigned char*>(a.y) + ((size_t)(m0 + rrow) * a.ldy + ch * 64) * (F32OUT ? 4 : 2) + (lnv % LPR) * 16;
        const size_t ystep = (size_t)RPI * a.ldy * (F32OUT ? 4 : 2);
        const unsigned char* srd = stage + rrow * SP + (lnv % LPR) * 16;
#pragma unroll
        for (int pass = 0; pass < MT / TPS; ++pass) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4 cc = *reinterpret_cast<const float4*>(mycst + nt * 16 + g * 4);
                const float4 cs = *reinterpret_cast<const float4*>(mycst + 64 + nt * 16 + g * 4);
#pragma unroll
                for (int mh = 0; mh < TPS; ++mh) {
                    const int mt = pass * TPS + mh;
                    const f32x4 t = acc[nt][mt];
                    const float mu_ = STATS_IN_REGS ? mu[STATS_IN_REGS ? mt : 0] : s_mean[mt * 16 + l15], rs_ = STATS_IN_REGS ? rs[STATS_IN_REGS ? mt : 0] : s_rstd[mt * 16 + l15];
                    float v0 = rs_ * (t[0] - mu_ * cs.x) + cc.x, v1 = rs_ * (t[1] - mu_ * cs.y) + cc.y;
                    float v2 = rs_ * (t[2] - mu_ * cs.z) + cc.z, v3 = rs_ * (t[3] - mu_ * cs.w) + cc.w;
                    if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    unsigned char* sp = stage + (mh * 16 + l15) * SP + (nt * 16 + g * 4) * (F32OUT ? 4 : 2);
                    if constexpr (F32OUT) *reinterpret_cast<float4*>(sp) = make_float4(v0, v1, v2, v3);
                    else *reinterpret_cast<uint2*>(sp) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // whole row pieces: bf16 8 lanes x 16 B = a row's 128 B (8 rows per instruction); float32 16 lanes x 16 B = 256 B (4 rows)
#pragma unroll
            for (int it = 0; it < TPS * 16 / RPI; ++it, yp += ystep, rrow += RPI) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(srd + it * RPI * SP);
                if (rrow < rows_live) *reinterpret_cast<u32x4*>(yp) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (nstamp < 8) RB_STAMP(a.dbg, wave, lane, 2 * nstamp + 1);        // chunk stored
        ++nstamp;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same sublayer on 80-ROW blocks, able to walk several of them (round 4; the default from 4 096 rows on).
// What round 4 measured on the 64-row kernel first (profiles/r04_ffn_*.txt, dev/exp/mb_rowblock.py ffn4, dev/exp/rb_ffn_stamps.py):
//   * its workgroup lives 110 k cycles of which 65.5 k are MFMA issue -- staging 11-18 k before the first chunk exists, closing epilogue 12.5 k
//     (residual in, rows out through a workgroup-wide LDS transposition, three barriers) -- and a row-block workgroup owns its CU;
//   * a persistent form that stages block j + 1 beside the consumers' last chunks of block j and closes blocks without loads (below) cut the
//     CU-time per block from 110 k to 87 k cycles -- and changed NOTHING with four launches in flight (44-46 us per 11 520-row launch, four
//     concurrent feed-forward streams, one, two or three blocks per workgroup): the chip is not CU-time-bound there.  Neither do 20 % fewer
//     weight bytes per row (these 80-row blocks) move that four-stream figure; what they move is the whole decode, +3 %, one block per
//     workgroup (more blocks per workgroup lose: the launch's latency chain grows).  The 64-row persistent kernel left the library again.
// What is kept from it:
//   * the consumers' accumulators START from the residual rows (x, loaded in the accumulator layout straight into the accumulator
//     registers -- for a next block right behind the staging of this block's tiles, register by register), so closing a block needs no
//     loads and no workgroup barrier: tiles go through 2.3 KB of the wavefront's own LDS (16 rows x 32 columns, 128-byte row pieces) + b_2
//     (from LDS: a global load there would sit in the vmcnt queue in front of the residual loads and every store would wait for those);
//   * the producers stage the next block as soon as THEY are through with this one's last chunk (a producers-only barrier in LDS);
//   * nothing in the closing code may be a loop invariant of the block loop (row pointers, bias): hoisted above the chunk loop it is spilled,
//     and a scratch reload is an s_waitcnt vmcnt(0) that drains the loads just issued (measured: 20 k cycles per block instead of 4 k);
//   * the consumer's weight stream as buffer loads (scalar resource + one vector offset): with 64-bit global addresses the allocator spills a
//     weight fragment inside the MFMA loop at 160 accumulator + 64 ring registers.
// Five row tiles are what the consumer's registers hold (8 x 5 accumulator tiles = 160, + the 64 of its weight ring); the hidden ring is three
// slots of 128 columns (60 KB) beside the 80 KB block.  Sums: a row's result is (x + sum of the MFMA partial sums in chunk order) + b_2,
// whatever block or workgroup it sits in.
//   producers (wavefronts 0-3): stage the block (20 rows each), then per chunk 32 hidden columns each: w_1 tiles (c*2 + (w >> 1), (w & 1)*2 ..+1)
//   consumers (wavefronts 4-7): 128 output columns each, K = the chunks as they arrive.
constexpr int R5_ROWS = 80, R5_HC = 128, R5_SLOTS = 3, R5_SLOT = R5_ROWS * 256;
#ifndef R5_PIPE
#define R5_PIPE true
#endif
constexpr int R5_HR = R5_ROWS * 1024;                                  // hidden ring behind the block
constexpr int R5_CST = R5_HR + R5_SLOTS * R5_SLOT, R5_CSTW = 16 * 144; // consumer staging: [4 wavefronts][16 rows][144 B]
constexpr int R5_PC = R5_CST + 4 * R5_CSTW;                            // producer constants: [4 wavefronts][c[32] | cs[32]] floats
constexpr int R5_STAT = R5_PC + 4 * 256;                               // s_mean[80], s_rstd[80]
constexpr int R5_FLAG = R5_STAT + 2 * R5_ROWS * 4;                     // full[3], empty[3], producer barrier
constexpr int R5_B2 = R5_FLAG + 64;                                    // b_2 [512]
constexpr int R5_LDS = R5_B2 + 2048;
constexpr int R5_PJS = R5_LDS;                                         // projection tail: per consumer wavefront (sum, sum of squares) of its 128 columns of every row [4][80]
constexpr int R5_LDS_PJ = R5_PJS + 4 * R5_ROWS * 8;
constexpr int R5_BO = R5_LDS_PJ;                                       // head segment: b_o [512]
constexpr int R5_LDS_HEAD = R5_BO + 2048;
static_assert(R5_LDS_HEAD <= 160 * 1024, "one workgroup per CU");
static_assert(8 * RbGemmCfg<false, 5>::STG + 8 * 512 <= R5_SLOTS * R5_SLOT, "the tail's staging and constants fit the hidden ring");

// The projection tail of rb_ffn5_kernel<PROJ>: pj_y = W_pj . LN(y) + c_pj for the block the consumers just closed (bf16 rows back in the block's LDS, their
// share of the row sums beside it), all eight wavefronts, chunks w, w + 8, ... as rb_gemm_kernel; staging and chunk constants live in the hidden ring.
// The wavefront's first weight steps are requested before the barrier: the producers get here while the consumers still close.  Called at the END OF EACH
// ROLE'S BRANCH (both reach the one barrier): behind the join of the two, the allocator spilled 43-340 registers in the feed-forward loops.
template <int ROLE>      // (a different instruction stream per caller: identical tails of the two branches are merged behind their join, see above)
__device__ __forceinline__ void rb_ffn5_proj_tail(const u32x4* pj_wp, const float* pj_c, const float* pj_cs, void* pj_y, int pj_ldy, int pj_N, int M, bf16x8 (&wb)[RB_PF * 4]) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_mean = reinterpret_cast<float*>(smem + R5_STAT);
    float* s_rstd = s_mean + R5_ROWS;
    int tid = threadIdx.x;
    if constexpr (ROLE == 0) asm volatile("; projection tail, producers" : "+v"(tid));      // (nothing of the tail is computed ahead of the feed-forward loops and kept in registers through them)
    else asm volatile("; projection tail, consumers" : "+v"(tid));
    const int wave = tid >> 6, lane = tid & 63;
    rb_prime<4>(pj_wp + (size_t)wave * (16 * 256) + lane, wb);
    __syncthreads();
    {   // every wavefront derives all 80 row statistics itself (the same values to the same words)
        const float2* pjs = reinterpret_cast<const float2*>(smem + R5_PJS);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * 64 + lane;
            if (r < R5_ROWS) {
                const float2 p0 = pjs[r], p1 = pjs[R5_ROWS + r], p2 = pjs[2 * R5_ROWS + r], p3 = pjs[3 * R5_ROWS + r];
                const float sm = (p0.x + p1.x) + (p2.x + p3.x), sq = (p0.y + p1.y) + (p2.y + p3.y);
                const float mean = sm * (1.0f / 512.0f);
                const float var = fmaxf((sq - sm * mean) * (1.0f / 511.0f), 0.f);
                s_mean[r] = mean;
                s_rstd[r] = 1.0f / (sqrtf(var) + 1e-6f);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    RbGemmArgs ga{};
    ga.wp = pj_wp; ga.c = pj_c; ga.cs = pj_cs; ga.y = pj_y; ga.ldy = pj_ldy; ga.M = M; ga.N = pj_N;
    rb_gemm_chunks<false, 5>(ga, smem, smem + R5_HR, reinterpret_cast<float*>(smem + R5_HR + 8 * RbGemmCfg<false, 5>::STG), s_mean, s_rstd,
                             (int)blockIdx.x * R5_ROWS, wave, lane, wave, 8, wb);
}

template <bool EXTRA, bool STAMPS, bool PROJ = false, bool ONE = PROJ, bool HEAD = false>   // ONE: one block per workgroup (no block loop: nothing is kept for a next block); EXTRA: the optional bf16 copy / partial sums are written; STAMPS: the developer timeline (BOFI_RB_DBG & 16);
                                                        // PROJ: the projection tail (a.pj_*; one block per workgroup)
                                                        // HEAD (round 6): the attention sublayer's W_o + residual in FRONT of the feed-forward sublayer (a.head_*; one block per workgroup): the producers stage the
                                                        // attention core's context rows as the block, the CONSUMERS -- whose accumulators start from the residual rows anyway -- run one 0.5-MB W_o segment over it
                                                        // (x1 = x + W_o . ctx + b_o stays in their accumulators as the feed-forward's residual), put x1 back into the block as bf16 with their share of its row sums,
                                                        // and the feed-forward runs as without a head.  The attention kernel in front of this one is then a light core (no W_o, no 80-120 KB of LDS, no idle weight
                                                        // stream), W_o runs at 80 rows per weight byte and x1 never makes the round trip through memory.
__global__ __launch_bounds__(512) void rb_ffn5_kernel(RbFfnArgs a) {
    static_assert(!HEAD || (ONE && !EXTRA && !STAMPS), "the head segment: one block per workgroup, plain outputs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xt = smem;                                  // [80][512] bf16, swizzled, row pitch 1 024 B
    unsigned char* hr = smem + R5_HR;                          // 3 slots x [80][128] bf16, swizzled, row pitch 256 B
    float* s_mean = reinterpret_cast<float*>(smem + R5_STAT);
    float* s_rstd = s_mean + R5_ROWS;
    unsigned* flags = reinterpret_cast<unsigned*>(smem + R5_FLAG);      // full[0..2], empty[3..5], producer barrier [6]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int nblocks = (a.M + R5_ROWS - 1) / R5_ROWS, nch = a.dff / R5_HC;
    const bool producer = wave < 4;
    const int w4 = wave & 3;
    const bool stamps = STAMPS && blockIdx.x == 0 && lane == 0 && w4 == 0;
    const int stamp_base = __builtin_amdgcn_readfirstlane(producer ? 0 : 128);
    int nst = 0;
    RB3_STAMP(stamps, nst);                                    // entry
    unsigned long long rt0 = 0;                                // (the 100 MHz constant clock beside the shader clock: the clock the chip holds under this load)
    if (stamps) rt0 = __builtin_amdgcn_s_memrealtime();

    // (developer diagnostic, stamped build only, BOFI_RB_DBG & 32: every wavefront re-reads the FIRST step of its weight stream -- the kernel without its
    // weight traffic, results meaningless)
    const int wmul = (STAMPS && (a.dbg & 32)) ? 0 : 1;
    bf16x8 wbuf[16];                                           // producer: 8 steps x 2 fragments; consumer: 2 steps x 8 fragments
    auto w1seg = [&](int c) { return a.w1p + (size_t)((c * 2 + (w4 >> 1)) * wmul) * (16 * 256) + (w4 & 1) * 128 + lane; };
    // the consumer's weight stream as BUFFER loads: a scalar resource over the wavefront's two 64-column chunks of w_2 (one after the other), the step in
    // a scalar offset, the lane's 16 bytes in ONE vector register -- global loads cost a 64-bit address pair per base, and at 160 accumulator + 64
    // ring registers the allocator then spills a weight fragment inside the loop
    const size_t w2j = (size_t)(a.dff >> 5) * 4096;           // bytes from a chunk to the next
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(a.w2p + (size_t)(2 * __builtin_amdgcn_readfirstlane(w4)) * (a.dff >> 5) * 256), 0, (int)(2 * w2j), 0x00020000);
    const int lo16 = lane * 16;
    auto w2frag = [&](int step, int f) {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2r, lo16 + (f & 3) * 1024, (int)((f >> 2) * w2j) + step * 4096 * wmul, 0));
    };
    // (HEAD) the consumer's W_o stream, the same way: its two 64-column chunks of W_o [8 chunks][16 steps][4 tiles][64 lanes][16 B] one after the other
    const __amdgpu_buffer_rsrc_t wor = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>((HEAD ? a.head_wop : a.w2p) + (size_t)(2 * __builtin_amdgcn_readfirstlane(w4)) * (16 * 256)), 0, 2 * 65536, 0x00020000);
    auto wofrag = [&](int step, int f) {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wor, lo16 + (f & 3) * 1024, (f >> 2) * 65536 + step * 4096, 0));
    };
    if (producer) rb_prime<2, 8>(w1seg(0), wbuf);
    else {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int f = 0; f < 8; ++f) wbuf[p * 8 + f] = HEAD ? wofrag(p, f) : w2frag(p, f);
    }
    if (tid < 16) flags[tid] = 0u;                             // (HEAD: [7] context block staged, [8] consumers through with it, [9] x1 back in the block)
    reinterpret_cast<float*>(smem + R5_B2)[tid] = a.b2[tid];
    if constexpr (HEAD) reinterpret_cast<float*>(smem + R5_BO)[tid] = a.head_bo[tid];
    __syncthreads();                                           // (the only workgroup barrier of the kernel)

    // (experiment, BOFI_RB_DBG & 64 / & 128: static issue priority for the producer / consumer wavefronts -- the two of a SIMD share its matrix pipe)
    if (producer ? (a.dbg & 64) : (a.dbg & 128)) __builtin_amdgcn_s_setprio(1);
    if (producer) {
        const int lbase = rb_lane_base(l15, g);
        float* mycst = reinterpret_cast<float*>(smem + R5_PC) + w4 * 64;
        unsigned q = 0, pb = 0;                                // chunks handed over so far / producer-barrier generation
#pragma unroll 1
        for (int blk = blockIdx.x; blk < nblocks; blk = ONE ? nblocks : blk + (int)gridDim.x) {      // (ONE: no back edge, nothing kept for a next block)
            const int m0 = blk * R5_ROWS;
            if (blk != (int)blockIdx.x) { ++pb; rb_signal(flags + 6, lane); rb_wait_ge(flags + 6, 4u * pb); }      // every producer is through with the previous block
            RB3_STAMP(stamps, nst);                            // staging starts
            if constexpr (HEAD) {
                // the attention core's context rows ARE the block (bf16 already: a copy into the swizzled layout); then the consumers run W_o over it and hand x1 back
                {   // wavefront w4: rows 20*w4 .. +19, eight at a time, eight lanes per row, eight 16-byte chunks per lane.  ALL THREE passes' loads are requested before the
                    // first row is written (96 registers the producers do not need yet): one memory round trip in front of the consumers' W_o segment instead of three
                    u32x4 v[3][8];
#pragma unroll
                    for (int pass = 0; pass < 3; ++pass) {
                        const int lr = pass * 8 + (lane >> 3), sub = lane & 7, m = m0 + w4 * 20 + lr;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            v[pass][j] = (lr < 20 && m < a.M) ? *reinterpret_cast<const u32x4*>(a.head_ctx + (size_t)m * a.head_ldc + (j * 8 + sub) * 8) : u32x4{0u, 0u, 0u, 0u};
                    }
                    __builtin_amdgcn_sched_barrier(0);             // (keeps the stores below from being interleaved with -- and waiting between -- the loads)
#pragma unroll
                    for (int pass = 0; pass < 3; ++pass) {
                        const int lr = pass * 8 + (lane >> 3), r = w4 * 20 + lr, sub = lane & 7;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (lr < 20) *reinterpret_cast<u32x4*>(xt + rb_off(r, j * 8 + sub)) = v[pass][j];
                    }
                }
                rb_signal(flags + 7, lane);                    // (the consumers wait for all four producers' rows)
                rb_wait_ge(flags + 9, 4u);                     // x1 is back in the block as bf16, the consumers' partial row sums beside it
                const float2* pjs = reinterpret_cast<const float2*>(smem + R5_PJS);
#pragma unroll
                for (int i = 0; i < 2; ++i) {                  // (every producer derives the statistics itself: the same values to the same words -- the tail reads them too)
                    const int r = i * 64 + lane;
                    if (r < R5_ROWS) {
                        const float2 p0 = pjs[r], p1 = pjs[R5_ROWS + r], p2 = pjs[2 * R5_ROWS + r], p3 = pjs[3 * R5_ROWS + r];
                        const float sm = (p0.x + p1.x) + (p2.x + p3.x), sq = (p0.y + p1.y) + (p2.y + p3.y);
                        const float mean = sm * (1.0f / 512.0f);
                        const float var = fmaxf((sq - sm * mean) * (1.0f / 511.0f), 0.f);
                        s_mean[r] = mean;
                        s_rstd[r] = 1.0f / (sqrtf(var) + 1e-6f);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {              // wavefront w4: rows 20*w4 .. +19, eight at a time, eight lanes per row
                const int lr = pass * 8 + (lane >> 3), r = w4 * 20 + lr, sub = lane & 7, m = m0 + r;
                const bool mine = lr < 20;
                float4 v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    v[j] = (mine && m < a.M) ? *reinterpret_cast<const float4*>(a.x + (size_t)m * a.ldx + j * 32 + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
                    sq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
                    if (mine) *reinterpret_cast<uint2*>(xt + rb_off(r, j * 4 + (sub >> 1)) + (sub & 1) * 8) = make_uint2(pack_bf16(v[j].x, v[j].y), pack_bf16(v[j].z, v[j].w));
                }
                sm = oct_sum(sm); sq = oct_sum(sq);
                if (mine && sub == 0) {
                    const float mean = sm * (1.0f / 512.0f);
                    const float var = fmaxf((sq - sm * mean) * (1.0f / 511.0f), 0.f);
                    s_mean[r] = mean;
                    s_rstd[r] = 1.0f / (sqrtf(var) + 1e-6f);
                }
            }
            ++pb; rb_signal(flags + 6, lane); rb_wait_ge(flags + 6, 4u * pb);                                      // block and statistics complete
            }
            RB3_STAMP(stamps, nst);                            // block staged
            float mu[5], rs[5];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) { mu[mt] = s_mean[mt * 16 + l15]; rs[mt] = s_rstd[mt * 16 + l15]; }
#pragma unroll 1
            for (int c = 0; c < nch; ++c, ++q) {
                const int col = c * R5_HC + w4 * 32 + (lane & 31);
                const float cv = lane < 32 ? a.c1[col] : a.cs1[col];                          // (requested before the segment's weight loads)
                f32x4 acc1[2][5];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) acc1[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (STAMPS) rb_segment<5, 2, 8, 5, R5_PIPE>(w1seg(c), w1seg(c + 1 < nch ? c + 1 : 0), wbuf, smem, lbase, acc1, 256 * wmul);
                else rb_segment<5, 2, 8, 5, R5_PIPE>(w1seg(c), w1seg(c + 1 < nch ? c + 1 : 0), wbuf, smem, lbase, acc1);
                if (!(c & 1)) RB3_STAMP(stamps, nst);          // segment done
                mycst[lane] = cv;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const unsigned slot = q % R5_SLOTS, round = q / R5_SLOTS;
                if (round) rb_wait_ge(flags + 3 + slot, 4u * round);                         // the consumers are through with chunk q - 3
                unsigned char* hs = hr + slot * R5_SLOT;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float4 cc = *reinterpret_cast<const float4*>(mycst + nt * 16 + g * 4);
                    const float4 cs = *reinterpret_cast<const float4*>(mycst + 32 + nt * 16 + g * 4);
                    const int ch = w4 * 4 + nt * 2 + (g >> 1);                               // 16-byte chunk of the slot's 256-byte rows
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) {
                        const f32x4 t = acc1[nt][mt];
                        const float h0 = fmaxf(rs[mt] * (t[0] - mu[mt] * cs.x) + cc.x, 0.f), h1 = fmaxf(rs[mt] * (t[1] - mu[mt] * cs.y) + cc.y, 0.f);
                        const float h2 = fmaxf(rs[mt] * (t[2] - mu[mt] * cs.z) + cc.z, 0.f), h3 = fmaxf(rs[mt] * (t[3] - mu[mt] * cs.w) + cc.w, 0.f);
                        const int row = mt * 16 + l15;
                        *reinterpret_cast<uint2*>(hs + row * 256 + ((ch ^ l15) << 4) + (g & 1) * 8) = make_uint2(pack_bf16(h0, h1), pack_bf16(h2, h3));
                    }
                }
                rb_signal(flags + slot, lane);
                if (!(c & 1)) RB3_STAMP(stamps, nst);          // chunk handed over
            }
        }
        if constexpr (PROJ) rb_ffn5_proj_tail<0>(a.pj_wp, a.pj_c, a.pj_cs, a.pj_y, a.pj_ldy, a.pj_N, a.M, wbuf);      // (the weights fly while the consumers close the block)
    } else {
        const int nsteps = nch * 4;
        const int er = lane >> 3, ec = lane & 7;               // closing layout: row it*8 + er of a 16-row tile, 16-byte piece ec of its 32 columns
        // the accumulators start from the residual rows (dead rows of a ragged last block: the batch's last row -- never stored)
        f32x4 acc2[8][5];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            const float* xn = a.x + (size_t)min((int)blockIdx.x * R5_ROWS + mt * 16 + l15, a.M - 1) * a.ldx + w4 * 128 + g * 4;
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const float4 v = *reinterpret_cast<const float4*>(xn + f * 16);
                acc2[f][mt] = f32x4{v.x, v.y, v.z, v.w};
            }
        }
        if constexpr (HEAD) {
            // ---- head segment: acc2 (= x) += W_o . ctx over the staged context block: 16 k-steps of this wavefront's 128 output columns
            rb_wait_ge(flags + 7, 4u);
            auto wo_steps = [&](int k4, bool more) {            // four k-steps of the segment; `more`: the ring is refilled two steps ahead (false: the segment's last two steps)
                int hl = l15, hg = g;
                asm volatile("" : "+v"(hl), "+v"(hg));
                const int lb = hl * 1024 + (((hl >> 2) << 6) | ((hg ^ (hl & 3)) << 4));
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const int kb = k4 * 4 + kq;
                    const unsigned char* xp = smem + (lb ^ (kb << 6));
#pragma unroll
                    for (int mb = 0; mb < 5; mb += 2) {
                        bf16x8 xa[2];
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2)
                            if (mb + m2 < 5) xa[m2] = *reinterpret_cast<const bf16x8*>(xp + (mb + m2) * 16384);
#pragma unroll
                        for (int f = 0; f < 8; ++f)
#pragma unroll
                            for (int m2 = 0; m2 < 2; ++m2)
                                if (mb + m2 < 5) acc2[f][mb + m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[(kq & 1) * 8 + f], xa[m2], acc2[f][mb + m2], 0, 0, 0);
                    }
                    if (more || kq < 2) {
#pragma unroll
                        for (int f = 0; f < 8; ++f) wbuf[(kq & 1) * 8 + f] = wofrag(kb + 2, f);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll 1
            for (int k4 = 0; k4 < 3; ++k4) wo_steps(k4, true);
            wo_steps(3, false);                                 // (the w_2 ring is primed behind the hand-back below: its 64 registers are free for it, and the first hidden chunk is a segment away anyway)
            rb_signal(flags + 8, lane);                        // this wavefront is through with the context block ...
            // x1 = acc2 + b_o
            {
                const float* bol = reinterpret_cast<const float*>(smem + R5_BO) + w4 * 128 + g * 4;
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    const float4 bb = *reinterpret_cast<const float4*>(bol + f * 16);
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) { acc2[f][mt][0] += bb.x; acc2[f][mt][1] += bb.y; acc2[f][mt][2] += bb.z; acc2[f][mt][3] += bb.w; }
                }
            }
            rb_wait_ge(flags + 8, 4u);                         // ... and so are the other three: the block may take x1
            {
                float2* pjs = reinterpret_cast<float2*>(smem + R5_PJS) + w4 * R5_ROWS;
                int lv = l15, gv = g;
                asm volatile("" : "+v"(lv), "+v"(gv));
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    const int r = mt * 16 + lv;
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int f = 0; f < 8; ++f) {
                        const f32x4 t = acc2[f][mt];
                        sm += (t[0] + t[1]) + (t[2] + t[3]);
                        sq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
                        *reinterpret_cast<uint2*>(xt + rb_off(r, w4 * 16 + f * 2 + (gv >> 1)) + (gv & 1) * 8) = make_uint2(pack_bf16(t[0], t[1]), pack_bf16(t[2], t[3]));
                    }
                    sm = xor32_sum(xor16_sum(sm)); sq = xor32_sum(xor16_sum(sq));
                    if (gv == 0) pjs[r] = make_float2(sm, sq);
                }
            }
            rb_signal(flags + 9, lane);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int f = 0; f < 8; ++f) wbuf[p * 8 + f] = w2frag(p, f);
        }
        unsigned q = 0;
#pragma unroll 1
        for (int blk = blockIdx.x; blk < nblocks; blk = ONE ? nblocks : blk + (int)gridDim.x) {      // (ONE: no back edge, nothing kept for a next block)
            const int m0 = blk * R5_ROWS, rows_live = min(R5_ROWS, a.M - m0);
            const int blk_next = ONE ? nblocks : blk + (int)gridDim.x;
#pragma unroll 1
            for (int c = 0; c < nch; ++c, ++q) {
                const unsigned slot = q % R5_SLOTS, round = q / R5_SLOTS;
                rb_wait_ge(flags + slot, 4u * (round + 1));                                 // chunk q is in its slot
                if (!(c & 1)) RB3_STAMP(stamps, nst);          // chunk arrived
                int hl = l15, hg = g;                          // (recomputed per chunk: kept across the loop it is spilled, and its reload drains the weight ring)
                asm volatile("" : "+v"(hl), "+v"(hg));
                int hb = R5_HR + (int)slot * R5_SLOT + hl * 256 + (((hl >> 2) << 6) | ((hg ^ (hl & 3)) << 4));
                asm volatile("" : "+v"(hb));
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const unsigned char* xp = smem + (hb ^ (kb << 6));
#pragma unroll
                    for (int mb = 0; mb < 5; mb += 2) {         // (row tiles 0-1, 2-3, 4: few operand registers -- the kernel sits at 256)
                        bf16x8 xa[2];
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2)
                            if (mb + m2 < 5) xa[m2] = *reinterpret_cast<const bf16x8*>(xp + (mb + m2) * 4096);
#pragma unroll
                        for (int f = 0; f < 8; ++f)
#pragma unroll
                            for (int m2 = 0; m2 < 2; ++m2)
                                if (mb + m2 < 5) acc2[f][mb + m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[(kb & 1) * 8 + f], xa[m2], acc2[f][mb + m2], 0, 0, 0);
                    }
                    int sn = c * 4 + kb + 2;                                                // (the stream wraps: the next block reads the same weights)
                    sn = sn >= nsteps ? sn - nsteps : sn;
#pragma unroll
                    for (int f = 0; f < 8; ++f) wbuf[(kb & 1) * 8 + f] = w2frag(sn, f);
                    __builtin_amdgcn_sched_barrier(0);
                }
                rb_signal(flags + 3 + slot, lane);
                if (!(c & 1)) RB3_STAMP(stamps, nst);          // chunk consumed
            }
            // ---- close the block: 16 rows x 32 columns of the wavefront's own LDS at a time (+ b_2) -> 128-byte row pieces -> memory; the freed
            // accumulator registers take the next block's residual rows at once
            int erv = er, ecv = ec, gv = g, lv = l15;
            asm volatile("" : "+v"(erv), "+v"(ecv), "+v"(gv), "+v"(lv));
            const int ws = __builtin_amdgcn_readfirstlane(w4);
            const size_t ystep = (size_t)8 * a.ldy;
            float* yp = a.y + (size_t)(m0 + erv) * a.ldy + ws * 128 + ecv * 4;
            int rr = erv;                                      // row of the block this lane closes next
            unsigned char* sw = smem + R5_CST + ws * R5_CSTW + lv * 144 + gv * 16;
            const unsigned char* sr = smem + R5_CST + ws * R5_CSTW + erv * 144 + ecv * 16;
            const float* b2l = reinterpret_cast<const float*>(smem + R5_B2) + ws * 128 + ecv * 4;
            // (projection tail: the closed rows also go back into the block as bf16 -- every producer is through with it once the last chunk is full --
            // and the wavefront's share of their LayerNorm sums into LDS)
            unsigned char* xw = smem + erv * 1024 + (ecv & 1) * 8;
            const int xc = ws * 16 + (ecv >> 1);
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                float ps1[2] = {0.f, 0.f}, ps2[2] = {0.f, 0.f};
                const float* xn = a.x + (size_t)min(blk_next * R5_ROWS + mt * 16 + lv, a.M - 1) * a.ldx + ws * 128 + gv * 4;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {               // columns qq*32 .. +31 of the wavefront's 128: fragments 2 qq, 2 qq + 1
#pragma unroll
                    for (int fq = 0; fq < 2; ++fq) {
                        const f32x4 t = acc2[qq * 2 + fq][mt];
                        *reinterpret_cast<float4*>(sw + fq * 64) = make_float4(t[0], t[1], t[2], t[3]);
                    }
                    if (blk_next < nblocks) {
#pragma unroll
                        for (int fq = 0; fq < 2; ++fq) {
                            const float4 v = *reinterpret_cast<const float4*>(xn + (qq * 2 + fq) * 16);
                            acc2[qq * 2 + fq][mt] = f32x4{v.x, v.y, v.z, v.w};
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const float4 bb = *reinterpret_cast<const float4*>(b2l + qq * 32);
                    float4 sv[2];
#pragma unroll
                    for (int it = 0; it < 2; ++it) sv[it] = *reinterpret_cast<const float4*>(sr + it * 8 * 144);
                    float* yq = yp;
                    int rq = rr;
#pragma unroll
                    for (int it = 0; it < 2; ++it, yq += ystep, rq += 8) {
                        const float4 o = make_float4(sv[it].x + bb.x, sv[it].y + bb.y, sv[it].z + bb.z, sv[it].w + bb.w);
                        const bool live = rq < rows_live;
                        if (live) *reinterpret_cast<float4*>(yq + qq * 32) = o;
                        if constexpr (PROJ) {                 // block row mt*16 + it*8 + er (its low four bits: it*8 + er), 16-byte chunk xc + qq*4
                            *reinterpret_cast<uint2*>(xw + (mt * 16 + it * 8) * 1024 + (((xc + qq * 4) ^ (it * 8 + erv)) << 4)) = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
                            ps1[it] += (o.x + o.y) + (o.z + o.w);
                            ps2[it] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                        }
                        if constexpr (EXTRA) {
                            const size_t m = live ? (size_t)(m0 + rq) : 0;
                            const int cb = ws * 128 + qq * 32 + ecv * 4;
                            if (live && a.yb) *reinterpret_cast<uint2*>(a.yb + m * 512 + cb) = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
                            if (a.stats_out) {                 // a 32-column group = the 8 lanes of a row piece
                                const float s1 = oct_sum((o.x + o.y) + (o.z + o.w)), s2 = oct_sum((o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w));
                                if (live && !ecv) reinterpret_cast<float2*>(a.stats_out + m * 32)[cb >> 5] = make_float2(s1, s2);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();           // (the next group rewrites the staging rows)
                }
                if constexpr (PROJ) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const float s1 = oct_sum(ps1[it]), s2 = oct_sum(ps2[it]);
                        if (!ecv) reinterpret_cast<float2*>(smem + R5_PJS)[ws * R5_ROWS + rr + it * 8] = make_float2(s1, s2);
                    }
                }
                yp += 2 * ystep; rr += 16;
            }
            RB3_STAMP(stamps, nst);                            // block closed
        }
        if constexpr (PROJ) rb_ffn5_proj_tail<1>(a.pj_wp, a.pj_c, a.pj_cs, a.pj_y, a.pj_ldy, a.pj_N, a.M, wbuf);
    }
    if (stamps && nst < 126) {
        g_rb_stamps[stamp_base + nst] = 0ull;                   // terminator
        g_rb_stamps[stamp_base + 126] = __builtin_amdgcn_s_memrealtime() - rt0;      // 10 ns ticks from entry to exit ...
        g_rb_stamps[stamp_base + 127] = __builtin_amdgcn_s_memtime();               // ... and the shader clock at exit (entry: stamp 0)
    }
}

int launch_rb_ffn(const RbFfnArgs& a, hipStream_t st) {
    if (a.M < 1 || a.dff < 512 || a.dff % 512 || a.dff > 2560 || !a.x || !a.w1p || !a.c1 || !a.cs1 || !a.w2p || !a.b2 || !a.y || a.ldx % 4 || a.ldy % 4)
        return BOFI_ERR_ARG;
    const size_t lds = 131072 + (size_t)a.dff * 8 + 2048 + 512 + 64;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<false, false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_ffn5_kernel<false, false, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return BOFI_ERR_HIP;
        attr_set = true;
    }
    // knobs (read again after bofi_reload_env): BOFI_RB_FFN_V = 5 (default): 80-row blocks -- +3 % on the decode with launches in flight; 2: one
    // 64-row block per workgroup (round 3's kernel) -- 6-12 us faster per launch when ONE decode runs alone (62 against 68 us at 11 520 rows);
    // BOFI_RB_FFN_BPW = row blocks a workgroup of the 80-row kernel walks (default 1; grid = blocks / that)
    static int env_seen = -1, version = 5, bpw = 1, v5_rows = 0, forced = 0, one_on = 1;
    if (env_seen != g_env_generation) {
        const char* e = getenv("BOFI_RB_FFN_V"); version = e ? atoi(e) : 5; forced = e != nullptr;
        e = getenv("BOFI_RB_FFN_BPW"); bpw = e ? max(1, atoi(e)) : 1;
        e = getenv("BOFI_RB_FFN_ONE"); one_on = e ? atoi(e) : 1;           // 0: the block-walking build of the kernel also at one block per workgroup
        e = getenv("BOFI_RB_FFN_V5_ROWS"); v5_rows = e ? atoi(e) : 0;      // rows from which the 80-row kernel runs (below: the 64-row kernel)
        env_seen = g_env_generation;
    }
    RbFfnArgs b = a;
    { const char* e = getenv("BOFI_RB_DBG"); b.dbg = e ? atoi(e) : 0; }
    if (a.pj_wp && (!a.pj_c || !a.pj_cs || !a.pj_y || a.pj_N < 512 || a.pj_N % 64 || a.pj_ldy % 8 || a.yb || a.stats_out)) return BOFI_ERR_ARG;
    if (a.head_wop) {                                         // with the attention sublayer's W_o + residual in front (and, optionally, the projection tail behind): one block per workgroup
        if (!a.head_ctx || !a.head_bo || a.head_ldc % 8 || a.yb || a.stats_out) return BOFI_ERR_ARG;
        if (a.pj_wp) hipLaunchKernelGGL((rb_ffn5_kernel<false, false, true, true, true>), dim3((a.M + R5_ROWS - 1) / R5_ROWS), dim3(512), R5_LDS_HEAD, st, b);
        else hipLaunchKernelGGL((rb_ffn5_kernel<false, false, false, true, true>), dim3((a.M + R5_ROWS - 1) / R5_ROWS), dim3(512), R5_LDS_HEAD, st, b);
        g_gemm_flops += 2.0 * a.M * 512.0 * 512.0 + (a.pj_wp ? 2.0 * a.M * 512.0 * a.pj_N : 0.0);
    } else if (a.pj_wp) {                                     // with the projection tail: the 80-row kernel, one block per workgroup
        hipLaunchKernelGGL((rb_ffn5_kernel<false, false, true>), dim3((a.M + R5_ROWS - 1) / R5_ROWS), dim3(512), R5_LDS_PJ, st, b);
        g_gemm_flops += 2.0 * a.M * 512.0 * a.pj_N;
    } else if (version == 2 || a.M < v5_rows || (a.alone && !forced)) hipLaunchKernelGGL(rb_ffn2_kernel, dim3((a.M + 63) / 64), dim3(512), lds, st, b);
    else {
        const int grid = ((a.M + R5_ROWS - 1) / R5_ROWS + bpw - 1) / bpw;
        if ((a.yb || a.stats_out) && bpw == 1 && one_on) hipLaunchKernelGGL((rb_ffn5_kernel<true, false, false, true>), dim3(grid), dim3(512), R5_LDS, st, b);
        else if (a.yb || a.stats_out) hipLaunchKernelGGL((rb_ffn5_kernel<true, false>), dim3(grid), dim3(512), R5_LDS, st, b);
        else if (b.dbg & 16) hipLaunchKernelGGL((rb_ffn5_kernel<false, true>), dim3(grid), dim3(512), R5_LDS, st, b);
        else if (bpw == 1 && one_on) hipLaunchKernelGGL((rb_ffn5_kernel<false, false, false, true>), dim3(grid), dim3(512), R5_LDS, st, b);
        else hipLaunchKernelGGL((rb_ffn5_kernel<false, false>), dim3(grid), dim3(512), R5_LDS, st, b);
    }
    g_gemm_flops += 4.0 * a.M * 512.0 * a.dff;
    return hipGetLastError() == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Attention sublayer: x <- x + W_o . concat_h softmax(q_h k_h^T / 8, key-prefix mask) v_h + b_o for G images per workgroup.
// Wavefront = head.  The attention itself is attn_bf16.hip's register-resident form (S^T = K Q^T with the key on the accumulator
// rows, the probabilities feeding O^T = V^T P^T from the same registers, V^T by transposing LDS reads); the next image's Q / K / V
// are requested while the current one is computed; a head's output goes to its 64 columns of the block in LDS instead of HBM.
// Then all eight wavefronts run the output projection over the block (rb_segment) and close the sublayer (rb_pass_store).

typedef __attribute__((ext_vector_type(4))) short rb_s16x4;
constexpr int RB_VROW = 80;                   // bf16 elements per staged V row (160 B: an odd multiple of 32 B, attn_bf16.hip)

// one (image, head): Q / K fragments straight from memory, V through the wavefront's LDS tile `sv` -> O^T in registers:
// ot[dt][qi][c] = O[q = qi*16 + l15][d = dt*16 + g*4 + c]
template <int NQT, int NKT>
__device__ __forceinline__ void rb_attn_head(const RbAttnArgs& a, int img, int h, int lane, bf16_t* sv, const bf16_t* zrow, f32x4 (&ot)[4][NQT]) {
    constexpr int VR = NKT == 4 ? 48 : 32;            // V rows staged (keys 48..63 of a four-tile call read the zero row)
    constexpr int NKL = NKT == 4 ? 3 : NKT;           // key tiles that can hold keys
    const int l15 = lane & 15, g = lane >> 4, Lk = a.Lk;
    const bf16_t* qg = a.q + (size_t)img * a.Lq * a.ldq + h * 64;
    const bf16_t* kg = a.k + (size_t)img * Lk * a.ldk + h * 64;
    const bf16_t* vg = a.v + (size_t)img * Lk * a.ldv + h * 64;
    const bf16x8 zero8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 bq[NQT][2], ak[NKL][2];
#pragma unroll
    for (int qi = 0; qi < NQT; ++qi) {
        const int r = qi * 16 + l15;
#pragma unroll
        for (int s = 0; s < 2; ++s) bq[qi][s] = r < a.Lq ? *reinterpret_cast<const bf16x8*>(qg + (size_t)r * a.ldq + s * 32 + g * 8) : zero8;
    }
#pragma unroll
    for (int kj = 0; kj < NKL; ++kj) {
        const int r = kj * 16 + l15;
#pragma unroll
        for (int s = 0; s < 2; ++s) ak[kj][s] = r < Lk ? *reinterpret_cast<const bf16x8*>(kg + (size_t)r * a.ldk + s * 32 + g * 8) : zero8;
    }
#pragma unroll
    for (int c = lane; c < VR * 8; c += 64) {
        const int r = c >> 3, ch = c & 7;
        *reinterpret_cast<u32x4*>(&sv[r * RB_VROW + ch * 8]) = r < Lk ? *reinterpret_cast<const u32x4*>(vg + (size_t)r * a.ldv + ch * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    // key counts.  Encoder, filling pass and cross-attention have ONE count per image (klen_sq == 0): it is read as a scalar and whole
    // key tiles are then either unmasked, masked per element (the one tile the count falls into) or skipped -- the softmax is most of
    // this phase's vector work.  Per-row counts (klen_sq != 0) mask every element.  All requested here, with the operands.
    const bool uni = a.klen_sq == 0;
    int klu = Lk, kls[NQT];
    {
        const int bi = a.klen_shared_last ? min(a.B, (img / a.klen_shared_last + 1) * a.klen_shared_last) - 1 : img;     // quirk Q1 per group
        if (a.klen && uni) klu = __builtin_amdgcn_readfirstlane(max(0, min(a.klen[bi * a.klen_sb] + a.klen_bias, Lk)));
#pragma unroll
        for (int qi = 0; qi < NQT; ++qi) {
            const int qrow = qi * 16 + l15;
            kls[qi] = klu;
            if (a.klen && !uni && qrow < a.Lq) kls[qi] = max(0, min(a.klen[bi * a.klen_sb + qrow * a.klen_sq] + a.klen_bias, Lk));
        }
    }
    // ---- S^T[key][q] = K Q^T over the NKL key tiles that can hold keys (a four-tile call has Lk <= 48: its last tile is padding)
    f32x4 st[NKL][NQT];
#pragma unroll
    for (int kj = 0; kj < NKL; ++kj)
#pragma unroll
        for (int qi = 0; qi < NQT; ++qi) {
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[kj][0], bq[qi][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[kj][1], bq[qi][1], c, 0, 0, 0);
            st[kj][qi] = c;
        }
    // ---- softmax over the keys of each query column; a lane holds keys kj*16 + g*4 + r.  softmax(s / 8) = exp2((s - max s) * log2(e) / 8) / sum
    constexpr float SC = 0.125f * 1.44269504088896340736f;
    bf16x8 bp[NQT][NKT / 2];
#pragma unroll
    for (int qi = 0; qi < NQT; ++qi) {
        const int kl = kls[qi];
        float m = -INFINITY;
#pragma unroll
        for (int kj = 0; kj < NKL; ++kj) {
            if (uni && kj * 16 >= klu) continue;                          // (scalar conditions: whole tiles)
            const bool whole = uni && kj * 16 + 16 <= klu;
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, (whole || kj * 16 + g * 4 + r < kl) ? st[kj][qi][r] : -INFINITY);
        }
        m = xor32_max(xor16_max(m));
        const float mb = m * SC;
        float sum = 0.f;
#pragma unroll
        for (int kj = 0; kj < NKL; ++kj) {
            if (uni && kj * 16 >= klu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kj][qi][r] = 0.f;
                continue;
            }
            const bool whole = uni && kj * 16 + 16 <= klu;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kj][qi][r], SC, -mb));
                if (!whole) e = (kj * 16 + g * 4 + r < kl) ? e : 0.f;
                st[kj][qi][r] = e;
                sum += e;
            }
        }
        sum = xor32_sum(xor16_sum(sum));
        const float inv_sum = 1.0f / sum;                 // an empty row: 0 * (1/0) = NaN for every key, as softmax over all -inf
#pragma unroll
        for (int s = 0; s < NKT / 2; ++s) {
            bf16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kj = 2 * s + (j >> 2), r = j & 3;
                float pv = 0.f;
                if (kj < NKL) {
                    pv = st[kj][qi][r] * inv_sum;
                    if ((kj + 1) * 16 > Lk && kl == 0 && kj * 16 + g * 4 + r >= Lk) pv = 0.f;      // padded keys of an empty row: V is 0 there
                }
                f[j] = (short)f32_to_bf16(pv);
            }
            bp[qi][s] = f;
        }
    }
    // ---- O^T[d][q] = V^T P^T
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qi = 0; qi < NQT; ++qi) ot[dt][qi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = l15 >> 2, tp = l15 & 3;
#pragma unroll
    for (int s = 0; s < NKT / 2; ++s)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16_t* a0p = &sv[(32 * s + 4 * g + tq) * RB_VROW + dt * 16 + 4 * tp];
            const bf16_t* a1p = (NKT == 4 && s == 1) ? zrow + dt * 16 + 4 * tp : a0p + 16 * RB_VROW;
            const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rb_s16x4*)a0p);
            const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rb_s16x4*)a1p);
            bf16x8 av;
            av[0] = lo[0]; av[1] = lo[1]; av[2] = lo[2]; av[3] = lo[3];
            av[4] = hi[0]; av[5] = hi[1]; av[6] = hi[2]; av[7] = hi[3];
#pragma unroll
            for (int qi = 0; qi < NQT; ++qi) ot[dt][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bp[qi][s], ot[dt][qi], 0, 0, 0);
        }
    __builtin_amdgcn_wave_barrier();                   // (the V tile is rewritten for the wavefront's next image only after these reads were issued)
}

// W wavefronts (16 or 8): wavefront (p, h) = wave >> 3, wave & 7 runs head h of images p, p + W/8, ... of the workgroup's G = NR * W/8 images (NR rounds);
// the V tiles and the block share LDS (the block is written once every wavefront is through with its V tile); then every wavefront takes
// 512 / W of the 512 output columns.  W = 8 (round 4): half the images per workgroup, at most half the LDS and registers of a CU -- TWO workgroups per
// CU, so one's attention phase (vector work and load latency: 26 k of a 16-wavefront workgroup's 54 k cycles, the MFMA pipes idle) runs beside the
// other's output projection; the price is the output projection's weights streamed once per G/2 images more.
// PJ (round 5): the projection that reads the sublayer's output next -- the decoder layer's LayerNorm-folded cross-attention query projection -- from the block while it
// sits in LDS: y as bf16 back into the block with its row statistics (a wavefront's share of a row's sums through LDS), the closing stores through staging rows of their
// own, then one more weight segment and the fold; the queries go to memory as 8-byte pieces.  The same weight bytes per row as the 96-row projection launch it replaces
// (0.5 MB per 80 rows), one launch and one pass over the stream less per decoder layer.
template <int NQT, int NKT, int NR, int W, bool PJ = false>
__global__ __launch_bounds__(W * 64, 4) void rb_attn_kernel(RbAttnArgs a) {      // (4 wavefronts per SIMD: 128 registers, so that two 8-wavefront workgroups share a CU)
    constexpr int IPR = W / 8, G = NR * IPR, VR = NKT == 4 ? 48 : 32;      // images per round / per workgroup
    constexpr int LQM = NR == 2 ? 20 : (NQT == 2 ? 32 : 40);              // query rows per image this instantiation is launched for
    constexpr int MT = (G * LQM + 15) / 16;
    constexpr int NT = 32 / W;                                             // 16-column tiles of the output projection per wavefront: 2 (W = 16) or 4 (W = 8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* blk = smem;                                              // [MT*16 rows][512] bf16, swizzled  (aliases the V tiles)
    bf16_t* svs = reinterpret_cast<bf16_t*>(smem);                          // W x [VR][RB_VROW]
    constexpr int BODY = (W * VR * RB_VROW * 2 > MT * 16384) ? W * VR * RB_VROW * 2 : MT * 16384;
    float* bos = reinterpret_cast<float*>(smem + BODY);                     // [512]
    bf16_t* zrow = reinterpret_cast<bf16_t*>(smem + BODY + 2048);           // 256 B of zeros
    // projection tail: staging rows of the closing stores (the block stays live), a wavefront's (sum, sum of squares) per row, the fold's constants
    unsigned char* pj_stg = smem + BODY + 2048 + 256;                       // [W][16 rows][144 B]
    float2* pj_part = reinterpret_cast<float2*>(pj_stg + W * 16 * 144);     // [W][MT*16]
    float* pj_cc = reinterpret_cast<float*>(pj_part + W * MT * 16);         // c[512] | cs[512]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int img0 = blockIdx.x * G, nimg = min(G, a.B - img0);
    RB_STAMP(a.dbg, wave, lane, 0);
    for (int i = tid; i < 512; i += W * 64) bos[i] = a.bo[i];
    if constexpr (PJ)
        for (int i = tid; i < 512; i += W * 64) { pj_cc[i] = a.pj_c[i]; pj_cc[512 + i] = a.pj_cs[i]; }
    if (tid < 64) reinterpret_cast<uint32_t*>(zrow)[tid] = 0u;
    __syncthreads();                                                        // (the zero row is read by every wavefront)

    RB_STAMP(a.dbg, wave, lane, 1);
    // ---- attention
    const int h = wave & 7, p = wave >> 3;
    bf16_t* sv = svs + wave * (VR * RB_VROW);
    f32x4 ot[NR][4][NQT];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (p + IPR * r < nimg && !(a.dbg & 1)) rb_attn_head<NQT, NKT>(a, img0 + p + IPR * r, h, lane, sv, zrow, ot[r]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the output projection's weights: columns wave*(512/W) .. = tiles of chunk (wave * NT) / 4
    const u32x4* wo = a.wop + (size_t)((wave * NT) >> 2) * (16 * 256) + ((wave * NT) & 3) * 64 + lane;
    RB_STAMP(a.dbg, wave, lane, 2);
    constexpr int OPF = NT == 4 ? 2 : RB_PF;                                // ring steps in flight: 32 registers either way (the kernel lives in 128)
    bf16x8 wb[OPF * NT];
    if constexpr (NT == 2) rb_prime<NT, OPF>(wo, wb);                       // (four tiles: primed once the heads' outputs have left the registers)
    __syncthreads();                                                        // every V tile is dead: the block may overwrite them
    RB_STAMP(a.dbg, wave, lane, 3);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (p + IPR * r >= nimg) continue;
        const int row0 = (p + IPR * r) * a.Lq;
#pragma unroll
        for (int qi = 0; qi < NQT; ++qi) {
            const int qrow = qi * 16 + l15;
            if (qrow >= a.Lq) continue;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<uint2*>(blk + rb_off(row0 + qrow, h * 8 + dt * 2 + (g >> 1)) + (g & 1) * 8) =
                    make_uint2(pack_bf16(ot[r][dt][qi][0], ot[r][dt][qi][1]), pack_bf16(ot[r][dt][qi][2], ot[r][dt][qi][3]));
        }
    }
    if constexpr (NT == 4) rb_prime<NT, OPF>(wo, wb);
    __syncthreads();
    RB_STAMP(a.dbg, wave, lane, 4);

    // ---- output projection over the block.  The accumulators START from the residual rows (loaded in the accumulator layout, 16 rows x 64 bytes
    // per instruction; in flight during the projection's first steps), so closing the sublayer needs no loads
    const int rows_live = (a.dbg & 4) ? 0 : nimg * a.Lq, m0 = img0 * a.Lq, c0 = wave * (512 / W);
    f32x4 acc[NT][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = mt * 16 + l15;
        const float* xr = a.x + (size_t)(m0 + min(r, max(rows_live, 1) - 1)) * a.ldx + c0 + g * 4;      // (dead rows: the block's last live row -- never stored)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4 v = *reinterpret_cast<const float4*>(xr + nt * 16);
            acc[nt][mt] = f32x4{v.x, v.y, v.z, v.w};
        }
    }
    const u32x4* wq = nullptr;
    if constexpr (PJ) wq = a.pj_wp + (size_t)((wave * NT) >> 2) * (16 * 256) + ((wave * NT) & 3) * 64 + lane;
    if (!(a.dbg & 2)) rb_segment<MT, NT, OPF>(wo, PJ ? wq : wo, wb, smem, rb_lane_base(l15, g), acc);      // (PJ: its last steps request the projection's first ones)
    RB_STAMP(a.dbg, wave, lane, 5);

    // ---- close the sublayer: the block is dead once every wavefront has left the segment; each wavefront then turns 32 columns of a row
    // tile at a time into 128-byte row pieces through 2.3 KB of LDS of its own (+ b_o) -- no further workgroup barrier (round 3: three passes of
    // 32 rows through a shared staging area, residual loads and two barriers per pass: 13 k of the kernel's 54 k cycles)
    __syncthreads();
    RB_STAMP(a.dbg, wave, lane, 6);
    if constexpr (PJ) {                                         // y = accumulators + b_o: into the block as bf16, this wavefront's share of the row sums beside it
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = mt * 16 + l15;
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bb = *reinterpret_cast<const float4*>(bos + c0 + nt * 16 + g * 4);
                f32x4 t = acc[nt][mt];
                t[0] += bb.x; t[1] += bb.y; t[2] += bb.z; t[3] += bb.w;
                acc[nt][mt] = t;
                sm += (t[0] + t[1]) + (t[2] + t[3]);
                sq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
                *reinterpret_cast<uint2*>(blk + rb_off(r, (c0 >> 3) + nt * 2 + (g >> 1)) + (g & 1) * 8) = make_uint2(pack_bf16(t[0], t[1]), pack_bf16(t[2], t[3]));
            }
            sm = xor32_sum(xor16_sum(sm)); sq = xor32_sum(xor16_sum(sq));
            if (g == 0) pj_part[wave * (MT * 16) + r] = make_float2(sm, sq);
        }
    }
    {
        unsigned char* stg = PJ ? pj_stg + wave * (16 * 144) : smem + wave * (16 * 144);
        const int er = lane >> 3, ec = lane & 7;
        const size_t ystep = (size_t)8 * a.ldy;
#pragma unroll
        for (int half = 0; half < NT / 2; ++half) {            // 32 columns (two tiles) at a time
            const int cb = c0 + half * 32 + ec * 4;
            const float4 bb = PJ ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(bos + cb);      // (PJ: the accumulators hold the bias already)
            float* yp = a.y + (size_t)(m0 + er) * a.ldy + cb;
            int rr = er;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 t = acc[half * 2 + nt][mt];
                    *reinterpret_cast<float4*>(stg + l15 * 144 + nt * 64 + g * 16) = make_float4(t[0], t[1], t[2], t[3]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float4 sv2[2];
#pragma unroll
                for (int it = 0; it < 2; ++it) sv2[it] = *reinterpret_cast<const float4*>(stg + (it * 8 + er) * 144 + ec * 16);
#pragma unroll
                for (int it = 0; it < 2; ++it, yp += ystep, rr += 8) {
                    const float4 o = make_float4(sv2[it].x + bb.x, sv2[it].y + bb.y, sv2[it].z + bb.z, sv2[it].w + bb.w);
                    const bool live = rr < rows_live;
                    const size_t m = live ? (size_t)(m0 + rr) : 0;
                    if (live) {
                        *reinterpret_cast<float4*>(yp) = o;
                        if (a.yb) *reinterpret_cast<uint2*>(a.yb + m * 512 + cb) = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
                    }
                    if (a.stats_out) {                     // 32 columns are one statistics group: the 8 lanes of a row piece
                        const float s1 = oct_sum((o.x + o.y) + (o.z + o.w)), s2 = oct_sum((o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w));
                        if (live && !ec) reinterpret_cast<float2*>(a.stats_out + m * 32)[cb >> 5] = make_float2(s1, s2);
                    }
                }
                __builtin_amdgcn_wave_barrier();           // (the next row tile rewrites the staging rows)
            }
        }
    }
    RB_STAMP(a.dbg, wave, lane, 7);
    if constexpr (PJ) {
        __syncthreads();                                        // the block holds y as bf16, the partial sums are in place
        float mu[MT], rs[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) { const float2 p = pj_part[w * (MT * 16) + mt * 16 + l15]; s1 += p.x; s2 += p.y; }
            const float mean = s1 * (1.0f / 512.0f);
            const float var = fmaxf((s2 - s1 * mean) * (1.0f / 511.0f), 0.f);
            mu[mt] = mean; rs[mt] = 1.0f / (sqrtf(var) + 1e-6f);
            asm volatile("" : "+v"(mu[mt]), "+v"(rs[mt]));      // (computed here, not behind the segment: the partial sums would wait in registers)
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        rb_segment<MT, NT, OPF>(wq, wq, wb, smem, rb_lane_base(l15, g), acc);
        int cofs = c0 + g * 4;
        asm volatile("" : "+v"(cofs));
        unsigned short* qy = a.pj_y + (size_t)m0 * a.pj_ldy + cofs;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4 cc = *reinterpret_cast<const float4*>(pj_cc + cofs + nt * 16);
            const float4 cs = *reinterpret_cast<const float4*>(pj_cc + 512 + cofs + nt * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 t = acc[nt][mt];
                const float v0 = rs[mt] * (t[0] - mu[mt] * cs.x) + cc.x, v1 = rs[mt] * (t[1] - mu[mt] * cs.y) + cc.y;
                const float v2 = rs[mt] * (t[2] - mu[mt] * cs.z) + cc.z, v3 = rs[mt] * (t[3] - mu[mt] * cs.w) + cc.w;
                const int r = mt * 16 + l15;
                if (r < rows_live) *reinterpret_cast<uint2*>(qy + (size_t)r * a.pj_ldy + nt * 16) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
            }
        }
        RB_STAMP(a.dbg, wave, lane, 8);
    }
}

template <int NQT, int NKT, int NR, int W, bool PJ = false>
static int launch_rb_attn_t(const RbAttnArgs& a, hipStream_t st) {
    constexpr int IPR = W / 8, G = NR * IPR, VR = NKT == 4 ? 48 : 32;
    constexpr int LQM = NR == 2 ? 20 : (NQT == 2 ? 32 : 40);              // query rows per image this instantiation is launched for
    constexpr int MT = (G * LQM + 15) / 16;
    constexpr size_t body = (W * VR * RB_VROW * 2 > MT * 16384) ? W * VR * RB_VROW * 2 : MT * 16384;
    constexpr size_t lds = body + 2048 + 256 + (PJ ? W * 16 * 144 + W * MT * 16 * 8 + 4096 : 0);
    static_assert(lds <= 160 * 1024, "fits a CU");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_attn_kernel<NQT, NKT, NR, W, PJ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return BOFI_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL((rb_attn_kernel<NQT, NKT, NR, W, PJ>), dim3((a.B + G - 1) / G), dim3(W * 64), lds, st, a);
    return hipGetLastError() == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}

// -1: shape not covered (the caller keeps attention + GEMM)
int launch_rb_attn(const RbAttnArgs& a, hipStream_t st) {
    if (!a.q || !a.k || !a.v || !a.wop || !a.bo || !a.x || !a.y || a.B < 1 || a.Lq < 1 || a.Lk < 1 || a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldx % 4 || a.ldy % 4)
        return BOFI_ERR_ARG;
    if (a.Lq > 40 || a.Lk > 48) return -1;
    // BOFI_RB_ATTN_W (developer knob, read again after bofi_reload_env) = 16: one 16-wavefront workgroup per CU (round 3), 8: 8-wavefront workgroups of
    // half the images, two per CU; default (0): 8 for query blocks of <= 20 rows (the filling pass: 25.9 -> 20.3 / 31.9 -> 24.7 us per launch at 320
    // images), 16 for the encoder's 36 (28.1 against 29.6 us; with four launches in flight the 8-wavefront form there costs 1-2 %)
    static int env_seen = -1, wenv = 0;
    if (env_seen != g_env_generation) { const char* e = getenv("BOFI_RB_ATTN_W"); wenv = e ? atoi(e) : 0; env_seen = g_env_generation; }
    // (round 5: with launches in flight what counts is the weight bytes a launch pulls through L2, not its latency: 16 wavefronts -- four images per W_o stream -- for
    // the filling pass too, +1.3 % on the headline, profiles/r05_weight_bytes_ab.txt; a launch that runs alone keeps the 8-wavefront form)
    const int w = wenv ? wenv : (a.Lq <= 20 && a.alone ? 8 : 16);
    int rc;
    if (a.pj_wp) {                                              // with the projection tail: the 16-wavefront form of the filling pass's self-attention only
        if (!a.pj_c || !a.pj_cs || !a.pj_y || a.pj_ldy % 4 || a.yb || a.stats_out) return BOFI_ERR_ARG;
        if (a.Lq > 20 || a.Lk > 32) return -1;
        rc = launch_rb_attn_t<2, 2, 2, 16, true>(a, st);
        if (rc == BOFI_OK) g_gemm_flops += 2 * 2.0 * a.B * a.Lq * 512.0 * 512.0;
        return rc;
    }
    if (w == 16) {
        if (a.Lq <= 20 && a.Lk <= 32) rc = launch_rb_attn_t<2, 2, 2, 16>(a, st);       // 4 images of <= 20 rows per workgroup
        else if (a.Lq <= 20) rc = launch_rb_attn_t<2, 4, 2, 16>(a, st);
        else if (a.Lq <= 32 && a.Lk <= 32) rc = launch_rb_attn_t<2, 2, 1, 16>(a, st);  // 2 images
        else if (a.Lq <= 32) rc = launch_rb_attn_t<2, 4, 1, 16>(a, st);
        else rc = launch_rb_attn_t<3, 4, 1, 16>(a, st);
    } else {
        if (a.Lq <= 20 && a.Lk <= 32) rc = launch_rb_attn_t<2, 2, 2, 8>(a, st);        // 2 images of <= 20 rows per workgroup
        else if (a.Lq <= 20) rc = launch_rb_attn_t<2, 4, 2, 8>(a, st);
        else if (a.Lq <= 32 && a.Lk <= 32) rc = launch_rb_attn_t<2, 2, 1, 8>(a, st);   // 1 image
        else if (a.Lq <= 32) rc = launch_rb_attn_t<2, 4, 1, 8>(a, st);
        else rc = launch_rb_attn_t<3, 4, 1, 8>(a, st);
    }
    if (rc == BOFI_OK) g_gemm_flops += 2.0 * a.B * a.Lq * 512.0 * 512.0;
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// y[M][N] = epilogue(LN(x) . W^T) for K = 512 and any N % 64 == 0: the LayerNorm-folded projections of the path (q|k|v, the
// cross-attention queries, the stacked cross K|V of all layers, the generator) as a row-block kernel: the block of the residual
// stream is staged once (float32 -> bf16, row statistics on the way: no bf16 copy and no statistics from the producer), the eight
// wavefronts take the 64-column chunks w, w + 8, ... of the weight (any chunk count), 16 k-steps each, and there is NO workgroup
// barrier after the staging: a wavefront's epilogue (fold, bias, rounding, its own LDS to turn the accumulator layout into whole
// 128- / 256-byte row pieces) runs beside its SIMD partner's MFMAs.
// MT = row tiles per block: 4 (64 rows) or 8 (128 rows, bf16 outputs).  Every workgroup streams the WHOLE weight through its CU's vector
// memory path (64 B per clock): at 64 rows that path and the MFMA pipes saturate together, and with every CU streaming the chip
// delivers ~16 TB/s of weights whatever the kernels do (round 4: four concurrent feed-forward streams, dev/exp/mb_rowblock.py ffn4) --
// the projections are bound by weight bytes per row.  128 rows per block halve them (128 KB of LDS for the block, 16 rows of staging per
// wavefront at a time, operands read four tiles at a time to stay within 256 registers).
template <bool F32OUT, int MT>
__global__ __launch_bounds__(512) void rb_gemm_kernel(RbGemmArgs a) {
    using Cfg = RbGemmCfg<F32OUT, MT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xt = smem;
    unsigned char* stage_all = smem + Cfg::XT;
    float* s_mean = reinterpret_cast<float*>(smem + Cfg::STAT);
    float* s_rstd = s_mean + Cfg::BR;
    float* cst = reinterpret_cast<float*>(smem + Cfg::CST);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // gridDim.y workgroups share a row block: workgroup y takes the chunk octets y, y + gridDim.y, ... (wide outputs on few row blocks:
    // the generator's 2.4 MB of float32 logits per block leave a CU at ~25 GB/s -- two workgroups per block halve that tail)
    const int m0 = blockIdx.x * Cfg::BR, nchunks = a.N >> 6, ch0 = wave + 8 * blockIdx.y, chstep = 8 * gridDim.y;
    auto seg = [&](int ch) { return a.wp + (size_t)ch * (16 * 256) + lane; };
    RB_STAMP(a.dbg, 8 + wave, lane, 0);                        // entry (rows 8-15: this kernel has 8 wavefronts)
    bf16x8 wb[RB_PF * 4];
    if (ch0 < nchunks) rb_prime<4>(seg(ch0), wb);
    constexpr int RPW = MT * 2;                                // rows a wavefront stages
#pragma unroll
    for (int pass = 0; pass < (RPW + 7) / 8; ++pass) {         // stage the block: wavefront w takes rows RPW*w .. +RPW-1, eight at a time, eight lanes per row
        const int lr = pass * 8 + (lane >> 3), r = wave * RPW + lr, sub = lane & 7, m = m0 + r;
        const bool mine = lr < RPW;
        float4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            v[j] = (mine && m < a.M) ? *reinterpret_cast<const float4*>(a.x + (size_t)m * a.ldx + j * 32 + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            sq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
            if (mine) *reinterpret_cast<uint2*>(xt + rb_off(r, j * 4 + (sub >> 1)) + (sub & 1) * 8) = make_uint2(pack_bf16(v[j].x, v[j].y), pack_bf16(v[j].z, v[j].w));
        }
        sm = oct_sum(sm); sq = oct_sum(sq);
        if (mine && sub == 0) {
            const float mean = sm * (1.0f / 512.0f);
            const float var = fmaxf((sq - sm * mean) * (1.0f / 511.0f), 0.f);
            s_mean[r] = mean;
            s_rstd[r] = 1.0f / (sqrtf(var) + 1e-6f);
        }
    }
    __syncthreads();
    RB_STAMP(a.dbg, 8 + wave, lane, 1);                        // block staged
    rb_gemm_chunks<F32OUT, MT>(a, smem, stage_all, cst, s_mean, s_rstd, m0, wave, lane, ch0, chstep, wb);
    RB_STAMP(a.dbg, 8 + wave, lane, 2);                        // exit
}

template <bool F32OUT, int MT>
static int launch_rb_gemm_t(const RbGemmArgs& a, hipStream_t st) {
    using Cfg = RbGemmCfg<F32OUT, MT>;
    static_assert(Cfg::LDS <= 160 * 1024, "one workgroup per CU");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_gemm_kernel<F32OUT, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return BOFI_ERR_HIP;
        attr_set = true;
    }
    const int blocks = (a.M + Cfg::BR - 1) / Cfg::BR, octets = (a.N / 64 + 7) / 8;
    int split = 1;                                            // workgroups per row block: fill the chip when the row blocks alone do not
    while (split < 4 && blocks * (split + 1) <= 256 && octets >= 6 * (split + 1)) ++split;
    hipLaunchKernelGGL((rb_gemm_kernel<F32OUT, MT>), dim3(blocks, split), dim3(512), Cfg::LDS, st, a);
    return hipGetLastError() == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}

int launch_rb_gemm(const RbGemmArgs& a, hipStream_t st) {
    if (a.M < 1 || a.N < 64 || a.N % 64 || !a.x || !a.wp || !a.c || !a.cs || !a.y || a.ldx % 4 || a.ldy % 8) return BOFI_ERR_ARG;
    // developer knob (read again after bofi_reload_env): BOFI_RB_GEMM_MT = 4: 64-row blocks everywhere; 6 (default): 96-row blocks for bf16
    // outputs from BOFI_RB_GEMM_MT8_ROWS rows on (below that the 64-row blocks' larger number of workgroups wins; round 4's 128-row form, <false, 8>, spilled 46 registers, never won and left in round 6)
    static int env_seen = -1, mt = 6, mt8_rows = 4096, mt_min_n = 0, gen6 = -1;
    if (env_seen != g_env_generation) {
        const char* e = getenv("BOFI_RB_GEMM_MT"); mt = e ? atoi(e) : 6;
        e = getenv("BOFI_RB_GEMM_MT8_ROWS"); mt8_rows = e ? atoi(e) : 4096;
        e = getenv("BOFI_RB_GEMM_MT_MIN_N"); mt_min_n = e ? atoi(e) : 0;   // output columns from which the larger blocks run
        e = getenv("BOFI_RB_GEN_MT6"); gen6 = e ? atoi(e) : -1;            // float32 outputs (the generator) on 96-row blocks too: 1 always, 0 never, default: with launches in flight (a third fewer weight bytes: +0.7 %)
        env_seen = g_env_generation;
    }
    int rc;
    if (a.y_f32) rc = (mt == 6 && (gen6 > 0 || (gen6 < 0 && !a.alone)) && a.M >= mt8_rows) ? launch_rb_gemm_t<true, 6>(a, st) : launch_rb_gemm_t<true, 4>(a, st);
    else if (mt == 6 && a.M >= mt8_rows && a.N >= mt_min_n && !(a.alone && a.N < 2048)) rc = launch_rb_gemm_t<false, 6>(a, st);      // (alone: 96-row blocks only where they win alone)
    else rc = launch_rb_gemm_t<false, 4>(a, st);
    if (rc == BOFI_OK) g_gemm_flops += 2.0 * a.M * 512.0 * a.N;
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// w [N][K] row-major bf16 -> fragment-major: [N/64 chunks][K/32 steps][4 tiles][64 lanes][8 bf16]; lane (l15, g) of a fragment holds
// row chunk*64 + tile*16 + l15, k = step*32 + g*8 .. +7
__global__ __launch_bounds__(256) void rb_pack_frag_kernel(const bf16_t* __restrict__ w, u32x4* __restrict__ out, int N, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * K / 8) return;
    const int lane = (int)(i & 63), nt = (int)((i >> 6) & 3);
    const size_t t = i >> 8;
    const int kb = (int)(t % (size_t)(K >> 5)), chunk = (int)(t / (size_t)(K >> 5));
    const int n = chunk * 64 + nt * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8;
    out[i] = *reinterpret_cast<const u32x4*>(w + (size_t)n * K + k);
}

int launch_rb_pack_frag(const void* w, void* out, int N, int K, hipStream_t st) {
    if (!w || !out || N < 64 || N % 64 || K < 32 || K % 32) return BOFI_ERR_ARG;
    const size_t n = (size_t)N * K / 8;
    hipLaunchKernelGGL(rb_pack_frag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w, (u32x4*)out, N, K);
    return hipGetLastError() == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}

}  // namespace bofi

extern "C" int bofi_rb_stamps(unsigned long long* host_out) {      // developer aid: the 16 x 16 stamps of the last BOFI_RB_DBG & 16 launch
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(bofi::g_rb_stamps), sizeof(unsigned long long) * 256) == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}

extern "C" int bofi_pack_frag(const void* w, void* out, int N, int K, void* stream) {
    return bofi::launch_rb_pack_frag(w, out, N, K, (hipStream_t)stream);
}

extern "C" int bofi_attn_block(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int Lq, int Lk, const int* klen, int klen_sb,
                               int klen_sq, int klen_bias, int klen_shared_last, const void* wop, const float* bo, const float* x, int ldx, float* y,
                               int ldy, void* yb, float* stats_out, void* stream) {
    bofi::RbAttnArgs a{};
    a.q = (const bofi::bf16_t*)q; a.ldq = ldq; a.k = (const bofi::bf16_t*)k; a.ldk = ldk; a.v = (const bofi::bf16_t*)v; a.ldv = ldv;
    a.B = B; a.Lq = Lq; a.Lk = Lk; a.klen = klen; a.klen_sb = klen_sb; a.klen_sq = klen_sq; a.klen_bias = klen_bias; a.klen_shared_last = klen_shared_last;
    a.wop = (const bofi::u32x4*)wop; a.bo = bo; a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.yb = (bofi::bf16_t*)yb; a.stats_out = stats_out;
    { const char* e = getenv("BOFI_RB_DBG"); a.dbg = e ? atoi(e) : 0; }
    a.alone = 1;                                               // (the direct entry: a launch of its own; BOFI_RB_ATTN_W picks the other form)
    const int rc = bofi::launch_rb_attn(a, (hipStream_t)stream);
    return rc < 0 ? BOFI_ERR_ARG : rc;
}

extern "C" int bofi_attn_linear_block(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int Lq, int Lk, const int* klen, int klen_sb,
                                      int klen_bias, int klen_shared_last, const void* wop, const float* bo, float* x, int ldx, const void* pj_wp, const float* pj_c,
                                      const float* pj_cs, void* pj_y, int pj_ldy, void* stream) {
    if (!pj_wp) return BOFI_ERR_ARG;
    bofi::RbAttnArgs a{};
    a.q = (const bofi::bf16_t*)q; a.ldq = ldq; a.k = (const bofi::bf16_t*)k; a.ldk = ldk; a.v = (const bofi::bf16_t*)v; a.ldv = ldv;
    a.B = B; a.Lq = Lq; a.Lk = Lk; a.klen = klen; a.klen_sb = klen_sb; a.klen_sq = 0; a.klen_bias = klen_bias; a.klen_shared_last = klen_shared_last;
    a.wop = (const bofi::u32x4*)wop; a.bo = bo; a.x = x; a.ldx = ldx; a.y = x; a.ldy = ldx;
    a.pj_wp = (const bofi::u32x4*)pj_wp; a.pj_c = pj_c; a.pj_cs = pj_cs; a.pj_y = (bofi::bf16_t*)pj_y; a.pj_ldy = pj_ldy;
    { const char* e = getenv("BOFI_RB_DBG"); a.dbg = e ? atoi(e) : 0; }
    const int rc = bofi::launch_rb_attn(a, (hipStream_t)stream);
    return rc < 0 ? BOFI_ERR_ARG : rc;
}

extern "C" int bofi_linear_block(const float* x, int ldx, const void* wp, const float* c, const float* cs, void* y, int ldy, int y_f32, int M, int N,
                                 int relu, void* stream) {
    bofi::RbGemmArgs a{};
    a.x = x; a.ldx = ldx; a.wp = (const bofi::u32x4*)wp; a.c = c; a.cs = cs; a.y = y; a.ldy = ldy; a.y_f32 = y_f32; a.M = M; a.N = N; a.relu = relu;
    { const char* e = getenv("BOFI_RB_DBG"); a.dbg = e ? atoi(e) : 0; }
    return bofi::launch_rb_gemm(a, (hipStream_t)stream);
}

extern "C" int bofi_ffn_block(const float* x, int ldx, const void* w1p, const float* c1, const float* cs1, const void* w2p, const float* b2, float* y,
                              int ldy, void* yb, float* stats_out, int M, int dff, void* stream) {
    bofi::RbFfnArgs a{};
    a.x = x; a.ldx = ldx; a.w1p = (const bofi::u32x4*)w1p; a.c1 = c1; a.cs1 = cs1; a.w2p = (const bofi::u32x4*)w2p; a.b2 = b2; a.y = y; a.ldy = ldy;
    a.yb = (bofi::bf16_t*)yb; a.stats_out = stats_out; a.M = M; a.dff = dff;
    return bofi::launch_rb_ffn(a, (hipStream_t)stream);
}

extern "C" int bofi_attn_out_ffn_block(const float* x, int ldx, const void* ctx, int ldc, const void* wop, const float* bo, const void* w1p, const float* c1, const float* cs1,
                                       const void* w2p, const float* b2, float* y, int ldy, int M, int dff, const void* pj_wp, const float* pj_c, const float* pj_cs, void* pj_y,
                                       int pj_ldy, int pj_N, void* stream) {
    if (!ctx || !wop || !bo) return BOFI_ERR_ARG;
    bofi::RbFfnArgs a{};
    a.x = x; a.ldx = ldx; a.w1p = (const bofi::u32x4*)w1p; a.c1 = c1; a.cs1 = cs1; a.w2p = (const bofi::u32x4*)w2p; a.b2 = b2; a.y = y; a.ldy = ldy;
    a.M = M; a.dff = dff;
    a.head_ctx = (const uint16_t*)ctx; a.head_ldc = ldc; a.head_wop = (const bofi::u32x4*)wop; a.head_bo = bo;
    if (pj_wp) { a.pj_wp = (const bofi::u32x4*)pj_wp; a.pj_c = pj_c; a.pj_cs = pj_cs; a.pj_y = pj_y; a.pj_ldy = pj_ldy; a.pj_N = pj_N; }
    return bofi::launch_rb_ffn(a, (hipStream_t)stream);
}

extern "C" int bofi_ffn_linear_block(const float* x, int ldx, const void* w1p, const float* c1, const float* cs1, const void* w2p, const float* b2, float* y,
                                     int ldy, int M, int dff, const void* pj_wp, const float* pj_c, const float* pj_cs, void* pj_y, int pj_ldy, int pj_N,
                                     void* stream) {
    if (!pj_wp) return BOFI_ERR_ARG;
    bofi::RbFfnArgs a{};
    a.x = x; a.ldx = ldx; a.w1p = (const bofi::u32x4*)w1p; a.c1 = c1; a.cs1 = cs1; a.w2p = (const bofi::u32x4*)w2p; a.b2 = b2; a.y = y; a.ldy = ldy;
    a.M = M; a.dff = dff;
    a.pj_wp = (const bofi::u32x4*)pj_wp; a.pj_c = pj_c; a.pj_cs = pj_cs; a.pj_y = pj_y; a.pj_ldy = pj_ldy; a.pj_N = pj_N;
    return bofi::launch_rb_ffn(a, (hipStream_t)stream);
}
