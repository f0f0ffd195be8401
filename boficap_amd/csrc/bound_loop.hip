// The whole bounding loop of core_NAIC (reference TransformerModel.py:1833-1869) for a GROUP of 16 images as ONE persistent workgroup
// (round 5): every iteration's five stages -- row-0 self-attention of the one-layer bounding network (LengthPredictor_UIC.forward
// :357-383 -> DecoderLayer_UIC :283-297), its output projection, the cross-attention over the image's regions, the feed-forward
// sublayer, the final norm + two heads + log-softmax + first-max pick (:375-383) and the slot bookkeeping (:1843-1869) -- run inside
// the workgroup, which leaves the loop when its own 16 images are finished.  No grid barrier, no co-residency requirement: workgroups
// never wait for each other.
//
// Why: as 5 launches per iteration (bound_ops.hip + naic.hip's tail kernel) the loop was 55 chip-wide bursts of 64-160 workgroups per
// 320-image decode, each holding its CUs while it waited on L2 -- 22 % of the time of a decode in flight for 2.3 % of its FLOPs
// (profiles/r04_ablations.txt).  Here a 320-image decode's loop occupies 20 CUs and nothing else; the other decodes in flight keep the
// rest of the chip.
//
// Shape of the work: one activation ROW per image.  The 16 rows of the group are the B operand (columns) of v_mfma_f32_16x16x32_f16,
// the weights -- fragment-major, the layout of rowblock.hip's rb_pack_frag_kernel -- are the A operand streamed from L2 straight into
// operand registers: the 8 wavefronts split the OUTPUT columns, a ring of BL_PF k-steps per wavefront stays in flight ACROSS stage
// boundaries (the last segment of a stage requests the first steps of the next stage's stream: a weight address never depends on data).
// Per iteration a workgroup streams 5.75 MB of weights (3 x 0.5 MB projections, 2 + 2 MB feed-forward, 0.25 MB head hidden layers) and
// reads its images' cross-attention K|V rows (16 x 36 x 2 KB).
//
// Precision: the bounding network's operands are FP16 here, not bf16 -- same bytes and MFMA rate, 8x finer rounding
// (weights |w| < 1, LayerNorm-ed activations O(1), everything clamped to the fp16 range on conversion; NaN passes).  LayerNorms are
// applied explicitly in float32 (two-pass mean / unbiased std, eps on the std: TransformerModel.py:1346-1349) on the float32 residual
// rows held in LDS, the gain folded into the consumer's weights and gain-bias into its bias (c = b + W . b_ln); the self-attention
// scores of row 0 come from a float32 table (position, label, head) and its values from a float32 V table (bound_tables_kernel below);
// softmaxes, the heads' output layers and log-softmax are float32.  What stays bf16: the cross-attention K|V rows (they are the
// encoder's kv_all projection, written by the row-block kernels).
#include "bofi_common.h"
#include "bofi_kernels.h"
#include "bofi_naic.h"

#include <cstdlib>

namespace bofi {

extern int g_env_generation;                   // bumped by bofi_reload_env (gemm_glds.hip)

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// developer aid (BOFI_BL_DBG=1): s_memtime stamps of workgroup 0, wavefront 0, for kernel entry ([0]), loop exit ([1]) and the stage ends of iteration
// a.dbg - 1 ([2 + i]); read back by bofi_bl_stamps
__device__ unsigned long long g_bl_stamps[32];
#define BL_STAMP(i) do { if (a.dbg && it == a.dbg - 1 && blockIdx.x == 0 && tid == 0) g_bl_stamps[2 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

constexpr int BL_G = 16;            // images per workgroup (one 16-column MFMA tile)
constexpr int BL_PF = 4;            // k-steps of the weight stream in flight per wavefront (divides 16)
constexpr int BL_YP = 516;          // float pitch of the float32 row tiles in LDS (2 064 B: the 16 rows of a tile spread over all banks)
constexpr int BL_HP = 260;          // float pitch of the heads' hidden rows
constexpr int BL_LMAX = 24;         // positions per image held in LDS (L = seq_length + 2 <= 24)
// LDS map (bytes)
constexpr int BL_OFF_Y = 0;                              // float32 [16][516] residual rows
constexpr int BL_OFF_X = BL_OFF_Y + BL_G * BL_YP * 4;    // fp16 [16][512] GEMM input rows, 16-byte chunks XOR-swizzled by row (rowblock.hip's rb_off)
constexpr int BL_OFF_BIG = BL_OFF_X + BL_G * 1024;       // 64 KiB: fp16 hidden rows [16][dff] | float32 queries + probabilities | heads' hidden + logits
constexpr int BL_OFF_SCT = BL_OFF_BIG + 65536;           // float32 [L*10][8] self-attention scores of row 0 by (position, label, head)
constexpr int BL_OFF_ST = BL_OFF_SCT + BL_LMAX * 10 * 8 * 4;       // int32 slot state: last, finished, phrase_num, att_len [16]; ext_syn, phrase_length, phrase_syn [16][24]; picks [16][2]
constexpr int BL_ST_INTS = 4 * BL_G + 3 * BL_G * BL_LMAX + 2 * BL_G + BL_G + 16;      // (... + s_act[16 + 1] + the pair's role word s_act[BL_G + 1])
constexpr int BL_SMEM = BL_OFF_ST + BL_ST_INTS * 4;

__device__ __forceinline__ float bl_clamp16(float v) { return v > 65504.f ? 65504.f : (v < -65504.f ? -65504.f : v); }      // (a NaN fails both tests and passes)
__device__ __forceinline__ uint32_t bl_pack(float lo, float hi) {
    const f32x2 v = {bl_clamp16(lo), bl_clamp16(hi)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// fp16 saturation bookkeeping (round 6): the largest magnitude about to be converted; fmaxf drops a NaN operand (a NaN is not a saturation: it passes through
// bl_clamp16 as the reference's NaN would); one branch per group of values, taken only when a clamp really fires
__device__ __forceinline__ float bl_amax(float m, float v) { return fmaxf(m, fabsf(v)); }
__device__ __forceinline__ void bl_note(int* sat, float m) { if (m > 65504.f) atomicOr(sat, 1); }
__device__ int g_bl_sat_scratch;               // where the kernel notes saturations when the caller gave no word
__device__ __forceinline__ float bl_relu(float v) { return v < 0.f ? 0.f : v; }      // torch.relu: a NaN stays a NaN (fmaxf would turn it into 0)
__device__ __forceinline__ f16x8 bl_ldw(const u32x4* p) { return __builtin_bit_cast(f16x8, *p); }
__device__ __forceinline__ float bl_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bl_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// byte offset of a lane's MFMA B-operand piece (row l15, 16-byte chunk kb*4 + g) in a swizzled tile of `pitch`-byte rows: base ^ (kb << 6)
__device__ __forceinline__ int bl_lane_base(int l15, int g, int pitch) { return l15 * pitch + (((l15 >> 2) << 6) | ((g ^ (l15 & 3)) << 4)); }

// the 16 B-operand pieces (K = 512) of this lane from a swizzled tile
__device__ __forceinline__ void bl_load_x(const unsigned char* tile, int lbase, f16x8 (&xb)[16]) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) xb[kb] = *reinterpret_cast<const f16x8*>(tile + (lbase ^ (kb << 6)));
}

// One SEGMENT of a wavefront's weight stream: 16 k-steps of four 16-column tiles (64 output columns x K = 512).  `cur` / `nxt`: this segment and the one
// that follows in the stream (lane offset included; step kb at + kb*256, tile nt at + nt*64); the ring wb holds steps kb .. kb + BL_PF - 1.
// Both are WAVE-UNIFORM pointers (scalar registers); the lane rides in the load's vector offset: no per-load address registers.
__device__ __forceinline__ void bl_seg(const u32x4* cur, const u32x4* nxt, int lane, f16x8 (&wb)[BL_PF * 4], const f16x8 (&xb)[16], f32x4 (&acc)[4]) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(kb % BL_PF) * 4 + nt], xb[kb], acc[nt], 0, 0, 0);
        const u32x4* src = kb + BL_PF < 16 ? cur + (kb + BL_PF) * 256 : nxt + (kb + BL_PF - 16) * 256;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wb[(kb % BL_PF) * 4 + nt] = bl_ldw(src + nt * 64 + lane);
        __builtin_amdgcn_sched_barrier(0);            // (the scheduler would sink the loads next to their use: the prefetch distance is the point)
    }
}

// ---- the pair's exchange (two workgroups per group): 16-byte write-through stores and agent-scope loads (cache policy sc1: the two workgroups may sit on different XCDs,
// whose L2s are not coherent with each other), relaxed agent-scope atomics for the state word and the iteration counters; no fences (dev/exp/grid_barrier_probe.hip: correct, 1.4 us)
__device__ __forceinline__ unsigned bl_cas(unsigned* p, unsigned expect, unsigned want) {
    __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;                                     // the value found
}

template <int NTT>      // the cross-attention's form: 5 = at most 36 regions (every BASELINE config), 8 = at most 64 (scores of an image in registers), 0 = any count (online softmax)
__global__ __launch_bounds__(512) void bound_loop_kernel(BoundLoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float* Y = reinterpret_cast<float*>(sm + BL_OFF_Y);
    unsigned char* X16 = sm + BL_OFF_X;
    unsigned char* BIG = sm + BL_OFF_BIG;
    float* Q = reinterpret_cast<float*>(BIG);                       // cross-attention queries [16][516]
    float* PSELF = reinterpret_cast<float*>(BIG);                   // self-attention probabilities [16][8][32]
    float* HID = reinterpret_cast<float*>(BIG);                     // heads' hidden rows [16][260]
    float* LG = HID + BL_G * BL_HP;                                 // logits [16][32]: 0..19 length, 20..29 label
    float* SCT = reinterpret_cast<float*>(sm + BL_OFF_SCT);
    int* s_last = reinterpret_cast<int*>(sm + BL_OFF_ST);
    int* s_fin = s_last + BL_G;
    int* s_pn = s_fin + BL_G;
    int* s_attl = s_pn + BL_G;
    int* s_ext = s_attl + BL_G;                                     // [16][24]
    int* s_plen = s_ext + BL_G * BL_LMAX;                           // [16][24] phrase_length by slot
    int* s_psyn = s_plen + BL_G * BL_LMAX;                          // [16][24] phrase_syn by slot
    int* s_pick = s_psyn + BL_G * BL_LMAX;                          // [16][2]
    int* s_act = s_pick + 2 * BL_G;                                 // [16] the group's unfinished images, compacted; [16] = their count

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;      // (wave in a scalar register: stream pointers stay scalar)
    const int B = a.B, L = a.L, S = a.S, R = a.R, hh = a.hh, dff = a.dff;
    // (one group per workgroup, spread over all eight XCDs by the dispatcher's round-robin: confining the groups to fewer L2s -- round 5's BOFI_BL_XCDS -- measured -4 / -11 / -17 %,
    // profiles/r05_bound_loop_xcds_ab.txt, and left in round 6)
    // Two workgroups per group (a.pair): the first to arrive claims the group (role 0) and runs on; the second, if it arrives before the first reaches its first
    // feed-forward stage, joins as role 1 -- else the first has gone solo by then (state 3) and the second leaves at once.  A workgroup only ever waits for a partner that
    // is RUNNING (state 4): no co-residency assumption, no deadlock whatever else occupies the chip.  Results do not depend on the mode: solo computes the same two partial
    // sums of y3 (hidden units [0, dff/2) and [dff/2, dff)) and adds them in the same order.
    int grp = blockIdx.x, role = 0, paired = 0;                      // paired: 0 solo, 1 first of a pair not formed yet, 2 pair formed
    unsigned* ctl = nullptr;
    if (a.pair) {
        grp = (int)blockIdx.x >> 1;
        ctl = a.xctl + (size_t)grp * 4;
        if (tid == 0) {
            const unsigned prev = bl_cas(ctl, 0u, 2u);
            int r = 0;
            if (prev != 0u) r = (prev == 2u && bl_cas(ctl, 2u, 4u) == 2u) ? 1 : -1;
            s_act[BL_G + 1] = r;
        }
        __syncthreads();
        role = s_act[BL_G + 1];
        if (role < 0) return;                                       // (uniform: the first workgroup runs this group alone)
        paired = role == 1 ? 2 : 1;
    }
    const int b0 = grp * BL_G;
    const BoundState st = a.st;
    if (tid == 0 && a.wsat && *a.wsat) atomicOr(a.sat, 2);          // the fp16 weight copies themselves were clamped when they were packed
    if (a.dbg && blockIdx.x == 0 && tid == 0) g_bl_stamps[0] = __builtin_amdgcn_s_memtime();

    // ---- this wavefront's weight streams (a weight address never depends on data: the ring runs ahead across stages)
    const int nc1 = dff >> 9;                                       // 64-column chunks of w_1 per wavefront (dff / 64 / 8) = 512-wide K segments of w_2
    const u32x4* wo_self_s = a.wo_self + (size_t)wave * 4096;       // (wave-uniform; the lane is added by the loads)
    const u32x4* wq_s = a.wq_src + (size_t)wave * 4096;
    const u32x4* wo_src_s = a.wo_src + (size_t)wave * 4096;
    // w_1: the FIRST chunk this workgroup's S6 runs (S5's ring requests it): hidden half 0, wavefront w's chunks there (half 1 for a pair's second workgroup); see S6
    const u32x4* w1_s = a.w1 + (size_t)((nc1 & 1) ? wave * nc1 : (role == 1 ? 4 * nc1 : 0) + wave * (nc1 >> 1)) * 4096;
    const u32x4* w2_s = a.w2 + (size_t)wave * (dff >> 5) * 256;     // K segment sg at + sg*4096
    const u32x4* wh_s = a.wh + (size_t)(wave & 3) * 4096;           // the heads' hidden layers: 256 columns, wavefronts 0-3
    f16x8 wb[BL_PF * 4];
#pragma unroll
    for (int p = 0; p < BL_PF; ++p)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wb[p * 4 + nt] = bl_ldw(wo_self_s + p * 256 + nt * 64 + lane0);

    // ---- tables and state into LDS
    for (int i = tid; i < L * 10 * 8; i += 512) SCT[i] = a.sctab[i];
    // output layers of both heads (float32): thread (o, q) = (tid >> 4, tid & 15) of the first 480 holds every 16th weight of output o for the whole loop
    constexpr int W2T = 8;                                          // terms per thread: hh <= 16 * W2T
    const int w2o = tid >> 4, w2q = tid & 15;
    float w2v[W2T];
#pragma unroll
    for (int t = 0; t < W2T; ++t) {
        const int k = w2q + 16 * t;
        w2v[t] = (w2o < 30 && k < hh) ? (w2o < 20 ? a.len_w2[w2o * hh + k] : a.syn_w2[(w2o - 20) * hh + k]) : 0.f;
    }
    const float w2b = w2o < 20 ? a.len_b2[w2o] : (w2o < 30 ? a.syn_b2[w2o - 20] : 0.f);
    {
        const int* ext_src = a.ext_syn_in ? a.ext_syn_in : st.ext_syn;
        for (int i = tid; i < BL_G * BL_LMAX; i += 512) {
            const int im = i / BL_LMAX, p = i - im * BL_LMAX;
            const bool in = b0 + im < B && p < L;
            s_ext[i] = in ? ext_src[(size_t)(b0 + im) * L + p] : 0;
            s_plen[i] = (in && a.update) ? st.phrase_length[(size_t)(b0 + im) * L + p] : 0;
            s_psyn[i] = (in && a.update) ? st.phrase_syn[(size_t)(b0 + im) * L + p] : 0;
        }
        if (tid < BL_G) {
            const int b = b0 + tid;
            const bool valid = b < B;
            s_last[tid] = valid ? (a.last_in ? a.last_in[b] : st.last[b]) : 1;
            s_fin[tid] = valid ? (a.update ? st.finished[b] : 0) : 1;
            s_pn[tid] = (valid && a.update) ? st.phrase_num[b] : 0;
            s_attl[tid] = valid ? (a.att_len ? max(0, min(a.att_len[b], R)) : R) : R;
        }
    }
    __syncthreads();

    int it_done = 0;

    // LayerNorm (no gain / bias: they live in the consumer's weights) of the 16 rows of Y -> X16; wavefront w owns rows 2w, 2w + 1
    auto norm_rows = [&](int lane) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = 2 * wave + u;
            const float4 v0 = *reinterpret_cast<const float4*>(Y + row * BL_YP + lane * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(Y + row * BL_YP + lane * 8 + 4);
            const float mean = wave_sum(((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w))) * (1.0f / 512.0f);
            const float d0 = v0.x - mean, d1 = v0.y - mean, d2 = v0.z - mean, d3 = v0.w - mean;
            const float d4 = v1.x - mean, d5 = v1.y - mean, d6 = v1.z - mean, d7 = v1.w - mean;
            const float var = wave_sum(((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7))) * (1.0f / 511.0f);
            const float rs = 1.0f / (sqrtf(var) + 1e-6f);
            u32x4 o;
            o[0] = bl_pack(d0 * rs, d1 * rs); o[1] = bl_pack(d2 * rs, d3 * rs); o[2] = bl_pack(d4 * rs, d5 * rs); o[3] = bl_pack(d6 * rs, d7 * rs);
            *reinterpret_cast<u32x4*>(X16 + row * 1024 + ((lane ^ (row & 15)) << 4)) = o;
        }
    };

    const int max_iters = a.max_iters;
#pragma unroll 1
    for (int it = 0; it < max_iters; ++it) {
        {   // every image of the group finished: the loop is over (TransformerModel.py:1869, per group)
            int nf = 0;
#pragma unroll
            for (int i = 0; i < BL_G; ++i) nf += s_fin[i];
            if (nf == BL_G) break;
        }
        ++it_done;
        // the unfinished images, compacted: the per-image stages (self-attention values, cross-attention) give wavefront w the entries w and w + 8, so a group
        // with <= 8 images left runs them in one round (the list is read behind the barrier that follows the self-attention probabilities)
        if (tid < 64) {
            const bool act = tid < BL_G && !s_fin[tid];
            const unsigned long long mask = __ballot(act);
            if (act) s_act[__popcll(mask & ((1ull << tid) - 1ull))] = tid;
            if (tid == 0) s_act[BL_G] = __popcll(mask);
        }
        // (lane-derived addresses are re-derived per iteration from a value the optimiser cannot see through: hoisted out of the loop they are
        // spilled at the cross-attention's register peak, and a scratch reload drains the weight ring)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, g = lane >> 4;
        const int xbase = bl_lane_base(l15, g, 1024);
        const int ncol = wave * 64 + g * 4;                         // first output column of tile 0 of this lane (tile nt: + nt*16) in a 512-wide stage
        BL_STAMP(0);
        // ================= S1: row-0 self-attention over the (position, label) tables =================
        {   // probabilities: a 32-lane half per (image, head), lane = key position (row 0 sees keys p < last: tgt_mask[j, 0, :last])
            const int hw = wave * 2 + (lane >> 5), li = lane & 31;
#pragma unroll 1
            for (int rnd = 0; rnd < 8; ++rnd) {
                const int pair = rnd * 16 + hw, i = pair >> 3, h = pair & 7;
                if (s_fin[i]) continue;                                   // (the wavefront's two halves hold the same image)
                const int n = min(s_last[i], L);
                const float sc = li < n ? SCT[(li * 10 + s_ext[i * BL_LMAX + li]) * 8 + h] : -INFINITY;
                const float m = xor16_max(row16_max(sc));
                const float e = li < n ? expf(sc - m) : 0.f;
                const float sum = xor16_sum(row16_sum(e));
                PSELF[(i * 8 + h) * 32 + li] = e / sum;
            }
        }
        __syncthreads();
        BL_STAMP(1);
        {   // ctx = P . V rows of the float32 table: wavefront w owns the active images w and w + 8, a lane 8 columns (head lane / 8)
            const int h = lane >> 3, n_act = s_act[BL_G];
#pragma unroll 1
            for (int idx = wave; idx < n_act; idx += 8) {
                const int i = s_act[idx], n = min(s_last[i], L);
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
                for (int j0 = 0; j0 < n; j0 += 8) {
                    float4 v0[8], v1[8];
                    float p[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int jj = min(j0 + t, n - 1);
                        const float* vr = a.vtab + (size_t)(jj * 10 + s_ext[i * BL_LMAX + jj]) * 512 + lane * 8;
                        v0[t] = *reinterpret_cast<const float4*>(vr);
                        v1[t] = *reinterpret_cast<const float4*>(vr + 4);
                        p[t] = j0 + t < n ? PSELF[(i * 8 + h) * 32 + j0 + t] : 0.f;
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        acc[0] = fmaf(p[t], v0[t].x, acc[0]); acc[1] = fmaf(p[t], v0[t].y, acc[1]); acc[2] = fmaf(p[t], v0[t].z, acc[2]); acc[3] = fmaf(p[t], v0[t].w, acc[3]);
                        acc[4] = fmaf(p[t], v1[t].x, acc[4]); acc[5] = fmaf(p[t], v1[t].y, acc[5]); acc[6] = fmaf(p[t], v1[t].z, acc[6]); acc[7] = fmaf(p[t], v1[t].w, acc[7]);
                    }
                }
                u32x4 o;
                bl_note(a.sat, bl_amax(bl_amax(bl_amax(fabsf(acc[0]), acc[1]), bl_amax(fabsf(acc[2]), acc[3])), bl_amax(bl_amax(fabsf(acc[4]), acc[5]), bl_amax(fabsf(acc[6]), acc[7]))));
                o[0] = bl_pack(acc[0], acc[1]); o[1] = bl_pack(acc[2], acc[3]); o[2] = bl_pack(acc[4], acc[5]); o[3] = bl_pack(acc[6], acc[7]);
                *reinterpret_cast<u32x4*>(X16 + i * 1024 + ((lane ^ (i & 15)) << 4)) = o;
            }
            // (rows of finished images keep what they held: their MFMA columns are never read)
        }
        __syncthreads();
        BL_STAMP(2);
        f16x8 xb[16];
        f32x4 acc[4];
        // ================= S2: y1 = (x0 + bo_self) + Wo_self . ctx =================
        {
            float4 cv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4*>(a.x0b + ncol + nt * 16);
            bl_load_x(X16, xbase, xb);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bl_seg(wo_self_s, wq_s, lane, wb, xb, acc);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<float4*>(Y + l15 * BL_YP + ncol + nt * 16) = make_float4(acc[nt][0] + cv[nt].x, acc[nt][1] + cv[nt].y, acc[nt][2] + cv[nt].z, acc[nt][3] + cv[nt].w);
        }
        __syncthreads();
        BL_STAMP(3);
        norm_rows(lane);
        __syncthreads();
        BL_STAMP(4);
        // ================= S3: q = Wq_src' . LN(y1) + c =================
        {
            float4 cv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4*>(a.cq + ncol + nt * 16);
            bl_load_x(X16, xbase, xb);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bl_seg(wq_s, wo_src_s, lane, wb, xb, acc);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<float4*>(Q + l15 * BL_YP + ncol + nt * 16) = make_float4(acc[nt][0] + cv[nt].x, acc[nt][1] + cv[nt].y, acc[nt][2] + cv[nt].z, acc[nt][3] + cv[nt].w);
        }
        __syncthreads();
        BL_STAMP(5);
        // ================= S4: cross-attention of the 16 query rows over their images' regions =================
        // Wavefront w owns the active images w and w + 8 of the compacted list.  A lane owns 8 columns (lane*8 .. +7: head lane / 8) of EVERY region row: a load instruction of the
        // wavefront is one whole 1-KiB K (or V) row.  A key's score is the sum over the 8 lanes of its head's octet (three DPP steps); the softmax runs
        // in the lane (the octet's lanes hold the same numbers), P.V accumulates in the lane: no cross-lane traffic besides the octet sums, no LDS.
        // Rows go through two register buffers of BR rows, NB batches per pass (NB even: the buffers alternate across the K pass, the V pass and
        // the next image without a drain); every load is unconditional (rows past R: the last row again, weighted 0) so that the waits count exactly.
        if constexpr (NTT == 0) {
            // any region count (real bottom-up features have up to 100, captioning/utils/opts.py:84): K and V rows in batches of 6 through two register buffers, an
            // ONLINE softmax per image (running maximum m and sum l, the accumulated P.V rescaled when the maximum moves): the scores of a whole image need not fit
            // the registers.  Same lane layout and load discipline as the fixed forms below.
            asm volatile("" : "+v"(lane));
            constexpr int BR = 6;
            const bf16_t* kbase = a.k + lane * 8;
            const bf16_t* vbase = a.v + lane * 8;
            u32x4 kb[2][BR], vb[2][BR];
            const int nb = (R + BR - 1) / BR;
            auto issue = [&](u32x4 (&kk)[BR], u32x4 (&vv)[BR], int bi, int j0) {
                const size_t base = (size_t)bi * R * a.ldkv;
#pragma unroll
                for (int r = 0; r < BR; ++r) kk[r] = *reinterpret_cast<const u32x4*>(kbase + base + (size_t)min(j0 + r, R - 1) * a.ldkv);
#pragma unroll
                for (int r = 0; r < BR; ++r) vv[r] = *reinterpret_cast<const u32x4*>(vbase + base + (size_t)min(j0 + r, R - 1) * a.ldkv);
            };
            const int n_act = s_act[BL_G];
            if (wave < n_act) issue(kb[0], vb[0], min(b0 + s_act[wave], B - 1), 0);
#pragma unroll 1
            for (int idx = wave; idx < n_act; idx += 8) {
                const int i = s_act[idx], bi = min(b0 + i, B - 1), kl = s_attl[i];
                const int bi_next = min(b0 + s_act[min(idx + 8, n_act - 1)], B - 1);
                const float4 qa = *reinterpret_cast<const float4*>(Q + i * BL_YP + lane * 8);
                const float4 qb = *reinterpret_cast<const float4*>(Q + i * BL_YP + lane * 8 + 4);
                float m = -INFINITY, l = 0.f;
                float o[8];
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) o[e8] = 0.f;
                auto compute = [&](const u32x4 (&kk)[BR], const u32x4 (&vv)[BR], int j0) {
                    float sc[BR];
                    float mb = -INFINITY;
#pragma unroll
                    for (int r = 0; r < BR; ++r) {
                        const u32x4 kv = kk[r];
                        float d = qa.x * bl_lo(kv[0]);
                        d = fmaf(qa.y, bl_hi(kv[0]), d); d = fmaf(qa.z, bl_lo(kv[1]), d); d = fmaf(qa.w, bl_hi(kv[1]), d);
                        d = fmaf(qb.x, bl_lo(kv[2]), d); d = fmaf(qb.y, bl_hi(kv[2]), d); d = fmaf(qb.z, bl_lo(kv[3]), d); d = fmaf(qb.w, bl_hi(kv[3]), d);
                        d = oct_sum(d) * 0.125f;
                        sc[r] = (j0 + r < kl) ? d : -INFINITY;           // (rows past R were loaded as the last row: j0 + r >= R >= kl masks them)
                        mb = fmaxf(mb, sc[r]);
                    }
                    const float m_new = fmaxf(m, mb);
                    const float scale = (m_new == -INFINITY) ? 1.f : expf(m - m_new);      // nothing seen yet: nothing to rescale
                    l *= scale;
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) o[e8] *= scale;
#pragma unroll
                    for (int r = 0; r < BR; ++r) {
                        const float p = (j0 + r < kl) ? expf(sc[r] - m_new) : 0.f;
                        l += p;
                        const u32x4 v = vv[r];
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) { o[2 * e4] = fmaf(p, bl_lo(v[e4]), o[2 * e4]); o[2 * e4 + 1] = fmaf(p, bl_hi(v[e4]), o[2 * e4 + 1]); }
                    }
                    m = m_new;
                };
#pragma unroll 1
                for (int bt = 0; bt < nb; bt += 2) {
                    issue(kb[1], vb[1], bi, min(bt + 1, nb - 1) * BR);          // (an odd count: the last batch again, not consumed)
                    compute(kb[0], vb[0], bt * BR);
                    const bool more = bt + 2 < nb;
                    issue(kb[0], vb[0], more ? bi : bi_next, more ? (bt + 2) * BR : 0);      // the next pair's first batch, or the next image's
                    if (bt + 1 < nb) compute(kb[1], vb[1], (bt + 1) * BR);
                }
                const float inv = 1.0f / l;                                   // no visible region: 0 * (1 / 0) = NaN, as softmax over an all-masked row of -inf
                u32x4 w;
                bl_note(a.sat, bl_amax(bl_amax(bl_amax(fabsf(o[0]), o[1]), bl_amax(fabsf(o[2]), o[3])), bl_amax(bl_amax(fabsf(o[4]), o[5]), bl_amax(fabsf(o[6]), o[7]))) * fabsf(inv));
                w[0] = bl_pack(o[0] * inv, o[1] * inv); w[1] = bl_pack(o[2] * inv, o[3] * inv); w[2] = bl_pack(o[4] * inv, o[5] * inv); w[3] = bl_pack(o[6] * inv, o[7] * inv);
                *reinterpret_cast<u32x4*>(X16 + i * 1024 + ((lane ^ (i & 15)) << 4)) = w;
            }
        } else
        {
            asm volatile("" : "+v"(lane));
            constexpr int BR = NTT == 5 ? 9 : 8, NB = NTT == 5 ? 4 : 8, RMAX = BR * NB;      // 36 rows (every BASELINE config) or 64
            const bf16_t* kbase = a.k + lane * 8;
            const bf16_t* vbase = a.v + lane * 8;
            u32x4 buf[2][BR];
            auto issue = [&](u32x4 (&bb)[BR], const bf16_t* base, int bi, int j0) {
                const bf16_t* p = base + (size_t)bi * R * a.ldkv;
#pragma unroll
                for (int r = 0; r < BR; ++r) bb[r] = *reinterpret_cast<const u32x4*>(p + (size_t)min(j0 + r, R - 1) * a.ldkv);
            };
            const int n_act = s_act[BL_G];
            if (wave < n_act) issue(buf[0], kbase, min(b0 + s_act[wave], B - 1), 0);
#pragma unroll 1
            for (int idx = wave; idx < n_act; idx += 8) {
                const int i = s_act[idx], bi = min(b0 + i, B - 1), kl = s_attl[i];
                const int bi_next = min(b0 + s_act[min(idx + 8, n_act - 1)], B - 1);       // (after the last image: a redundant batch of its own rows)
                const float4 qa = *reinterpret_cast<const float4*>(Q + i * BL_YP + lane * 8);
                const float4 qb = *reinterpret_cast<const float4*>(Q + i * BL_YP + lane * 8 + 4);
                float sc[RMAX];
                float m = -INFINITY;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    if (bt + 1 < NB) issue(buf[(bt + 1) & 1], kbase, bi, (bt + 1) * BR);
                    else issue(buf[0], vbase, bi, 0);
#pragma unroll
                    for (int r = 0; r < BR; ++r) {
                        const u32x4 kv = buf[bt & 1][r];
                        float d = qa.x * bl_lo(kv[0]);
                        d = fmaf(qa.y, bl_hi(kv[0]), d); d = fmaf(qa.z, bl_lo(kv[1]), d); d = fmaf(qa.w, bl_hi(kv[1]), d);
                        d = fmaf(qb.x, bl_lo(kv[2]), d); d = fmaf(qb.y, bl_hi(kv[2]), d); d = fmaf(qb.z, bl_lo(kv[3]), d); d = fmaf(qb.w, bl_hi(kv[3]), d);
                        d = oct_sum(d) * 0.125f;                              // / sqrt(d_k), d_k = 64
                        const int j = bt * BR + r;
                        sc[j] = j < kl ? d : -INFINITY;
                        m = fmaxf(m, sc[j]);
                    }
                }
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < RMAX; ++j) { sc[j] = j < kl ? expf(sc[j] - m) : 0.f; sum += sc[j]; }
                const float inv = 1.0f / sum;                                 // no visible region: 0 * (1 / 0) = NaN, as softmax over an all-masked row of -inf
                float o[8];
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) o[e8] = 0.f;
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    if (bt + 1 < NB) issue(buf[(bt + 1) & 1], vbase, bi, (bt + 1) * BR);
                    else issue(buf[0], kbase, bi_next, 0);                    // the next image's first K rows
#pragma unroll
                    for (int r = 0; r < BR; ++r) {
                        const u32x4 v = buf[bt & 1][r];
                        const float p = sc[bt * BR + r] * inv;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) { o[2 * e4] = fmaf(p, bl_lo(v[e4]), o[2 * e4]); o[2 * e4 + 1] = fmaf(p, bl_hi(v[e4]), o[2 * e4 + 1]); }
                    }
                }
                u32x4 w;
                bl_note(a.sat, bl_amax(bl_amax(bl_amax(fabsf(o[0]), o[1]), bl_amax(fabsf(o[2]), o[3])), bl_amax(bl_amax(fabsf(o[4]), o[5]), bl_amax(fabsf(o[6]), o[7]))));
                w[0] = bl_pack(o[0], o[1]); w[1] = bl_pack(o[2], o[3]); w[2] = bl_pack(o[4], o[5]); w[3] = bl_pack(o[6], o[7]);
                *reinterpret_cast<u32x4*>(X16 + i * 1024 + ((lane ^ (i & 15)) << 4)) = w;
            }
        }
        __syncthreads();
        BL_STAMP(6);
        // ================= S5: y2 = y1 + Wo_src . ctx2 + bo =================
        {
            float4 cv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4*>(a.bo_src + ncol + nt * 16);
            bl_load_x(X16, xbase, xb);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bl_seg(wo_src_s, w1_s, lane, wb, xb, acc);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float4* yp = reinterpret_cast<float4*>(Y + l15 * BL_YP + ncol + nt * 16);
                const float4 y = *yp;
                *yp = make_float4(y.x + (acc[nt][0] + cv[nt].x), y.y + (acc[nt][1] + cv[nt].y), y.z + (acc[nt][2] + cv[nt].z), y.w + (acc[nt][3] + cv[nt].w));
            }
        }
        __syncthreads();
        BL_STAMP(7);
        norm_rows(lane);
        __syncthreads();
        BL_STAMP(8);
        // ================= S6: h = relu(W1' . LN(y2) + c1) =================
        // The hidden units come in two HALVES, [0, dff/2) and [dff/2, dff) (dff a multiple of 1 024; else one "half" holds them all): a pair's workgroup takes the half of its
        // role, a workgroup that runs alone both, one after the other -- the two partial sums of y3 are kept apart either way and added in the same order.
        // Half hf, wavefront w: hidden chunks (of 64 columns) hf*4*nc1 + w*ncw + cc, cc < ncw = nc1 / 2.
        if (paired == 1) {                                            // the first workgroup of a pair decides HERE, once: has the partner arrived (4) or not (2 -> 3: alone from now on)?
            if (tid == 0) s_act[BL_G + 1] = bl_cas(ctl, 2u, 3u) == 2u ? 0 : 2;
            __syncthreads();
            paired = s_act[BL_G + 1];
        }
        const bool split = !(nc1 & 1);
        const int ncw = split ? nc1 >> 1 : nc1;                       // chunks of a wavefront per half / 512-unit segments of w_2 per half
        const int nrun = (split && paired != 2) ? 2 * ncw : ncw;      // chunks this workgroup's wavefronts run: one half (a pair's workgroup) or both
        // chunk j of this workgroup's walk -> index of the 64-column hidden chunk (half hf: chunks hf*4*nc1 + w*ncw + c)
        auto chunk_of = [&](int j) {
            if (!split) return wave * nc1 + j;
            const int hf = paired == 2 ? role : (j >= ncw ? 1 : 0);
            return hf * 4 * nc1 + wave * ncw + (j >= ncw ? j - ncw : j);
        };
        const int sg_first = (split && paired == 2) ? role * ncw : 0;   // the first w_2 segment S7 runs (its first steps are requested by S6's last chunk)
        bl_load_x(X16, xbase, xb);
#pragma unroll 1
        for (int j = 0; j < nrun; ++j) {
            const int ch = chunk_of(j);
            const int hc = ch * 64 + g * 4;
            float4 cv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4*>(a.c1 + hc + nt * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bl_seg(a.w1 + (size_t)ch * 4096, j + 1 < nrun ? a.w1 + (size_t)chunk_of(j + 1) * 4096 : w2_s + (size_t)sg_first * 4096, lane, wb, xb, acc);
            float hmax = 0.f;                                                 // (hidden values are >= 0 after the ReLU: only the upper end can saturate)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = hc + nt * 16;                                   // 4 columns n .. n + 3 of row l15: half of 16-byte chunk n >> 3
                const float h0 = bl_relu(acc[nt][0] + cv[nt].x), h1 = bl_relu(acc[nt][1] + cv[nt].y), h2 = bl_relu(acc[nt][2] + cv[nt].z), h3 = bl_relu(acc[nt][3] + cv[nt].w);
                hmax = fmaxf(fmaxf(hmax, fmaxf(h0, h1)), fmaxf(h2, h3));
                const uint2 o = make_uint2(bl_pack(h0, h1), bl_pack(h2, h3));
                *reinterpret_cast<uint2*>(BIG + l15 * (dff * 2) + (((n >> 3) ^ l15) << 4) + (n & 4) * 2) = o;
            }
            bl_note(a.sat, hmax);
        }
        __syncthreads();
        BL_STAMP(9);
        // ================= S7: y3 = ((y2 + p0) + p1) + b2, p_hf = W2[:, half hf] . h[half hf] =================
        // ONE summation order whatever the mode: a workgroup that runs alone adds p0 into the residual rows when half 0's segments are through and p1 (+ b2) after half 1's;
        // a pair's workgroup holds its own partial sum in the accumulators and takes the partner's from the exchange.
        {
            const int hbase = bl_lane_base(l15, g, dff * 2);
            const int nseg = (split && paired != 2) ? 2 * ncw : ncw;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int j = 0; j < nseg; ++j) {
                const int sg = sg_first + j;
                bl_load_x(BIG, hbase ^ (sg << 10), xb);
                const u32x4* nxt = j + 1 < nseg ? w2_s + (size_t)(sg + 1) * 4096 : (wave < 4 ? wh_s : wo_self_s);
                bl_seg(w2_s + (size_t)sg * 4096, nxt, lane, wb, xb, acc);
                if (split && paired != 2 && j == ncw - 1) {               // (alone: half 0 is through -> y2 + p0 into the residual rows; the accumulators start p1)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        float4* yp = reinterpret_cast<float4*>(Y + l15 * BL_YP + ncol + nt * 16);
                        const float4 y = *yp;
                        *yp = make_float4(y.x + acc[nt][0], y.y + acc[nt][1], y.z + acc[nt][2], y.w + acc[nt][3]);
                        acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            f32x4 othr[4];
            if (paired == 2) {
                // ---- the pair's exchange: my partial sum out (write-through), the partner's in; double-buffered by the iteration's parity (a workgroup passes exchange i only
                // after its partner has WRITTEN partial i, i.e. has read partial i - 1: buffer (i + 1) & 1 is free by then)
                float* slot = a.xbuf + ((size_t)grp * 2 + (it_done & 1)) * 2 * (BL_G * 512);      // (wave-uniform: a scalar buffer resource over both roles' tiles; the lane rides in the offset)
                const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(slot, 0, 2 * BL_G * 512 * 4, 0x00020000);
                const int lofs = (l15 * 512 + ncol) * 4, mine_o = role * (BL_G * 512 * 4) + lofs, othr_o = (1 - role) * (BL_G * 512 * 4) + lofs;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[nt]), xr, mine_o + nt * 64, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    __hip_atomic_fetch_add(ctl + 1 + role, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool ok = false;
                    for (int spin = 0; spin < (1 << 21); ++spin) {      // (the partner is running: this wait is its lag; the bound is a safety net against a lost workgroup)
                        if (__hip_atomic_load(ctl + 2 - role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)it_done) { ok = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (!ok) atomicOr(a.sat, 4);                          // reported like a saturation: the caller decodes again without the loop kernel
                }
                __syncthreads();
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) othr[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, othr_o + nt * 64, 0, 16));
            }
            float4 cv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4*>(a.b2 + ncol + nt * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float4* yp = reinterpret_cast<float4*>(Y + l15 * BL_YP + ncol + nt * 16);
                const float4 y = *yp;
                f32x4 t;
                if (paired == 2) {                                         // ((y2 + p0) + p1) + b2
                    const f32x4 p0 = role ? othr[nt] : acc[nt], p1 = role ? acc[nt] : othr[nt];
                    t = f32x4{(y.x + p0[0]) + p1[0], (y.y + p0[1]) + p1[1], (y.z + p0[2]) + p1[2], (y.w + p0[3]) + p1[3]};
                } else t = f32x4{y.x + acc[nt][0], y.y + acc[nt][1], y.z + acc[nt][2], y.w + acc[nt][3]};      // (alone: the rows hold y2 + p0 already, acc = p1; no halves: acc = the whole sum)
                *yp = make_float4(t[0] + cv[nt].x, t[1] + cv[nt].y, t[2] + cv[nt].z, t[3] + cv[nt].w);
            }
        }
        __syncthreads();
        BL_STAMP(10);
        norm_rows(lane);
        __syncthreads();
        BL_STAMP(11);
        // ================= S8: heads.  hidden = relu(W1h' . LN(y3) + c) (both heads side by side, 256 columns: wavefronts 0-3) =================
        if (wave < 4) {
            float4 cv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4*>(a.ch + ncol + nt * 16);
            bl_load_x(X16, xbase, xb);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bl_seg(wh_s, wo_self_s, lane, wb, xb, acc);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<float4*>(HID + l15 * BL_HP + ncol + nt * 16) = make_float4(bl_relu(acc[nt][0] + cv[nt].x), bl_relu(acc[nt][1] + cv[nt].y),
                                                                                              bl_relu(acc[nt][2] + cv[nt].z), bl_relu(acc[nt][3] + cv[nt].w));
        }
        __syncthreads();
        BL_STAMP(12);
        {   // output layers (float32): 16 lanes per output, their weights in registers; one DPP row reduction per image.  (Terms k >= hh carry weight 0 and
            // read the row's other columns: finite ReLU outputs, or NaN on a NaN row, where every logit is NaN anyway.)
            const float* hv = HID + (w2o < 20 ? 0 : hh) + w2q;
#pragma unroll 4
            for (int i = 0; i < BL_G; ++i) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int t = 0; t < W2T; t += 2) { s0 = fmaf(w2v[t], hv[i * BL_HP + 16 * t], s0); s1 = fmaf(w2v[t + 1], hv[i * BL_HP + 16 * t + 16], s1); }
                s0 = row16_sum(s0 + s1);
                if (w2q == 0 && w2o < 30) LG[i * 32 + w2o] = s0 + w2b;
            }
        }
        __syncthreads();
        BL_STAMP(13);
        {   // log-softmax and first-max pick: a 32-lane half per (image, head); torch.max semantics: the first NaN wins, else the first maximum (:380-383)
            const int hw = wave * 2 + (lane >> 5), li = lane & 31;
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                const int unit = rnd * 16 + hw, i = unit >> 1, head = unit & 1, nout = head ? 10 : 20;
                const bool on = li < nout;
                const float v = on ? LG[i * 32 + head * 20 + li] : -INFINITY;
                const float m = xor16_max(row16_max(v));                      // fmaxf ignores a NaN unless every input is one
                const float e = on ? expf(v - m) : 0.f;
                const float sum = xor16_sum(row16_sum(e));
                const float lp = (v - m) - logf(sum);
                if (it == 0 && on && b0 + i < B) {
                    if (!head && a.len_logp) a.len_logp[(size_t)(b0 + i) * 20 + li] = lp;
                    if (head && a.syn_logp) a.syn_logp[(size_t)(b0 + i) * 10 + li] = lp;
                }
                const bool isnan_ = on && (lp != lp);
                float cand = isnan_ ? (float)li : 1e9f;
                cand = -xor16_max(row16_max(-cand));
                float cmax = (on && v == m) ? (float)li : 1e9f;
                cmax = -xor16_max(row16_max(-cmax));
                const int best = cand < 1e8f ? (int)cand : (cmax < 1e8f ? (int)cmax : 0);
                if (li == 0) s_pick[i * 2 + head] = best;
            }
        }
        __syncthreads();
        BL_STAMP(14);
        if (tid < BL_G && a.update && b0 + tid < B && !s_fin[tid]) {      // slot bookkeeping of core_NAIC (TransformerModel.py:1843-1869), one thread per image, in LDS
            const int i = tid;
            int ln = s_pick[i * 2];
            const int sn = s_pick[i * 2 + 1], la = s_last[i];
            bool fin = false;
            if (ln == 0 || sn < 4 || sn > 6) {                              // EOS (:1846-1849)
                fin = true;
            } else {
                if (ln + la >= S + 1) { ln = S + 1 - la; fin = true; }      // truncate (:1850-1855)
                const int slot = s_pn[i];                                   // == iteration index while unfinished (Q3)
                s_plen[i * BL_LMAX + slot] = ln;
                s_psyn[i * BL_LMAX + slot] = sn;
                s_pn[i] = slot + 1;
                for (int p = la; p < la + ln; ++p) s_ext[i * BL_LMAX + p] = sn;
                s_last[i] = la + ln;                                        // (tgt_mask[j, 0, :la+ln] = True :1859-1867: row 0 sees keys p < last)
            }
            if (fin) s_fin[i] = 1;
        }
        __syncthreads();
        BL_STAMP(15);
    }
    if (a.dbg && blockIdx.x == 0 && tid == 0) { g_bl_stamps[1] = __builtin_amdgcn_s_memtime(); g_bl_stamps[31] = (unsigned long long)it_done; }
    if (role == 1) return;                                        // (the pair's second workgroup holds the same state: the first one writes it back)
    if (a.update) {      // the group's slot state back to the engine's arrays (the filling pass and the export read them)
        for (int i = tid; i < BL_G * BL_LMAX; i += 512) {
            const int im = i / BL_LMAX, p = i - im * BL_LMAX;
            if (b0 + im < B && p < L) {
                const size_t o = (size_t)(b0 + im) * L + p;
                st.ext_syn[o] = s_ext[i]; st.phrase_length[o] = s_plen[i]; st.phrase_syn[o] = s_psyn[i];
            }
        }
        if (tid < BL_G && b0 + tid < B) { st.last[b0 + tid] = s_last[tid]; st.finished[b0 + tid] = s_fin[tid]; st.phrase_num[b0 + tid] = s_pn[tid]; }
    }
    if (tid == 0 && a.update) {
        atomicMax(&st.counters[1], it_done);                               // iterations in which some image (of any group) was active
        int nf = 0;
        for (int i = 0; i < BL_G; ++i) nf += (b0 + i < B) ? s_fin[i] : 0;
        atomicAdd(&st.counters[0], nf);
        if (paired == 2) atomicAdd(&st.counters[5], 1);                    // (diagnostic: groups whose loop ran as a pair of workgroups; read through bofi_engine_debug_copy "counters")
    }
}

__global__ void bl_zero_words_kernel(int* w, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) w[i] = 0; }

int launch_bound_loop(const BoundLoopArgs& a, hipStream_t s) {
    if (a.B < 1 || a.R < 1 || a.R > 128 || a.L < 3 || a.L > BL_LMAX || a.S != a.L - 2 || a.hh < 1 || a.hh > 128 || 2 * a.hh > 256 || a.dff < 512 || a.dff > 2048 ||
        a.dff % 512 || a.ldkv % 8 || a.max_iters < 1)
        return BOFI_ERR_ARG;
    if (!a.wo_self || !a.x0b || !a.wq_src || !a.cq || !a.wo_src || !a.bo_src || !a.w1 || !a.c1 || !a.w2 || !a.b2 || !a.wh || !a.ch || !a.len_w2 || !a.len_b2 ||
        !a.syn_w2 || !a.syn_b2 || !a.sctab || !a.vtab || !a.k || !a.v)
        return BOFI_ERR_ARG;
    if (a.update && (!a.st.last || !a.st.finished || !a.st.phrase_num || !a.st.phrase_length || !a.st.phrase_syn || !a.st.ext_syn || !a.st.counters)) return BOFI_ERR_ARG;
    if (!a.update && (!a.ext_syn_in || !a.last_in)) return BOFI_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(bound_loop_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, BL_SMEM) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(bound_loop_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, BL_SMEM) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(bound_loop_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, BL_SMEM) != hipSuccess)
            return BOFI_ERR_HIP;
        attr_set = true;
    }
    BoundLoopArgs v = a;
    if (!v.sat) {
        static int* scratch = nullptr;
        if (!scratch && hipGetSymbolAddress(reinterpret_cast<void**>(&scratch), HIP_SYMBOL(g_bl_sat_scratch)) != hipSuccess) return BOFI_ERR_HIP;
        v.sat = scratch;
    }
    v.dbg = BOFI_ENV_INT("BOFI_BL_DBG", 0);
    const int groups = (a.B + BL_G - 1) / BL_G;
    // two workgroups per group (BOFI_BL_PAIR, re-read after bofi_reload_env: 0 = never, 1 (default) = launches of at most BOFI_BL_PAIR_MAX_B = 384 images, 2 = always): the
    // whole loop only (the stage API evaluates one iteration), hidden units in two halves of whole 512-unit segments (dff 1 024 or 2 048), and the caller's exchange
    // buffers.  The pair shortens the loop's LATENCY (815 -> 674 us per 320-image launch alone: each workgroup streams 3.75 instead of 5.75 MB per iteration, one 64-KB
    // exchange) at the price of twice the CUs and of the iteration's other stages run twice: +2.1 % on the 20-step region (4 launches of 320 images, where the four loops
    // coincide and most of the chip waits for them), -1.2 % at 640 and -2.5 % at 1 024 images per launch, where the loops hide under other launches' work
    // (profiles/r06_bound_loop_pair_ab.txt).  Results are bit-identical either way.
    const int pair_knob = BOFI_ENV_INT("BOFI_BL_PAIR", 1);
    v.pair = (a.update && a.xbuf && a.xctl && !((a.dff >> 9) & 1) && pair_knob != 0 && (pair_knob == 2 || a.B <= BOFI_ENV_INT("BOFI_BL_PAIR_MAX_B", 384))) ? 1 : 0;
    if (v.pair && a.st.pair_ctl != a.xctl)                  // (the engine's launch_bound_init zeroes the control words with the slot state: no launch of its own)
        hipLaunchKernelGGL(bl_zero_words_kernel, dim3((groups * 4 + 255) / 256), dim3(256), 0, s, reinterpret_cast<int*>(a.xctl), groups * 4);
    const int grid = v.pair ? 2 * groups : groups;
    if (a.R <= 36) hipLaunchKernelGGL(bound_loop_kernel<5>, dim3(grid), dim3(512), BL_SMEM, s, v);
    else if (a.R <= 64) hipLaunchKernelGGL(bound_loop_kernel<8>, dim3(grid), dim3(512), BL_SMEM, s, v);
    else hipLaunchKernelGGL(bound_loop_kernel<0>, dim3(grid), dim3(512), BL_SMEM, s, v);
    BOFI_CHECK_LAUNCH();
    // FLOP tally: the GEMM work of the iterations is data-dependent; counted as skippable work of max_iters iterations like the launches it replaces
    g_gemm_flops_skippable += (double)a.max_iters * a.B * (2.0 * 3 * 512 * 512 + 4.0 * 512 * a.dff);
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// fp16 fragment-major copies of the bounding network's weights, straight from the float32 parameters (finalize: uploaded temporaries; refresh:
// the trainer's tensors): out[i] of [Npad/64 chunks][K/32 steps][4 tiles][64 lanes][8 halves] = fp16(w[n][k .. k+7] * gain[k .. k+7]), rows >= n_each*nsrc zero
__global__ __launch_bounds__(256) void pack_frag16_kernel(Pack16Table t) {
    int e = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < t.n && (int)blockIdx.x >= t.e[i].blk0) e = i;
    const Pack16Entry& d = t.e[e];
    const size_t i = (size_t)((int)blockIdx.x - d.blk0) * 256 + threadIdx.x;
    if (i >= (size_t)d.Npad * d.K / 8) return;
    const int lane = (int)(i & 63), nt = (int)((i >> 6) & 3);
    const size_t tt = i >> 8;
    const int kb = (int)(tt % (size_t)(d.K >> 5)), chunk = (int)(tt / (size_t)(d.K >> 5));
    const int n = chunk * 64 + nt * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (n < d.n_each * d.nsrc) {
        const int src = n / d.n_each, r = n - src * d.n_each;
        const float* row = d.w[src] + (size_t)r * d.K + k;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = d.gain ? row[j] * d.gain[k + j] : row[j];
    }
    u32x4 o;
    if (t.sat) bl_note(t.sat, bl_amax(bl_amax(bl_amax(fabsf(v[0]), v[1]), bl_amax(fabsf(v[2]), v[3])), bl_amax(bl_amax(fabsf(v[4]), v[5]), bl_amax(fabsf(v[6]), v[7]))));
    o[0] = bl_pack(v[0], v[1]); o[1] = bl_pack(v[2], v[3]); o[2] = bl_pack(v[4], v[5]); o[3] = bl_pack(v[6], v[7]);
    reinterpret_cast<u32x4*>(d.out)[i] = o;
}
__global__ void bl_zero_word_kernel(int* w) { *w = 0; }

int launch_pack_frag16(const Pack16Table& t, hipStream_t s) {
    if (t.n < 1 || t.n > 8) return BOFI_ERR_ARG;
    Pack16Table v = t;
    int blocks = 0;
    for (int i = 0; i < v.n; ++i) {
        Pack16Entry& d = v.e[i];
        if (!d.w[0] || !d.out || d.nsrc < 1 || d.nsrc > 2 || (d.nsrc == 2 && !d.w[1]) || d.K % 32 || d.Npad % 64 || d.n_each * d.nsrc > d.Npad) return BOFI_ERR_ARG;
        d.blk0 = blocks;
        blocks += (int)(((size_t)d.Npad * d.K / 8 + 255) / 256);
    }
    if (v.sat) hipLaunchKernelGGL(bl_zero_word_kernel, dim3(1), dim3(1), 0, s, v.sat);       // (a kernel, not a memset node: this runs inside captured refreshes too)
    hipLaunchKernelGGL(pack_frag16_kernel, dim3(blocks), dim3(256), 0, s, v);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Input-independent float32 tables of the bounding layer's row-0 self-attention: the layer input at (position p, label s) is
// xt[p*10 + s] = lut_syn[s]*sqrt(d) + pe[p]; with xn = LN_0(xt[row]) (sublayer[0].norm, TransformerModel.py:1346-1349):
//   q0 = Wq . LN_0(x0) + bq                       (x0: row (0, [LEN]))
//   sctab[row][h] = q0[h] . (Wk . xn + bk)[h] / 8 (the score row 0 gives key `row` in head h)
//   vtab[row] = Wv . xn + bv
// d = 512, 8 heads.  Grid: rows + 1 workgroups of 512 threads; workgroup `rows` computes q0 first -- the others need it: two launches.
__device__ __forceinline__ void bl_ln_row(const float* x, const float* gain, const float* bias, float* xs, int tid, float* red) {
    const float v = x[tid];
    float s = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) * (1.0f / 512.0f);
    const float dlt = v - mean;
    float q = wave_sum(dlt * dlt);
    if ((tid & 63) == 0) red[8 + (tid >> 6)] = q;
    __syncthreads();
    const float var = (((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]))) * (1.0f / 511.0f);
    xs[tid] = gain[tid] * dlt / (sqrtf(var) + 1e-6f) + bias[tid];
    __syncthreads();
}
__device__ __forceinline__ float bl_dot512(const float* wrow, const float* xs, int lane) {      // every lane ends with the sum
    const float4 w0 = *reinterpret_cast<const float4*>(wrow + lane * 8), w1 = *reinterpret_cast<const float4*>(wrow + lane * 8 + 4);
    const float4 x0 = *reinterpret_cast<const float4*>(xs + lane * 8), x1 = *reinterpret_cast<const float4*>(xs + lane * 8 + 4);
    float s = w0.x * x0.x;
    s = fmaf(w0.y, x0.y, s); s = fmaf(w0.z, x0.z, s); s = fmaf(w0.w, x0.w, s);
    s = fmaf(w1.x, x1.x, s); s = fmaf(w1.y, x1.y, s); s = fmaf(w1.z, x1.z, s); s = fmaf(w1.w, x1.w, s);
    return wave_sum(s);
}
__global__ __launch_bounds__(512) void bound_q0_kernel(BoundTablesArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[512];
    __shared__ float red[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    bl_ln_row(a.x0, a.n0g, a.n0b, xs, tid, red);
    for (int j = 0; j < 64; ++j) {
        const int n = wave * 64 + j;
        const float s = bl_dot512(a.wq + (size_t)n * 512, xs, lane);
        if (lane == 0) a.q0[n] = s + a.bq[n];
    }
}
__global__ __launch_bounds__(512) void bound_tables_kernel(BoundTablesArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[512];
    __shared__ float red[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = blockIdx.x;
    bl_ln_row(a.xt + (size_t)row * 512, a.n0g, a.n0b, xs, tid, red);
    float sc = 0.f;
    for (int j = 0; j < 64; ++j) {            // head `wave` of K
        const int n = wave * 64 + j;
        const float kv = bl_dot512(a.wk + (size_t)n * 512, xs, lane) + a.bk[n];
        sc = fmaf(a.q0[n], kv, sc);
    }
    if (lane == 0) a.sctab[(size_t)row * 8 + wave] = sc * 0.125f;
    for (int j = 0; j < 64; ++j) {
        const int n = wave * 64 + j;
        const float vv = bl_dot512(a.wv + (size_t)n * 512, xs, lane) + a.bv[n];
        if (lane == 0) a.vtab[(size_t)row * 512 + n] = vv;
    }
}

int launch_bound_tables(const BoundTablesArgs& a, hipStream_t s) {
    if (a.rows < 1 || !a.xt || !a.x0 || !a.n0g || !a.n0b || !a.wq || !a.bq || !a.wk || !a.bk || !a.wv || !a.bv || !a.q0 || !a.sctab || !a.vtab) return BOFI_ERR_ARG;
    hipLaunchKernelGGL(bound_q0_kernel, dim3(1), dim3(512), 0, s, a);
    hipLaunchKernelGGL(bound_tables_kernel, dim3(a.rows), dim3(512), 0, s, a);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi

extern "C" int bofi_bl_stamps(unsigned long long* host_out) {      // developer aid: the 32 stamps of the last BOFI_BL_DBG launch
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(bofi::g_bl_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? BOFI_OK : BOFI_ERR_HIP;
}
