// Small kernels of the NAIC bound+fill decode that are not GEMM/attention/LayerNorm:
//   * slot-state initialisation and export                (core_NAIC TransformerModel.py:1823-1838, 1876)
//   * row-0 self-attention of the bound layer over the precomputed (position, label) K/V table
//   * bound heads + greedy pick + slot bookkeeping          (TransformerModel.py:375-383, 1843-1869)
//   * fill-pass input embedding                              (decode_NA TransformerModel.py:570-577)
//   * vocabulary log-softmax + greedy pick + pad-after-length (AttModel.py:206-207, 421-423)
// All integer state stays on the device; no kernel here needs a host round trip.
#include "bofi_common.h"
#include "bofi_kernels.h"
#include "bofi_naic.h"

namespace bofi {

// ------------------------------------------------------------------------------------------------
__global__ void bound_init_kernel(BoundState st, int B, int L, int pad_idx, int len_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 8) st.counters[i] = 0;
    if (st.pair_ctl && i < ((B + 15) / 16) * 4) st.pair_ctl[i] = 0u;      // (the loop kernel's pair state and iteration counters: B * L >= B / 4 threads are there)
    if (i < B) { st.last[i] = 1; st.finished[i] = 0; st.phrase_num[i] = 0; }
    if (i < B * L) {
        st.phrase_length[i] = 0;
        st.phrase_syn[i] = pad_idx;
        st.ext_syn[i] = (i % L == 0) ? len_idx : pad_idx;     // position 0 is the [LEN] marker
        if (st.klen) st.klen[i] = 1;                          // tgt_mask[:, :, 0] = True (TransformerModel.py:1836)
    }
}

int launch_bound_init(const BoundState& st, int B, int L, int pad_idx, int len_idx, hipStream_t s) {
    const int n = max(B * L, 8);
    hipLaunchKernelGGL(bound_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, B, L, pad_idx, len_idx);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

__global__ void bound_export_kernel(BoundState st, int B, int L, int S, int* phrase_num, int* phrase_length,
                                    int64_t* phrase_syn, int* iters, int* live_max, int* sat_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && iters) *iters = st.counters[1];
    if (i == 0 && sat_out) *sat_out = st.counters[4];                 // fp16 saturation status of this decode's bounding loop (0 unless the loop kernel clamped something)
    if (i == 0 && live_max) atomicMax(live_max, st.counters[1]);      // the largest live-iteration count of every decode since the caller cleared the word
    if (i < B && phrase_num) phrase_num[i] = st.phrase_num[i];
    if (i < B * S) {
        const int b = i / S, t = i - b * S;                      // reference returns [:, :-2]
        if (phrase_length) phrase_length[i] = st.phrase_length[b * L + t];
        if (phrase_syn) phrase_syn[i] = st.phrase_syn[b * L + t];
    }
}

int launch_bound_export(const BoundState& st, int B, int L, int S, int* phrase_num, int* phrase_length,
                        int64_t* phrase_syn, int* iters, hipStream_t s, int* live_max, int* sat_out) {
    hipLaunchKernelGGL(bound_export_kernel, dim3((B * S + 255) / 256), dim3(256), 0, s, st, B, L, S, phrase_num,
                       phrase_length, phrase_syn, iters, live_max, sat_out);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-image tail of one bounding iteration, fused with the head of the next one:
//   (HEADS) final norm of row 0 -> two 2-layer heads -> log-softmax -> first-max argmax
//           (LengthPredictor_UIC.forward TransformerModel.py:375-383)
//   (UPDATE) slot bookkeeping of core_NAIC (TransformerModel.py:1843-1869)
//   (ATTN)  row-0 self-attention sublayer of the bound layer for the NEXT iteration, output projection and residual
//           included: y1 = x0 + Wo . attn(q0, K, V) + bo.
// The self-attention is exact for a one-layer bound network: the layer input at position p is
// lut_syn[label_p]*sqrt(d) + pe[p], a function of (p, label) only, so K and V of every possible row are tabulated once per
// model ("kvtab", [L*10, 2d]) and the query of row 0 ([LEN] at position 0) is a constant vector q0.  Row 0 sees keys
// p < last[b] (tgt_mask[j, 0, :last] = True, TransformerModel.py:1859/1867).  The output projection is linear in V, so it is
// tabulated as well: votab[row][h] = Wo[:, h-block] . V[row, h-block] (a d-vector per (row, head)), and
//   y1 = (x0 + bo) + sum_h sum_j p[h][j] * votab[row_j][h]
// -- one launch (and one chip-wide dependency) less per iteration than attention -> ctx -> GEMM.
// One workgroup (512 threads) per image.  The hidden layer of the two heads is 0.1 M weights per head pair: packed per
// thread ([slice][group][k][4 outputs], compute dtype) so that a thread's whole share is 32 16-byte loads, issued at kernel
// entry -- they are in flight while the row of y arrives and is normalised.
// developer aid (BOFI_TAIL_DBG & 16, with BOFI_DBG_PART): cycle stamps of the phases, relative to kernel entry, from thread 0 of every image
#define TAIL_STAMP(i)                                                                                         \
    do {                                                                                                      \
        if ((dbgm & 16) && a.dbg_part && tid == 0)                                                            \
            a.dbg_part[(size_t)blockIdx.x * (8 * 2 * a.hh + a.d) + (i)] = (float)(__builtin_amdgcn_s_memtime() - t_entry); \
    } while (0)

// NLD = hidden-layer weight loads a thread keeps in flight.  32 (one batch in bf16, all requested at kernel entry) holds 128 VGPRs of
// weights: one workgroup per CU, so launches of more than 256 images run in two rounds and nothing else shares the CU.  16 (two
// batches, the second requested after the first has been summed: one more L2 round trip) fits 128 VGPRs in all: two workgroups per
// CU.  Measured (tools/exp/ab_bench_c.sh): alone, the lean variant is 1 % slower per decode at 64 images, equal at 128 / 256 and
// 3 % faster at 320; with four launches in flight it is 1.5 % faster at 128 and 256 as well.  It runs above 64 images.
template <typename T, int NLD, int MINW>
__global__ __launch_bounds__(512, MINW) void bound_tail_kernel(BoundTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    const int flags = a.flags, B = a.B, L = a.L, S = a.S, d = a.d, hh = a.hh, H = a.H;
    const BoundState st = a.st;
    if ((flags & BOUND_EARLY) && ((flags & BOUND_SAIC) ? st.counters[2] >= 1 : st.counters[0] >= B)) return;
    const int nh = 2 * hh;
    float* xs = smem;                 // [d] normalised row
    float* part = xs + d;             // [8][nh] partial hidden sums
    float* hid = part + 8 * nh;       // [nh]
    float* lg = hid + nh;             // [32] logits: 0..19 length, 20..29 label
    float* red = lg + 32;             // [16]
    int* sint = reinterpret_cast<int*>(red + 16);  // [0] = last, [1] = finished, [2..2+L) = ext_syn row
    float* w2s = red + 16 + 64;       // (unused: the output layers are read straight into registers)
    float* ps = w2s + 30 * (hh + 1);  // [H][64] attention probabilities
    float* ysum = ps + H * 64;        // [512 / (d/8)][d] partial sums of the self-attention sublayer output per key subset (16 KB)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dbgm = flags >> 8;       // developer ablations (BOFI_TAIL_DBG): 1 = no w2 staging / logits, 2 = no hidden loop, 4 = no serial head, 16 = stamps
#ifdef BOFI_TAIL_POISON
    for (int i = tid; i < d + 8 * nh + nh + 32 + 16 + 64 + 30 * (hh + 1) + H * 64 + 4096; i += 512) smem[i] = __builtin_nanf("");
    __syncthreads();
#endif
    const T* __restrict__ w1p = static_cast<const T*>(a.w.w1p);
    const T* __restrict__ q0 = static_cast<const T*>(a.q0);
    const T* __restrict__ kvtab = static_cast<const T*>(a.kvtab);
    const T* __restrict__ votab = static_cast<const T*>(a.votab);

    // ---- hidden-layer weights of this thread: (slice of K, group of 4 outputs); requested before anything else
    constexpr int EPL = 16 / sizeof(T);           // elements per 16-byte load
    constexpr int KPL = EPL / 4;                  // k values per load (4 outputs each)
    const int ng = nh / 4, slice = tid / ng, grp = tid - slice * ng, kps = d / 8;
    const int nld = kps / KPL;                    // loads per thread in all
    const bool hw = (flags & BOUND_HEADS) && slice < 8;
    // piece u of thread (slice, grp) sits at [(slice * nld + u) * ng + grp]: the 16-byte pieces a wave-instruction loads are adjacent
    const T* wbase = w1p + ((size_t)slice * nld * ng + grp) * EPL;
    const size_t wstep = (size_t)ng * EPL;
    u32x4 wv[NLD];

    // the few words everything else waits for go first (vmcnt retires in order: behind the weight stream they would wait for all of it)
    const int* ext_src = a.ext_syn_in ? a.ext_syn_in : st.ext_syn;
    const int* last_src = a.last_in ? a.last_in : st.last;
    int ext_v = 0, last_v = 0, fin_v = 0, pn_v = 0;
    if (tid < L) ext_v = ext_src[b * L + tid];
    if (tid == 0) { last_v = last_src[b]; fin_v = (flags & BOUND_UPDATE) ? st.finished[b] : 0; pn_v = (flags & BOUND_UPDATE) ? st.phrase_num[b] : 0; }

    if (!(flags & BOUND_HEADS)) {
        if (tid < L) sint[2 + tid] = ext_v;
        if (tid == 0) { sint[0] = last_v; sint[1] = fin_v; }
    }
    if (flags & BOUND_HEADS) {
        const BoundHeadWeights& w = a.w;
        const float* __restrict__ y = a.y;
        // the row of y (possibly split-K partial slabs [yparts][B][d], summed in fixed order), the norm vectors, the hidden bias
        constexpr int KPT = 4;                        // columns per thread: d <= 512 * KPT
        float yv[KPT], gv[KPT], bvn[KPT];
#pragma unroll
        for (int c = 0; c < KPT; ++c) {
            const int k = tid + c * 512;
            yv[c] = 0.f; gv[c] = 0.f; bvn[c] = 0.f;
            if (k < d) {
                const size_t ys = a.y_stride ? (size_t)a.y_stride : (size_t)d;
                float acc = y[(size_t)b * ys + k];
                if (a.yparts <= 4) {                  // the slabs are requested together (a loop over a run-time count waits for each in turn)
                    float pv[3];
#pragma unroll
                    for (int pz = 1; pz < 4; ++pz) pv[pz - 1] = pz < a.yparts ? y[((size_t)pz * B + b) * ys + k] : 0.f;
#pragma unroll
                    for (int pz = 1; pz < 4; ++pz) if (pz < a.yparts) acc += pv[pz - 1];
                } else {
                    for (int pz = 1; pz < a.yparts; ++pz) acc += y[((size_t)pz * B + b) * ys + k];
                }
                yv[c] = acc; gv[c] = w.norm_gain[k]; bvn[c] = w.norm_bias[k];
            }
        }
        const float b1v = tid < nh ? w.b1[tid] : 0.f;
        // output layers of both heads: thread (o, q) of the first 480 sums every 16th term of output o (coalesced rows of 100 floats)
        constexpr int W2T = 7;                         // terms per thread: hh <= 16 * W2T
        const int w2o = tid >> 4, w2q = tid & 15;
        float w2v[W2T];
#pragma unroll
        for (int t = 0; t < W2T; ++t) {
            const int k = w2q + 16 * t;
            w2v[t] = (w2o < 30 && k < hh) ? (w2o < 20 ? w.len_w2[w2o * hh + k] : w.syn_w2[(w2o - 20) * hh + k]) : 0.f;
        }
        const float w2b = w2o < 20 ? w.len_b2[w2o] : (w2o < 30 ? w.syn_b2[w2o - 20] : 0.f);
        if (hw) {                                     // the hidden layer's weight stream, behind everything the first phases wait for
#pragma unroll
            for (int u = 0; u < NLD; ++u) wv[u] = (u < nld) ? *reinterpret_cast<const u32x4*>(wbase + (size_t)u * wstep) : u32x4{0u, 0u, 0u, 0u};
        }
        TAIL_STAMP(1);
        if (tid < L) sint[2 + tid] = ext_v;           // (first use is behind the barriers below: the wait for these words sits here, after every load has been issued)
        if (tid == 0) { sint[0] = last_v; sint[1] = fin_v; }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KPT; ++c) s += yv[c];
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const float mean = (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) / (float)d;
        TAIL_STAMP(2);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < KPT; ++c) if (tid + c * 512 < d) { const float t = yv[c] - mean; q += t * t; }
        q = wave_sum(q);
        if (lane == 0) red[8 + wave] = q;
        __syncthreads();
        const float den = sqrtf((((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]))) / (float)(d - 1)) + 1e-6f;
#pragma unroll
        for (int c = 0; c < KPT; ++c) if (tid + c * 512 < d) xs[tid + c * 512] = gv[c] * (yv[c] - mean) / den + bvn[c];
        __syncthreads();
        TAIL_STAMP(3);
        // hidden layer of both heads: thread (slice, group) sums 4 outputs over an eighth of K, k ascending
        if (hw && !(dbgm & 2)) {
            const int k0 = slice * kps;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c0 = 0; c0 < nld; c0 += NLD) {
                if (c0 > 0) {
#pragma unroll
                    for (int u = 0; u < NLD; ++u)
                        wv[u] = (c0 + u < nld) ? *reinterpret_cast<const u32x4*>(wbase + (size_t)(c0 + u) * wstep) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < NLD; ++u) {
                    if (c0 + u >= nld) break;
                    union { u32x4 v; T e[EPL]; } wu;
                    wu.v = wv[u];
#pragma unroll
                    for (int kk = 0; kk < KPL; ++kk) {
                        const float xv = xs[k0 + (c0 + u) * KPL + kk];
                        acc.x = fmaf(ElemOps<T>::to_f32(wu.e[kk * 4 + 0]), xv, acc.x); acc.y = fmaf(ElemOps<T>::to_f32(wu.e[kk * 4 + 1]), xv, acc.y);
                        acc.z = fmaf(ElemOps<T>::to_f32(wu.e[kk * 4 + 2]), xv, acc.z); acc.w = fmaf(ElemOps<T>::to_f32(wu.e[kk * 4 + 3]), xv, acc.w);
                    }
                }
            }
            *reinterpret_cast<float4*>(part + slice * nh + grp * 4) = acc;
        }
        TAIL_STAMP(4);
        __syncthreads();
        TAIL_STAMP(5);
        if (a.dbg_part && !(dbgm & 16)) {
            for (int i = tid; i < 8 * nh; i += 512) a.dbg_part[(size_t)b * (8 * nh + d) + i] = part[i];
            for (int i = tid; i < d; i += 512) a.dbg_part[(size_t)b * (8 * nh + d) + 8 * nh + i] = xs[i];
        }
        if (tid < nh)
            hid[tid] = fmaxf((((part[tid] + part[nh + tid]) + (part[2 * nh + tid] + part[3 * nh + tid])) +
                              ((part[4 * nh + tid] + part[5 * nh + tid]) + (part[6 * nh + tid] + part[7 * nh + tid]))) + b1v, 0.f);
        __syncthreads();
        {   // output layers: 16 lanes per output, DPP row reduction (all 512 threads take part; outputs >= 30 are padding)
            float acc = 0.f;
            if (!(dbgm & 1)) {
                const float* hv = w2o < 20 ? hid : (hid + hh);
#pragma unroll
                for (int t = 0; t < W2T; ++t) { const int k = w2q + 16 * t; acc = fmaf(w2v[t], k < hh ? hv[k] : 0.f, acc); }
            }
            acc = row16_sum(acc);
            if (w2q == 0 && w2o < 30) lg[w2o] = acc + w2b;
        }
        __syncthreads();
        TAIL_STAMP(6);
        // log-softmax and first-max pick of both heads in wavefront 0: lanes 0..19 the length head, lanes 32..41 the label head
        // (each head inside one 32-lane half: reductions by row16 + xor16 steps).  torch.max semantics: the first NaN wins, else
        // the first maximum (TransformerModel.py:380-383).
        if (wave == 0 && !(dbgm & 4)) {
            const int head = lane >> 5, li = lane & 31, nout = head ? 10 : 20;
            const bool on = li < nout;
            const float v = on ? lg[head * 20 + li] : -INFINITY;
            float m = xor16_max(row16_max(v));                                     // fmaxf ignores a NaN unless every input is one
            float e = on ? expf(v - m) : 0.f;
            const float sum = xor16_sum(row16_sum(e));
            const float lp = (v - m) - logf(sum);
            float* out = head ? (a.syn_logp ? a.syn_logp + (size_t)b * 10 : nullptr) : (a.len_logp ? a.len_logp + (size_t)b * 20 : nullptr);
            if (on && out) out[li] = lp;
            const bool isnan_ = on && (lp != lp);
            // index of the first NaN (if any), else of the first maximum: min over the half of a candidate index
            float cand = isnan_ ? (float)li : 1e9f;
            cand = -xor16_max(row16_max(-cand));
            float cmax = (on && v == m) ? (float)li : 1e9f;
            cmax = -xor16_max(row16_max(-cmax));
            const int best = cand < 1e8f ? (int)cand : (cmax < 1e8f ? (int)cmax : 0);
            if (li == 0) reinterpret_cast<int*>(lg)[30 + head] = best;        // (lg holds 30 logits; its last two words carry the picks)
        }
        __syncthreads();
        if (tid == 0 && !(dbgm & 4)) {
            const SaicState& sa = a.sa;
            const int iter = a.iter;
            const int pick[2] = {reinterpret_cast<const int*>(lg)[30], reinterpret_cast<const int*>(lg)[31]};
            if (flags & BOUND_SAIC) {
                // core_SAIC bookkeeping of iteration `iter` (TransformerModel.py:1910-1948)
                if (b == 0) st.counters[1] += 1;
                const int pl = sint[0];
                if (!sint[1]) {
                    int ln = pick[0];
                    const int sn = pick[1];
                    bool fin = false;
                    if (ln == 0 || sn < 4 || sn > 6) {
                        fin = true;
                    } else {
                        if (ln + pl >= S + 1) { ln = S + 1 - pl; fin = true; }
                        st.phrase_length[b * L + iter] = ln;
                        st.phrase_syn[b * L + iter] = sn;
                        st.phrase_num[b] += 1;
                    }
                    if (fin) { st.finished[b] = 1; atomicAdd(&st.counters[0], 1); }
                }
                const int cur = st.phrase_length[b * L + iter];
                if (cur != 0) {
                    const int prev = st.phrase_length[b * L + iter - 1], sl = sa.seq_last[b], sy = st.phrase_syn[b * L + iter];
                    for (int p = pl; p < pl + cur; ++p) st.ext_syn[b * L + p] = sy;
                    if (cur <= prev) {                        // :1934-1936
                        const int pre_pad = prev - cur;
                        for (int k = 0; k < cur; ++k) sa.ext_phrase[b * L + pl + k] = sa.seq[b * L + sl + pre_pad + k];
                    } else {                                  // :1937-1947 position-wise stretch
                        const int pre_less = prev - (cur % prev), times = cur / prev;
                        int copied = 0;
                        for (int k = 0; k < prev; ++k) {
                            const int nrep = k < pre_less ? times : times + 1, tokv = sa.seq[b * L + sl + k];
                            for (int c = 0; c < nrep; ++c) sa.ext_phrase[b * L + pl + copied + c] = tokv;
                            copied += nrep;
                        }
                    }
                    for (int r = pl; r < L; ++r) sa.klen_dec[b * L + r] = pl + cur;      // phrase_mask[j, pl:, :pl+cur] = True
                }
            } else if (flags & BOUND_UPDATE) {
                if (b == 0) st.counters[1] += 1;              // iterations in which some image was active
                if (!sint[1]) {
                    int ln = pick[0];
                    const int sn = pick[1], la = sint[0];
                    bool fin = false;
                    if (ln == 0 || sn < 4 || sn > 6) {        // EOS (TransformerModel.py:1846-1849)
                        fin = true;
                    } else {
                        if (ln + la >= S + 1) { ln = S + 1 - la; fin = true; }      // truncate (:1850-1855)
                        const int slot = pn_v;                // == iteration index while unfinished (Q3); read at kernel entry
                        st.phrase_length[b * L + slot] = ln;
                        st.phrase_syn[b * L + slot] = sn;
                        st.phrase_num[b] = slot + 1;
                        for (int p = la; p < la + ln; ++p) { st.ext_syn[b * L + p] = sn; sint[2 + p] = sn; }
                        st.last[b] = la + ln;
                        sint[0] = la + ln;
                        if (st.klen) {                        // tgt_mask[j, la:, :la+ln] = True; tgt_mask[j, 0, :la+ln] = True (:1859-1867)
                            for (int r = la; r < L; ++r) st.klen[b * L + r] = la + ln;
                            st.klen[b * L] = la + ln;
                        }
                    }
                    if (fin) { st.finished[b] = 1; sint[1] = 1; atomicAdd(&st.counters[0], 1); }
                }
            }
        }
    }
    __syncthreads();
    TAIL_STAMP(7);
    if (!(flags & BOUND_ATTN) || sint[1]) return;           // a finished image's state is frozen: no further steps matter

    // ---- row-0 self-attention over the (position, label) table: scores and softmax, one wavefront per head
    const int n = min(sint[0], L);
    for (int h = wave; h < H; h += 8) {
        float sc = -INFINITY;
        if (lane < n) {
            const int row = lane * 10 + sint[2 + lane];
            const T* kr = kvtab + (size_t)row * 2 * d + h * 64;
            const T* qr = q0 + h * 64;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 64 / EPL; ++c) {
                union { u32x4 v; T e[EPL]; } ku, qu;
                ku.v = *reinterpret_cast<const u32x4*>(kr + c * EPL);
                qu.v = *reinterpret_cast<const u32x4*>(qr + c * EPL);
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc = fmaf(ElemOps<T>::to_f32(qu.e[e]), ElemOps<T>::to_f32(ku.e[e]), acc);
            }
            sc = acc * 0.125f;
        }
        const float m = wave_max(sc);
        const float e = (lane < n) ? expf(sc - m) : 0.f;
        const float sum = wave_sum(e);
        ps[h * 64 + lane] = ElemOps<T>::to_f32(ElemOps<T>::from_f32(e / sum));     // P is held in compute dtype, as in attn.hip
    }
    __syncthreads();
    TAIL_STAMP(8);
    // ---- y1 = (x0 + bo) + sum_j sum_h P[h][j] * votab[row_j][h].  All table rows are requested in ONE round trip: a thread owns
    // 8 columns (one 16-byte piece of a row in bf16) and every (512 / (d/8))-th key; its <= 3 keys x H heads are independent loads.
    // The key subsets are then summed through LDS in fixed order.
    {
        constexpr int CPT = 8, LPT = CPT / EPL;            // columns per thread, 16-byte loads per row piece
        constexpr int JB = NLD >= 32 ? 3 : 2;              // keys per thread and batch (8 x JB loads in flight)
        const int ncg = d / CPT, njs = 512 / ncg;          // column groups; key subsets (d = 512: 64 and 8)
        const int cg = tid % ncg, js = tid / ncg;
        float acc[CPT];
#pragma unroll
        for (int e = 0; e < CPT; ++e) acc[e] = 0.f;
        if (js < njs) {
            for (int jb = js; jb < n; jb += njs * JB) {
                for (int h0 = 0; h0 < H; h0 += 8) {
                    u32x4 vv[JB][8][LPT];
#pragma unroll
                    for (int t = 0; t < JB; ++t) {
                        const int j = jb + t * njs;
                        const int row = (j < n) ? j * 10 + sint[2 + j] : 0;
                        const T* vr = votab + ((size_t)row * H) * d + cg * CPT;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int q = 0; q < LPT; ++q)
                                vv[t][u][q] = (j < n && h0 + u < H) ? *reinterpret_cast<const u32x4*>(vr + (size_t)(h0 + u) * d + q * EPL) : u32x4{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int t = 0; t < JB; ++t) {
                        const int j = jb + t * njs;
                        if (j >= n) break;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (h0 + u >= H) break;
                            const float pj = ps[(h0 + u) * 64 + j];
#pragma unroll
                            for (int q = 0; q < LPT; ++q) {
                                union { u32x4 v; T e[EPL]; } vu;
                                vu.v = vv[t][u][q];
#pragma unroll
                                for (int e = 0; e < EPL; ++e) acc[q * EPL + e] = fmaf(pj, ElemOps<T>::to_f32(vu.e[e]), acc[q * EPL + e]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < CPT; ++e) ysum[js * d + cg * CPT + e] = acc[e];
        }
    }
    __syncthreads();
    TAIL_STAMP(9);
    for (int c0 = 0; c0 < d; c0 += 512) {
        const int c = c0 + tid;
        if (c0 + (wave << 6) >= d) break;                  // whole wavefronts drop out (d % 64 == 0): the DPP reductions below need full ones
        float acc = a.x0b[c];
        const int njs = 512 / (d / 8);
        for (int q = 0; q < njs; ++q) acc += ysum[q * d + c];
        a.y1[(size_t)b * d + c] = acc;
        if (a.y1t) ElemOps<T>::store(static_cast<T*>(a.y1t) + (size_t)b * d + c, acc);
        float s1 = acc, s2 = acc * acc;                    // partial (sum, sumsq) per 32 columns, as the GEMM epilogues write them
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        s1 = xor16_sum(s1); s2 = xor16_sum(s2);
        if ((tid & 31) == 0) reinterpret_cast<float2*>(a.stats)[(size_t)b * (d >> 5) + (c >> 5)] = make_float2(s1, s2);
    }
    TAIL_STAMP(10);
}

int launch_bound_tail(const BoundTailArgs& a, int dtype, hipStream_t s) {
    const int d = a.d, hh = a.hh, L = a.L;
    if (d > 2048 || (2 * hh) % 4 || d % 64 || (2 * hh / 4) * 8 > 512 || 2 * hh > 512 || L > 60 || a.H > 64 || a.H * 64 != d) return BOFI_ERR_ARG;
    if ((d / 8) % (dtype == BOFI_DT_F32 ? 1 : 2)) return BOFI_ERR_ARG;
    if ((a.flags & BOUND_ATTN) && (!a.votab || !a.x0b || !a.y1 || !a.stats || !a.q0 || !a.kvtab)) return BOFI_ERR_ARG;
    if ((a.flags & BOUND_HEADS) && (!a.y || !a.w.w1p)) return BOFI_ERR_ARG;
    BoundTailArgs v = a;
    if (v.yparts < 1) v.yparts = 1;
    if (d % 8 || 512 % (d / 8) || d / 8 > 256) return BOFI_ERR_ARG;
    const size_t shm = (size_t)(d + 8 * 2 * hh + 2 * hh + 32 + 16 + 64 + 30 * (hh + 1) + a.H * 64 + 4096) * sizeof(float);
    const int small_at = BOFI_ENV_INT("BOFI_TAIL_SMALL_AT", 65);   // developer knob: images from which the two-per-CU variant runs
    const bool small = a.B >= small_at;
    if (dtype == BOFI_DT_F32) {                               // (float32 rows are twice as wide: no variant of it fits 128 VGPRs)
        hipLaunchKernelGGL((bound_tail_kernel<float, 32, 1>), dim3(a.B), dim3(512), shm, s, v);
    } else {
        if (small) hipLaunchKernelGGL((bound_tail_kernel<bf16_t, 16, 4>), dim3(a.B), dim3(512), shm, s, v);
        else hipLaunchKernelGGL((bound_tail_kernel<bf16_t, 32, 1>), dim3(a.B), dim3(512), shm, s, v);
    }
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Fill-pass input: pos_embed(tgt_embed(word) + syn_embed(label)), each Embeddings scaled by sqrt(d)
// (TransformerModel.py:576-577, 1486-1487, 1505-1507).  word = BOS everywhere unless tok != NULL.
template <typename T>
__global__ __launch_bounds__(128) void embed_fill_kernel(const float* __restrict__ lut_tok, const float* __restrict__ lut_syn,
                                                         const float* __restrict__ pe, const int* __restrict__ ext_syn,
                                                         const int64_t* tok, int B, int S, int L, int d, int bos_idx, float sqrt_d,
                                                         float* __restrict__ x, T* __restrict__ xt, float* __restrict__ stats,
                                                         const T* __restrict__ qkv_tab, T* __restrict__ qkv_out, int nq) {
    const int row = blockIdx.x;                       // (b, t)
    const int b = row / S, t = row - b * S;
    const int syn = ext_syn[b * L + t + 1];           // extend_phrase_syn[:, 1:-1]
    const int64_t word = tok ? tok[row] : bos_idx;
    const float* tr = lut_tok + (size_t)word * d;
    const float* sr = lut_syn + (size_t)syn * d;
    const float* pr = pe + (size_t)t * d;
    for (int k = threadIdx.x; k < d; k += 128) {      // d % 128 == 0: every 32-lane half handles one 32-column group
        const float v = (tr[k] * sqrt_d + sr[k] * sqrt_d) + pr[k];
        x[(size_t)row * d + k] = v;
        if (xt) ElemOps<T>::store(xt + (size_t)row * d + k, v);
        if (stats) {                                  // partial (sum, sumsq) per 32 columns, as the GEMM epilogues write them
            float ps = v, pq = v * v;
            ps = row16_sum(ps); pq = row16_sum(pq);
            ps = xor16_sum(ps); pq = xor16_sum(pq);
            if ((threadIdx.x & 31) == 0) reinterpret_cast<float2*>(stats)[(size_t)row * (d >> 5) + (k >> 5)] = make_float2(ps, pq);
        }
    }
    if (qkv_tab) {                                    // layer 0's q|k|v row of (label, position): word = BOS everywhere in this call
        const u32x4* src = reinterpret_cast<const u32x4*>(qkv_tab + ((size_t)syn * S + t) * nq);
        u32x4* dst = reinterpret_cast<u32x4*>(qkv_out + (size_t)row * nq);
        for (int k = threadIdx.x; k < nq * (int)sizeof(T) / 16; k += 128) dst[k] = src[k];
    }
}

int launch_embed_fill(const float* lut_tok, const float* lut_syn, const float* pe, const int* ext_syn, const int64_t* tok,
                      int B, int S, int L, int d, int bos_idx, float* x, void* xt, int dtype, float* stats, hipStream_t s, const void* qkv_tab, void* qkv_out, int nq) {
    if (d % 128 || (qkv_tab && (tok || !qkv_out || nq % 8))) return BOFI_ERR_ARG;
    const float sqrt_d = (float)sqrt((double)d);
    if (dtype == BOFI_DT_F32)
        hipLaunchKernelGGL((embed_fill_kernel<float>), dim3(B * S), dim3(128), 0, s, lut_tok, lut_syn, pe, ext_syn, tok, B, S, L, d,
                           bos_idx, sqrt_d, x, (float*)xt, stats, (const float*)qkv_tab, (float*)qkv_out, nq);
    else
        hipLaunchKernelGGL((embed_fill_kernel<bf16_t>), dim3(B * S), dim3(128), 0, s, lut_tok, lut_syn, pe, ext_syn, tok, B, S, L, d,
                           bos_idx, sqrt_d, x, (bf16_t*)xt, stats, (const bf16_t*)qkv_tab, (bf16_t*)qkv_out, nq);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Generic embedding rows for the SAIC path: pos_embed(tgt_embed(tok) [+ syn_embed(syn)]) with int32 ids read from
// the slot state (bound input TransformerModel.py:515-518, decoder input :520-530).
template <typename T>
__global__ __launch_bounds__(128) void embed_rows_kernel(const float* __restrict__ lut_tok, const float* __restrict__ lut_syn,
                                                         const float* __restrict__ pe, const int* tok, const int* syn, int ld, int off,
                                                         int Tn, int d, int bos_idx, float sqrt_d, float* __restrict__ x,
                                                         T* __restrict__ xt, float* __restrict__ stats, const int* halt) {
    if (halt && *halt >= 1) return;
    const int row = blockIdx.x, b = row / Tn, t = row - b * Tn;
    const int word = tok ? tok[b * ld + off + t] : bos_idx;
    const float* tr = lut_tok + (size_t)word * d;
    const float* sr = syn ? lut_syn + (size_t)syn[b * ld + off + t] * d : nullptr;
    const float* pr = pe + (size_t)t * d;
    for (int k = threadIdx.x; k < d; k += 128) {
        const float v = sr ? (tr[k] * sqrt_d + sr[k] * sqrt_d) + pr[k] : tr[k] * sqrt_d + pr[k];
        x[(size_t)row * d + k] = v;
        if (xt) ElemOps<T>::store(xt + (size_t)row * d + k, v);
        if (stats) {
            float ps = v, pq = v * v;
            ps = row16_sum(ps); pq = row16_sum(pq);
            ps = xor16_sum(ps); pq = xor16_sum(pq);
            if ((threadIdx.x & 31) == 0) reinterpret_cast<float2*>(stats)[(size_t)row * (d >> 5) + (k >> 5)] = make_float2(ps, pq);
        }
    }
}

int launch_embed_rows(const float* lut_tok, const float* lut_syn, const float* pe, const int* tok, const int* syn, int ld, int off,
                      int B, int Tn, int d, int bos_idx, float* x, void* xt, int dtype, float* stats, const int* halt, hipStream_t s) {
    if (d % 128) return BOFI_ERR_ARG;
    const float sqrt_d = (float)sqrt((double)d);
    if (dtype == BOFI_DT_F32)
        hipLaunchKernelGGL((embed_rows_kernel<float>), dim3(B * Tn), dim3(128), 0, s, lut_tok, lut_syn, pe, tok, syn, ld, off, Tn, d, bos_idx,
                           sqrt_d, x, (float*)xt, stats, halt);
    else
        hipLaunchKernelGGL((embed_rows_kernel<bf16_t>), dim3(B * Tn), dim3(128), 0, s, lut_tok, lut_syn, pe, tok, syn, ld, off, Tn, d, bos_idx,
                           sqrt_d, x, (bf16_t*)xt, stats, halt);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// The decoder rows an iteration of the semi-autoregressive loop has to compute: the positions of the phrase each unfinished
// image placed in this iteration (TransformerModel.py:1933-1948).  Every earlier row's input and key set are final once its phrase
// is placed, so its K / V in every decoder layer are too; later rows are never attended.  One wavefront; rows image-major.
__global__ __launch_bounds__(64) void saic_rows_kernel(BoundState st, int B, int L, int S, int iter, int* rows, int* n_rows) {
    if (st.counters[2] >= 1) return;
    int n = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + threadIdx.x;
        const int cur = b < B ? st.phrase_length[b * L + iter] : 0, pl = b < B ? st.last[b] : 0;
        int incl = cur;                                         // inclusive prefix sum over the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((int)threadIdx.x >= o) incl += t; }
        const int base = n + incl - cur;
        for (int k = 0; k < cur; ++k) rows[base + k] = b * S + pl - 1 + k;
        n += __shfl(incl, 63, 64);
    }
    if (threadIdx.x == 0) *n_rows = n;
}

int launch_saic_rows(const BoundState& st, int B, int L, int S, int iter, int* rows, int* n_rows, hipStream_t s) {
    hipLaunchKernelGGL(saic_rows_kernel, dim3(1), dim3(64), 0, s, st, B, L, S, iter, rows, n_rows);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// zero fill of a float buffer as a kernel of this library: inside a captured launch sequence a hipMemsetAsync becomes a memset
// NODE, and replays of such a node were seen to fill with a stale pattern (pointer-like values) after host-side allocations --
// a plain kernel node carries its arguments by value
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}
int launch_zero_f32(float* p, size_t n, hipStream_t s) {
    if (!n) return BOFI_OK;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, p, n);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

__global__ void set_u64_kernel(uint64_t* p, uint64_t v) { *p = v; }
int launch_set_u64(uint64_t* p, uint64_t v, hipStream_t s) {
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, s, p, v);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

__global__ void saic_init_kernel(BoundState st, SaicState sa, int B, int L, int pad_idx, int bos_idx, int len_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 8) st.counters[i] = 0;
    if (i < B) { st.last[i] = 1; st.finished[i] = 0; st.phrase_num[i] = 0; sa.seq_last[i] = 0; }
    if (i < B * L) {
        const bool first = (i % L) == 0;
        st.phrase_length[i] = first ? 1 : 0;                   // phrase_length[:, 0] = 1 (TransformerModel.py:1902)
        st.phrase_syn[i] = pad_idx;
        st.ext_syn[i] = pad_idx;
        sa.seq[i] = first ? bos_idx : pad_idx;
        sa.ext_len[i] = first ? len_idx : pad_idx;
        sa.ext_phrase[i] = pad_idx;
        sa.klen_dec[i] = 0;
    }
}

int launch_saic_init(const BoundState& st, const SaicState& sa, int B, int L, int pad_idx, int bos_idx, int len_idx, hipStream_t s) {
    const int n = max(B * L, 8);
    hipLaunchKernelGGL(saic_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, sa, B, L, pad_idx, bos_idx, len_idx);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// tok/logp: ids and log-probs of this iteration's decoder pass (all S positions, or the new phrases' rows).  Grid (S + 1, B): block
// (k < S, b) copies the log-prob row of the k-th token of image b's new phrase, block (S, b) its token ids.  No block writes what
// another reads (st.last, st.phrase_length[iter]); the positions advance in saic_halt_kernel, the next launch.
__global__ __launch_bounds__(256) void saic_copy_kernel(BoundState st, SaicState sa, const int64_t* __restrict__ tok,
                                                        const float* __restrict__ logp, float* __restrict__ seq_logprob, int B,
                                                        int L, int S, int V, int iter) {
    if (st.counters[2] >= 1) return;
    const int b = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
    const bool nan_seen = st.counters[3] != 0;                  // "phrase nan!": return before the copy (TransformerModel.py:1956-1958)
    const int cur = st.phrase_length[b * L + iter], pl = st.last[b];
    if (nan_seen || cur == 0) return;
    if (k < S) {
        if (k >= cur || !seq_logprob) return;
        const float* src = logp + ((size_t)b * S + pl - 1 + k) * V;
        float* dst = seq_logprob + ((size_t)b * S + pl - 1 + k) * V;          // seq_logprobs[j, pl+k] -> returned slice [:, 1:-1]
        for (int i = tid; i < V; i += 256) dst[i] = src[i];
        return;
    }
    for (int kk = tid; kk < cur; kk += 256) {
        const int t = (int)tok[(size_t)b * S + pl - 1 + kk];
        sa.seq[b * L + pl + kk] = t;
        sa.ext_len[b * L + pl + kk] = t;
    }
}
// after every block of saic_copy_kernel has read the old state: advance the images' positions, then the loop exit conditions
__global__ void saic_halt_kernel(BoundState st, SaicState sa, int B, int L, int iter) {
    if (st.counters[2] >= 1) return;
    const bool nan_seen = st.counters[3] != 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int cur = st.phrase_length[b * L + iter];
        if (!nan_seen && cur != 0) {
            st.last[b] += cur;
            sa.seq_last[b] += st.phrase_length[b * L + iter - 1];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && (nan_seen || st.counters[0] >= B)) st.counters[2] = 1;
}
// the second half of launch_saic_copy alone: an iteration whose words come from the caller (BOFI_FLAG_SAIC_LAYOUT_ONLY: no decoder pass ran)
int launch_saic_halt(const BoundState& st, const SaicState& sa, int B, int L, int iter, hipStream_t s) {
    hipLaunchKernelGGL(saic_halt_kernel, dim3(1), dim3(256), 0, s, st, sa, B, L, iter);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}
int launch_saic_copy(const BoundState& st, const SaicState& sa, const int64_t* tok, const float* logp, float* seq_logprob, int B, int L,
                     int S, int V, int iter, hipStream_t s) {
    hipLaunchKernelGGL(saic_copy_kernel, dim3(S + 1, B), dim3(256), 0, s, st, sa, tok, logp, seq_logprob, B, L, S, V, iter);
    hipLaunchKernelGGL(saic_halt_kernel, dim3(1), dim3(256), 0, s, st, sa, B, L, iter);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

__global__ void saic_export_kernel(BoundState st, SaicState sa, int B, int L, int S, int64_t* seq, int* phrase_num, int* phrase_length,
                                   int64_t* phrase_syn, int* iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && iters) *iters = st.counters[1];
    if (i < B && phrase_num) phrase_num[i] = st.phrase_num[i];
    if (i < B * S) {
        const int b = i / S, t = i - b * S;                      // reference returns [:, 1:-1]
        seq[i] = sa.seq[b * L + t + 1];
        if (phrase_length) phrase_length[i] = st.phrase_length[b * L + t + 1];
        if (phrase_syn) phrase_syn[i] = st.phrase_syn[b * L + t + 1];
    }
}

// The tokens emitted so far replaced by the caller's (seq int64 [B, S], the layout of the exported seq): positions 1 .. last - 1 of sa.seq and of
// the bounding step's input sa.ext_len.  Between two calls that each enqueue part of the loop (bofi_engine_set_saic_range): a caller that draws
// the phrase's words itself -- from its own distribution over the same layout -- and lets the loop continue on THOSE words.
__global__ void saic_put_words_kernel(BoundState st, SaicState sa, const int64_t* __restrict__ seq, int B, int L, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S, p = i - b * S + 1;
    if (p < st.last[b]) { const int t = (int)seq[i]; sa.seq[b * L + p] = t; sa.ext_len[b * L + p] = t; }
}
int launch_saic_put_words(const BoundState& st, const SaicState& sa, const int64_t* seq, int B, int L, int S, hipStream_t s) {
    hipLaunchKernelGGL(saic_put_words_kernel, dim3((B * S + 255) / 256), dim3(256), 0, s, st, sa, seq, B, L, S);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

int launch_saic_export(const BoundState& st, const SaicState& sa, int B, int L, int S, int64_t* seq, int* phrase_num,
                       int* phrase_length, int64_t* phrase_syn, int* iters, hipStream_t s) {
    hipLaunchKernelGGL(saic_export_kernel, dim3((B * S + 255) / 256), dim3(256), 0, s, st, sa, B, L, S, seq, phrase_num, phrase_length,
                       phrase_syn, iters);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Vocabulary epilogue, one workgroup per (image, position) row of V logits:
//   log_softmax (in place), greedy argmax with torch.max's CPU semantics (lowest index among
//   equal maxima; NaN beats everything and the first NaN is returned), pad after the image's
//   token count.  HBM-bound: V*4 bytes read + V*4 written per row.
template <int NPT>   // values per thread held in registers: V <= 512 * NPT (one HBM read + one write per value)
__global__ __launch_bounds__(512) void vocab_finalize_kernel(float* __restrict__ logits, int V, int S, int log_softmax,
                                                             const int* ntok, int ntok_bias, int pad_idx, int64_t* seq,
                                                             int* nan_flag, const int* halt, const int* row_idx, const int* n_rows,
                                                             const float* __restrict__ src, int ld_src, float* __restrict__ row_plogp,
                                                             float* __restrict__ row_chosen) {
    __shared__ float red[16];
    __shared__ int redi[16];
    __shared__ float s_tpad;                                    // the log-prob at pad_idx (the emitted id of a row past the image's token count)
    if (halt && *halt >= 1) return;
    if (n_rows && (int)blockIdx.x >= *n_rows) return;           // row list: rows row_idx[0 .. *n_rows) only
    const int row = row_idx ? row_idx[blockIdx.x] : blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* x = logits + (size_t)row * V;
    const float* xin = src ? src + (size_t)row * ld_src : x;
    float v[NPT];
    float m = -INFINITY;
    int first_nan = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int idx = tid + i * 512;
        v[i] = idx < V ? xin[idx] : -INFINITY;
        if (v[i] != v[i]) first_nan = min(first_nan, idx);
        m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first_nan = min(first_nan, __shfl_xor(first_nan, o, 64));
    if (lane == 0) { red[wave] = m; redi[wave] = first_nan; }
    __syncthreads();
    m = red[0]; first_nan = redi[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) { m = fmaxf(m, red[w]); first_nan = min(first_nan, redi[w]); }
    __syncthreads();
    float lse = 0.f, plogp = 0.f;
    if (log_softmax) {
        float s = 0.f, u = 0.f;                                 // u: sum of e_i * (x_i - max) -- with it sum_i p_i log p_i = u / s - lse, no second exponential, no second pass
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const float e = (tid + i * 512 < V) ? expf(v[i] - m) : 0.f;
            s += e;
            if (row_plogp && tid + i * 512 < V) u += e * (v[i] - m);
        }
        s = wave_sum(s);
        if (row_plogp) u = wave_sum(u);
        if (lane == 0) { red[wave] = s; red[8 + wave] = u; }
        __syncthreads();
        const float sum = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
        lse = logf(sum);
        plogp = (((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]))) / sum - lse;
        __syncthreads();
    }
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int idx = tid + i * 512;
        if (idx >= V) continue;
        float t = v[i];
        if (log_softmax) {
            t = (first_nan != 0x7fffffff) ? __builtin_nanf("") : (t - m) - lse;    // one NaN poisons the row's softmax
            if (log_softmax != 2) x[idx] = t;           // 2: the ids only (a refinement round whose log-probs the next round overwrites): the same
                                                        // comparison values, no store
        } else if (src) {
            x[idx] = t;                                 // raw logits asked for: the copy to the caller's pitch
        }
        if (row_chosen && idx == pad_idx) s_tpad = t;
        if (t > bv) { bv = t; bi = idx; }             // idx ascends per thread: first max kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 8; ++w)
            if (red[w] > bv || (red[w] == bv && redi[w] < bi)) { bv = red[w]; bi = redi[w]; }
        if (first_nan != 0x7fffffff) { bi = log_softmax ? 0 : first_nan; bv = __builtin_nanf(""); if (nan_flag) atomicOr(nan_flag, 1); }
        if (bi == 0x7fffffff) bi = 0;                 // all -inf row: torch.max returns index 0
        bool padded = false;
        if (ntok) {
            const int b = row / S, t = row - b * S;
            if (t >= ntok[b] + ntok_bias) { bi = pad_idx; padded = true; }
        }
        seq[row] = bi;
        if (row_plogp && log_softmax) {                 // what bofi_vocab_stats computes from the finished tensor: sum_v p log p and the log-prob of the emitted id
            row_plogp[row] = first_nan != 0x7fffffff ? __builtin_nanf("") : plogp;
            row_chosen[row] = padded ? s_tpad : bv;
        }
    }
}

int launch_vocab_finalize(float* logits, int rows, int V, int S, int log_softmax, const int* ntok, int ntok_bias, int pad_idx,
                          int64_t* seq, hipStream_t st, int* nan_flag, const int* halt, const int* row_idx, const int* n_rows, const float* src,
                          int ld_src, float* row_plogp, float* row_chosen) {
    if (!logits || !seq || rows < 0 || V <= 0 || S <= 0 || (src && ld_src < V) || (!row_plogp != !row_chosen) || (row_plogp && (pad_idx < 0 || pad_idx >= V))) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    if (V <= 512 * 4) hipLaunchKernelGGL((vocab_finalize_kernel<4>), dim3(rows), dim3(512), 0, st, logits, V, S, log_softmax, ntok, ntok_bias, pad_idx, seq, nan_flag, halt, row_idx, n_rows, src, ld_src, row_plogp, row_chosen);
    else if (V <= 512 * 20) hipLaunchKernelGGL((vocab_finalize_kernel<20>), dim3(rows), dim3(512), 0, st, logits, V, S, log_softmax, ntok, ntok_bias, pad_idx, seq, nan_flag, halt, row_idx, n_rows, src, ld_src, row_plogp, row_chosen);
    else if (V <= 512 * 64) hipLaunchKernelGGL((vocab_finalize_kernel<64>), dim3(rows), dim3(512), 0, st, logits, V, S, log_softmax, ntok, ntok_bias, pad_idx, seq, nan_flag, halt, row_idx, n_rows, src, ld_src, row_plogp, row_chosen);
    else return BOFI_ERR_ARG;
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Consumers of the finished [rows, V] log-prob tensor that do not need it in user memory.
//
// vocab_stats: per row, sum_v p_v * logp_v and the log-prob of the emitted id -- the two reductions eval needs for its
// per-image entropy / perplexity (captioning/utils/eval_utils.py:463-464), so a caller can skip materialising the
// 48.6 MB tensor.
__global__ __launch_bounds__(256) void vocab_stats_kernel(const float* __restrict__ logp, const int64_t* __restrict__ seq, int V,
                                                          float* __restrict__ row_plogp, float* __restrict__ row_chosen) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = logp + (size_t)row * V;
    float s = 0.f;
    for (int i = tid; i < V; i += 256) { const float t = x[i]; s += expf(t) * t; }       // softmax(logp) == exp(logp)
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        row_plogp[row] = (red[0] + red[1]) + (red[2] + red[3]);
        row_chosen[row] = x[seq[row]];
    }
}

// vocab_sample: `n` independent draws per row from Categorical(logits = logp / temperature) (CaptionModel.py:419-425; a
// NaN log-prob counts as -10 as there) by the Gumbel-max rule with a counter-hash uniform; draw c of image b goes to
// row b * n + c of the output (models/utils.py:3-14 repeats each image n times), ids past the image's token count are pad.
__global__ __launch_bounds__(256) void vocab_sample_kernel(const float* __restrict__ logp, int V, int S, int n, float inv_temp, uint64_t seed,
                                                           const int* __restrict__ ntok, int pad_idx, int64_t* __restrict__ out,
                                                           const int* halt, const int* row_idx, const int* n_rows,
                                                           const uint64_t* seed_dev) {
    __shared__ float redv[4];
    __shared__ int redi[4];
    if (halt && *halt >= 1) return;
    if (n_rows && (int)blockIdx.x >= *n_rows) return;
    if (seed_dev) seed += *seed_dev;                           // per-call part of the seed (a captured launch keeps the per-iteration part)
    const int row = row_idx ? row_idx[blockIdx.x] : blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = row / S, t = row - b * S;
    const float* x = logp + (size_t)row * V;
    for (int c = 0; c < n; ++c) {
        float bv = -INFINITY;
        int bi = 0;
        const uint64_t base = ((uint64_t)row * n + c) * (uint64_t)V;
        for (int i = tid; i < V; i += 256) {
            float lp = x[i];
            if (lp != lp) lp = -10.f;
            const float u = ((float)hash64_hi(seed, base + i) + 0.5f) * (1.0f / 4294967296.0f);
            const float gv = lp * inv_temp - logf(-logf(u));
            if (gv > bv) { bv = gv; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { redv[wave] = bv; redi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (redv[w] > bv || (redv[w] == bv && redi[w] < bi)) { bv = redv[w]; bi = redi[w]; }
            if (ntok && t >= ntok[b]) bi = pad_idx;
            out[((size_t)b * n + c) * S + t] = bi;
        }
        __syncthreads();
    }
}

int launch_vocab_sample(const float* logp, int rows, int V, int S, int n, float temperature, uint64_t seed, const int* ntok, int pad_idx,
                        int64_t* out, hipStream_t st, const int* halt, const int* row_idx, const int* n_rows, const uint64_t* seed_dev) {
    if (!logp || !out || rows < 0 || V <= 0 || S <= 0 || rows % S || n <= 0 || !(temperature > 0.f)) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    hipLaunchKernelGGL(vocab_sample_kernel, dim3(rows), dim3(256), 0, st, logp, V, S, n, 1.0f / temperature, seed, ntok, pad_idx, out, halt, row_idx, n_rows, seed_dev);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi

extern "C" int bofi_vocab_stats(const float* logp, const int64_t* seq, int rows, int V, float* row_plogp, float* row_chosen, void* stream) {
    if (!logp || !seq || !row_plogp || !row_chosen || rows < 0 || V <= 0) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    hipLaunchKernelGGL(bofi::vocab_stats_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logp, seq, V, row_plogp, row_chosen);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_vocab_sample(const float* logp, int rows, int V, int S, int n, float temperature, uint64_t seed, const int* ntok,
                                 int pad_idx, int64_t* out, void* stream) {
    return bofi::launch_vocab_sample(logp, rows, V, S, n, temperature, seed, ntok, pad_idx, out, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int bofi_vocab_finalize(float* logits, int rows, int V, int S, int log_softmax, const int* ntok, int pad_idx,
                                   int64_t* seq, void* stream) {
    return bofi::launch_vocab_finalize(logits, rows, V, S, log_softmax, ntok, 0, pad_idx, seq, (hipStream_t)stream);
}
