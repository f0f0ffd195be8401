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
    if (i < 4) st.counters[i] = 0;
    if (i < B) { st.last[i] = 1; st.finished[i] = 0; st.phrase_num[i] = 0; }
    if (i < B * L) {
        st.phrase_length[i] = 0;
        st.phrase_syn[i] = pad_idx;
        st.ext_syn[i] = (i % L == 0) ? len_idx : pad_idx;     // position 0 is the [LEN] marker
    }
}

int launch_bound_init(const BoundState& st, int B, int L, int pad_idx, int len_idx, hipStream_t s) {
    const int n = max(B * L, 4);
    hipLaunchKernelGGL(bound_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, B, L, pad_idx, len_idx);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

__global__ void bound_export_kernel(BoundState st, int B, int L, int S, int* phrase_num, int* phrase_length,
                                    int64_t* phrase_syn, int* iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && iters) *iters = st.counters[1];
    if (i < B && phrase_num) phrase_num[i] = st.phrase_num[i];
    if (i < B * S) {
        const int b = i / S, t = i - b * S;                      // reference returns [:, :-2]
        if (phrase_length) phrase_length[i] = st.phrase_length[b * L + t];
        if (phrase_syn) phrase_syn[i] = st.phrase_syn[b * L + t];
    }
}

int launch_bound_export(const BoundState& st, int B, int L, int S, int* phrase_num, int* phrase_length,
                        int64_t* phrase_syn, int* iters, hipStream_t s) {
    hipLaunchKernelGGL(bound_export_kernel, dim3((B * S + 255) / 256), dim3(256), 0, s, st, B, L, S, phrase_num,
                       phrase_length, phrase_syn, iters);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Row-0 self-attention of the bound layer.  Exact for a one-layer bound network:
// the layer input at position p is lut_syn[label_p]*sqrt(d) + pe[p], a function of (p, label) only,
// so K and V of every possible row are tabulated once per model ("kvtab", [L*10, 2d]) and the
// query of row 0 ([LEN] at position 0) is a constant vector q0.  Row 0 sees keys p < last[b]
// (tgt_mask[j, 0, :last] = True, TransformerModel.py:1859/1867).
// One workgroup per image, one wavefront per head (looping if H > 4).
template <typename T>
__global__ __launch_bounds__(256) void bound_selfattn_kernel(const T* __restrict__ q0, const T* __restrict__ kvtab,
                                                             BoundState st, const int* ext_syn, const int* last, int L,
                                                             int d, int H, T* __restrict__ ctx, int B) {
    if (st.counters && st.counters[0] >= B) return;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = min(last[b], L);
    for (int h = wave; h < H; h += 4) {
        float s = -INFINITY;
        int row = 0;
        if (lane < n) {
            row = lane * 10 + ext_syn[b * L + lane];
            const T* kr = kvtab + (size_t)row * 2 * d + h * 64;
            const T* qr = q0 + h * 64;
            float acc = 0.f;
#pragma unroll 8
            for (int i = 0; i < 64; ++i) acc = fmaf(ElemOps<T>::to_f32(qr[i]), ElemOps<T>::to_f32(kr[i]), acc);
            s = acc * 0.125f;
        }
        const float m = wave_max(s);
        float e = (lane < n) ? expf(s - m) : 0.f;
        const float sum = wave_sum(e);
        float pr = ElemOps<T>::to_f32(ElemOps<T>::from_f32(e / sum));     // P is held in compute dtype, as in attn.hip
        float o = 0.f;
        for (int j = 0; j < n; ++j) {
            const float pj = __shfl(pr, j, 64);
            const int rj = __shfl(row, j, 64);
            o = fmaf(pj, ElemOps<T>::to_f32(kvtab[(size_t)rj * 2 * d + d + h * 64 + lane]), o);
        }
        ElemOps<T>::store(ctx + (size_t)b * d + h * 64 + lane, o);
    }
}

int launch_bound_selfattn(const void* q0, const void* kvtab, int dtype, const BoundState& st, const int* ext_syn,
                          const int* last, int B, int L, int d, int H, void* ctx, bool early_out, hipStream_t s) {
    BoundState s2 = st;
    if (!early_out) s2.counters = nullptr;
    if (dtype == BOFI_DT_F32)
        hipLaunchKernelGGL((bound_selfattn_kernel<float>), dim3(B), dim3(256), 0, s, (const float*)q0, (const float*)kvtab, s2,
                           ext_syn, last, L, d, H, (float*)ctx, B);
    else
        hipLaunchKernelGGL((bound_selfattn_kernel<bf16_t>), dim3(B), dim3(256), 0, s, (const bf16_t*)q0, (const bf16_t*)kvtab,
                           s2, ext_syn, last, L, d, H, (bf16_t*)ctx, B);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Final norm of row 0 -> two 2-layer heads -> log-softmax -> first-max argmax -> slot bookkeeping.
// One workgroup per image; everything in float32 (the heads are 0.1 M parameters).
__global__ __launch_bounds__(256) void bound_heads_kernel(const float* __restrict__ y, BoundHeadWeights w, BoundState st, int B,
                                                          int L, int S, int d, int hh, int update, float* len_logp_out,
                                                          float* syn_logp_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (update && st.counters[0] >= B) return;
    float* xs = smem;                 // [d] normalised row
    float* hid = xs + d;              // [2*hh]
    float* lg = hid + 2 * hh;         // [32] logits: 0..19 length, 20..29 label
    float* stat = lg + 32;            // [2]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* yr = y + (size_t)b * d;
    if (wave == 0) {
        float s = 0.f;
        for (int k = lane; k < d; k += 64) s += yr[k];
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
        for (int k = lane; k < d; k += 64) { const float t = yr[k] - mean; q += t * t; }
        q = wave_sum(q);
        if (lane == 0) { stat[0] = mean; stat[1] = sqrtf(q / (float)(d - 1)) + 1e-6f; }
    }
    __syncthreads();
    for (int k = tid; k < d; k += 256) xs[k] = w.norm_gain[k] * (yr[k] - stat[0]) / stat[1] + w.norm_bias[k];
    __syncthreads();
    for (int j = wave; j < 2 * hh; j += 4) {
        const float* wr = w.w1 + (size_t)j * d;
        float acc = 0.f;
        for (int k = lane; k < d; k += 64) acc = fmaf(wr[k], xs[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) hid[j] = fmaxf(acc + w.b1[j], 0.f);
    }
    __syncthreads();
    if (tid < 30) {
        const bool is_len = tid < 20;
        const float* wr = is_len ? (w.len_w2 + tid * hh) : (w.syn_w2 + (tid - 20) * hh);
        const float* hv = is_len ? hid : (hid + hh);
        float acc = 0.f;
        for (int k = 0; k < hh; ++k) acc = fmaf(wr[k], hv[k], acc);
        lg[tid] = acc + (is_len ? w.len_b2[tid] : w.syn_b2[tid - 20]);
    }
    __syncthreads();
    if (tid != 0) return;
    int pick[2];
    for (int head = 0; head < 2; ++head) {
        const int n = head ? 10 : 20;
        float* v = lg + (head ? 20 : 0);
        float m = v[0];
        for (int i = 1; i < n; ++i) m = fmaxf(m, v[i]);
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += expf(v[i] - m);
        const float lse = logf(sum);
        int best = 0;
        float bv = -INFINITY;
        float* out = head ? (syn_logp_out ? syn_logp_out + (size_t)b * 10 : nullptr)
                          : (len_logp_out ? len_logp_out + (size_t)b * 20 : nullptr);
        for (int i = 0; i < n; ++i) {
            const float lp = (v[i] - m) - lse;
            if (out) out[i] = lp;
            if (lp > bv || (lp != lp && bv == bv)) { bv = lp; best = i; }     // first max; a NaN wins once
        }
        pick[head] = best;
    }
    if (!update) return;
    if (b == 0) st.counters[1] += 1;                      // iterations in which some image was active
    if (st.finished[b]) return;
    int ln = pick[0];
    const int sn = pick[1], la = st.last[b];
    bool fin = false;
    if (ln == 0 || sn < 4 || sn > 6) {                    // EOS (TransformerModel.py:1846-1849)
        fin = true;
    } else {
        if (ln + la >= S + 1) { ln = S + 1 - la; fin = true; }          // truncate (:1850-1855)
        const int slot = st.phrase_num[b];                // == iteration index while unfinished (Q3)
        st.phrase_length[b * L + slot] = ln;
        st.phrase_syn[b * L + slot] = sn;
        st.phrase_num[b] = slot + 1;
        for (int p = la; p < la + ln; ++p) st.ext_syn[b * L + p] = sn;
        st.last[b] = la + ln;
    }
    if (fin) { st.finished[b] = 1; atomicAdd(&st.counters[0], 1); }
}

int launch_bound_heads(const float* y, const BoundHeadWeights& w, const BoundState& st, int B, int L, int S, int d, int hh,
                       int update, float* len_logp, float* syn_logp, hipStream_t s) {
    const size_t shm = (size_t)(d + 2 * hh + 32 + 2) * sizeof(float);
    hipLaunchKernelGGL(bound_heads_kernel, dim3(B), dim3(256), shm, s, y, w, st, B, L, S, d, hh, update, len_logp, syn_logp);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Fill-pass input: pos_embed(tgt_embed(word) + syn_embed(label)), each Embeddings scaled by sqrt(d)
// (TransformerModel.py:576-577, 1486-1487, 1505-1507).  word = BOS everywhere unless tok != NULL.
__global__ void embed_fill_kernel(const float* __restrict__ lut_tok, const float* __restrict__ lut_syn,
                                  const float* __restrict__ pe, const int* __restrict__ ext_syn, const int64_t* tok, int B,
                                  int S, int L, int d, int bos_idx, float sqrt_d, float* __restrict__ x) {
    const int row = blockIdx.x;                       // (b, t)
    const int b = row / S, t = row - b * S;
    const int syn = ext_syn[b * L + t + 1];           // extend_phrase_syn[:, 1:-1]
    const int64_t word = tok ? tok[row] : bos_idx;
    const float* tr = lut_tok + (size_t)word * d;
    const float* sr = lut_syn + (size_t)syn * d;
    const float* pr = pe + (size_t)t * d;
    for (int k = threadIdx.x; k < d; k += blockDim.x) x[(size_t)row * d + k] = (tr[k] * sqrt_d + sr[k] * sqrt_d) + pr[k];
}

int launch_embed_fill(const float* lut_tok, const float* lut_syn, const float* pe, const int* ext_syn, const int64_t* tok,
                      int B, int S, int L, int d, int bos_idx, float* x, hipStream_t s) {
    const float sqrt_d = (float)sqrt((double)d);
    hipLaunchKernelGGL(embed_fill_kernel, dim3(B * S), dim3(128), 0, s, lut_tok, lut_syn, pe, ext_syn, tok, B, S, L, d, bos_idx,
                       sqrt_d, x);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Vocabulary epilogue, one workgroup per (image, position) row of V logits:
//   log_softmax (in place), greedy argmax with torch.max's CPU semantics (lowest index among
//   equal maxima; NaN beats everything and the first NaN is returned), pad after the image's
//   token count.  HBM-bound: V*4 bytes read + V*4 written per row.
__global__ __launch_bounds__(256) void vocab_finalize_kernel(float* __restrict__ logits, int V, int S, int log_softmax,
                                                             const int* ntok, int ntok_bias, int pad_idx, int64_t* seq) {
    __shared__ float red[8];
    __shared__ int redi[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* x = logits + (size_t)row * V;
    float m = -INFINITY;
    int first_nan = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float v = x[i];
        if (v != v) first_nan = min(first_nan, i);
        m = fmaxf(m, v);
    }
    m = wave_max(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first_nan = min(first_nan, __shfl_xor(first_nan, o, 64));
    if (lane == 0) { red[wave] = m; redi[wave] = first_nan; }
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    first_nan = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
    __syncthreads();
    float lse = 0.f;
    if (log_softmax) {
        float s = 0.f;
        for (int i = tid; i < V; i += 256) s += expf(x[i] - m);
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        lse = logf((red[0] + red[1]) + (red[2] + red[3]));
        __syncthreads();
    }
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        float v = x[i];
        if (log_softmax) {
            v = (first_nan != 0x7fffffff) ? __builtin_nanf("") : (v - m) - lse;    // one NaN poisons the row's softmax
            x[i] = v;
        }
        if (v > bv) { bv = v; bi = i; }               // i ascends per thread: first max kept
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
        for (int w = 1; w < 4; ++w)
            if (red[w] > bv || (red[w] == bv && redi[w] < bi)) { bv = red[w]; bi = redi[w]; }
        if (first_nan != 0x7fffffff) bi = log_softmax ? 0 : first_nan;
        if (bi == 0x7fffffff) bi = 0;                 // all -inf row: torch.max returns index 0
        if (ntok) {
            const int b = row / S, t = row - b * S;
            if (t >= ntok[b] + ntok_bias) bi = pad_idx;
        }
        seq[row] = bi;
    }
}

int launch_vocab_finalize(float* logits, int rows, int V, int S, int log_softmax, const int* ntok, int ntok_bias, int pad_idx,
                          int64_t* seq, hipStream_t st) {
    if (!logits || !seq || rows < 0 || V <= 0 || S <= 0) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    hipLaunchKernelGGL(vocab_finalize_kernel, dim3(rows), dim3(256), 0, st, logits, V, S, log_softmax, ntok, ntok_bias, pad_idx,
                       seq);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi

extern "C" int bofi_vocab_finalize(float* logits, int rows, int V, int S, int log_softmax, const int* ntok, int pad_idx,
                                   int64_t* seq, void* stream) {
    return bofi::launch_vocab_finalize(logits, rows, V, S, log_softmax, ntok, 0, pad_idx, seq, (hipStream_t)stream);
}
