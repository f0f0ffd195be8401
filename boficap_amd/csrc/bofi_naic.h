// Device-resident slot state of the bounding pass and the launchers of naic.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bofi {

// Per-image state of core_NAIC (reference TransformerModel.py:1825-1831), all int32 on the device.
// L = seq_length + 2 positions per image.
struct BoundState {
    int* last;            // [B]   end of the laid-out slots in extend_phrase_syn (starts at 1)
    int* finished;        // [B]
    int* phrase_num;      // [B]
    int* phrase_length;   // [B, L]
    int* phrase_syn;      // [B, L]
    int* ext_syn;         // [B, L]  extend_phrase_syn: [LEN] id at 0, label of the slot covering p
    int* counters;        // [4]: 0 = images finished, 1 = bound iterations executed
};

// Extra per-image state of core_SAIC (reference TransformerModel.py:1879-1896), int32 on the device.
// BoundState.last doubles as phrase_last; BoundState.counters[2] is the halt word (all finished or NaN seen),
// counters[3] the NaN flag.
struct SaicState {
    int* seq_last;        // [B]
    int* seq;             // [B, L]  tokens, position 0 = BOS
    int* ext_len;         // [B, L]  extend_phrase_len: [LEN] id at 0, then the tokens emitted so far
    int* ext_phrase;      // [B, L]  decoder input tokens (position-wise copy of the previous phrase)
    int* klen_dec;        // [B, L]  key-prefix length of row r of phrase_mask
};

struct BoundHeadWeights {   // float32
    const float* norm_gain; const float* norm_bias;      // length_predictor.norm
    const float* w1t; const float* b1;                   // TRANSPOSED [d, 2*hh]: Length_classifier1 | Syntactic_classifier1
    const float* len_w2; const float* len_b2;            // [20, hh]
    const float* syn_w2; const float* syn_b2;            // [10, hh]
};

int launch_bound_init(const BoundState& st, int B, int L, int pad_idx, int len_idx, hipStream_t s);
int launch_bound_export(const BoundState& st, int B, int L, int S, int* phrase_num, int* phrase_length,
                        int64_t* phrase_syn, int* iters, hipStream_t s);
// flags of launch_bound_tail
#define BOUND_HEADS 1    /* final norm + heads + argmax on y */
#define BOUND_UPDATE 2   /* apply the slot bookkeeping (needs BOUND_HEADS) */
#define BOUND_ATTN 4     /* row-0 self-attention of the next iteration -> ctx */
#define BOUND_EARLY 8    /* return at once when every image is finished */
#define BOUND_SAIC 16    /* SAIC bookkeeping (TransformerModel.py:1910-1948) instead of NAIC's; early-out on the halt word */
int launch_bound_tail(const float* y, const BoundHeadWeights& w, const BoundState& st, const int* ext_syn_in, const int* last_in,
                      const void* q0, const void* kvtab, void* ctx, int dtype, int B, int L, int S, int d, int hh, int H, int flags,
                      float* len_logp, float* syn_logp, hipStream_t s, const SaicState* sa = nullptr, int iter = 0, int yparts = 1);
// rows of pos_embed(tgt_embed(tok) [+ syn_embed(syn)]): row r = (b, t), ids read at [b*ld + off + t]; tok == NULL -> BOS
int launch_embed_rows(const float* lut_tok, const float* lut_syn, const float* pe, const int* tok, const int* syn, int ld, int off,
                      int B, int T, int d, int bos_idx, float* x, void* xt, int dtype, float* stats, const int* halt, hipStream_t s);
int launch_saic_init(const BoundState& st, const SaicState& sa, int B, int L, int pad_idx, int bos_idx, int len_idx, hipStream_t s);
// after the decoder pass of iteration `iter`: copy the new phrase's tokens / log-probs (TransformerModel.py:1968-1977), set halt
int launch_saic_copy(const BoundState& st, const SaicState& sa, const int64_t* tok, const float* logp, float* seq_logprob, int B, int L,
                     int S, int V, int iter, hipStream_t s);
int launch_saic_export(const BoundState& st, const SaicState& sa, int B, int L, int S, int64_t* seq, int* phrase_num,
                       int* phrase_length, int64_t* phrase_syn, int* iters, hipStream_t s);
int launch_embed_fill(const float* lut_tok, const float* lut_syn, const float* pe, const int* ext_syn, const int64_t* tok,
                      int B, int S, int L, int d, int bos_idx, float* x, void* xt, int dtype, float* stats, hipStream_t s);

}  // namespace bofi
