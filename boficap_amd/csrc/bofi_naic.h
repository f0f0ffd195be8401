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
int launch_bound_tail(const float* y, const BoundHeadWeights& w, const BoundState& st, const int* ext_syn_in, const int* last_in,
                      const void* q0, const void* kvtab, void* ctx, int dtype, int B, int L, int S, int d, int hh, int H, int flags,
                      float* len_logp, float* syn_logp, hipStream_t s);
int launch_embed_fill(const float* lut_tok, const float* lut_syn, const float* pe, const int* ext_syn, const int64_t* tok,
                      int B, int S, int L, int d, int bos_idx, float* x, void* xt, int dtype, float* stats, hipStream_t s);

}  // namespace bofi
