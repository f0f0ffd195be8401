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
    int* counters;        // [8]: 0 = images finished, 1 = bound iterations executed, (2, 3: the semi-autoregressive loop's halt / NaN words), 4 = fp16 saturation word, 5 = groups that ran as a pair of workgroups (diagnostic) of the
                          //      persistent bounding-loop kernel (BoundLoopArgs.sat), zeroed by launch_bound_init like the others
    int* klen;            // [B, L]  keys row r of the bound sequence may attend (tgt_mask rows are key prefixes, TransformerModel.py:1859-1867):
                          //         maintained for the dense (N_len >= 2) bounding pass; may be NULL
    unsigned* pair_ctl;   // [ceil(B / 16) * 4] control words of the loop kernel's workgroup pairs (BoundLoopArgs.xctl), zeroed by launch_bound_init with the rest; may be NULL
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

struct BoundHeadWeights {   // float32 unless noted
    const float* norm_gain; const float* norm_bias;      // length_predictor.norm
    const float* w1t; const float* b1;                   // TRANSPOSED [d, 2*hh]: Length_classifier1 | Syntactic_classifier1
    const float* len_w2; const float* len_b2;            // [20, hh]
    const float* syn_w2; const float* syn_b2;            // [10, hh]
    const void* w1p;                                     // w1t in the compute dtype, packed per thread of the tail kernel:
                                                         // [8 K-slices][2*hh/4 groups][d/8 k][4 outputs] (launch_pack_w1p)
};

// arguments of the per-image tail kernel of a bounding iteration (naic.hip)
struct BoundTailArgs {
    const float* y; int yparts;          // HEADS: FFN output of row 0, [yparts][B][y_stride] float32 partial slabs (summed in fixed order)
    int y_stride;                        // elements between the row-0 vectors of consecutive images (0: d)
    BoundHeadWeights w;
    BoundState st; SaicState sa;
    const int* ext_syn_in; const int* last_in;   // non-NULL: a given slot layout instead of the engine's state (stage API)
    const void* q0; const void* kvtab;   // ATTN: query of row 0 [d]; K|V of every (position, label) row [L*10][2d]; compute dtype
    const void* votab;                   // ATTN: Wo_self[:, h-block] . V[row, h-block]: [L*10][H][d], compute dtype
    const float* x0b;                    // ATTN: x0 + bo_self [d]
    float* y1; void* y1t; float* stats;  // ATTN outputs: y1 [B,d] float32, its compute-dtype copy (or NULL), partial (sum, sumsq) [B][d/32][2]
    int B, L, S, d, hh, H, flags, iter;
    float* len_logp; float* syn_logp;    // HEADS: optional outputs [B,20] / [B,10]
    float* dbg_part;                     // developer aid: [B][8*2*hh + d] partial hidden sums and the normalised row, or NULL
};

// arguments of the persistent bounding-loop kernel (bound_loop.hip): one workgroup per 16 images runs every iteration of core_NAIC's loop
typedef __attribute__((ext_vector_type(4))) uint32_t bl_u32x4;
struct BoundLoopArgs {
    // fp16 fragment-major weights of the bounding layer (launch_pack_frag16) and their float32 epilogue vectors
    const bl_u32x4* wo_self; const float* x0b;           // y1 = (x0 + bo_self) + Wo_self . ctx                  [512][512], [512]
    const bl_u32x4* wq_src; const float* cq;              // q = Wq_src' . LN(y1) + c (sublayer[1].norm folded)   [512][512], [512]
    const bl_u32x4* wo_src; const float* bo_src;          // y2 = y1 + Wo_src . ctx2 + bo
    const bl_u32x4* w1; const float* c1;                  // h = relu(W1' . LN(y2) + c) (sublayer[2].norm folded) [dff][512], [dff]
    const bl_u32x4* w2; const float* b2;                  // y3 = y2 + W2 . h + b2                                [512][dff], [512]
    const bl_u32x4* wh; const float* ch;                  // heads' hidden layers, length | label, final norm folded [256][512] (rows >= 2*hh zero), [256]
    const float* len_w2; const float* len_b2; const float* syn_w2; const float* syn_b2;      // output layers, float32 [20][hh], [20], [10][hh], [10]
    const float* sctab; const float* vtab;                // launch_bound_tables: [L*10][8], [L*10][512] float32
    const uint16_t* k; const uint16_t* v; int ldkv;       // cross-attention K and V rows of the images [B*R][ldkv] bf16
    const int* att_len;                                   // regions per image (NULL: R each)
    BoundState st;                                        // update != 0: slot state in (after launch_bound_init) and out
    const int* ext_syn_in; const int* last_in;            // update == 0 (stage API): the slot layout to evaluate
    float* len_logp; float* syn_logp;                     // optional [B][20] / [B][10]: log-probabilities of the first iteration run
    int B, R, L, S, hh, dff;
    int max_iters;                                        // iterations at most (seq_length; 1 for the stage API)
    int update;                                           // apply the slot bookkeeping and loop until the group's images are finished
    int dbg;                                              // developer aid (BOFI_BL_DBG = i + 1: in-kernel stamps of iteration i), set by the launcher
    int* sat;                                             // fp16 saturation word (round 6; NULL: a scratch word of the library): bit 0 is OR-ed in when an activation (attention context, hidden
                                                          // row) was clamped to +-65 504 on its way into an fp16 MFMA operand, bit 1 when the fp16 weight copies were clamped at pack time (*wsat != 0)
    const int* wsat;                                      // the pack-time word of launch_pack_frag16 (may be NULL)
    // two workgroups per group (round 6): the feed-forward's hidden units split in halves over a PAIR of workgroups -- each streams half of w_1 and w_2
    // (2 of the iteration's 5.75 MB less per workgroup) and the two partial sums of y3 meet once per iteration through `xbuf` (write-through stores, agent-scope
    // loads, no fences).  `xctl` [groups][4] unsigned, zeroed by the launcher: [0] the pair's state (0 unclaimed, 2 first workgroup running, 3 first went solo, 4 pair
    // formed), [1] / [2] iteration counters of role 0 / 1.  `xbuf` [groups][2 parities][2 roles][16][512] float32.  NULL / NULL: one workgroup per group.
    float* xbuf; unsigned* xctl;
    int pair;                                             // set by the launcher: the grid holds two workgroups per group
};
int launch_bound_loop(const BoundLoopArgs& a, hipStream_t s);
struct Pack16Entry { const float* w[2]; const float* gain; void* out; int n_each, nsrc, K, Npad, blk0; };      // blk0: set by the launcher
struct Pack16Table { Pack16Entry e[8]; int n; int* sat; };      // sat (may be NULL): set to 1 when a weight (times its folded gain) left the fp16 range and was clamped; the launcher zeroes it first
int launch_pack_frag16(const Pack16Table& t, hipStream_t s);
struct BoundTablesArgs {
    const float* xt; const float* x0; int rows;           // layer inputs [rows][512], the row-0 input [512]
    const float* n0g; const float* n0b;                   // sublayer[0].norm
    const float* wq; const float* bq; const float* wk; const float* bk; const float* wv; const float* bv;      // self_attn.linears.0-2, float32 [512][512]
    float* q0; float* sctab; float* vtab;                 // out: [512], [rows][8], [rows][512]
};
int launch_bound_tables(const BoundTablesArgs& a, hipStream_t s);

int launch_bound_init(const BoundState& st, int B, int L, int pad_idx, int len_idx, hipStream_t s);
int launch_bound_export(const BoundState& st, int B, int L, int S, int* phrase_num, int* phrase_length,
                        int64_t* phrase_syn, int* iters, hipStream_t s, int* live_max = nullptr, int* sat_out = nullptr);
// flags of launch_bound_tail
#define BOUND_HEADS 1    /* final norm + heads + argmax on y */
#define BOUND_UPDATE 2   /* apply the slot bookkeeping (needs BOUND_HEADS) */
#define BOUND_ATTN 4     /* row-0 self-attention sublayer of the next iteration -> y1 (+ copy, + row statistics) */
#define BOUND_EARLY 8    /* return at once when every image is finished */
#define BOUND_SAIC 16    /* SAIC bookkeeping (TransformerModel.py:1910-1948) instead of NAIC's; early-out on the halt word */
int launch_bound_tail(const BoundTailArgs& a, int dtype, hipStream_t s);
// derived tables of the bound layer (repack.hip): the packed hidden weights of the heads, the projected V table, x0 + bo
int launch_pack_w1p(const float* w1t, void* w1p, int dtype, int d, int nh, hipStream_t s);
int launch_votab(const void* kvtab, const void* wo, const float* x0, const float* bo, void* votab, float* x0b, int dtype, int rows, int d, int H,
                 hipStream_t s);
// rows of pos_embed(tgt_embed(tok) [+ syn_embed(syn)]): row r = (b, t), ids read at [b*ld + off + t]; tok == NULL -> BOS
int launch_embed_rows(const float* lut_tok, const float* lut_syn, const float* pe, const int* tok, const int* syn, int ld, int off,
                      int B, int T, int d, int bos_idx, float* x, void* xt, int dtype, float* stats, const int* halt, hipStream_t s);
int launch_saic_init(const BoundState& st, const SaicState& sa, int B, int L, int pad_idx, int bos_idx, int len_idx, hipStream_t s);
// after the decoder pass of iteration `iter`: copy the new phrase's tokens / log-probs (TransformerModel.py:1968-1977), set halt
int launch_saic_halt(const BoundState& st, const SaicState& sa, int B, int L, int iter, hipStream_t s);
int launch_saic_copy(const BoundState& st, const SaicState& sa, const int64_t* tok, const float* logp, float* seq_logprob, int B, int L,
                     int S, int V, int iter, hipStream_t s);
// decoder rows of the phrases placed in iteration `iter` (image-major), and their count
int launch_saic_rows(const BoundState& st, int B, int L, int S, int iter, int* rows, int* n_rows, hipStream_t s);
int launch_saic_put_words(const BoundState& st, const SaicState& sa, const int64_t* seq, int B, int L, int S, hipStream_t s);
int launch_saic_export(const BoundState& st, const SaicState& sa, int B, int L, int S, int64_t* seq, int* phrase_num,
                       int* phrase_length, int64_t* phrase_syn, int* iters, hipStream_t s);
int launch_embed_fill(const float* lut_tok, const float* lut_syn, const float* pe, const int* ext_syn, const int64_t* tok,
                      int B, int S, int L, int d, int bos_idx, float* x, void* xt, int dtype, float* stats, hipStream_t s, const void* qkv_tab = nullptr, void* qkv_out = nullptr, int nq = 0);

}  // namespace bofi
