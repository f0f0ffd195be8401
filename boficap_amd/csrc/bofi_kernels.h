// Internal C++ launch API shared by the translation units of libboficap_hip.so.
// The public C ABI is include/boficap_hip.h; these are the same operations with typed streams.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bofi {

int launch_layernorm(const float* x, const float* gain, const float* bias, void* y, int y_dtype, int rows, int d,
                     hipStream_t st, const int* skip_if_ge = nullptr, int skip_threshold = 0);

int launch_cast_bf16(const float* x, void* y, size_t n, hipStream_t st);

struct LinearArgs {
    const void* x; int x_dtype; int ldx;
    const void* w; int w_dtype;
    const float* bias;
    const float* residual; int ldr;
    void* y; int y_dtype; int ldy;
    int M, N, K;
    int relu;
    const int* row_len; int rows_per_group;
    // optional fused LayerNorm on the rows of x (x must be float32, K == d_model): the A tile is
    // normalised while it is staged into LDS
    const float* ln_gain; const float* ln_bias;
    // optional early-out word: the kernel returns at once when *skip_if_ge >= skip_threshold
    const int* skip_if_ge; int skip_threshold;
    // LayerNorm folded into the GEMM (LDS-DMA kernel only): x is the raw residual stream in the compute
    // dtype, w already carries the LN gain, bias carries c[n] = bias[n] + sum_k b_ln[k] W[n][k];
    // ln_stats: float2 [M][K/32] partial (sum, sumsq) of the rows of the fp32 residual stream,
    // ln_colsum: float [N] column sums of the (rounded) scaled weight
    const float* ln_stats; const float* ln_colsum;
    int ln_groups;            // partial-sum pairs per row of ln_stats (0: K / 32; the direct-operand kernels of bound_ops.hip write K / 16)
    float* stats_out;         // write partial (sum, sumsq) of the OUTPUT rows: float2 [M][N/32]
    void* y2; int ldy2;       // second copy of the output in the compute dtype (feeds the next folded GEMM)
    int splitk;               // > 1: K split over workgroups; y receives `splitk` float32 partial slabs [splitk][M][ldy]
    // training: y = residual + keep(act(x w^T + bias)) / (1 - p), keep = drop_hash(drop_seed, m * N + n) >= drop_thresh (0: off)
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed;
    const uint64_t* drop_step;    // device word added to drop_seed when set (a captured graph replays with fresh masks)
    // training backward through relu (+ dropout): != 0 turns `residual` into a MASK -- y = residual > 0 ? (x w^T) * mask_scale : 0
    // (residual = the forward activation: a dropped or clipped unit is 0 there; mask_scale = 1 / (1 - p))
    float mask_scale;
    // row list (LDS-DMA kernel only): the GEMM covers rows row_idx[0 .. *m_dev) of x (both device memory; M is the capacity) and writes
    // the same rows of y / y2 / stats_out; residual and ln_stats are read at those rows too
    const int* row_idx; const int* m_dev;
};
int launch_linear(const LinearArgs& a, hipStream_t st);
extern double g_gemm_flops, g_gemm_flops_skippable;   // GEMM FLOPs enqueued since the last reset (host-side tally; gemm.hip)
int launch_linear_glds(const LinearArgs& a, hipStream_t st);      // gemm_glds.hip; -1 = not eligible

struct AttnArgs {
    const void* q; int ldq;
    const void* k; int ldk;
    const void* v; int ldv;
    void* out; int ldo;
    int dtype;
    int B, H, Lq, Lk;
    const int* klen; int klen_sb, klen_sq;
    int klen_bias;            // effective length = klen[...] + klen_bias (e.g. -1 for last-1)
    int klen_shared_last;     // quirk Q1, group size G (0: off): item b uses entry min(B, (b/G + 1)*G) - 1 of klen; G >= B: entry B-1 for all
    const int* skip_if_ge; int skip_threshold;
    int kdiv;                 // key/value batch item = b / kdiv (0 or 1: one per query batch item)
    // training: dropout on the attention probabilities (TransformerModel.py:1430-1431), bf16 kernel only;
    // keep(b, h, q, k) = drop_hash(seed, ((b*H + h)*Lq + q)*Lk + k) >= drop_thresh (0: off)
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed; const uint64_t* drop_step;
    // training with unpadded captions: batch item b owns the q_count[b] query rows that start at row q_start[b] (B entries each;
    // NULL: the dense layout b * Lq, Lq rows); k_ragged: its keys / values are laid out the same way (self-attention) instead of
    // (b / kdiv) * Lk.  klen is then indexed by the global query row.  Lq / Lk stay the maxima (they size the kernel).
    const int* q_start; const int* q_count; int k_ragged;
    int q_rows;                   // > 0 with q_start: rows of the q / out buffers; the bf16 kernels clear the rows behind the last item
};
int launch_attention(const AttnArgs& a, hipStream_t st);

// bound_ops.hip: the row-0 stages of a bounding iteration for at most 64 images (bf16, d_model 512)
struct BoundQAttnArgs {
    const uint16_t* x;            // y1 in bf16 [B][d]
    const float* stats;           // its row partial sums [B][d/32][2]
    const uint16_t* wq; const float* bias; const float* colsum;     // Wq_src with sublayer[1].norm folded in (Lin of engine.hip)
    const uint16_t* k; const uint16_t* v; int ldkv;                  // cross-attention K and V rows [B*R][ldkv]
    const int* att_len;           // regions per image (NULL: R each)
    uint16_t* out;                // attention output [B][d] (heads concatenated)
    int B, R, d, H;
    const int* skip_if_ge; int skip_threshold;
    // query rows from a device-side list: rows row_idx[0 .. *n_rows) of x / stats / out (B = capacity); rows_per_image > 0: row r belongs
    // to image r / rows_per_image (several query rows per image: the decoder's positions), else row = image
    const int* row_idx; const int* n_rows; int rows_per_image;
    int stats_groups;             // partial-sum pairs per row of `stats` (0: d / 32)
};
int launch_bound_qattn(const BoundQAttnArgs& a, hipStream_t st);
struct RowGemmArgs {
    const uint16_t* x; int ldx;   // [M][K] bf16
    const uint16_t* w;            // [N][K] bf16
    const float* bias;
    const float* stats; int stats_groups; const float* colsum;     // LayerNorm fold on x: [M][stats_groups][2] partial sums of the fp32 rows
    const float* residual; int ldr;
    float* y; int ldy;            // float32 output (split-K: `splitk` partial slabs [splitk][M][ldy], bias and residual in slab 0)
    uint16_t* yb; int ldyb;       // bf16 output / copy
    float* stats_out;             // [M][N/16][2] partial sums of the output rows
    int M, N, K, splitk, relu;
    const int* skip_if_ge; int skip_threshold;
    const int* row_idx; const int* m_dev;     // row list: rows row_idx[0 .. *m_dev) of x / residual / statistics in, the same rows of the outputs
    int kchunks;                              // (set by the launcher) K / (512 * splitk)
};
int launch_rowgemm(const RowGemmArgs& a, hipStream_t st);
int launch_attention_bf16(const AttnArgs& a, hipStream_t st);     // attn_bf16.hip; -1 = not eligible

int launch_vocab_finalize(float* logits, int rows, int V, int S, int log_softmax, const int* ntok, int ntok_bias,
                          int pad_idx, int64_t* seq, hipStream_t st, int* nan_flag = nullptr, const int* halt = nullptr,
                          const int* row_idx = nullptr, const int* n_rows = nullptr,      // row list: rows row_idx[0 .. *n_rows) only
                          const float* src = nullptr, int ld_src = 0,                    // src: the logits are read from src (row pitch ld_src), results go to `logits` (pitch V)
                          float* row_plogp = nullptr, float* row_chosen = nullptr);      // optional (log_softmax != 0): per row sum_v p log p and the log-prob of the emitted id (what bofi_vocab_stats reads back out of the tensor)

// ---- row-block sublayer kernels (rowblock.hip): bf16, d_model 512, weights in the fragment-major layout of launch_rb_pack_frag
typedef __attribute__((ext_vector_type(4))) uint32_t rb_u32x4;
struct RbFfnArgs {
    const float* x; int ldx;                  // residual stream in [M][512] float32
    const rb_u32x4* w1p; const float* c1; const float* cs1;      // w_1 (pre-norm folded in) fragment-major; its folded bias and column sums [dff]
    const rb_u32x4* w2p; const float* b2;     // w_2 fragment-major; bias [512]
    float* y; int ldy;                        // residual stream out (may be x: a workgroup reads its rows before it writes them)
    uint16_t* yb; float* stats_out;           // optional: bf16 copy [M][512], partial sums [M][16][2]
    int M, dff;
    int dbg;                                  // developer aid (BOFI_RB_DBG & 16): in-kernel stamps
    int alone;                                // 1: this launch runs with nothing beside it (bofi_engine_set_decodes_in_flight(1)): the 64-row kernel's shorter chain
    // optional: the LayerNorm-folded projection that reads this sublayer's output (the next layer's q|k|v, the stacked cross K|V) computed from the
    // closed block while it is still in LDS: pj_y[M][pj_ldy] bf16 = W_pj . LN(y) + c_pj (what launch_rb_gemm would compute from y)
    const rb_u32x4* pj_wp; const float* pj_c; const float* pj_cs; void* pj_y; int pj_ldy, pj_N;
    // optional HEAD segment (round 6): the attention sublayer's output projection and residual in front of the feed-forward sublayer, from the attention CORE's context rows:
    //   x1 = x + W_o . ctx + b_o;   y = x1 + w_2 . relu(w_1 . LN(x1) + b_1) + b_2
    // (SublayerConnection around MultiHeadedAttention's last Linear, TransformerModel.py:1361-1363, 1467, then :1477-1478): head_ctx [M][head_ldc] bf16 (the heads' outputs
    // side by side, what launch_attention writes), head_wop W_o fragment-major, head_bo [512].  x1 never exists in memory.  The 80-row kernel, one block per workgroup.
    const uint16_t* head_ctx; int head_ldc; const rb_u32x4* head_wop; const float* head_bo;
};
struct RbAttnArgs {
    const uint16_t* q; int ldq;               // [B*Lq][ldq], head h at columns h*64
    const uint16_t* k; int ldk;               // [B*Lk][ldk]
    const uint16_t* v; int ldv;
    int B, Lq, Lk;
    const int* klen; int klen_sb, klen_sq, klen_bias, klen_shared_last;     // as AttnArgs
    const rb_u32x4* wop; const float* bo;     // output projection [512][512] fragment-major, bias [512]
    const float* x; int ldx;                  // residual stream in [B*Lq][512] float32
    float* y; int ldy;                        // out (may be x)
    uint16_t* yb; float* stats_out;           // optional, as RbFfnArgs
    int dbg;                                  // developer ablation (BOFI_RB_DBG): 1 = no attention, 2 = no output projection, 4 = no closing stores, 16 = stamps
    int alone;                                // 1: nothing runs beside this launch (query blocks of <= 20 rows then take the 8-wavefront workgroups: shorter, but W_o streamed twice as often)
    // optional projection tail (query blocks of <= 20 rows, keys <= 32, the 16-wavefront form): pj_y[B*Lq][pj_ldy] bf16 = W' . LN(y) + c for a LayerNorm-folded
    // [512][512] projection of the sublayer's OUTPUT (the decoder layer's cross-attention queries), computed from each block while it sits in LDS
    const rb_u32x4* pj_wp; const float* pj_c; const float* pj_cs; uint16_t* pj_y; int pj_ldy;
};
struct RbGemmArgs {
    const float* x; int ldx;                  // [M][512] float32 residual stream (the LayerNorm is folded into w / c / cs)
    const rb_u32x4* wp; const float* c; const float* cs;      // fragment-major [N][512]; folded bias, column sums [N]
    void* y; int ldy; int y_f32;              // bf16 (or float32) [M][ldy]
    int M, N, relu;
    int dbg;                                  // developer aid (BOFI_RB_DBG & 16: in-kernel stamps, set by the C entry)
    int alone;                                // 1: nothing runs beside this launch: 64-row blocks (more, shorter workgroups)
};
int launch_rb_gemm(const RbGemmArgs& a, hipStream_t st);
int launch_rb_ffn(const RbFfnArgs& a, hipStream_t st);
int launch_rb_attn(const RbAttnArgs& a, hipStream_t st);           // -1: shape not covered
int launch_rb_pack_frag(const void* w, void* out, int N, int K, hipStream_t st);

// ---- device-side weight repack (repack.hip)
struct PackLinArgs {
    const float* w[16]; const float* b[16]; int nsrc, n_each, K;
    const float* gain; const float* bln;            // pre-norm LayerNorm to fold in (or null)
    float* bout; float* cs;
};
int launch_pack_lin(const PackLinArgs& a, void* wout, int dtype, hipStream_t st);
// the whole refresh of an engine's Linears as ONE launch each for the packed weights and for their fragment-major copies, and one for the
// small float32 vectors (norm vectors, head layers): descriptor tables in device memory (repack.hip)
struct PackLinDesc {
    PackLinArgs a; void* wout; void* wp;          // wp: fragment-major copy of wout [Npad][K] bf16 (or null)
    int Npad;
    int row0;                                     // first row of this entry in the launch's row numbering (a multiple of 4)
    int blk0;                                     // first 256-thread block of this entry in the fragment-major launch
};
struct CopyDesc { const float* src; float* dst; int n; int blk0; };      // blk0: first 256-element block in the launch
int launch_pack_lin_multi(const PackLinDesc* tab_dev, int n_ent, int total_rows, int dtype, hipStream_t st);
int launch_pack_frag_multi(const PackLinDesc* tab_dev, int n_ent, int total_blocks, hipStream_t st);
int launch_copy_multi(const CopyDesc* tab_dev, int n_ent, int total_blocks, hipStream_t st);
int launch_pack_heads(const float* lw1, const float* sw1, const float* lb1, const float* sb1, float* w1t, float* b1, int d, int hh, hipStream_t st);
int launch_bound_table(const float* lut_syn, const float* lut_tok, const float* pe, float* xt, float* x0, float* x0_sa, int L, int d, int len_idx,
                       hipStream_t st);

int launch_vocab_sample(const float* logp, int rows, int V, int S, int n, float temperature, uint64_t seed, const int* ntok, int pad_idx,
                        int64_t* out, hipStream_t st, const int* halt = nullptr, const int* row_idx = nullptr, const int* n_rows = nullptr,
                        const uint64_t* seed_dev = nullptr);      // seed_dev: device word added to `seed` (so that a captured launch can draw anew)
int launch_set_u64(uint64_t* p, uint64_t v, hipStream_t s);
int launch_zero_f32(float* p, size_t n, hipStream_t s);

}  // namespace bofi
