/*
 * boficap_hip.h -- C ABI of libboficap_hip.so, the MI355X (gfx950) implementation of BoFiCap's
 * bound+fill caption decoder hot path.
 *
 * The reference (ChangxinWang/BoFiCap) is pure Python/PyTorch and has no FFI layer; its hot path
 * is a chain of stock torch ops.  Each entry point below therefore names the reference function
 * (file:line under /root/reference) whose arithmetic it replaces.  All pointers are DEVICE
 * pointers unless a name ends in _host; all sizes are element counts; `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  Every function returns 0 on success and a
 * BOFI_ERR_* code otherwise; nothing here allocates or synchronises except the engine
 * constructor/finaliser.  Thread-safety: one engine per host thread / per GPU process (the
 * reference is single-threaded per replica, tools/train.py:99-101).
 *
 * dtype codes: 0 = float32, 1 = bfloat16 (raw 16-bit).  "compute dtype" is the type GEMM and
 * attention operands are held in; accumulation, softmax, LayerNorm statistics and the residual
 * stream are always float32.
 */
#ifndef BOFICAP_HIP_H
#define BOFICAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BOFI_OK 0
#define BOFI_ERR_ARG 1    /* bad shape / dtype / NULL pointer */
#define BOFI_ERR_HIP 2    /* a HIP runtime call failed */
#define BOFI_ERR_STATE 3  /* engine used before finalize, missing weight, ... */

#define BOFI_DT_F32 0
#define BOFI_DT_BF16 1

/* version of this ABI; bumped on any signature change */
int bofi_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Operator level (stateless).  Used by the engine and by the per-kernel parity tests.
 * ------------------------------------------------------------------------------------------- */

/* BoFiCap LayerNorm: y = gain * (x - mean) / (std_unbiased + 1e-6) + bias, per row.
 * Replaces captioning/models/TransformerModel.py:1346-1349 (NOT nn.LayerNorm: N-1 variance, eps
 * added to std).  x: float32 [rows, d];  y: y_dtype [rows, d];  d % 64 == 0, d <= 2048. */
int bofi_layernorm(const float* x, const float* gain, const float* bias, void* y, int y_dtype,
                   int rows, int d, void* stream);

/* y = act(x . w^T + bias) [+ residual], the nn.Linear of TransformerModel.py:1454-1456,1467
 * (attention projections), :1477-1478 (FFN), :1642-1645 (att_embed Linear+ReLU), :1316-1319
 * (generator.proj).
 *   x: x_dtype [M, K] row stride ldx (x_dtype = w_dtype, or float32 converted on load)
 *   w: w_dtype [N, K] contiguous (the nn.Linear weight as stored);  bias: float32 [N] or NULL
 *   residual: float32 [M, N] row stride ldr or NULL, added AFTER bias/activation
 *             (SublayerConnection, TransformerModel.py:1361-1363)
 *   y: y_dtype [M, N] row stride ldy (float32 or w_dtype)
 *   relu: 0/1.   row_len/rows_per_group: if row_len != NULL, output row m is forced to exact 0
 *   when (m % rows_per_group) >= row_len[m / rows_per_group]  (pack_padded/pad_packed semantics of
 *   AttModel.py:46-51).   K % 64 == 0 for bf16, K % 32 == 0 for float32. */
int bofi_linear(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* bias,
                const float* residual, int ldr, void* y, int y_dtype, int ldy, int M, int N, int K,
                int relu, const int* row_len, int rows_per_group, void* stream);

/* Multi-head scaled-dot-product attention core, softmax(q k^T / sqrt(64) masked) v, head dim 64.
 * Replaces attention() TransformerModel.py:1421-1432 for prefix-structured masks: query row i of
 * batch item b may attend keys j < klen[b*klen_sb + i*klen_sq] (every mask the reference builds
 * for this path has that form).  A row with klen 0 yields NaN like the reference's
 * softmax over all -inf.  klen == NULL means all Lk keys.
 *   q: dtype, row (b,i) at q + (b*Lq+i)*ldq, head h at column h*64;  k, v likewise with Lk, ldk, ldv
 *   out: dtype [B*Lq, H*64] row stride ldo.    Lq, Lk <= 128. */
int bofi_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                   int ldo, int dtype, int B, int H, int Lq, int Lk, const int* klen, int klen_sb,
                   int klen_sq, void* stream);

/* log_softmax over the vocabulary + greedy pick + pad-after-length, one row per (image, position).
 * Replaces F.log_softmax(self.logit(phrase), dim=2) AttModel.py:206-207, torch.max(logprobs, 2)
 * CaptionModel.py:388-390 (first maximal index; a NaN wins and the first NaN is returned, as
 * torch.max does on CPU) and seq[b, sum(phrase_length[b]):] = pad AttModel.py:422-423.
 *   logits: float32 [rows, V], overwritten in place with log-probabilities when log_softmax != 0
 *   ntok: int32 [rows / S] valid token count per image or NULL;  seq: int64 [rows]. */
int bofi_vocab_finalize(float* logits, int rows, int V, int S, int log_softmax, const int* ntok,
                        int pad_idx, int64_t* seq, void* stream);

/* bofi_attention with the two extra knobs of the training path: effective key count =
 * klen[...] + klen_bias, and key/value batch item = b / kdiv (the seq_per_img captions of one image
 * attend the image's memory without repeating it; the reference repeats it, models/utils.py:3-14).
 * drop_p > 0 (bf16, Lk <= 64 only): dropout on the attention probabilities (TransformerModel.py:1430-1431),
 * keep(b, h, q, k) = hash(drop_seed + *drop_step, ((b*H + h)*Lq + q)*Lk + k) >= drop_p * 2^32, kept ones / (1 - drop_p).
 * q_start / q_count (int32 [B] each, or both NULL): unpadded rows -- batch item b owns the q_count[b] query rows starting at
 * row q_start[b]; with k_ragged its keys / values are laid out the same way (self-attention), else (b / kdiv) * Lk as usual;
 * q_rows > 0 (bf16 kernels, the MFMA backward): total rows of the q / out buffers -- the rows behind the last item, which no item
 * owns, are written as zeros (0: left untouched);
 * klen is then indexed by the global query row; Lq / Lk are the maxima.  Same three arguments on the two backward entry points. */
int bofi_attention_ex(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                      int ldo, int dtype, int B, int H, int Lq, int Lk, int kdiv, const int* klen,
                      int klen_sb, int klen_sq, int klen_bias, float drop_p, uint64_t drop_seed,
                      const uint64_t* drop_step, const int* q_start, const int* q_count, int k_ragged, int q_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training ops (float32).  The reference trains by running torch autograd over
 * TransformerModel._forward (tools/train.py:212-227); these are the hand-written backward (and the
 * few extra forward) kernels the XE step needs.  Parameter gradients that are sums over rows are
 * ACCUMULATED into the given buffers (zero them first).
 * ------------------------------------------------------------------------------------------- */
/* backward of bofi_layernorm: dx [rows, d] (+ add [rows, d] when given: the gradient that reaches x along the residual
 * connection, so that the sum of the two branches costs no extra pass); dgain, dbias [d] accumulated */
int bofi_layernorm_bwd(const float* x, const float* gain, const float* dy, const float* add, float* dx, float* dgain,
                       float* dbias, int rows, int d, void* stream);
/* the same, additionally writing dz_bf16 [rows, d] (d = 512 or 128; may be NULL) = bf16(keep(dx) / (1 - drop_p)) with the
 * dropout mask of bofi_linear_ex(drop_p, drop_seed, drop_step) over [rows, d]: when x = residual + dropout(h W^T + b) came out of
 * that linear's epilogue, dz is the gradient its backward GEMMs read, and the separate mask-and-cast pass is not needed */
int bofi_layernorm_bwd_ex(const float* x, const float* gain, const float* dy, const float* add, float* dx, float* dgain,
                          float* dbias, int rows, int d, void* dz_bf16, float drop_p, uint64_t drop_seed,
                          const uint64_t* drop_step, void* stream);
/* backward of bofi_attention_ex (Lq, Lk <= 64): dq like q, dk/dv like k/v (accumulated when kdiv > 1); q_start / q_count /
 * k_ragged as in bofi_attention_ex */
int bofi_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                       const float* dout, int ldo, float* dq, float* dk, float* dv, int B, int H, int Lq,
                       int Lk, int kdiv, const int* klen, int klen_sb, int klen_sq, int klen_bias,
                       const int* q_start, const int* q_count, int k_ragged, void* stream);
/* the same backward on the matrix cores: q/k/v float32 or bf16 (in_dtype), products in bf16 with fp32 accumulation,
 * operands gathered with transposing LDS reads; dq row stride lddq, dk/dv row stride lddk, float32 or bf16 (dq_dtype /
 * dkv_dtype: bf16 when the gradient feeds a GEMM directly), WRITTEN (not accumulated) also when kdiv > 1: one workgroup owns
 * an (image, head) and sums over its captions in registers / LDS; Lq <= 64 unless q_start is given with k_ragged == 0 (then
 * the key owner's rows are one contiguous run walked in chunks and Lq is only an upper bound);
 * drop_*: the dropout of the forward's attention probabilities, regenerated from the same (seed, index) */
int bofi_attention_bwd_mfma(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int in_dtype,
                            const float* dout, int ldo, void* dq, int lddq, void* dk, void* dv, int lddk, int dq_dtype,
                            int dkv_dtype, int B, int H, int Lq, int Lk, int kdiv, const int* klen, int klen_sb, int klen_sq,
                            int klen_bias, float drop_p, uint64_t drop_seed, const uint64_t* drop_step, const int* q_start,
                            const int* q_count, int k_ragged, int q_rows, void* stream);
/* backward of log_softmax given the log-probabilities y: dx = dy - exp(y) * rowsum(dy) */
int bofi_logsoftmax_bwd(const float* y, const float* dy, float* dx, int rows, int V, void* stream);
/* The loader's phrase-aware collate (captioning/data/dataloader.py:343-428) of SAMPLED captions, semi-autoregressive half, on the device: seq int64 [N, S] (tokens),
 * phrase_length int32 / phrase_syn int64 [N, S] (slot layout, real phrases first) -> per position the decoder input token (sa_seq: the previous phrase squeezed or
 * stretched over this one, BOS before the first), the phrase's label (sa_syn) and the number of visible keys (sa_klen).  The self-critical step's per-phrase forwards
 * read these (loss_wrapper.py:193-209 behind TransformerModel.py:1903-1984); no host round trip, capturable. */
/* The self-critical step's bookkeeping behind a draw: positions of phrases [p0, p1) of every caption (phrase_length int32 [N, S]) take tok int64 [N, S] into seq, the
 * token's log-prob lp[n, t, tok] (float32 [N, S, V]) into drawn, and set mask (bool [N, S], or NULL); the other positions are left alone. */
int bofi_rl_take_draws(const float* lp, const int64_t* tok, const int* phrase_length, int N, int S, int V, int p0, int p1, int64_t* seq, float* drawn, void* mask,
                       void* stream);
int bofi_saic_collate(const int64_t* seq, const int* phrase_length, const int64_t* phrase_syn, int N, int S, int bos_idx, int64_t* sa_syn, int64_t* sa_seq,
                      int* sa_klen, void* stream);
/* backward of picked[r] = y[r][labels[r]] straight through the log_softmax that made y (the token terms of
 * LanguageModelCriterion_UIC, losses.py:341-347: gather on the log-probs): dx[r][c] = dpicked[r] * ((c == labels[r]) - exp(y[r][c])).
 * labels must lie in [0, V).  dx: float32 [rows, V] (lddx == V), or bf16 [rows, lddx] with lddx >= V and the columns V..lddx-1
 * written as zeros (the operand the vocabulary projection's backward GEMMs read, K granule 64). */
int bofi_nll_bwd(const float* y, const int64_t* labels, const float* dpicked, void* dx, int dx_dtype, int lddx, int rows, int V,
                 void* stream);

/* LanguageModelCriterion_UIC (captioning/modules/losses.py:319-369, reduction 'mean') in one launch, for the paired training
 * forward: the four slot outputs [N, Pm, c_len | c_syn] (log-probs), the loader's phrase_num [N], phrase_length / phrase_syn
 * [N, L] (int64; slot p is counted iff p < phrase_num[n], its labels sit at column p + 1), the picked token log-probs [T] with
 * the SA / NA row weights.  out8 = {sa_len, sa_tok, sa_syn, na_len, na_tok, na_syn parts, their sum (the loss), sum(w_sa)}.
 * The backward takes dL/d(loss) (device scalar) and writes dense gradients of the four slot outputs and d_picked [T]. */
int bofi_uic_criterion(const float* sa_len, const float* sa_syn, const float* na_len, const float* na_syn, int N, int Pm, int c_len,
                       int c_syn, const int64_t* phrase_num, const int64_t* phrase_length, const int64_t* phrase_syn, int L,
                       const float* picked, const float* w_sa, const float* w_na, int T, float* out8, void* stream);
int bofi_uic_criterion_bwd(const float* sa_len, const float* sa_syn, const float* na_len, const float* na_syn, int N, int Pm,
                           int c_len, int c_syn, const int64_t* phrase_num, const int64_t* phrase_length,
                           const int64_t* phrase_syn, int L, const float* picked, const float* w_sa, const float* w_na, int T,
                           const float* g_loss, const float* out8, float* d_sa_len, float* d_sa_syn, float* d_na_len,
                           float* d_na_syn, float* d_picked, void* stream);
/* out[n] += sum_m x[m][n]  (bias gradients) */
int bofi_colsum_add(const float* x, float* out, int M, int N, void* stream);
/* x[r] = sqrt(d) * (lut_tok[tok[r]] + lut_syn[syn[r]]) + pe[pos ? pos[r] : r % L]; tok or syn may be NULL, a negative id
 * leaves that term out for the row (also in bofi_embed_bwd)
 * (Embeddings + PositionalEncoding, TransformerModel.py:1484-1511) and its backward into one table */
int bofi_embed_rows(const float* lut_tok, const float* lut_syn, const float* pe, const int64_t* tok,
                    const int64_t* syn, const int64_t* pos, int rows, int L, int d, float* x, void* stream);
int bofi_embed_bwd(const float* dx, const int64_t* ids, float* dlut, int rows, int d, float scale, void* stream);
/* Weight-gradient GEMM: c[i][j] += sum_m a[m][i] * b[m][j], i < NI, j < NJ (dW = dz^T x without transposed copies).
 * a, b: bf16 row-major [M, a_cols | b_cols] (cols a multiple of 8, zero beyond NI | NJ; 16-byte aligned rows);
 * c: float32 [NI, NJ] row stride ldc, ACCUMULATED into with atomics (zero it for a plain product).
 * colsum (may be NULL): colsum[i] += sum_m a[m][i], taken from the A tiles while they sit in LDS (dz's column sums = the
 * bias gradient; a few atomics per workgroup instead of a reduction pass of its own). */
int bofi_gemm_tn_acc(const void* a, int lda, int a_cols, const void* b, int ldb, int b_cols, float* c, int ldc, int M,
                     int NI, int NJ, float* colsum, void* stream);

/* n weight-gradient GEMMs (each as bofi_gemm_tn_acc: c[e] += a[e]^T b[e], colsum[e] += column sums of a[e]; colsum or its
 * entries may be NULL) in as few launches as the kernel-argument space allows (40 problems each).  All arrays are HOST arrays
 * of n entries; the problems are passed to the kernel by value.  The training step defers the weight gradients of all its
 * Linear layers (nn.Linear's grad_weight, one addmm each in the reference) to one such call after backward. */
int bofi_gemm_tn_grouped(int n, const void* const* a, const int* lda, const int* a_cols, const void* const* b, const int* ldb,
                         const int* b_cols, float* const* c, const int* ldc, const int* M, const int* NI, const int* NJ,
                         float* const* colsum, void* stream);
/* xt[n][m] = x[m][n], zero for M <= m < Mpad, written as out_dtype: operand layout of the weight-gradient GEMM.
 * colsum (may be NULL): colsum[n] += sum_m x[m][n], the bias gradient, taken from the tiles while they are in LDS */
int bofi_transpose_pad(const float* x, int ldx, void* xt, int out_dtype, int M, int N, int Mpad, float* colsum, void* stream);

/* All weight operands of the backward's input-gradient GEMMs in one launch.  table[e] = {src_off, dst_off, N, K, Np} (int64,
 * element offsets): the bf16 matrix [N, K] at src + src_off is written transposed as [K, Np] at dst + dst_off (pad columns
 * N..Np-1 untouched).  tile_first[e] = number of 64 x 64 tiles of the entries before e (tile_first[n] = total_tiles).
 * Replaces one bofi_transpose_pad per weight and step; same role as the .t() views autograd takes of nn.Linear weights. */
int bofi_transpose_many(const void* src_bf16, void* dst_bf16, const int64_t* table, const int* tile_first, int n,
                        int total_tiles, void* stream);
/* y[m][n] = bf16(x[m][n]) for n < N, zero for N <= n < ldy: a GEMM operand in the bf16 compute dtype.
 * relu_y (may be NULL): forward output of a ReLU layer, same layout as x: x is masked where relu_y <= 0 first.
 * drop_p > 0: then the dropout mask of bofi_linear_ex(drop_p, drop_seed, drop_step) is applied (x' = keep(x) / (1 - p)).
 * colsum (may be NULL): colsum[n] += sum_m x'[m][n] on the way (the bias gradient when x is dz) */
int bofi_cast_bf16(const float* x, int ldx, void* y, int ldy, int M, int N, float* colsum, const float* relu_y, float drop_p,
                   uint64_t drop_seed, const uint64_t* drop_step, void* stream);
/* bofi_linear with every epilogue option of the training path.  Dropout (drop_p 0: off):
 * y = residual + keep(act(x w^T + bias)) / (1 - drop_p), keep(m, n) = hash(seed, m * N + n) >= drop_p * 2^32 with
 * seed = drop_seed + (drop_step ? *drop_step : 0) -- drop_step is a DEVICE word, so a captured hipGraph draws fresh masks
 * on every replay (SublayerConnection x + dropout(sublayer(norm(x))), TransformerModel.py:1361-1363; FFN dropout :1478);
 * y2 (may be NULL): a second copy of the output in the compute dtype, row stride ldy2 (N % 32 == 0) -- the operand of
 * the next GEMM / attention kernel, so that no separate cast pass is needed. */
int bofi_linear_ex(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* bias,
                   const float* residual, int ldr, void* y, int y_dtype, int ldy, int M, int N, int K, int relu,
                   const int* row_len, int rows_per_group, float drop_p, uint64_t drop_seed, const uint64_t* drop_step, void* y2,
                   int ldy2, void* stream);

/* The fused decode-path linear on bf16 operands (nn.Linear behind a pre-norm SublayerConnection, TransformerModel.py:1337-1363):
 * ln_stats != NULL: x is the RAW residual stream in bf16 and the LayerNorm is folded in -- w must be W * gain, bias = W . beta + b,
 * ln_colsum[n] = sum_k w[n][k], ln_stats = [M][ln_groups][2] partial (sum, sum of squares) pairs of the float32 rows (ln_groups 0: K / 32),
 * y = rstd[m] * (x w^T - mean[m] * colsum) + bias;  then ReLU, then + residual (float32, may alias y);  y2 (may be NULL): a bf16 copy,
 * stats_out (may be NULL): [M][N / 32][2] partial sums of the OUTPUT rows for the next folded LayerNorm.
 * Large shapes (>= 200 tiles of 256 x 128, N % 128 == 0) run as persistent workgroups (gemm_pers.hip), the others one tile per
 * workgroup (gemm_glds.hip): bit-identical results either way. */
int bofi_linear_fused(const void* x, int ldx, const void* w, const float* bias, const float* residual, int ldr, void* y, int y_dtype, int ldy,
                      void* y2, int ldy2, const float* ln_stats, const float* ln_colsum, int ln_groups, float* stats_out, int M, int N, int K,
                      int relu, void* stream);

/* Developer knobs read from the environment (BOFI_GEMM_PERS, BOFI_GEMM_PERS_MIN) are cached at first use: after changing them in a
 * running process call this to have them read again. */
void bofi_reload_env(void);

/* y = mask > 0 ? (x w^T) * scale : 0 (no bias): the input gradient of a linear whose INPUT came out of relu (+ dropout with
 * scale = 1 / (1 - p)) -- mask [M, N] float32 is that forward activation (a clipped or dropped unit is 0 there), so the
 * gradient arrives already masked, e.g. in bf16 for the previous layer's backward GEMMs (nn.Linear backward followed by the
 * relu / dropout backward in the reference).  Operands in the compute dtype, N % 4 == 0. */
/* bofi_linear over a ROW LIST: rows row_idx[0 .. *n_rows) of x (both in device memory; M = capacity the launch is sized for) are
 * multiplied and written to the same rows of y (residual read at those rows); other rows of y stay untouched.  The count is read on the
 * device, so a captured launch serves any list (the semi-autoregressive decode's per-iteration rows).  x and w in one dtype, K a
 * multiple of the 128-byte slab. */
int bofi_linear_rows(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* bias, const float* residual, int ldr,
                     void* y, int y_dtype, int ldy, int M, int N, int K, int relu, const int* row_idx, const int* n_rows, void* stream);
int bofi_linear_masked(const void* x, int x_dtype, int ldx, const void* w, int w_dtype, const float* mask, int ldm,
                       float scale, void* y, int y_dtype, int ldy, int M, int N, int K, void* stream);

/* y = (residual or 0) + keep(x) / (1 - p), keep mask = hash(seed, index) (nn.Dropout of the sublayers,
 * TransformerModel.py:1361-1363; the backward is the same call on dy with the same seed) */
int bofi_dropout(const float* x, const float* residual, float* y, int64_t n, float p, uint64_t seed, const uint64_t* drop_step,
                 void* stream);
/* One optimiser step over a flat float32 bucket (n % 4 == 0, 16-byte aligned): g' = clamp(g * grad_scale, +-clip_value)
 * (clip_value <= 0: no clamp; train.py:225-226), then torch.optim.Adam's update with bias correction for `step` >= 1
 * (misc.py:245-251).  shadow_bf16 (may be NULL) receives a bf16 copy of the new parameters. */
int bofi_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                   float beta2, float eps, int step, float clip_value, float grad_scale, void* stream);
/* dx = dy where y > 0 */
int bofi_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);

/* Per row sum_v exp(logp_v) * logp_v and logp[row, seq[row]]: the reductions behind eval's per-image entropy and
 * perplexity (captioning/utils/eval_utils.py:463-464) without reading the distribution back in user code. */
int bofi_vocab_stats(const float* logp, const int64_t* seq, int rows, int V, float* row_plogp, float* row_chosen, void* stream);
/* n draws per row from Categorical(logits = logp / temperature) (sample_next_word, CaptionModel.py:405-425; NaN counts as
 * -10): Gumbel-max with a counter-hash uniform.  out int64 [(rows / S) * n, S], draw c of image b in row b * n + c
 * (the reference repeats each image n times, models/utils.py:3-14); positions >= ntok[b] get pad_idx (AttModel.py:422-423). */
int bofi_vocab_sample(const float* logp, int rows, int V, int S, int n, float temperature, uint64_t seed, const int* ntok,
                      int pad_idx, int64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Engine level: the whole NAIC bound+fill decode as one call.
 * ------------------------------------------------------------------------------------------- */
typedef struct bofi_engine bofi_engine_t;

typedef struct bofi_config {
    int vocab;        /* tgt_vocab = vocab_size + 4 (AttModel.py:79) */
    int feat;         /* att_feat_size (2048) */
    int d_model;      /* 512; d_model / heads must be 64 */
    int d_ff;         /* 2048 */
    int heads;        /* 8 */
    int n_enc;        /* 6 */
    int n_dec;        /* 6 */
    int seq_length;   /* 20 (S); the bound sequence has S+2 positions */
    int pad_idx, bos_idx, eos_idx, len_idx;   /* 0 1 2 3 */
    int head_hidden;  /* 100: width of Length/Syntactic_classifier1 */
    int max_batch;    /* workspace is sized for this many images per call */
    int max_regions;  /* and this many regions per image (<= 128) */
    int dtype;        /* compute dtype: BOFI_DT_F32 (parity) or BOFI_DT_BF16 (throughput) */
    int n_len;        /* layers of the bounding network (LengthPredictor_UIC N_len, TransformerModel.py:357-375): 1 (configs/uic_sd.yml) takes
                         the row-0-only incremental form; >= 2 (configs/uic_sd_N2.yml) the dense form -- all S+2 rows through every layer
                         per iteration, as the reference computes it (a field since ABI version 2; the library is at version 3: bofi_abi_version) */
} bofi_config_t;

/* Allocates device weights + workspace for one model replica on the current HIP device. */
int bofi_engine_create(const bofi_config_t* cfg, bofi_engine_t** out);
void bofi_engine_destroy(bofi_engine_t* e);

/* Hands the engine one state_dict entry by its reference name (the 311-key schema of
 * TransformerModel.state_dict()), float32 on the HOST.  The engine keeps a copy. */
int bofi_engine_set_weight(bofi_engine_t* e, const char* name, const float* data_host, int64_t numel);

/* Packs/convert the weights for the kernels (fused QKV, stacked cross-attention K/V, bf16 copies)
 * and precomputes the input-independent tables of the bound network.  Must follow the last
 * set_weight and precede decode; may be called again after weights change.  Synchronises. */
int bofi_engine_finalize(bofi_engine_t* e);

/* A second engine that SHARES the parent's packed weights and has its own workspace, so that several
 * decodes can be in flight on different streams (the bounding loop is a latency chain that leaves most
 * CUs idle; a batch in another stream fills them).  The parent must outlive its forks and must not be
 * re-finalized while they exist.  Destroy a fork with bofi_engine_destroy. */
int bofi_engine_fork(bofi_engine_t* parent, bofi_engine_t** out);
/* The same with a workspace for up to max_batch images per call (0 = the parent's): dynamic batching puts several loader batches into one decode call
 * (bofi_engine_set_q1_group), which a model built for one batch per call has no workspace for -- its weights serve either size. */
int bofi_engine_fork_sized(bofi_engine_t* parent, int max_batch, bofi_engine_t** out);

/* Re-pack the engine's weights from float32 parameters that live on the DEVICE (reference names, as set_weight): the
 * packing of bofi_engine_finalize (q|k|v stacking, LayerNorm folding, compute-dtype cast, bound tables) as kernels on
 * `stream`, into the existing allocations -- captured graphs and forks stay valid.  For a training loop that decodes
 * with its current weights every step (the self-critical phase, periodic evaluation).  Requires a finalized engine. */
int bofi_engine_refresh_device(bofi_engine_t* e, int n, const char* const* names, const float* const* ptrs, const int64_t* numels,
                               void* stream);

/* Several independent batches in ONE decode call (dynamic batching for serving): with group > 0 the B images of a call are
 * consecutive batches of `group` images and quirk Q1 (BOFI_FLAG_STRICT_Q1) applies inside each batch, so every batch's result
 * equals its own separate decode.  0 (default): the whole call is one batch. */
int bofi_engine_set_q1_group(bofi_engine_t* e, int group);

/* A hint, not a semantic: how many decodes the caller keeps in flight on this device (engine + forks on their own streams).  1: this engine's
 * launches run with nothing beside them -- from 4 096 rows on the sublayer kernels then take 64-row blocks (more, shorter workgroups: 62 against 68 us
 * for the encoder's feed-forward sublayer at 11 520 rows) where the default, n = 0 (unknown) or n > 1, takes the 80- / 96-row blocks that move fewer weight
 * bytes per row (+3.5 % images/s with four launches in flight).  Results under either choice agree as bf16 kernels of different summation order do. */
int bofi_engine_set_decodes_in_flight(bofi_engine_t* e, int n);

/* 1 when a bofi_engine_decode_naic of R regions per image would run core_NAIC's bounding loop (TransformerModel.py:1833-1869) as the persistent
 * kernel of round 5 -- one workgroup per 16 images runs every iteration (row-0 self-attention, cross-attention, feed-forward, heads, slot bookkeeping)
 * and leaves when its images are finished; fp16 operands derived from the float32 parameters -- under the engine's current decodes-in-flight hint:
 * bf16 engine at the reference's width (d_model 512, 8 heads, d_ff <= 2048 a multiple of 512, seq_length <= 22, N_len = 1), R <= 128, and the
 * hint is not 1 (a decode that runs alone keeps the five launches per iteration: shorter latency); environment knob BOFI_BOUND_LOOP: 0 = never,
 * 2 = whatever the hint (read again by bofi_reload_env).  The bound-iteration cap (bofi_engine_set_bound_iter_cap) does not apply to that kernel.  0 otherwise. */
int bofi_engine_bound_loop_active(bofi_engine_t* e, int R);

/* Temperature (> 0) and seed of the token draws of a decode called with BOFI_FLAG_SAMPLE. */
int bofi_engine_set_sampling(bofi_engine_t* e, float temperature, uint64_t seed);
/* The iterations the following bofi_engine_decode_saic calls enqueue: it_begin .. it_end of core_SAIC's loop (TransformerModel.py:1903-1984; 1-based,
 * it_end 0 = seq_length; the default is the whole loop, 1 .. seq_length).  A call with it_begin == 1 encodes and initialises; one with
 * it_begin > 1 continues the PREVIOUS call's loop on the state left in the engine's workspace (same arguments, no other decode in between) and
 * exports again: [1, c] followed by [c + 1, seq_length] is the same computation as the whole loop.  Iterations past the last live one return
 * at once but still cost their launches; *bound_iters (live iterations so far) < c after [1, c] means the loop has ended. */
int bofi_engine_set_saic_range(bofi_engine_t* e, int it_begin, int it_end);
/* Between two partial bofi_engine_decode_saic calls: the tokens emitted so far (positions before the end of the last placed phrase) are replaced by
 * seq int64 [B, seq_length] (device memory, the layout of the exported seq), in the emitted caption and in the bounding step's input
 * (extend_phrase_len, TransformerModel.py:1959-1966): the loop continues on the caller's words.  For a caller that draws each phrase itself over the
 * engine's layout -- the self-critical step's reference-estimator mode draws from the TRAINING forward's dropout-perturbed distribution
 * (loss_wrapper.py:193-209 samples in train mode), boficap_amd/trainer.py. */
int bofi_engine_saic_put_words(bofi_engine_t* e, const int64_t* seq, int B, void* stream);
/* Bounding iterations the following bofi_engine_decode_naic calls enqueue (core_NAIC's loop, TransformerModel.py:1843-1869; 0 = all seq_length,
 * the default).  The loop exits when every image is finished; enqueued iterations past that point return at once but still cost their five
 * launches.  With a cap c the decode is the reference's if and only if the loop ended within c iterations: *bound_iters (iterations in which
 * some image was live) < c.  Otherwise decode again with cap 0. */
int bofi_engine_set_bound_iter_cap(bofi_engine_t* e, int cap);
/* With a pipeline of capped decodes whose results are consumed later: `live_max` (device int32, NULL = off) receives, by atomic max, the
 * live-iteration count of every following bofi_engine_decode_naic -- one read after the last decode tells whether ALL of them ended inside the
 * cap (the caller clears the word first). */
int bofi_engine_set_live_iterations_max(bofi_engine_t* e, int* live_max);

/* fp16 saturation status of the persistent bounding-loop kernel (round 6; ABI version 4).  That kernel computes the bounding network with FP16 operands
 * (fp16 copies of the float32 parameters, activations converted on their way into the MFMAs) and clamps to +-65 504 on conversion; a model whose bounding
 * layer leaves that range -- none here does: |y| < 30 -- would get a silently different slot layout where the bf16 kernels (same exponent range as float32) would not.
 * `word` (device int32, NULL = off -- the default, also of a fork): every following bofi_engine_decode_naic WRITES its status there (beside bound_iters): 0 = nothing was
 * clamped (always 0 when the decode ran the five-launch bf16 iterations), bit 0 = an activation (attention context or hidden row of the bounding layer) was clamped
 * during this decode, bit 1 = the fp16 weight copies were clamped when they were packed (finalize / refresh_device), bit 2 = the kernel's pair exchange timed out (two workgroups share a group's
 * feed-forward weights: a safety net against a lost workgroup, never observed).  A caller that reads a non-zero word decodes that
 * batch again under bofi_engine_set_bound_loop(e, 0) (boficap_amd/engine.py does, with a warning).  Per-call state, part of the graph key.
 * Replaces nothing in the reference (TransformerModel.py:357-383 runs in float32 there); it guards this library's own precision choice. */
int bofi_engine_set_saturation_out(bofi_engine_t* e, int* word);
/* Per-engine choice of the bounding loop's form for the following decodes: -1 (default, also of a fork) = BOFI_BOUND_LOOP and the decodes-in-flight hint decide
 * (bofi_engine_bound_loop_active), 0 = the five-launch bf16 iterations, 2 = the persistent fp16 loop kernel whatever the hint.  Part of the graph key. */
int bofi_engine_set_bound_loop(bofi_engine_t* e, int mode);

/* Device pointer of the engine's own [max_batch * S, V] float32 log-prob workspace: where a decode called WITHOUT a
 * seq_logprob buffer leaves the distribution (valid until the next decode on this engine) -- input of
 * bofi_vocab_stats / bofi_vocab_sample when the 48.6 MB tensor is not wanted in user memory. */
const float* bofi_engine_logprob(bofi_engine_t* e);
/* The same two per-position figures straight out of a NAIC decode's vocabulary epilogue (no second pass over the [B, S, V] tensor): float32 [B * S] device buffers
 * that every following bofi_engine_decode_naic of this engine fills (greedy decodes with log-softmax; NULL, NULL: off -- the default, also of a fork).  Per-call state,
 * part of the graph key.  sum_v p log p is formed as sum_v e_v (x_v - max) / sum_v e_v - lse in the pass that sums the exponentials. */
int bofi_engine_set_row_stats_out(bofi_engine_t* e, float* row_plogp, float* row_chosen);

/* A HIP stream owned by the engine (hipStream_t as void*), for callers that keep one decode per engine in
 * flight and want each on its own hardware queue. */
void* bofi_engine_stream(bofi_engine_t* e);

#define BOFI_FLAG_STRICT_Q1 1      /* reproduce TransformerModel.py:1872-1873: every image's fill mask
                                      uses the LAST image's length.  Default ON in the Python wrapper. */
#define BOFI_FLAG_RAW_LOGITS 2     /* output_logsoftmax = 0 (AttModel.py:208-209) */
#define BOFI_FLAG_GRAPH 4          /* replay the call from a captured hipGraph when possible */
#define BOFI_FLAG_SAMPLE 16         /* decode_saic: draw each phrase's tokens from Categorical(logits / temperature) instead of
                                      the argmax (sample_method 'sample'); parameters from bofi_engine_set_sampling */
#define BOFI_FLAG_REFINE_SHIFT 8   /* bits 8..11: extra filling rounds; round r > 0 feeds round r-1's ids back as the
                                      decoder input tokens (decode_NA's glat_input, TransformerModel.py:570-574).  The
                                      reference has no refinement loop: parity of rounds > 0 is pinned to the oracle only. */
/* bofi_engine_decode_naic in PHASES (none of the three bits = the whole decode).  A caller that pipelines decodes enqueues the phases of one decode
 * as separate calls on the same engine -- encode, then bound, then fill, ordered by the caller's streams / events -- so that the bounding loop (a few
 * workgroups for ~1 ms) can run on a side stream while the launch stream already encodes the next batch on another fork: */
#define BOFI_FLAG_PHASE_ENCODE 32   /* _prepare_feature + Encoder + the stacked cross K|V (TransformerModel.py:1674-1690, 1332-1336) */
#define BOFI_FLAG_PHASE_BOUND 64    /* the bounding loop of core_NAIC (:1833-1869) on the preceding encode of this engine */
#define BOFI_FLAG_SAIC_LAYOUT_ONLY 4096 /* decode_saic: every enqueued iteration lays its phrase out (bounding step + bookkeeping, TransformerModel.py:1903-1948) and stops there --
                                          no decoder pass, no tokens: the caller draws the phrase's words from its own distribution and hands them back with
                                          bofi_engine_saic_put_words before the next call (the reference-estimator self-critical step) */
#define BOFI_FLAG_PHASE_FILL 128    /* decode_NA + logit + greedy pick + the slot-state export (:570-587, 1872-1876) on the preceding two */

/* model(fc, att, att_masks, opt={'train_mode':'NAIC','sample_method':'greedy'}, mode='sample'):
 * AttModel._sample AttModel.py:307-338,419-429 -> _prepare_feature TransformerModel.py:1674-1690
 * -> Encoder :1332-1336 -> core_NAIC :1823-1876 (bound loop + decode_NA :570-587) -> logit/
 * log_softmax AttModel.py:203-210 -> greedy pick CaptionModel.py:388-390 -> pad tail.
 *   att_feats: feats_dtype [B, R, feat] contiguous;  att_len: int32 [B] regions per image or NULL
 *   seq: int64 [B, S];  seq_logprob: float32 [B, S, vocab] or NULL (then only ids are produced);
 *   phrase_num: int32 [B];  phrase_length: int32 [B, S];  phrase_syn: int64 [B, S];
 *   memory_out: float32 [B, R, d_model] or NULL (encoder output, for _prepare_feature parity);
 *   bound_iters: int32 [1] or NULL (number of bound iterations in which some image was active). */
int bofi_engine_decode_naic(bofi_engine_t* e, const void* att_feats, int feats_dtype, const int* att_len,
                            int B, int R, int flags, int64_t* seq, float* seq_logprob, int* phrase_num,
                            int* phrase_length, int64_t* phrase_syn, float* memory_out, int* bound_iters,
                            void* stream);

/* model(..., opt={'train_mode':'SAIC','sample_method':'greedy'}, mode='sample'): the semi-autoregressive decode
 * core_SAIC TransformerModel.py:1878-1986 (AttModel.py:430-437) -- per phrase one bounding step on the words
 * emitted so far (:1907-1926), the position-wise copy of the previous phrase (:1928-1948), a full decoder
 * pass (decode_SA :520-530) and the copy of the new phrase's tokens and log-probs (:1968-1977); stops
 * when every image is finished or a NaN appears (:1956-1958).  Outputs as for decode_naic
 * (seq_logprob rows never written stay 0, as in the reference).  From the second phrase on only the new phrase's rows go
 * through the decoder's GEMMs (their K / V join a per-layer cache; results identical to the all-rows form).  flags:
 * BOFI_FLAG_RAW_LOGITS, BOFI_FLAG_SAMPLE (bofi_engine_set_sampling; the seed lives in device memory, so a replayed graph
 * draws anew), BOFI_FLAG_GRAPH (captured once per argument set, as decode_naic). */
int bofi_engine_decode_saic(bofi_engine_t* e, const void* att_feats, int feats_dtype, const int* att_len,
                            int B, int R, int flags, int64_t* seq, float* seq_logprob, int* phrase_num,
                            int* phrase_length, int64_t* phrase_syn, int* bound_iters, void* stream);

/* Stages of the call above, exposed for module-level parity tests (SURVEY.md §4 pyramid level 2). */
int bofi_engine_encode(bofi_engine_t* e, const void* att_feats, int feats_dtype, const int* att_len,
                       int B, int R, float* memory_out, void* stream);
/* One bound step from a given slot layout: ext_syn int32 [B, S+2], last int32 [B] ->
 * len_logp float32 [B, 20], syn_logp float32 [B, 10] (LengthPredictor_UIC.forward
 * TransformerModel.py:357-383 on the memory of the preceding bofi_engine_encode). */
int bofi_engine_bound_step(bofi_engine_t* e, const int* ext_syn, const int* last, int B, int R,
                           const int* att_len, float* len_logp, float* syn_logp, void* stream);

/* The row-0 stages of a bounding iteration as operators (bound_ops.hip; bf16, d_model 512, 8 heads): one activation row per image.
 * bofi_rowgemm: y[M, N] = epilogue(x[M, K] . w[N, K]^T), K = 512 * splitk.  stats (may be NULL): partial (sum, sum of squares) of the
 * float32 rows behind x, [M][stats_groups][2] -> the LayerNorm folded into w / bias / colsum (y = rstd (acc - mean colsum) + bias);
 * relu; residual float32 [M, ldr]; outputs: y float32 (splitk > 1: `splitk` partial slabs [splitk][M][ldy], bias and residual in
 * slab 0), yb bf16 copy, stats_out [M][N/16][2] partial sums of the output rows.  Without split-K any K = 512 * c is walked in chunks.
 * skip (may be NULL): the launch returns at once when *skip >= skip_threshold.  row_idx / n_rows (both or neither; device memory): the
 * GEMM covers rows row_idx[0 .. *n_rows) of x (and of residual / stats) and writes the same rows of the outputs; M is then the capacity.
 * bofi_bound_qattn: out[B, 512] = per head softmax(q k^T / 8) v with q = LayerNorm-folded Wq x (as above, stats [B][16][2]), keys and
 * values k, v [B * R, ldkv] (R <= 64 regions per image, att_len int32 [B] or NULL = R each).  row_idx / n_rows (both or neither): the query
 * rows are rows row_idx[0 .. *n_rows) of x / stats / out (B = capacity); with rows_per_image > 0 row r attends the regions of image
 * r / rows_per_image (a decoder layer's cross-attention over the rows of a list). */
int bofi_rowgemm(const void* x, int ldx, const void* w, const float* bias, const float* stats, int stats_groups, const float* colsum,
                 const float* residual, int ldr, float* y, int ldy, void* yb, int ldyb, float* stats_out, int M, int N, int K, int splitk,
                 int relu, const int* skip, int skip_threshold, const int* row_idx, const int* n_rows, void* stream);
int bofi_bound_qattn(const void* x, const float* stats, const void* wq, const float* bias, const float* colsum, const void* k, const void* v,
                     int ldkv, const int* att_len, void* out, int B, int R, const int* skip, int skip_threshold, const int* row_idx,
                     const int* n_rows, int rows_per_image, void* stream);

/* Row-block sublayer kernels (rowblock.hip; bf16 operands, d_model 512): a workgroup keeps a block of activation rows in LDS for a
 * whole sublayer and streams the weights from L2 straight into MFMA operand registers.
 * bofi_pack_frag: w [N, K] row-major bf16 -> the fragment-major layout those kernels stream
 *   ([N/64][K/32][4 tiles][64 lanes][8 bf16]; N % 64 == 0, K % 32 == 0); out holds N*K bf16.
 * bofi_ffn_block: y = x + w_2 relu(w_1 LN(x) + b_1) + b_2 -- PositionwiseFeedForward behind SublayerConnection
 *   (TransformerModel.py:1477-1478, 1361-1377), one launch.  x, y float32 [M, 512] (y may be x); w1p = bofi_pack_frag of the
 *   [dff, 512] weight with the norm's gain folded in, c1 = b_1 + w_1 b_ln, cs1 = column sums of the rounded folded weight
 *   (float32 [dff]: the fold of gemm_glds.hip); w2p = bofi_pack_frag of the [512, dff] weight, b2 float32 [512]; dff % 512 == 0,
 *   dff <= 2560.  Optional outputs (NULL to skip): yb bf16 [M, 512], stats_out float32 [M][16][2] partial (sum, sum of squares)
 *   per 32 columns of y (what a LayerNorm-folded consumer GEMM reads).
 * bofi_attn_block: y = x + W_o concat_h softmax(q_h k_h^T / 8) v_h + b_o -- MultiHeadedAttention (TransformerModel.py:1421-1432,
 *   1454-1467) behind the sublayer's residual, one launch: 8 heads of 64, q [B*Lq, ldq], k / v [B*Lk, ldk / ldv] bf16 (head h at
 *   columns h*64), Lq, Lk <= 48.  Key-prefix masks as bofi_attention: klen int32 (NULL: all Lk keys), entry
 *   klen[bi * klen_sb + q * klen_sq] + klen_bias for query row q of image b, bi = b, or with klen_shared_last = G > 0 the last
 *   image of b's group of G (quirk Q1, TransformerModel.py:1872-1873); a row without keys is NaN as in the reference.
 *   wop = bofi_pack_frag of the [512, 512] output projection, bo float32 [512]; x, y float32 [B*Lq, 512] (y may be x); yb /
 *   stats_out as bofi_ffn_block.
 * bofi_linear_block: y[M, N] = act(W' LN(x) + b) for a LayerNorm-folded projection with K = 512 (the q|k|v, cross-query, stacked cross K|V
 *   and generator projections: TransformerModel.py:1454-1456, AttModel.py:203-210 behind their pre-norms): x float32 [M, 512] -- the
 *   row statistics are computed in the kernel --, wp = bofi_pack_frag of the folded [N, 512] weight (N % 64 == 0), c / cs its folded
 *   bias and column sums, y bf16 or float32 (y_f32) [M, ldy], relu 0 / 1. */
int bofi_pack_frag(const void* w, void* out, int N, int K, void* stream);
int bofi_linear_block(const float* x, int ldx, const void* wp, const float* c, const float* cs, void* y, int ldy, int y_f32, int M, int N,
                      int relu, void* stream);
int bofi_attn_block(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int Lq, int Lk, const int* klen,
                    int klen_sb, int klen_sq, int klen_bias, int klen_shared_last, const void* wop, const float* bo, const float* x,
                    int ldx, float* y, int ldy, void* yb, float* stats_out, void* stream);
/* bofi_attn_linear_block: bofi_attn_block (in place, one key count per image, Lq <= 20, Lk <= 32: the filling pass's self-attention) followed by bofi_linear_block
 *   (N = 512) on its output, ONE launch -- the decoder layer's self-attention sublayer and the LayerNorm-folded query projection of its cross-attention
 *   (TransformerModel.py:1408-1413 around :1454-1456): x as bofi_attn_block's y, pj_y bf16 [B*Lq, pj_ldy] = W_pj' LN(x) + c_pj.  Each 80-row block (four images per
 *   workgroup) is projected while it sits in LDS; the projection differs from the two-launch form in the summation order of the row statistics. */
int bofi_attn_linear_block(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int Lq, int Lk, const int* klen, int klen_sb,
                           int klen_bias, int klen_shared_last, const void* wop, const float* bo, float* x, int ldx, const void* pj_wp, const float* pj_c,
                           const float* pj_cs, void* pj_y, int pj_ldy, void* stream);
/* (bofi_decoder_attn_block -- both attention sublayers of a decoder layer as one launch, round 5 -- lost 3 % with launches in flight and left the ABI in round 6:
 *   dev/exp/rb_dec_attn_kernel.inc, profiles/r05_dec_attn_fused.txt, last shipped at commit 45884c1.) */
int bofi_ffn_block(const float* x, int ldx, const void* w1p, const float* c1, const float* cs1, const void* w2p, const float* b2,
                   float* y, int ldy, void* yb, float* stats_out, int M, int dff, void* stream);
/* bofi_ffn_linear_block: bofi_ffn_block followed by bofi_linear_block on its output, ONE launch -- the feed-forward sublayer and the
 *   LayerNorm-folded projection that reads the stream next (the next layer's q|k|v, TransformerModel.py:1454-1456 behind :1361-1377; after the
 *   last encoder layer the stacked cross K|V): y as bofi_ffn_block, pj_y bf16 [M, pj_ldy] = W_pj' LN(y) + c_pj (pj_wp / pj_c / pj_cs as
 *   bofi_linear_block's wp / c / cs; pj_N % 64 == 0, pj_N >= 512, pj_ldy % 8 == 0).  Each 80-row block is projected while it is still in LDS. */
int bofi_ffn_linear_block(const float* x, int ldx, const void* w1p, const float* c1, const float* cs1, const void* w2p, const float* b2,
                          float* y, int ldy, int M, int dff, const void* pj_wp, const float* pj_c, const float* pj_cs, void* pj_y,
                          int pj_ldy, int pj_N, void* stream);
/* bofi_attn_out_ffn_block (round 6): the SECOND half of an attention sublayer -- output projection, bias, residual (MultiHeadedAttention.forward's last Linear
 *   TransformerModel.py:1467 behind SublayerConnection :1361-1363) -- as the HEAD segment of the feed-forward sublayer that follows it (:1477-1478; EncoderLayer :1374-1377,
 *   DecoderLayer :1408-1413), ONE launch:   x1 = x + W_o ctx + b_o;   y = x1 + w_2 relu(w_1 LN(x1) + b_1) + b_2   [; pj_y = W_pj' LN(y) + c_pj as bofi_ffn_linear_block]
 *   ctx: bf16 [M, ldc] -- the heads' attention outputs side by side, what bofi_attention writes (the attention CORE: no weights, a light kernel);  wop: W_o [512][512] in
 *   bofi_pack_frag's layout, bo float32 [512];  the rest as bofi_ffn_block / bofi_ffn_linear_block (pj_wp NULL: no projection tail).  x1 never exists in memory: the
 *   kernel's consumer wavefronts start their accumulators from x, run the W_o segment over the staged context block, and keep x1 as the feed-forward's residual.
 *   Equal to bofi_attn_block's W_o half + bofi_ffn_block up to the summation order of the row statistics behind LN(x1).  80-row blocks, one per workgroup. */
int bofi_attn_out_ffn_block(const float* x, int ldx, const void* ctx, int ldc, const void* wop, const float* bo, const void* w1p, const float* c1,
                            const float* cs1, const void* w2p, const float* b2, float* y, int ldy, int M, int dff, const void* pj_wp, const float* pj_c,
                            const float* pj_cs, void* pj_y, int pj_ldy, int pj_N, void* stream);

/* Developer aid: copy one of the bounding iteration's workspace buffers ("by1", "byb", "st_b", "bq2", "bctx2", "by2", "bh", "by3")
 * into user memory (device to device, on `stream`). */
int bofi_engine_debug_copy(bofi_engine_t* e, const char* name, void* dst, int64_t bytes, void* stream);

/* Measurement aid: GEMM FLOPs (2 M N K per launch of bofi_linear* / the weight-gradient GEMMs, engine calls included) enqueued
 * by this process since the last reset; host-side tally, not thread-safe.  Launches that carry an early-out word (loop
 * iterations that return at once when every image is finished) are tallied apart, into *skippable (may be NULL).
 * bench.py prices the EXECUTED work of a step with it. */
double bofi_gemm_flops(int reset, double* skippable);

/* The filling pass alone on a GIVEN slot layout (teacher-forced): decode_NA TransformerModel.py:570-587 -> logit /
 * log_softmax AttModel.py:203-210 -> greedy pick CaptionModel.py:388-390 -> pad tail AttModel.py:422-423, on the memory of
 * the preceding bofi_engine_encode.  ext_syn int32 [B, S+2] (extend_phrase_syn of core_NAIC :1829: position 0 = [LEN],
 * placed slots = their label), last int32 [B] (1 + tokens laid out).  flags: BOFI_FLAG_STRICT_Q1 / RAW_LOGITS / refinement
 * rounds as for decode_naic.  Lets a reduced-precision engine be compared with the oracle position by position on the
 * oracle's own layout, whatever its own bounding pass would have picked. */
int bofi_engine_fill_naic(bofi_engine_t* e, const int* ext_syn, const int* last, int B, int R, const int* att_len, int flags,
                          int64_t* seq, float* seq_logprob, void* stream);

/* Last HIP error string seen by this library on the calling thread (for exceptions in the host). */
const char* bofi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* BOFICAP_HIP_H */
