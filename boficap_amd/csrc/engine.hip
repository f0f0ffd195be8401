// Host-side engine: owns one model replica's packed weights and workspace in HBM and enqueues the
// whole NAIC bound+fill decode (reference AttModel._sample AttModel.py:307-338,419-429 ->
// TransformerModel._prepare_feature :1674-1690 -> core_NAIC :1823-1876 -> logit/log_softmax ->
// greedy pick) as one stream of HIP kernels, optionally replayed from a captured hipGraph.
// No host<->device synchronisation happens inside a decode call: the slot state of the bounding
// loop lives on the device and finished batches make the remaining bound kernels return at once.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/boficap_hip.h"
#include "bofi_common.h"
#include "bofi_kernels.h"
#include "bofi_naic.h"


namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define ENG_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) return fail(BOFI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define ENG_OK(call)                                                                           \
    do {                                                                                       \
        int r__ = (call);                                                                      \
        if (r__ != BOFI_OK) { if (g_err.empty() || r__ != BOFI_ERR_HIP) g_err = std::string(#call) + " failed"; return r__; } \
    } while (0)

uint16_t host_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

struct Lin { void* w = nullptr; float* b = nullptr; float* cs = nullptr; int N = 0, K = 0; int Npad = 0; void* wp = nullptr; void* wp16 = nullptr; };   // cs: column sums when a LayerNorm is folded in; Npad > N: rows N .. Npad of w / b / cs exist and are zero; wp: fragment-major copy of w for the row-block kernels (rowblock.hip), bf16 engine only; wp16: fragment-major FP16 copy made from the float32 parameters (bound_loop.hip), the bounding layer's Linears of a bf16 engine only
struct Norm { float* g = nullptr; float* b = nullptr; };
struct EncLayer { Lin qkv, o, w1, w2; Norm n0, n1; };
struct DecLayer { Lin qkv, o, q_src, o_src, w1, w2; Norm n0, n1, n2; };

// how a packed weight was made from the named parameters: replayed on the device by bofi_engine_refresh_device
struct LinRecipe { Lin* out; std::vector<std::string> prefixes; int n_each, K; std::string fold; };

struct NormRecipe { Norm* out; std::string prefix; int d; };

struct GraphEntry {
    std::vector<uintptr_t> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

}  // namespace

struct bofi_engine {
    bofi_config_t cfg{};
    std::map<std::string, std::vector<float>> host;
    std::vector<void*> allocs;
    bool finalized = false;
    int L = 0;                       // seq_length + 2
    size_t tsz = 4;                  // bytes per compute-dtype element

    // weights
    Lin att_embed;
    std::vector<EncLayer> enc; Norm enc_norm;
    std::vector<DecLayer> dec; Norm dec_norm;
    std::vector<DecLayer> blay;      // the bounding network's layers as full layers (dense form: N_len >= 2, or forced for a self-check)
    int n_len = 1;
    bool bound_dense = false;
    Lin kv_all;                      // stacked cross-attention K|V: bound layer, then decoder layers
    Lin gen;
    float *lut_syn = nullptr, *lut_tok = nullptr, *pe = nullptr;
    // bound layer
    Lin b_o_self, b_q_src, b_o_src, b_w1, b_w2;
    Norm b_n0, b_n1, b_n2;
    bofi::BoundHeadWeights heads{};
    void *b_q0 = nullptr, *b_kvtab = nullptr;     // compute dtype: [d], [L*10, 2d]
    float* b_x0 = nullptr;                        // [d] residual input of row 0
    void *b_votab = nullptr, *b_w1p = nullptr;    // compute dtype: Wo_self . V per (row, head) [L*10, H, d]; packed head hidden weights
    float* b_x0b = nullptr;                       // [d] x0 + bo_self
    // the persistent bounding-loop kernel (bound_loop.hip; bf16 engine at the reference's width): float32 tables of the row-0 self-attention
    Lin b_heads;                                  // hidden layers of both heads stacked, length_predictor.norm folded in (its wp16 and folded bias are what the loop kernel reads)
    float *b_q0_32 = nullptr, *b_sctab = nullptr, *b_vtab = nullptr;      // [d], [L*10][H], [L*10][d]
    // filling pass, round 0: every word is BOS (decode_NA TransformerModel.py:570-577), so the first decoder layer's input row -- and with it its q|k|v row -- depends on
    // (label, position) only: f_qkv0[(label*S + t)][3d] in the compute dtype, made by the SAME row-block projection kernel a decode would run on those rows (a row's
    // result does not depend on the launch), gathered by embed_fill instead of a 10-GFLOP projection per 320 images.  Rebuilt by finalize / refresh_device.
    void* f_qkv0 = nullptr; float* f_x0 = nullptr; int* f_syn = nullptr; bool fill_tab_ready = false;
    // per-call (bofi_engine_set_row_stats_out; part of the graph key): where the filling pass's epilogue leaves, per position, sum_v p log p and the log-prob of the
    // emitted id -- eval's entropy / perplexity (eval_utils.py:463-464) without a second pass over the log-probs
    float *row_plogp_out = nullptr, *row_chosen_out = nullptr;
    bool loop_ready = false;
    float* dbg_part = nullptr;

    // workspace
    float *x_enc = nullptr, *x_fill = nullptr, *logits = nullptr;
    float* logits_pad = nullptr;         // bf16 engine: the generator's output with a pitch of gen.Npad (whole 128-column tiles: the persistent GEMM), read by vocab_finalize
    void *qkv = nullptr, *ctx = nullptr, *hdn = nullptr, *mem = nullptr, *kv = nullptr, *qs = nullptr, *xn = nullptr;
    void* feats_t = nullptr;                                          // bf16 copy of float32 input features
    void *xb_enc = nullptr, *xb_fill = nullptr, *byb = nullptr;       // compute-dtype copies of the residual streams
    float *st_enc = nullptr, *st_fill = nullptr, *st_b = nullptr;     // row partial sums [rows][d/32][2]
    float* st_fill16 = nullptr;                                       // the same for the decoder rows [Bm*Sq][d/16][2] (row-list iterations of the SAIC decode)
    float* st_b16 = nullptr;                                          // row partial sums per 16 columns [Bm][d/16][2] (bound_ops.hip)
    float *by1 = nullptr, *by2 = nullptr, *by3 = nullptr;
    void *bctx = nullptr, *bq2 = nullptr, *bctx2 = nullptr, *bh = nullptr;
    bofi::BoundState st{};
    bofi::SaicState sa{};
    float* xw = nullptr; void* xwb = nullptr; float* st_w = nullptr;   // SAIC bound input rows [B*L, d] (+copy, +stats)
    void* kvs = nullptr;                                                // their K|V [B*L, 2d]
    int64_t* tok64 = nullptr;                                           // greedy ids of one decoder pass [B*S]
    std::vector<void*> qkv_dec;                                         // SAIC: q|k|v of every decoder layer [B*S, 3d] (the K / V cache)
    int *sa_rows = nullptr, *sa_nrows = nullptr;                        // decoder rows of the iteration's new phrases, their count
    uint64_t* d_seed = nullptr;                                         // per-call sampling seed (read by the captured sampling kernels)
    bool saic_cache = true, saic_lean = true;
    Lin b_kv_self;                                                      // bound self-attention K|V with sublayer.0.norm folded in
    Lin t_kvself, t_qself; float* d_xt = nullptr; Norm head_norm;       // setup-time operands of the bound tables, kept for refreshes
    std::vector<LinRecipe> lin_recipes; std::vector<NormRecipe> norm_recipes;
    void* b_q0_sa = nullptr; float* b_x0_sa = nullptr;                  // row-0 constants when position 0 holds tgt_embed([LEN])

    hipStream_t cap_stream = nullptr;
    int q1_group = 0;                     // > 0: the call carries several independent batches of this many images (quirk Q1 per batch)
    float sample_temperature = 1.0f;      // BOFI_FLAG_SAMPLE: token draws inside the semi-autoregressive loop
    uint64_t sample_seed = 0;
    int saic_it_begin = 1, saic_it_end = 0;   // iterations the next semi-autoregressive decode enqueues (bofi_engine_set_saic_range; end 0 = seq_length)
    int bound_iter_cap = 0;               // bounding iterations the non-autoregressive decode enqueues (bofi_engine_set_bound_iter_cap; 0 = seq_length)
    int in_flight = 0;                    // decodes the caller keeps in flight (bofi_engine_set_decodes_in_flight; 0 = unknown: throughput forms)
    int* live_max = nullptr;              // optional device word: max over decodes of their live-iteration counts (bofi_engine_set_live_iterations_max)
    int* sat_out = nullptr;               // optional device word: fp16 saturation status of the following NAIC decodes' bounding loop (bofi_engine_set_saturation_out)
    int loop_mode = -1;                   // bofi_engine_set_bound_loop: -1 = BOFI_BOUND_LOOP / hint decide, 0 = five-launch iterations, 2 = the persistent loop kernel
    float* bl_xbuf = nullptr; unsigned* bl_xctl = nullptr;      // the loop kernel's pair exchange: partial sums and control words (workspace: one set per engine / fork)
    int* b_wsat = nullptr;                // [1] set by pack_frag16 when an fp16 weight copy was clamped (shared with forks, like the weights)
    hipStream_t run_stream = nullptr;     // a stream of the engine's own, offered to callers that keep several decodes in flight
    std::vector<GraphEntry> graphs;

    template <typename U> int dalloc(U** p, size_t n_elems, size_t elem = sizeof(U)) {
        void* q = nullptr;
        const size_t bytes = (n_elems * elem + 255) & ~(size_t)255;
        ENG_HIP(hipMalloc(&q, bytes ? bytes : 256));
        ENG_HIP(hipMemset(q, 0, bytes ? bytes : 256));
        allocs.push_back(q);
        *p = (U*)q;
        return BOFI_OK;
    }
    int upload_f32(float** p, const std::vector<float>& v) {
        ENG_OK(dalloc(p, v.size()));
        ENG_HIP(hipMemcpy(*p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        return BOFI_OK;
    }
    int upload_t(void** p, const std::vector<float>& v) {           // compute dtype
        if (cfg.dtype == BOFI_DT_F32) return upload_f32((float**)p, v);
        std::vector<uint16_t> h(v.size());
        for (size_t i = 0; i < v.size(); ++i) h[i] = host_bf16(v[i]);
        uint16_t* q = nullptr;
        ENG_OK(dalloc(&q, h.size()));
        ENG_HIP(hipMemcpy(q, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        *p = q;
        return BOFI_OK;
    }
    const std::vector<float>* get(const std::string& name, size_t numel) {
        auto it = host.find(name);
        if (it == host.end()) { g_err = "missing weight " + name; return nullptr; }
        if (it->second.size() != numel) {
            g_err = "weight " + name + " has " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(numel);
            return nullptr;
        }
        return &it->second;
    }
    // stack several [n_i, K] matrices (and their biases) into one Lin.  With fold_norm the pre-norm
    // LayerNorm that feeds this layer is folded in (gemm_glds.hip): w <- w * a_2 (per input column),
    // bias <- bias + w . b_2, cs[n] <- sum_k of the ROUNDED scaled weight (so that the mean term of
    // the epilogue cancels exactly what the MFMA accumulated).
    int make_lin(Lin* out, const std::vector<std::string>& prefixes, int n_each, int K, const std::string& fold_norm = "", int pad_to = 0, bool frag = false) {
        std::vector<float> w, b;
        for (const auto& p : prefixes) {
            const auto* pw = get(p + ".weight", (size_t)n_each * K);
            const auto* pb = get(p + ".bias", (size_t)n_each);
            if (!pw || !pb) return BOFI_ERR_STATE;
            w.insert(w.end(), pw->begin(), pw->end());
            b.insert(b.end(), pb->begin(), pb->end());
        }
        out->N = n_each * (int)prefixes.size();
        out->K = K;
        out->Npad = pad_to > 0 ? ((out->N + pad_to - 1) / pad_to) * pad_to : out->N;      // zero rows behind the N real ones (the repack on the device writes the first N only)
        if (!fold_norm.empty()) {
            const auto* g = get(fold_norm + ".a_2", K);
            const auto* bb = get(fold_norm + ".b_2", K);
            if (!g || !bb) return BOFI_ERR_STATE;
            std::vector<float> cs(out->N);
            for (int n = 0; n < out->N; ++n) {
                double c = b[n], s = 0.0;
                float* row = w.data() + (size_t)n * K;
                for (int k = 0; k < K; ++k) {
                    c += (double)(*bb)[k] * (double)row[k];
                    row[k] = row[k] * (*g)[k];
                    float r = row[k];
                    if (cfg.dtype == BOFI_DT_BF16) { uint32_t u = (uint32_t)host_bf16(r) << 16; std::memcpy(&r, &u, 4); }
                    s += (double)r;
                }
                b[n] = (float)c;
                cs[n] = (float)s;
            }
            cs.resize(out->Npad, 0.f);
            ENG_OK(upload_f32(&out->cs, cs));
        }
        w.resize((size_t)out->Npad * K, 0.f);
        b.resize(out->Npad, 0.f);
        ENG_OK(upload_t(&out->w, w));
        ENG_OK(upload_f32(&out->b, b));
        out->wp = nullptr;
        out->wp16 = nullptr;
        if (frag && cfg.dtype == BOFI_DT_BF16 && out->Npad % 64 == 0 && K % 32 == 0) {      // the layout the row-block kernels stream (zero rows included)
            ENG_OK(dalloc((char**)&out->wp, (size_t)out->Npad * K, 2));
            ENG_OK(bofi::launch_rb_pack_frag(out->w, out->wp, out->Npad, K, nullptr));
        }
        lin_recipes.push_back(LinRecipe{out, prefixes, n_each, K, fold_norm});
        return BOFI_OK;
    }
    int make_norm(Norm* out, const std::string& prefix, int d) {
        const auto* g = get(prefix + ".a_2", d);
        const auto* b = get(prefix + ".b_2", d);
        if (!g || !b) return BOFI_ERR_STATE;
        ENG_OK(upload_f32(&out->g, *g));
        ENG_OK(upload_f32(&out->b, *b));
        norm_recipes.push_back(NormRecipe{out, prefix, d});
        return BOFI_OK;
    }

    // ---- kernels -------------------------------------------------------------------------------
    struct LinOpt {
        const float* residual = nullptr; int ldr = 0;
        int relu = 0;
        const int* row_len = nullptr; int rpg = 0;
        bool early = false;
        bool halt = false;                   // SAIC: early-out on the halt word (counters[2] >= 1)
        int splitk = 1;                      // K split over workgroups, float32 partial slabs out
        const Norm* ln = nullptr;            // explicit LayerNorm kernel on x first (setup-time use only)
        const float* ln_stats = nullptr;     // LayerNorm folded into the GEMM (Lin built with fold_norm)
        float* stats_out = nullptr;          // emit row partial sums of the output
        void* y2 = nullptr;                  // compute-dtype copy of the output
        const int* row_idx = nullptr; const int* m_dev = nullptr;     // row list (LinearArgs)
        int ln_groups = 0;                   // partial-sum pairs per row of ln_stats (0: K / 32)
    };
    int linear(const void* x, int x_dtype, int ldx, const Lin& l, void* y, int y_dtype, int ldy, int M, const LinOpt& o, hipStream_t s) {
        if (o.ln) {
            if (x_dtype != BOFI_DT_F32 || ldx != l.K) return BOFI_ERR_ARG;
            int rc = bofi::launch_layernorm((const float*)x, o.ln->g, o.ln->b, xn, cfg.dtype, M, l.K, s, nullptr, 0);
            if (rc != BOFI_OK) return rc;
            x = xn; x_dtype = cfg.dtype;
        }
        if (o.ln_stats && !l.cs) return BOFI_ERR_STATE;
        bofi::LinearArgs a{};
        a.x = x; a.x_dtype = x_dtype; a.ldx = ldx; a.w = l.w; a.w_dtype = cfg.dtype; a.bias = l.b;
        a.residual = o.residual; a.ldr = o.ldr; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy;
        a.M = M; a.N = l.N; a.K = l.K; a.relu = o.relu; a.row_len = o.row_len; a.rows_per_group = o.rpg;
        a.ln_stats = o.ln_stats; a.ln_colsum = o.ln_stats ? l.cs : nullptr; a.ln_groups = o.ln_groups;
        a.stats_out = o.stats_out; a.y2 = o.y2; a.ldy2 = l.N; a.splitk = o.splitk; a.row_idx = o.row_idx; a.m_dev = o.m_dev;
        if (o.early) { a.skip_if_ge = st.counters; a.skip_threshold = cur_B; }
        if (o.halt) { a.skip_if_ge = st.counters + 2; a.skip_threshold = 1; }
        return bofi::launch_linear(a, s);
    }
    // residual stream in the compute dtype: the fp32 engine reads the stream itself
    const void* stream_t(const float* x32, const void* xt) const { return cfg.dtype == BOFI_DT_F32 ? (const void*)x32 : xt; }
    void* copy_t(void* xt) const { return cfg.dtype == BOFI_DT_F32 ? nullptr : xt; }
    int cur_B = 0;
    bool is_fork = false;
    size_t n_weight_allocs = 0;          // allocs[0 .. n) are weights (owned by the parent), the rest workspace
    // bofi_engine_refresh_device: descriptor tables of the batched repack launches (device copy, pinned staging, what was uploaded last)
    void* rt_dev = nullptr; void* rt_pin = nullptr; size_t rt_bytes = 0; std::vector<char> rt_cache;

    // workspace of one in-flight decode (a forked engine has its own, and shares the weights)
    int alloc_workspace() {
    const bofi_config_t& c = cfg;
    const size_t Bm = c.max_batch, Rm = c.max_regions, Sq = c.seq_length;
    const int d = c.d_model, dff = c.d_ff;
    const size_t Lm = Sq + 2;
    const size_t rows = Bm * (Rm > Lm ? Rm : Lm);
    ENG_OK(dalloc(&x_enc, Bm * Rm * d));
    ENG_OK(dalloc(&x_fill, Bm * Sq * d));
    ENG_OK(dalloc(&logits, Bm * Sq * c.vocab));
    if (c.dtype == BOFI_DT_BF16 && gen.Npad > gen.N) ENG_OK(dalloc(&logits_pad, Bm * Sq * (size_t)gen.Npad));
    ENG_OK(dalloc((char**)&qkv, rows * 3 * d, tsz));
    ENG_OK(dalloc((char**)&ctx, rows * d, tsz));
    ENG_OK(dalloc((char**)&hdn, rows * dff, tsz));
    ENG_OK(dalloc((char**)&mem, Bm * Rm * d, tsz));
    ENG_OK(dalloc((char**)&kv, Bm * Rm * (size_t)kv_all.N, tsz));
    ENG_OK(dalloc((char**)&qs, Bm * Sq * d, tsz));
    ENG_OK(dalloc((char**)&xn, (rows > (size_t)L * 10 ? rows : (size_t)L * 10) * d, tsz));
    ENG_OK(dalloc((char**)&xb_enc, Bm * Rm * d, tsz)); ENG_OK(dalloc((char**)&xb_fill, Bm * Sq * d, tsz));
    ENG_OK(dalloc((char**)&byb, Bm * d, tsz));
    if (c.dtype == BOFI_DT_BF16) ENG_OK(dalloc((char**)&feats_t, Bm * Rm * (size_t)c.feat, 2));
    ENG_OK(dalloc(&st_enc, Bm * Rm * (d / 32) * 2)); ENG_OK(dalloc(&st_fill, Bm * Sq * (d / 32) * 2));
    ENG_OK(dalloc(&st_b, Bm * (d / 32) * 2)); ENG_OK(dalloc(&st_b16, Bm * (d / 16) * 2));
    ENG_OK(dalloc(&by1, Bm * d)); ENG_OK(dalloc(&by2, Bm * d)); ENG_OK(dalloc(&by3, 4 * Bm * d));
    ENG_OK(dalloc((char**)&bctx, Bm * d, tsz)); ENG_OK(dalloc((char**)&bq2, Bm * d, tsz));
    ENG_OK(dalloc((char**)&bctx2, Bm * d, tsz)); ENG_OK(dalloc((char**)&bh, Bm * dff, tsz));
    ENG_OK(dalloc(&st.last, Bm)); ENG_OK(dalloc(&st.finished, Bm)); ENG_OK(dalloc(&st.phrase_num, Bm));
    ENG_OK(dalloc(&st.phrase_length, Bm * L)); ENG_OK(dalloc(&st.phrase_syn, Bm * L));
    ENG_OK(dalloc(&st.ext_syn, Bm * L)); ENG_OK(dalloc(&st.counters, 8)); ENG_OK(dalloc(&st.klen, Bm * L));
    if (c.dtype == BOFI_DT_BF16) {                                  // the loop kernel's pair exchange (bound_loop.hip): per group of 16 images 2 x 2 partial tiles + 4 control words
        const size_t groups = (Bm + 15) / 16;
        ENG_OK(dalloc(&bl_xbuf, groups * 4 * 16 * 512)); ENG_OK(dalloc(&bl_xctl, groups * 4));
        st.pair_ctl = bl_xctl;
    }
    ENG_OK(dalloc(&sa.seq_last, Bm)); ENG_OK(dalloc(&sa.seq, Bm * L)); ENG_OK(dalloc(&sa.ext_len, Bm * L));
    ENG_OK(dalloc(&sa.ext_phrase, Bm * L)); ENG_OK(dalloc(&sa.klen_dec, Bm * L));
    ENG_OK(dalloc(&xw, Bm * L * d)); ENG_OK(dalloc((char**)&xwb, Bm * L * d, tsz)); ENG_OK(dalloc(&st_w, Bm * L * (d / 32) * 2));
    ENG_OK(dalloc((char**)&kvs, Bm * L * 2 * (size_t)d, tsz)); ENG_OK(dalloc(&tok64, Bm * Sq));
    ENG_OK(dalloc(&sa_rows, Bm * Sq)); ENG_OK(dalloc(&sa_nrows, 4)); ENG_OK(dalloc(&d_seed, 2));
    ENG_OK(dalloc(&st_fill16, Bm * Sq * (d / 16) * 2));
    qkv_dec.assign(c.n_dec, nullptr);                  // allocated by the first semi-autoregressive decode

        return BOFI_OK;
    }

    // the per-image tail of a bounding iteration (naic.hip): heads on the FFN output `y`, bookkeeping, next self-attention sublayer
    int bound_tail(const float* y, int yparts, const int* ext_syn_in, const int* last_in, int B, int flags, float* len_logp, float* syn_logp,
                   hipStream_t s, bool saic = false, int iter = 0, int y_stride = 0) {
        bofi::BoundTailArgs a{};
        a.y = y; a.yparts = yparts; a.y_stride = y_stride; a.w = heads; a.st = st; a.sa = saic ? sa : bofi::SaicState{};
        a.ext_syn_in = ext_syn_in; a.last_in = last_in; a.q0 = b_q0; a.kvtab = b_kvtab; a.votab = b_votab; a.x0b = b_x0b;
        a.y1 = by1; a.y1t = copy_t(byb); a.stats = st_b;
        const int tail_dbg = BOFI_ENV_INT("BOFI_TAIL_DBG", 0);     // developer ablations
        a.B = B; a.L = L; a.S = cfg.seq_length; a.d = cfg.d_model; a.hh = cfg.head_hidden; a.H = cfg.heads; a.flags = flags | (tail_dbg << 8); a.iter = iter;
        a.len_logp = len_logp; a.syn_logp = syn_logp;
        const bool want_dbg = BOFI_ENV_INT("BOFI_DBG_PART", 0) != 0;
        if (want_dbg && !dbg_part) ENG_OK(dalloc(&dbg_part, (size_t)cfg.max_batch * (16 * cfg.head_hidden + cfg.d_model)));
        a.dbg_part = dbg_part;
        return bofi::launch_bound_tail(a, cfg.dtype, s);
    }
    // tables derived from the bound layer's weights once kvtab exists (finalize and refresh_device)
    int derive_bound_tables(hipStream_t s) {
        ENG_OK(bofi::launch_pack_w1p(heads.w1t, b_w1p, cfg.dtype, cfg.d_model, 2 * cfg.head_hidden, s));
        ENG_OK(bofi::launch_votab(b_kvtab, b_o_self.w, b_x0, b_o_self.b, b_votab, b_x0b, cfg.dtype, L * 10, cfg.d_model, cfg.heads, s));
        return derive_fill_table(s);
    }
    int derive_fill_table(hipStream_t s) {
        fill_tab_ready = false;
        if (!f_qkv0 || !rb_ok() || dec.empty() || !dec[0].qkv.wp || !dec[0].qkv.cs || dec[0].qkv.Npad != 3 * cfg.d_model) return BOFI_OK;
        const int d = cfg.d_model, S = cfg.seq_length;
        // the input rows of 10 pseudo-images whose slots all carry one label, then their q|k|v by the row-block projection kernel
        ENG_OK(bofi::launch_embed_fill(lut_tok, lut_syn, pe, f_syn, nullptr, 10, S, L, d, cfg.bos_idx, f_x0, nullptr, cfg.dtype, nullptr, s));
        bofi::RbGemmArgs a{};
        a.x = f_x0; a.ldx = d; a.wp = (const bofi::u32x4*)dec[0].qkv.wp; a.c = dec[0].qkv.b; a.cs = dec[0].qkv.cs; a.y = f_qkv0; a.ldy = 3 * d; a.y_f32 = 0; a.M = 10 * S; a.N = 3 * d;
        ENG_OK(bofi::launch_rb_gemm(a, s));
        fill_tab_ready = true;
        return BOFI_OK;
    }
    // ---- the persistent bounding-loop kernel's operands: fp16 fragment-major copies of the bounding layer's Linears and the float32 self-attention
    // tables, all derived on the device from the float32 parameters (finalize: uploaded temporaries; refresh_device: the caller's tensors)
    struct BoundSrc {
        const float *wq_self, *bq_self, *wk_self, *bk_self, *wv_self, *bv_self, *n0g, *n0b;
        const float *wo_self, *wq_src, *n1g, *wo_src, *w1, *n2g, *w2, *lw1, *sw1, *hng;
    };
    bool loop_config_ok() const {
        return cfg.dtype == BOFI_DT_BF16 && cfg.d_model == 512 && cfg.heads == 8 && cfg.d_ff % 512 == 0 && cfg.d_ff <= 2048 && L <= 24 && cfg.head_hidden <= 128 &&
               n_len == 1 && !bound_dense;
    }
    int derive_bound_f16(const BoundSrc& b, hipStream_t s) {
        if (!loop_config_ok()) return BOFI_OK;
        const int d = cfg.d_model, dff = cfg.d_ff, hh = cfg.head_hidden;
        bofi::Pack16Table t{};
        auto ent = [&](const float* w0, const float* w1_, int n_each, int nsrc, int K, int Npad, const float* gain, void* out) {
            bofi::Pack16Entry& q = t.e[t.n++];
            q.w[0] = w0; q.w[1] = w1_; q.gain = gain; q.out = out; q.n_each = n_each; q.nsrc = nsrc; q.K = K; q.Npad = Npad; q.blk0 = 0;
        };
        ent(b.wo_self, nullptr, d, 1, d, d, nullptr, b_o_self.wp16);
        ent(b.wq_src, nullptr, d, 1, d, d, b.n1g, b_q_src.wp16);
        ent(b.wo_src, nullptr, d, 1, d, d, nullptr, b_o_src.wp16);
        ent(b.w1, nullptr, dff, 1, d, dff, b.n2g, b_w1.wp16);
        ent(b.w2, nullptr, d, 1, dff, d, nullptr, b_w2.wp16);
        ent(b.lw1, b.sw1, hh, 2, d, 256, b.hng, b_heads.wp16);
        t.sat = b_wsat;
        ENG_OK(bofi::launch_pack_frag16(t, s));
        bofi::BoundTablesArgs a{};
        a.xt = d_xt; a.x0 = b_x0; a.rows = L * 10; a.n0g = b.n0g; a.n0b = b.n0b;
        a.wq = b.wq_self; a.bq = b.bq_self; a.wk = b.wk_self; a.bk = b.bk_self; a.wv = b.wv_self; a.bv = b.bv_self;
        a.q0 = b_q0_32; a.sctab = b_sctab; a.vtab = b_vtab;
        ENG_OK(bofi::launch_bound_tables(a, s));
        loop_ready = true;
        return BOFI_OK;
    }
    // BOFI_BOUND_LOOP (re-read after bofi_reload_env): 0 = the five-launch iterations of rounds 2-4 always, 1 (default) = the persistent loop kernel unless the caller
    // said its decodes run alone (bofi_engine_set_decodes_in_flight(1): the five launches are the shorter chain there -- 0.65 against 0.95 ms per 320 images),
    // 2 = the loop kernel whatever the hint
    static int bound_loop_knob() { return BOFI_ENV_INT("BOFI_BOUND_LOOP", 1); }
    bool bound_loop_ok(int R) const {
        const int k = loop_mode >= 0 ? loop_mode : bound_loop_knob();
        return loop_ready && loop_config_ok() && R <= 128 && k != 0 && (k == 2 || in_flight != 1);
    }
    // update != 0: the whole loop on the engine's slot state (after launch_bound_init); update == 0: one iteration on a given layout, log-probabilities out
    int bound_loop(int B, int R, const int* att_len, const int* ext_syn_in, const int* last_in, int update, float* len_logp, float* syn_logp, hipStream_t s) {
        bofi::BoundLoopArgs a{};
        a.wo_self = (const bofi::bl_u32x4*)b_o_self.wp16; a.x0b = b_x0b;
        a.wq_src = (const bofi::bl_u32x4*)b_q_src.wp16; a.cq = b_q_src.b;
        a.wo_src = (const bofi::bl_u32x4*)b_o_src.wp16; a.bo_src = b_o_src.b;
        a.w1 = (const bofi::bl_u32x4*)b_w1.wp16; a.c1 = b_w1.b;
        a.w2 = (const bofi::bl_u32x4*)b_w2.wp16; a.b2 = b_w2.b;
        a.wh = (const bofi::bl_u32x4*)b_heads.wp16; a.ch = b_heads.b;
        a.len_w2 = heads.len_w2; a.len_b2 = heads.len_b2; a.syn_w2 = heads.syn_w2; a.syn_b2 = heads.syn_b2;
        a.sctab = b_sctab; a.vtab = b_vtab;
        a.k = (const uint16_t*)kv; a.v = (const uint16_t*)kv + cfg.d_model; a.ldkv = kv_all.N; a.att_len = att_len;
        a.st = st; a.ext_syn_in = ext_syn_in; a.last_in = last_in; a.len_logp = len_logp; a.syn_logp = syn_logp;
        a.B = B; a.R = R; a.L = L; a.S = cfg.seq_length; a.hh = cfg.head_hidden; a.dff = cfg.d_ff;
        a.max_iters = update ? cfg.seq_length : 1; a.update = update;
        a.sat = st.counters + 4; a.wsat = b_wsat;             // (zeroed by launch_bound_init; the export hands it to the caller's word)
        a.xbuf = bl_xbuf; a.xctl = bl_xctl;                    // (two workgroups per group when the launcher's conditions hold: BOFI_BL_PAIR)
        return bofi::launch_bound_loop(a, s);
    }
    // y1 (by1 / byb / st_b) -> y3 partial slabs (by3): query projection + cross-attention, Wo_src, FFN, as the direct-operand kernels
    // of bound_ops.hip.  Returns -1 when the configuration is not theirs (the caller takes the general kernels), else a status;
    // *parts = number of slabs in by3.  `skip`: early-out word and threshold (NULL: none).
    int bound_chain_lean(int B, int R, const int* att_len, const int* skip, int skip_thr, hipStream_t s, int* parts) {
        const bool lean_on = BOFI_ENV_INT("BOFI_BOUND_LEAN", 1) != 0;
        const int d = cfg.d_model;
        if (!(lean_on && cfg.dtype == BOFI_DT_BF16 && d == 512 && cfg.heads == 8 && R <= 64 && cfg.d_ff % 512 == 0 && cfg.d_ff / 512 <= 4)) return -1;
        {   bofi::BoundQAttnArgs a{};
            a.x = (const uint16_t*)byb; a.stats = st_b; a.wq = (const uint16_t*)b_q_src.w; a.bias = b_q_src.b; a.colsum = b_q_src.cs;
            a.k = (const uint16_t*)kv; a.v = (const uint16_t*)kv + d; a.ldkv = kv_all.N; a.att_len = att_len; a.out = (uint16_t*)bctx2;
            a.B = B; a.R = R; a.d = d; a.H = cfg.heads; a.skip_if_ge = skip; a.skip_threshold = skip_thr;
            ENG_OK(bofi::launch_bound_qattn(a, s)); }
        {   bofi::RowGemmArgs a{};                   // y2 = y1 + Wo_src . ctx2 + bo
            a.x = (const uint16_t*)bctx2; a.ldx = d; a.w = (const uint16_t*)b_o_src.w; a.bias = b_o_src.b; a.residual = by1; a.ldr = d;
            a.y = by2; a.ldy = d; a.yb = (uint16_t*)byb; a.ldyb = d; a.stats_out = st_b16; a.M = B; a.N = d; a.K = d; a.splitk = 1;
            a.skip_if_ge = skip; a.skip_threshold = skip_thr;
            ENG_OK(bofi::launch_rowgemm(a, s)); }
        {   bofi::RowGemmArgs a{};                   // h = relu(W1 . LN(y2) + b1)
            a.x = (const uint16_t*)byb; a.ldx = d; a.w = (const uint16_t*)b_w1.w; a.bias = b_w1.b; a.stats = st_b16; a.stats_groups = d / 16;
            a.colsum = b_w1.cs; a.yb = (uint16_t*)bh; a.ldyb = cfg.d_ff; a.M = B; a.N = cfg.d_ff; a.K = d; a.splitk = 1; a.relu = 1;
            a.skip_if_ge = skip; a.skip_threshold = skip_thr;
            ENG_OK(bofi::launch_rowgemm(a, s)); }
        *parts = cfg.d_ff / 512;
        {   bofi::RowGemmArgs a{};                   // y3 = y2 + W2 . h + b2 as `parts` partial slabs
            a.x = (const uint16_t*)bh; a.ldx = cfg.d_ff; a.w = (const uint16_t*)b_w2.w; a.bias = b_w2.b; a.residual = by2; a.ldr = d;
            a.y = by3; a.ldy = d; a.M = B; a.N = d; a.K = cfg.d_ff; a.splitk = *parts;
            a.skip_if_ge = skip; a.skip_threshold = skip_thr;
            ENG_OK(bofi::launch_rowgemm(a, s)); }
        return BOFI_OK;
    }
    // ---- row-block sublayer kernels (rowblock.hip, bf16 engine at the reference's width): the attention sublayer (attention + output
    // projection + residual) and the feed-forward sublayer as one launch each.  Return -1 when the configuration or the shape is not
    // theirs (the caller then runs the separate attention / GEMM launches), else a status.  want_copy: the next consumer is a
    // LayerNorm-folded GEMM (it reads the compute-dtype copy and the row statistics); a following ffn_sublayer reads the stream itself.
    bool rb_ok() const { return cfg.dtype == BOFI_DT_BF16 && cfg.d_model == 512 && cfg.heads == 8; }
    // the row-block kernels are a fixed latency chain per workgroup (one block of rows, the sublayer's whole weight stream): they pay
    // from a few thousand rows on, where the tiled GEMMs' prologue / epilogue and the hidden tensor's round trip cost more
    // (BOFI_RB_MIN_ROWS: 0 = always, a huge value = never; re-read after bofi_reload_env.  Kernel family and launch size are then
    // decoupled: under ONE family a row's result does not depend on what else is in the launch -- bit for bit)
    static int rb_min_rows() {
        return BOFI_ENV_INT("BOFI_RB_MIN_ROWS", 4096);
    }
    // timing-only ablation switches (results INVALID): compiled in only by `BOFI_EXPERIMENTS=1 python -m boficap_amd.build --force`
#ifdef BOFI_EXPERIMENTS
    static bool exp_skip(const char* what) {
        static const char* v = [] { const char* e = getenv("BOFI_EXP_SKIP"); if (e) fprintf(stderr, "[boficap_hip] BOFI_EXP_SKIP=%s: kernels skipped, RESULTS INVALID\n", e); return e; }();
        return v && strstr(v, what);
    }
#else
    static constexpr bool exp_skip(const char*) { return false; }
#endif
    // pj (optional): the LayerNorm-folded [512, 512] projection that reads this sublayer's output next (the decoder layer's cross-attention queries), computed by the SAME
    // launch from each block while it sits in LDS (pj_y bf16, pitch pj_ldy); attn_proj_ok says when the attention kernel takes it (BOFI_RB_ATTN_PROJ, re-read after
    // bofi_reload_env: 0 = never, 1 (default) = when launches overlap, 2 = always)
    bool attn_proj_ok(const bofi::AttnArgs& at, const Lin& o, const Lin& pj) const {
        const int v = BOFI_ENV_INT("BOFI_RB_ATTN_PROJ", 1);
        const int M = at.B * at.Lq;
        return v && (v == 2 || in_flight != 1) && BOFI_ENV_INT("BOFI_RB_ATTN", 1) != 0 && BOFI_ENV_INT("BOFI_RB_ATTN_W", 0) == 0 && rb_ok() && o.wp && fold_rb_ok(pj, M) &&
               pj.Npad == 512 && !at.skip_if_ge && at.kdiv <= 1 && !at.q_start && !at.drop_thresh && !at.klen_sq && at.Lq <= 20 && at.Lk <= 32 && M >= rb_min_rows() &&
               !exp_skip("attn") && !exp_skip("qkv");
    }
    int attn_sublayer(const bofi::AttnArgs& at, const Lin& o, float* x, void* xb, float* stats, bool want_copy, hipStream_t s, const Lin* pj = nullptr, void* pj_y = nullptr,
                      int pj_ldy = 0) {
        const bool on = BOFI_ENV_INT("BOFI_RB_ATTN", 1) != 0;
        if (exp_skip("attn")) return BOFI_OK;
        if (!on || !rb_ok() || !o.wp || at.skip_if_ge || at.kdiv > 1 || at.q_start || at.drop_thresh || at.B * at.Lq < rb_min_rows()) return -1;
        bofi::RbAttnArgs a{};
        a.q = (const uint16_t*)at.q; a.ldq = at.ldq; a.k = (const uint16_t*)at.k; a.ldk = at.ldk; a.v = (const uint16_t*)at.v; a.ldv = at.ldv;
        a.B = at.B; a.Lq = at.Lq; a.Lk = at.Lk; a.klen = at.klen; a.klen_sb = at.klen_sb; a.klen_sq = at.klen_sq; a.klen_bias = at.klen_bias;
        a.klen_shared_last = at.klen_shared_last; a.wop = (const bofi::u32x4*)o.wp; a.bo = o.b; a.x = x; a.ldx = cfg.d_model; a.y = x; a.ldy = cfg.d_model;
        a.yb = want_copy ? (uint16_t*)xb : nullptr; a.stats_out = want_copy ? stats : nullptr;
        a.alone = in_flight == 1;
        if (pj) { a.pj_wp = (const bofi::u32x4*)pj->wp; a.pj_c = pj->b; a.pj_cs = pj->cs; a.pj_y = (uint16_t*)pj_y; a.pj_ldy = pj_ldy; a.yb = nullptr; a.stats_out = nullptr; }
        return bofi::launch_rb_attn(a, s);
    }
    // a LayerNorm-folded projection (K = d_model) of the residual stream x32 as a row-block kernel: it reads the float32 stream itself
    // (no compute-dtype copy, no row statistics from the producer).  -1: not its configuration.
    bool fold_rb_ok(const Lin& l, int M) const {
        const bool on = BOFI_ENV_INT("BOFI_RB_GEMM", 1) != 0;
        return on && rb_ok() && l.wp && l.cs && l.K == 512 && M >= rb_min_rows();
    }
    int fold_linear_rb(const float* x32, const Lin& l, void* y, int y_f32, int ldy, int M, hipStream_t s) {
        if (!fold_rb_ok(l, M)) return -1;
        if (exp_skip(y_f32 ? "gen" : l.Npad <= 1536 ? "qkv" : "kv")) return BOFI_OK;
        bofi::RbGemmArgs a{};
        a.x = x32; a.ldx = cfg.d_model; a.wp = (const bofi::u32x4*)l.wp; a.c = l.b; a.cs = l.cs; a.y = y; a.ldy = ldy; a.y_f32 = y_f32; a.M = M; a.N = l.Npad; a.relu = 0;
        a.alone = in_flight == 1;
        return bofi::launch_rb_gemm(a, s);
    }
    bool ffn_sublayer_ok(const Lin& w1, const Lin& w2, int M) const {
        const bool on = BOFI_ENV_INT("BOFI_RB_FFN", 1) != 0;
        return on && rb_ok() && w1.wp && w2.wp && w1.cs && cfg.d_ff % 512 == 0 && cfg.d_ff <= 2560 && M >= rb_min_rows();
    }
    // pj (optional): the LayerNorm-folded projection that reads this sublayer's output next (the next layer's q|k|v, the stacked cross K|V), computed by
    // the SAME launch from each closed block while it sits in LDS (pj_y, pitch pj_ldy); ffn_proj_ok says when the feed-forward kernel takes it
    // (BOFI_RB_FFN_PROJ, re-read after bofi_reload_env: 0 = never, 1 (default) = when launches overlap, 2 = always)
    bool ffn_proj_ok(const Lin& w1, const Lin& w2, const Lin& pj, int M) const {
        const int v = BOFI_ENV_INT("BOFI_RB_FFN_PROJ", 1);
        // (a decode running alone keeps the separate launches of 64-row blocks: measured 0.515 against 0.523 ms per batch with the narrow projections fused,
        // 0.536 with all of them)
        const int maxn = BOFI_ENV_INT("BOFI_RB_FFN_PROJ_MAXN", 0);          // developer knob: projections wider than this stay launches of their own (0: no limit)
        return v && (v == 2 || in_flight != 1) && ffn_sublayer_ok(w1, w2, M) && fold_rb_ok(pj, M) && pj.Npad >= 512 && (!maxn || pj.Npad <= maxn) && !exp_skip("ffn") &&
               !exp_skip("qkv") && !exp_skip("kv");
    }
    // The attention sublayer SPLIT (round 6): the attention core as the light kernel of attn_bf16.hip (context rows to memory as bf16: no W_o stream, no
    // 80-120 KB of LDS, many workgroups per CU) and W_o + residual as the HEAD segment of the feed-forward launch that follows (rb_ffn5_kernel<.., HEAD>: 80 rows per weight
    // byte, x1 never in memory) instead of rb_attn_kernel's attention + W_o + residual in one workgroup.  Gate (profiles/r06_gate_attn_split.txt, four streams): the encoder's
    // sublayer 21.4 -> 11.6 + 1.0 us, the filling pass's cross-attention 14.8 -> 7.9 + 2.2 us per 320 images -- four streams of ONE kernel each.  Inside the decode
    // (profiles/r06_attn_split_ab.txt, r06_split{0,1}_kernel_stats.txt) the cores save 132 us of kernel time per 320 images (30.1 -> 19.6 and 37.8 -> 22.4 us per launch
    // in flight) and the head segment costs 116 (115 -> 126 us per feed-forward launch: the producers wait while the consumers run W_o): +1.2 % at 1 024 images per
    // launch, where a launch's blocks come in several waves per CU and hide the longer chain, -1 ... -4 % at 320, where they do not.  Hence BOFI_RB_ATTN_SPLIT (re-read
    // after bofi_reload_env): 0 = never, 1 (default) = when launches overlap AND the launch holds at least BOFI_RB_ATTN_SPLIT_MIN_B (512) images, 2 = always.  Needs the
    // feed-forward sublayer directly behind the attention sublayer and no consumer of the bf16 copy / statistics.
    // Shapes the fused sublayer kernel does not take (more than 48 keys or 40 queries: real bottom-up features have up to 100 regions) are split at EVERY launch size: the
    // alternative there is the attention core + a tiled GEMM for W_o + the residual stream's round trip (profiles/r06_regions_sweep.txt).
    bool attn_split_ok(const bofi::AttnArgs& at, const Lin& o, const Lin& w1, const Lin& w2) const {
        const int v = BOFI_ENV_INT("BOFI_RB_ATTN_SPLIT", 1);
        const int M = at.B * at.Lq;
        const bool beyond = (at.Lq > 40 || at.Lk > 48) && M >= rb_min_rows();
        return v && (v == 2 || beyond || (in_flight != 1 && at.B >= BOFI_ENV_INT("BOFI_RB_ATTN_SPLIT_MIN_B", 512))) && BOFI_ENV_INT("BOFI_RB_ATTN", 1) != 0 && rb_ok() && o.wp &&
               ffn_sublayer_ok(w1, w2, M) && !at.skip_if_ge && at.kdiv <= 1 &&
               !at.q_start && !at.drop_thresh && at.Lq <= 128 && at.Lk <= 128 && !exp_skip("attn") && !exp_skip("ffn");
    }
    int ffn_sublayer(const Lin& w1, const Lin& w2, float* x, void* xb, float* stats, int M, hipStream_t s, const Lin* pj = nullptr, void* pj_y = nullptr,
                     int pj_ldy = 0, const Lin* head = nullptr, const void* head_ctx = nullptr) {
        if (!ffn_sublayer_ok(w1, w2, M)) return -1;
        if (exp_skip("ffn")) return BOFI_OK;
        bofi::RbFfnArgs a{};
        if (head) { a.head_wop = (const bofi::u32x4*)head->wp; a.head_bo = head->b; a.head_ctx = (const uint16_t*)head_ctx; a.head_ldc = cfg.d_model; }
        if (pj) { a.pj_wp = (const bofi::u32x4*)pj->wp; a.pj_c = pj->b; a.pj_cs = pj->cs; a.pj_y = pj_y; a.pj_ldy = pj_ldy; a.pj_N = pj->Npad; }
        a.x = x; a.ldx = cfg.d_model; a.w1p = (const bofi::u32x4*)w1.wp; a.c1 = w1.b; a.cs1 = w1.cs; a.w2p = (const bofi::u32x4*)w2.wp; a.b2 = w2.b;
        a.y = x; a.ldy = cfg.d_model; a.yb = (uint16_t*)xb; a.stats_out = stats; a.M = M; a.dff = cfg.d_ff;
        a.alone = in_flight == 1;
        return bofi::launch_rb_ffn(a, s);
    }
    int enqueue_encode(const void* feats, int feats_dtype, const int* att_len, int B, int R, float* memory_out, hipStream_t s);
    int enqueue_bound_iter(int B, int R, const int* att_len, const int* ext_syn, const int* last, int update, float* len_logp,
                           float* syn_logp, bool early, hipStream_t s);
    int enqueue_fill(const int* att_len, int B, int R, int flags, int64_t* seq, float* seq_logprob, hipStream_t s);
    int enqueue_bound_dense(const int* att_len, int B, int R, hipStream_t s);
    int enqueue_decode_saic(const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags, int64_t* seq,
                            float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn, int* bound_iters, hipStream_t s);
    int enqueue_decode(const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags, int64_t* seq,
                       float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn, float* memory_out,
                       int* bound_iters, hipStream_t s);
};

// ================================================================================================
int bofi_engine::enqueue_encode(const void* feats, int feats_dtype, const int* att_len, int B, int R, float* memory_out,
                                hipStream_t s) {
    const int d = cfg.d_model, dt = cfg.dtype, M = B * R;
    cur_B = B;
    if (exp_skip("encoder")) return BOFI_OK;           // (experiments build: the whole encoder phase)
    // The residual stream x_enc stays float32; every GEMM that closes a sublayer also writes a copy in
    // the compute dtype (xb_enc) and per-row partial sums (st_enc), from which the next pre-norm
    // LayerNorm is applied inside the consuming GEMM's epilogue (no LayerNorm launches).
    if (feats_dtype != dt) {                         // float32 features into a bf16 engine: one conversion pass
        ENG_OK(bofi::launch_cast_bf16((const float*)feats, feats_t, (size_t)M * cfg.feat, s));
        feats = feats_t; feats_dtype = dt;
    }
    {   // att_embed: Linear + ReLU, rows past an image's region count forced to 0 (AttModel.py:46-51)
        LinOpt o; o.relu = 1; o.row_len = att_len; o.rpg = R; o.stats_out = st_enc; o.y2 = copy_t(xb_enc);
        ENG_OK(linear(feats, feats_dtype, cfg.feat, att_embed, x_enc, BOFI_DT_F32, d, M, o, s));
    }
    const void* xa = stream_t(x_enc, xb_enc);
    const bool memory_out_needs_copy = false;           // (memory_out is a LayerNorm of the float32 stream itself)
    bool proj_made = false;                              // this layer's q|k|v (after the last layer: the cross K|V) came out of the previous feed-forward launch
    for (size_t li = 0; li < enc.size(); ++li) {
        auto& l = enc[li];
        if (!proj_made) {
            int rc = fold_linear_rb(x_enc, l.qkv, qkv, 0, 3 * d, M, s);
            if (rc > 0) return rc;
            if (rc < 0) { LinOpt o; o.ln_stats = st_enc; ENG_OK(linear(xa, dt, d, l.qkv, qkv, dt, 3 * d, M, o, s)); } }
        bofi::AttnArgs a{};
        a.q = qkv; a.k = (char*)qkv + (size_t)d * tsz; a.v = (char*)qkv + (size_t)2 * d * tsz;
        a.ldq = a.ldk = a.ldv = 3 * d; a.out = ctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = R; a.Lk = R;
        a.klen = att_len; a.klen_sb = 1; a.klen_sq = 0;
        const bool ffn_rb = ffn_sublayer_ok(l.w1, l.w2, M);
        const bool need_copy = !(fold_rb_ok(l.qkv, M) && fold_rb_ok(kv_all, M)) || memory_out_needs_copy;      // (consumers of this layer's output that read the copy + statistics: tiled GEMMs)
        const bool split = ffn_rb && !need_copy && (BOFI_ENV_INT("BOFI_RB_ATTN_SPLIT_WHICH", 3) & 1) && attn_split_ok(a, l.o, l.w1, l.w2);      // attention core now, W_o + residual as the head of the feed-forward launch
        int rc = split ? bofi::launch_attention(a, s) : attn_sublayer(a, l.o, x_enc, xb_enc, st_enc, !ffn_rb, s);
        if (rc > 0) return rc;
        if (rc < 0) {
            ENG_OK(bofi::launch_attention(a, s));
            LinOpt o; o.residual = x_enc; o.ldr = d; o.stats_out = st_enc; o.y2 = copy_t(xb_enc);
            ENG_OK(linear(ctx, dt, d, l.o, x_enc, BOFI_DT_F32, d, M, o, s));
        }
        {   // the consumers of this layer's output: the next layer's q|k|v (or the stacked cross K|V): tiled GEMMs read the copy + statistics
            const bool last = li + 1 == enc.size();
            const Lin& nxt = last ? kv_all : enc[li + 1].qkv;
            proj_made = !need_copy && ffn_proj_ok(l.w1, l.w2, nxt, M);
            const Lin* head = split ? &l.o : nullptr;
            rc = !ffn_rb ? -1 : proj_made ? ffn_sublayer(l.w1, l.w2, x_enc, nullptr, nullptr, M, s, &nxt, last ? kv : qkv, last ? kv_all.N : 3 * d, head, ctx)
                                          : ffn_sublayer(l.w1, l.w2, x_enc, need_copy ? xb_enc : nullptr, need_copy ? st_enc : nullptr, M, s, nullptr, nullptr, 0, head, ctx); }
        if (rc > 0) return rc;
        if (rc < 0) {
            { LinOpt o; o.relu = 1; o.ln_stats = st_enc; ENG_OK(linear(xa, dt, d, l.w1, hdn, dt, cfg.d_ff, M, o, s)); }
            { LinOpt o; o.residual = x_enc; o.ldr = d; o.stats_out = st_enc; o.y2 = copy_t(xb_enc);
              ENG_OK(linear(hdn, dt, cfg.d_ff, l.w2, x_enc, BOFI_DT_F32, d, M, o, s)); }
        }
    }
    if (memory_out) ENG_OK(bofi::launch_layernorm(x_enc, enc_norm.g, enc_norm.b, memory_out, BOFI_DT_F32, M, d, s));
    // cross-attention K|V of the bound layer and of every decoder layer in one GEMM on memory =
    // encoder.norm(x_enc), the norm folded in
    if (!proj_made) {
        int rc = fold_linear_rb(x_enc, kv_all, kv, 0, kv_all.N, M, s);
        if (rc > 0) return rc;
        if (rc < 0) { LinOpt o; o.ln_stats = st_enc; ENG_OK(linear(xa, dt, d, kv_all, kv, dt, kv_all.N, M, o, s)); } }
    return BOFI_OK;
}

int bofi_engine::enqueue_bound_iter(int B, int R, const int* att_len, const int* ext_syn, const int* last, int update,
                                    float* len_logp, float* syn_logp, bool early, hipStream_t s) {
    // One bounding iteration AFTER the row-0 self-attention sublayer output y1 (by1, its copy byb and row statistics st_b)
    // has been produced by the previous bound_tail(..., BOUND_ATTN).
    const int d = cfg.d_model, dt = cfg.dtype;
    cur_B = B;
    if (exp_skip("loop")) return BOFI_OK;
    {   // bf16 at the reference's width: the four stages as the direct-operand kernels of bound_ops.hip
        int parts = 0;
        const int rc = bound_chain_lean(B, R, att_len, early ? st.counters : nullptr, B, s, &parts);
        if (rc > 0) return rc;
        if (rc == 0) {
            const int flags = BOUND_HEADS | (update ? (BOUND_UPDATE | BOUND_ATTN) : 0) | (early ? BOUND_EARLY : 0);
            return bound_tail(by3, parts, update ? nullptr : ext_syn, update ? nullptr : last, B, flags, len_logp, syn_logp, s);
        }
    }
    { LinOpt o; o.early = early; o.ln_stats = st_b; ENG_OK(linear(stream_t(by1, byb), dt, d, b_q_src, bq2, dt, d, B, o, s)); }
    {
    bofi::AttnArgs a{};
    a.q = bq2; a.ldq = d; a.k = kv; a.v = (char*)kv + (size_t)d * tsz; a.ldk = a.ldv = kv_all.N;
    a.out = bctx2; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = 1; a.Lk = R;
    a.klen = att_len; a.klen_sb = 1; a.klen_sq = 0;
    if (early) { a.skip_if_ge = st.counters; a.skip_threshold = B; }
    ENG_OK(bofi::launch_attention(a, s));
    }
    { LinOpt o; o.residual = by1; o.ldr = d; o.early = early; o.stats_out = st_b; o.y2 = copy_t(byb);
      ENG_OK(linear(bctx2, dt, d, b_o_src, by2, BOFI_DT_F32, d, B, o, s)); }
    { LinOpt o; o.relu = 1; o.early = early; o.ln_stats = st_b; ENG_OK(linear(stream_t(by2, byb), dt, d, b_w1, bh, dt, cfg.d_ff, B, o, s)); }
    const int w2parts = (cfg.d_ff % (4 * 128) == 0) ? 4 : 1;     // K = d_ff split 4 ways: 4x the workgroups, a quarter of the K loop
    { LinOpt o; o.residual = by2; o.ldr = d; o.early = early; o.splitk = w2parts; ENG_OK(linear(bh, dt, cfg.d_ff, b_w2, by3, BOFI_DT_F32, d, B, o, s)); }
    // heads + bookkeeping, fused with the next iteration's row-0 self-attention
    const int flags = BOUND_HEADS | (update ? (BOUND_UPDATE | BOUND_ATTN) : 0) | (early ? BOUND_EARLY : 0);
    ENG_OK(bound_tail(by3, w2parts, update ? nullptr : ext_syn, update ? nullptr : last, B, flags, len_logp, syn_logp, s));
    return BOFI_OK;
}

// The bounding loop in its dense form (LengthPredictor_UIC.forward TransformerModel.py:357-383 as the reference runs it): per
// iteration the whole bound sequence -- all L = S + 2 rows, pos_embed(syn_embed(extend_phrase_syn)) -- goes through every one of
// the N_len layers (self-attention under tgt_mask, whose rows are key prefixes: BoundState.klen; cross-attention over the
// memory; FFN), then the final norm, the heads and the bookkeeping on row 0 (the tail kernel without its self-attention part).
// Needed for N_len >= 2, where the upper layers read the lower layers' outputs of every visible row; with
// BOFI_BOUND_DENSE=1 it also serves N_len = 1 as a cross-check of the incremental form.
int bofi_engine::enqueue_bound_dense(const int* att_len, int B, int R, hipStream_t s) {
    const int d = cfg.d_model, dt = cfg.dtype, S = cfg.seq_length, M = B * L;
    cur_B = B;
    const void* xa = stream_t(xw, xwb);
    for (int it = 0; it < S; ++it) {
        // input rows: the syntactic table stands where launch_embed_rows takes the word table (no second term)
        ENG_OK(bofi::launch_embed_rows(lut_syn, nullptr, pe, st.ext_syn, nullptr, L, 0, B, L, d, 0, xw, copy_t(xwb), dt, st_w, nullptr, s));
        for (size_t li = 0; li < blay.size(); ++li) {
            auto& l = blay[li];
            { LinOpt o; o.early = true; o.ln_stats = st_w; ENG_OK(linear(xa, dt, d, l.qkv, qkv, dt, 3 * d, M, o, s)); }
            bofi::AttnArgs a{};
            a.q = qkv; a.k = (char*)qkv + (size_t)d * tsz; a.v = (char*)qkv + (size_t)2 * d * tsz;
            a.ldq = a.ldk = a.ldv = 3 * d; a.out = ctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = L; a.Lk = L;
            a.klen = st.klen; a.klen_sb = L; a.klen_sq = 1; a.skip_if_ge = st.counters; a.skip_threshold = B;
            ENG_OK(bofi::launch_attention(a, s));
            { LinOpt o; o.early = true; o.residual = xw; o.ldr = d; o.stats_out = st_w; o.y2 = copy_t(xwb);
              ENG_OK(linear(ctx, dt, d, l.o, xw, BOFI_DT_F32, d, M, o, s)); }
            { LinOpt o; o.early = true; o.ln_stats = st_w; ENG_OK(linear(xa, dt, d, l.q_src, qs, dt, d, M, o, s)); }
            bofi::AttnArgs c{};
            c.q = qs; c.ldq = d;
            c.k = (char*)kv + (size_t)li * 2 * d * tsz; c.v = (char*)kv + ((size_t)li * 2 * d + d) * tsz;
            c.ldk = c.ldv = kv_all.N; c.out = ctx; c.ldo = d; c.dtype = dt; c.B = B; c.H = cfg.heads; c.Lq = L; c.Lk = R;
            c.klen = att_len; c.klen_sb = 1; c.klen_sq = 0; c.skip_if_ge = st.counters; c.skip_threshold = B;
            ENG_OK(bofi::launch_attention(c, s));
            { LinOpt o; o.early = true; o.residual = xw; o.ldr = d; o.stats_out = st_w; o.y2 = copy_t(xwb);
              ENG_OK(linear(ctx, dt, d, l.o_src, xw, BOFI_DT_F32, d, M, o, s)); }
            { LinOpt o; o.early = true; o.relu = 1; o.ln_stats = st_w; ENG_OK(linear(xa, dt, d, l.w1, hdn, dt, cfg.d_ff, M, o, s)); }
            { LinOpt o; o.early = true; o.residual = xw; o.ldr = d; o.stats_out = st_w; o.y2 = copy_t(xwb);
              ENG_OK(linear(hdn, dt, cfg.d_ff, l.w2, xw, BOFI_DT_F32, d, M, o, s)); }
        }
        ENG_OK(bound_tail(xw, 1, nullptr, nullptr, B, BOUND_HEADS | BOUND_UPDATE | BOUND_EARLY, nullptr, nullptr, s, false, 0, L * d));
    }
    return BOFI_OK;
}

int bofi_engine::enqueue_decode(const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags, int64_t* seq,
                                float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn,
                                float* memory_out, int* bound_iters, hipStream_t s) {
    const int S = cfg.seq_length;
    // phases (BOFI_FLAG_PHASE_*): none of the bits = the whole decode; else only the named parts (a pipelining caller enqueues them as separate calls)
    const int ph = flags & (BOFI_FLAG_PHASE_ENCODE | BOFI_FLAG_PHASE_BOUND | BOFI_FLAG_PHASE_FILL);
    const bool do_enc = !ph || (ph & BOFI_FLAG_PHASE_ENCODE), do_bound = !ph || (ph & BOFI_FLAG_PHASE_BOUND), do_fill = !ph || (ph & BOFI_FLAG_PHASE_FILL);
    cur_B = B;
    if (do_enc) ENG_OK(enqueue_encode(feats, feats_dtype, att_len, B, R, memory_out, s));
    // ---- bounding pass (core_NAIC TransformerModel.py:1833-1870)
    if (do_bound) {
    ENG_OK(bofi::launch_bound_init(st, B, L, cfg.pad_idx, cfg.len_idx, s));
    if (bound_dense) {
        ENG_OK(enqueue_bound_dense(att_len, B, R, s));
    } else if (bound_loop_ok(R) && !exp_skip("loop")) {
        // one launch: a workgroup per 16 images runs every iteration and leaves when its images are finished (no iteration budget to enqueue)
        ENG_OK(bound_loop(B, R, att_len, nullptr, nullptr, 1, nullptr, nullptr, s));
    } else {
        ENG_OK(bound_tail(nullptr, 1, nullptr, nullptr, B, BOUND_ATTN, nullptr, nullptr, s));
#ifdef BOFI_EXPERIMENTS
        static const int exp_iters = [] { const char* v = getenv("BOFI_EXP_ITERS"); if (v) fprintf(stderr, "[boficap_hip] BOFI_EXP_ITERS=%s: bounding loop truncated, RESULTS INVALID\n", v); return v ? atoi(v) : 0; }();
#else
        constexpr int exp_iters = 0;                 // (timing experiment, experiments build only)
#endif
        // (bofi_engine_set_bound_iter_cap: a caller that knows how many iterations its captions take enqueues that many + a margin instead of
        // all S -- an iteration past the last live one is five launches that return at once -- and decodes again without the cap when the
        // count of live iterations, *bound_iters, reaches the cap: the loop may then not have ended)
        const int n_iters = exp_iters ? exp_iters : (bound_iter_cap > 0 && bound_iter_cap < S ? bound_iter_cap : S);
        for (int it = 0; it < n_iters; ++it)
            ENG_OK(enqueue_bound_iter(B, R, att_len, st.ext_syn, st.last, 1, nullptr, nullptr, true, s));
    }
    }
    if (do_fill) {
        ENG_OK(enqueue_fill(att_len, B, R, flags, seq, seq_logprob, s));
        ENG_OK(bofi::launch_bound_export(st, B, L, S, phrase_num, phrase_length, phrase_syn, bound_iters, s, live_max, sat_out));
    }
    return BOFI_OK;
}

// ---- filling pass (decode_NA :570-587) on the slot layout in st.ext_syn / st.last and the cross K|V of the preceding
// encode; with refinement the pass is repeated with the previous round's ids as decoder input tokens (the glat_input hook
// of decode_NA :570-574 -- the reference has no refinement loop itself)
int bofi_engine::enqueue_fill(const int* att_len, int B, int R, int flags, int64_t* seq, float* seq_logprob, hipStream_t s) {
    const int d = cfg.d_model, dt = cfg.dtype, S = cfg.seq_length, M = B * S;
    if (exp_skip("filling")) return BOFI_OK;           // (experiments build: the whole filling pass incl. the vocabulary epilogue)
    const int rounds = 1 + ((flags >> BOFI_FLAG_REFINE_SHIFT) & 15);
    const void* xa = stream_t(x_fill, xb_fill);
    float* lg = seq_logprob ? seq_logprob : logits;
    const int gen_pad = BOFI_ENV_INT("BOFI_GEN_PAD", 1);      // developer knob: 0 = in place, one-tile kernel
    const bool gen_rb = gen_pad && logits_pad && fold_rb_ok(gen, M);
    for (int round = 0; round < rounds; ++round) {
    // (round 0 under the row-block family: layer 0's q|k|v rows come out of the (label, position) table with the embedding launch -- BOFI_FILL_QKV_TAB=0: the projection)
    const bool qkv_tab = round == 0 && fill_tab_ready && !dec.empty() && fold_rb_ok(dec[0].qkv, M) && BOFI_ENV_INT("BOFI_FILL_QKV_TAB", 1) != 0 && !exp_skip("qkv");
    ENG_OK(bofi::launch_embed_fill(lut_tok, lut_syn, pe, st.ext_syn, round ? seq : nullptr, B, S, L, d, cfg.bos_idx, x_fill, copy_t(xb_fill), dt,
                                   st_fill, s, qkv_tab ? f_qkv0 : nullptr, qkv_tab ? qkv : nullptr, 3 * d));
    bool proj_made = qkv_tab;                            // this layer's q|k|v came out of the previous layer's feed-forward launch (layer 0: out of the table)
    for (size_t li = 0; li < dec.size(); ++li) {
        auto& l = dec[li];
        if (!proj_made) {
            int rc = fold_linear_rb(x_fill, l.qkv, qkv, 0, 3 * d, M, s);
            if (rc > 0) return rc;
            if (rc < 0) { LinOpt o; o.ln_stats = st_fill; ENG_OK(linear(xa, dt, d, l.qkv, qkv, dt, 3 * d, M, o, s)); } }
        bofi::AttnArgs a{};
        a.q = qkv; a.k = (char*)qkv + (size_t)d * tsz; a.v = (char*)qkv + (size_t)2 * d * tsz;
        a.ldq = a.ldk = a.ldv = 3 * d; a.out = ctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = S; a.Lk = S;
        // syn_mask[i, :, :last-1] = True; strict mode reproduces the stale index of :1872-1873 (quirk Q1)
        a.klen = st.last; a.klen_sb = 1; a.klen_sq = 0; a.klen_bias = -1;
        a.klen_shared_last = (flags & BOFI_FLAG_STRICT_Q1) ? (q1_group > 0 ? q1_group : B) : 0;
        const bool q_rb = fold_rb_ok(l.q_src, M);
        const bool ffn_rb = ffn_sublayer_ok(l.w1, l.w2, M);
        const bool need_copy_out = !(fold_rb_ok(l.qkv, M) && gen_rb);      // (a consumer of this layer's output reads the bf16 copy + statistics: a tiled GEMM)
        bool cross_split = false;
        // (round 5's one-launch form of both attention sublayers, rb_dec_attn_kernel, lost 3 % in flight and left the library in round 6: dev/exp/rb_dec_attn_kernel.inc)
        int rc;
        {                                                    // attention sublayer (+ folded query projection) + attention sublayer
        const bool q_tail = attn_proj_ok(a, l.o, l.q_src);      // the cross-attention's query projection rides the self-attention launch
        rc = q_tail ? attn_sublayer(a, l.o, x_fill, nullptr, nullptr, false, s, &l.q_src, qs, d)
                    : attn_sublayer(a, l.o, x_fill, xb_fill, st_fill, !q_rb, s);   // (the query projection behind it is a folded GEMM)
        if (rc > 0) return rc;
        if (rc < 0) {
            ENG_OK(bofi::launch_attention(a, s));
            LinOpt o; o.residual = x_fill; o.ldr = d; o.stats_out = st_fill; o.y2 = copy_t(xb_fill);
            ENG_OK(linear(ctx, dt, d, l.o, x_fill, BOFI_DT_F32, d, M, o, s));
        }
        if (!(q_tail && rc == 0)) {
        rc = fold_linear_rb(x_fill, l.q_src, qs, 0, d, M, s);
        if (rc > 0) return rc;
        if (rc < 0) { LinOpt o; o.ln_stats = st_fill; ENG_OK(linear(xa, dt, d, l.q_src, qs, dt, d, M, o, s)); }
        }
        bofi::AttnArgs c{};
        c.q = qs; c.ldq = d;
        c.k = (char*)kv + (size_t)(n_len + li) * 2 * d * tsz; c.v = (char*)kv + ((size_t)(n_len + li) * 2 * d + d) * tsz;
        c.ldk = c.ldv = kv_all.N; c.out = ctx; c.ldo = d; c.dtype = dt; c.B = B; c.H = cfg.heads; c.Lq = S; c.Lk = R;
        c.klen = att_len; c.klen_sb = 1; c.klen_sq = 0;
        cross_split = ffn_rb && !need_copy_out && (BOFI_ENV_INT("BOFI_RB_ATTN_SPLIT_WHICH", 3) & 2) && attn_split_ok(c, l.o_src, l.w1, l.w2);      // the cross-attention's core now, its W_o + residual as the head of the feed-forward launch
        rc = cross_split ? bofi::launch_attention(c, s) : attn_sublayer(c, l.o_src, x_fill, xb_fill, st_fill, !ffn_rb, s);
        if (rc > 0) return rc;
        if (rc < 0) {
            ENG_OK(bofi::launch_attention(c, s));
            LinOpt o; o.residual = x_fill; o.ldr = d; o.stats_out = st_fill; o.y2 = copy_t(xb_fill);
            ENG_OK(linear(ctx, dt, d, l.o_src, x_fill, BOFI_DT_F32, d, M, o, s));
        }
        }
        {   // next consumer: the next layer's q|k|v or the generator
            const bool need_copy = need_copy_out;
            proj_made = !need_copy && li + 1 < dec.size() && ffn_proj_ok(l.w1, l.w2, dec[li + 1].qkv, M);
            const Lin* head = cross_split ? &l.o_src : nullptr;
            rc = !ffn_rb ? -1 : proj_made ? ffn_sublayer(l.w1, l.w2, x_fill, nullptr, nullptr, M, s, &dec[li + 1].qkv, qkv, 3 * d, head, ctx)
                                          : ffn_sublayer(l.w1, l.w2, x_fill, need_copy ? xb_fill : nullptr, need_copy ? st_fill : nullptr, M, s, nullptr, nullptr, 0, head, ctx); }
        if (rc > 0) return rc;
        if (rc < 0) {
            { LinOpt o; o.relu = 1; o.ln_stats = st_fill; ENG_OK(linear(xa, dt, d, l.w1, hdn, dt, cfg.d_ff, M, o, s)); }
            { LinOpt o; o.residual = x_fill; o.ldr = d; o.stats_out = st_fill; o.y2 = copy_t(xb_fill);
              ENG_OK(linear(hdn, dt, cfg.d_ff, l.w2, x_fill, BOFI_DT_F32, d, M, o, s)); }
        }
    }
    // ---- vocabulary projection (decoder.norm folded in), log-softmax, greedy pick, pad tail
    // generator.  V = 9 491 is not a whole number of 128-column tiles and its rows are not 16-byte aligned: the bf16 engine runs the GEMM
    // over the zero-padded weight rows into a buffer of pitch gen.Npad (persistent kernel, vector epilogue) and vocab_finalize reads
    // that, writing the log-probs at the caller's pitch V -- the same reads and writes as in place.
    const float* lsrc = nullptr;
    if (gen_pad && logits_pad) {
        int rc = fold_linear_rb(x_fill, gen, logits_pad, 1, gen.Npad, M, s);
        if (rc > 0) return rc;
        if (rc < 0) {
            Lin gp = gen; gp.N = gen.Npad;
            LinOpt o; o.ln_stats = st_fill;
            ENG_OK(linear(xa, dt, d, gp, logits_pad, BOFI_DT_F32, gen.Npad, M, o, s));
        }
        lsrc = logits_pad;
    } else {
        LinOpt o; o.ln_stats = st_fill; ENG_OK(linear(xa, dt, d, gen, lg, BOFI_DT_F32, cfg.vocab, M, o, s));
    }
    // (a round that another one follows: its ids are all the next round reads -- the log-probs it would write are overwritten: not stored)
    const int ids_only_on = BOFI_ENV_INT("BOFI_REFINE_IDS_ONLY", 1);      // developer knob: 0 = every round stores its log-probs
    const int lsm = (flags & BOFI_FLAG_RAW_LOGITS) ? 0 : ((ids_only_on && round + 1 < rounds && lsrc) ? 2 : 1);
    ENG_OK(bofi::launch_vocab_finalize(lg, M, cfg.vocab, S, lsm, st.last, -1, cfg.pad_idx, seq, s, nullptr, nullptr, nullptr, nullptr,
                                       lsrc, gen.Npad, lsm ? row_plogp_out : nullptr, lsm ? row_chosen_out : nullptr));
    }
    return BOFI_OK;
}

// Semi-autoregressive decode (reference core_SAIC TransformerModel.py:1878-1986, greedy): per phrase one bounding step on
// the WORDS emitted so far, then a full decoder pass over all S positions whose new phrase is kept.
int bofi_engine::enqueue_decode_saic(const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags, int64_t* seq,
                                     float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn, int* bound_iters,
                                     hipStream_t s) {
    const int d = cfg.d_model, dt = cfg.dtype, S = cfg.seq_length, M = B * S;
    const int* halt = st.counters + 2;
    // iterations it_first .. it_last of the loop (bofi_engine_set_saic_range).  Every iteration past the last live one returns at once but still
    // costs its ~60 launches' dispatch; a caller that knows how long its captions run enqueues fewer and, when the count of live iterations
    // (bound_iters) says the loop may not be through, the REST in a second call: the loop's state lives in the workspace, so [1, c] then
    // [c + 1, S] is the same computation as [1, S]
    const int it_first = saic_it_begin > 1 ? saic_it_begin : 1, it_last = saic_it_end > 0 && saic_it_end < S ? saic_it_end : S;
    if (it_first == 1) {
    ENG_OK(enqueue_encode(feats, feats_dtype, att_len, B, R, nullptr, s));
    ENG_OK(bofi::launch_saic_init(st, sa, B, L, cfg.pad_idx, cfg.bos_idx, cfg.len_idx, s));
    if (seq_logprob) ENG_OK(bofi::launch_zero_f32(seq_logprob, (size_t)M * cfg.vocab, s));   // seq_logprobs = zeros (:1883); not a memset node
    }
    const void* xwa = stream_t(xw, xwb);
    const void* xa = stream_t(x_fill, xb_fill);
    for (int it = it_first; it <= it_last; ++it) {
        // From the second iteration on only the rows of the phrases placed in this iteration go through the decoder's GEMMs and the
        // vocabulary epilogue (row list; launch_saic_rows): a placed row's input, key set and therefore its K / V in every layer
        // never change again, later rows are never attended, and only the new phrase's rows are copied out (:1968-1977).  The
        // first iteration runs every row: it is the only one in which a row without any key (an image that opened no phrase)
        // can raise the reference's "phrase nan!" return.
        const bool rowlist = saic_cache && it >= 2;
        const int* ri = rowlist ? sa_rows : nullptr;
        const int* rn = rowlist ? sa_nrows : nullptr;
        // ---- bounding step on the words: K|V of all L rows, row-0 query attends keys < phrase_last
        ENG_OK(bofi::launch_embed_rows(lut_tok, nullptr, pe, sa.ext_len, nullptr, L, 0, B, L, d, cfg.bos_idx, xw, copy_t(xwb), dt, st_w, halt, s));
        { LinOpt o; o.halt = true; o.ln_stats = st_w; ENG_OK(linear(xwa, dt, d, b_kv_self, kvs, dt, 2 * d, B * L, o, s)); }
        {
            bofi::AttnArgs a{};
            a.q = b_q0_sa; a.ldq = 0; a.k = kvs; a.v = (char*)kvs + (size_t)d * tsz; a.ldk = a.ldv = 2 * d;
            a.out = bctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = 1; a.Lk = L;
            a.klen = st.last; a.klen_sb = 1; a.klen_sq = 0; a.skip_if_ge = halt; a.skip_threshold = 1;
            ENG_OK(bofi::launch_attention(a, s));
        }
        { LinOpt o; o.residual = b_x0_sa; o.ldr = 0; o.halt = true; o.stats_out = st_b; o.y2 = copy_t(byb);
          ENG_OK(linear(bctx, dt, d, b_o_self, by1, BOFI_DT_F32, d, B, o, s)); }
        int w2parts = 0;
        const int lean_rc = bound_chain_lean(B, R, att_len, halt, 1, s, &w2parts);
        if (lean_rc > 0) return lean_rc;
        if (lean_rc < 0) {
            { LinOpt o; o.halt = true; o.ln_stats = st_b; ENG_OK(linear(stream_t(by1, byb), dt, d, b_q_src, bq2, dt, d, B, o, s)); }
            {
                bofi::AttnArgs a{};
                a.q = bq2; a.ldq = d; a.k = kv; a.v = (char*)kv + (size_t)d * tsz; a.ldk = a.ldv = kv_all.N;
                a.out = bctx2; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = 1; a.Lk = R;
                a.klen = att_len; a.klen_sb = 1; a.klen_sq = 0; a.skip_if_ge = halt; a.skip_threshold = 1;
                ENG_OK(bofi::launch_attention(a, s));
            }
            { LinOpt o; o.residual = by1; o.ldr = d; o.halt = true; o.stats_out = st_b; o.y2 = copy_t(byb);
              ENG_OK(linear(bctx2, dt, d, b_o_src, by2, BOFI_DT_F32, d, B, o, s)); }
            { LinOpt o; o.relu = 1; o.halt = true; o.ln_stats = st_b; ENG_OK(linear(stream_t(by2, byb), dt, d, b_w1, bh, dt, cfg.d_ff, B, o, s)); }
            w2parts = (cfg.d_ff % (4 * 128) == 0) ? 4 : 1;
            { LinOpt o; o.residual = by2; o.ldr = d; o.halt = true; o.splitk = w2parts; ENG_OK(linear(bh, dt, cfg.d_ff, b_w2, by3, BOFI_DT_F32, d, B, o, s)); }
        }
        ENG_OK(bound_tail(by3, w2parts, nullptr, nullptr, B, BOUND_HEADS | BOUND_UPDATE | BOUND_SAIC | BOUND_EARLY, nullptr, nullptr, s, true, it));
        if (flags & BOFI_FLAG_SAIC_LAYOUT_ONLY) {      // the caller draws this phrase's words itself (bofi_engine_saic_put_words before the next call): layout + positions only
            ENG_OK(bofi::launch_saic_halt(st, sa, B, L, it, s));
            continue;
        }
        // ---- decoder pass over all S positions (decode_SA :520-530) with the phrase-block mask as per-row key prefixes
        ENG_OK(bofi::launch_embed_rows(lut_tok, lut_syn, pe, sa.ext_phrase, st.ext_syn, L, 1, B, S, d, cfg.bos_idx, x_fill, copy_t(xb_fill), dt,
                                       st_fill, halt, s));
        if (rowlist) ENG_OK(bofi::launch_saic_rows(st, B, L, S, it, sa_rows, sa_nrows, s));
        // the row-list iterations in bf16 at the reference's width: a layer's six GEMMs as direct-operand row GEMMs (bound_ops.hip: no LDS
        // staging, ~5 us per launch whatever the few hundred rows), query projection + cross-attention as one launch; their row
        // statistics come in 16-column groups (st_fill16)
        const bool lean = rowlist && saic_lean && dt == BOFI_DT_BF16 && d == 512 && cfg.heads == 8 && R <= 64 && cfg.d_ff % 512 == 0;
        const float* fin_stats = st_fill; int fin_groups = 0;
        if (lean) {
            const float* cst = st_fill; int cg = d / 32;               // statistics of the rows' current stream: where, and in how many groups
            auto rg = [&](const void* x, int ldx, const Lin& l, const float* stats, int groups, const float* residual, float* y, void* yb,
                          int ldyb, float* stats_out, int relu) {
                bofi::RowGemmArgs a{};
                a.x = (const uint16_t*)x; a.ldx = ldx; a.w = (const uint16_t*)l.w; a.bias = l.b; a.stats = stats; a.stats_groups = groups;
                a.colsum = stats ? l.cs : nullptr; a.residual = residual; a.ldr = d; a.y = y; a.ldy = d; a.yb = (uint16_t*)yb; a.ldyb = ldyb;
                a.stats_out = stats_out; a.M = M; a.N = l.N; a.K = l.K; a.splitk = 1; a.relu = relu; a.skip_if_ge = halt; a.skip_threshold = 1;
                a.row_idx = ri; a.m_dev = rn;
                return bofi::launch_rowgemm(a, s);
            };
            for (size_t li = 0; li < dec.size(); ++li) {
                auto& l = dec[li];
                void* qkv = qkv_dec[li];
                ENG_OK(rg(xb_fill, d, l.qkv, cst, cg, nullptr, nullptr, qkv, 3 * d, nullptr, 0));
                bofi::AttnArgs a{};
                a.q = qkv; a.k = (char*)qkv + (size_t)d * tsz; a.v = (char*)qkv + (size_t)2 * d * tsz;
                a.ldq = a.ldk = a.ldv = 3 * d; a.out = ctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = S; a.Lk = S;
                a.klen = sa.klen_dec + 1; a.klen_sb = L; a.klen_sq = 1; a.klen_bias = -1; a.skip_if_ge = halt; a.skip_threshold = 1;
                ENG_OK(bofi::launch_attention(a, s));
                ENG_OK(rg(ctx, d, l.o, nullptr, 0, x_fill, x_fill, xb_fill, d, st_fill16, 0));
                cst = st_fill16; cg = d / 16;
                {   bofi::BoundQAttnArgs q{};
                    q.x = (const uint16_t*)xb_fill; q.stats = cst; q.stats_groups = cg; q.wq = (const uint16_t*)l.q_src.w; q.bias = l.q_src.b; q.colsum = l.q_src.cs;
                    q.k = (const uint16_t*)kv + (size_t)(n_len + li) * 2 * d; q.v = q.k + d; q.ldkv = kv_all.N; q.att_len = att_len; q.out = (uint16_t*)ctx;
                    q.B = M; q.R = R; q.d = d; q.H = cfg.heads; q.skip_if_ge = halt; q.skip_threshold = 1;
                    q.row_idx = ri; q.n_rows = rn; q.rows_per_image = S;
                    ENG_OK(bofi::launch_bound_qattn(q, s)); }
                ENG_OK(rg(ctx, d, l.o_src, nullptr, 0, x_fill, x_fill, xb_fill, d, st_fill16, 0));
                ENG_OK(rg(xb_fill, d, l.w1, cst, cg, nullptr, nullptr, hdn, cfg.d_ff, nullptr, 1));
                ENG_OK(rg(hdn, cfg.d_ff, l.w2, nullptr, 0, x_fill, x_fill, xb_fill, d, st_fill16, 0));
            }
            fin_stats = st_fill16; fin_groups = d / 16;
        } else
        for (size_t li = 0; li < dec.size(); ++li) {
            auto& l = dec[li];
            void* qkv = qkv_dec[li];
            { LinOpt o; o.halt = true; o.ln_stats = st_fill; o.row_idx = ri; o.m_dev = rn; ENG_OK(linear(xa, dt, d, l.qkv, qkv, dt, 3 * d, M, o, s)); }
            bofi::AttnArgs a{};
            a.q = qkv; a.k = (char*)qkv + (size_t)d * tsz; a.v = (char*)qkv + (size_t)2 * d * tsz;
            a.ldq = a.ldk = a.ldv = 3 * d; a.out = ctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = S; a.Lk = S;
            // phrase_mask[:, 1:-1, 1:-1]: row t <- row t+1, one key column dropped
            a.klen = sa.klen_dec + 1; a.klen_sb = L; a.klen_sq = 1; a.klen_bias = -1; a.skip_if_ge = halt; a.skip_threshold = 1;
            ENG_OK(bofi::launch_attention(a, s));
            { LinOpt o; o.halt = true; o.residual = x_fill; o.ldr = d; o.stats_out = st_fill; o.y2 = copy_t(xb_fill); o.row_idx = ri; o.m_dev = rn;
              ENG_OK(linear(ctx, dt, d, l.o, x_fill, BOFI_DT_F32, d, M, o, s)); }
            { LinOpt o; o.halt = true; o.ln_stats = st_fill; o.row_idx = ri; o.m_dev = rn; ENG_OK(linear(xa, dt, d, l.q_src, qs, dt, d, M, o, s)); }
            bofi::AttnArgs c{};
            c.q = qs; c.ldq = d;
            c.k = (char*)kv + (size_t)(n_len + li) * 2 * d * tsz; c.v = (char*)kv + ((size_t)(n_len + li) * 2 * d + d) * tsz;
            c.ldk = c.ldv = kv_all.N; c.out = ctx; c.ldo = d; c.dtype = dt; c.B = B; c.H = cfg.heads; c.Lq = S; c.Lk = R;
            c.klen = att_len; c.klen_sb = 1; c.klen_sq = 0; c.skip_if_ge = halt; c.skip_threshold = 1;
            ENG_OK(bofi::launch_attention(c, s));
            { LinOpt o; o.halt = true; o.residual = x_fill; o.ldr = d; o.stats_out = st_fill; o.y2 = copy_t(xb_fill); o.row_idx = ri; o.m_dev = rn;
              ENG_OK(linear(ctx, dt, d, l.o_src, x_fill, BOFI_DT_F32, d, M, o, s)); }
            { LinOpt o; o.halt = true; o.relu = 1; o.ln_stats = st_fill; o.row_idx = ri; o.m_dev = rn; ENG_OK(linear(xa, dt, d, l.w1, hdn, dt, cfg.d_ff, M, o, s)); }
            { LinOpt o; o.halt = true; o.residual = x_fill; o.ldr = d; o.stats_out = st_fill; o.y2 = copy_t(xb_fill); o.row_idx = ri; o.m_dev = rn;
              ENG_OK(linear(hdn, dt, cfg.d_ff, l.w2, x_fill, BOFI_DT_F32, d, M, o, s)); }
        }
        { LinOpt o; o.halt = true; o.ln_stats = fin_stats; o.ln_groups = fin_groups; o.row_idx = ri; o.m_dev = rn;
          ENG_OK(linear(xa, dt, d, gen, logits, BOFI_DT_F32, cfg.vocab, M, o, s)); }
        ENG_OK(bofi::launch_vocab_finalize(logits, M, cfg.vocab, S, (flags & BOFI_FLAG_RAW_LOGITS) ? 0 : 1, nullptr, 0, cfg.pad_idx, tok64, s,
                                           st.counters + 3, halt, ri, rn));
        if (flags & BOFI_FLAG_SAMPLE)          // sample_next_word 'sample' (CaptionModel.py:405-425): the drawn ids feed the next bound step
            ENG_OK(bofi::launch_vocab_sample(logits, M, cfg.vocab, S, 1, sample_temperature, (uint64_t)it * 0x9E3779B97F4A7C15ull, nullptr,
                                             cfg.pad_idx, tok64, s, halt, ri, rn, d_seed));
        ENG_OK(bofi::launch_saic_copy(st, sa, tok64, logits, seq_logprob, B, L, S, cfg.vocab, it, s));
    }
    ENG_OK(bofi::launch_saic_export(st, sa, B, L, S, seq, phrase_num, phrase_length, phrase_syn, bound_iters, s));
    return BOFI_OK;
}

// Replays the captured launch sequence stored under `key`, capturing it first (through `enqueue`, on the engine's capture stream)
// when the key is new.  At most 16 captures are kept per engine (a pipeline over ragged loader batches holds two inputs x a few region buckets).
template <typename F>
static int run_graphed(bofi_engine* e, const std::vector<uintptr_t>& key, hipStream_t s, F enqueue) {
    for (auto& g : e->graphs)
        if (g.key == key) { ENG_HIP(hipGraphLaunch(g.exec, s)); return BOFI_OK; }
    GraphEntry g;
    g.key = key;
    // the capture stream is created lazily so that the run streams of an engine and its forks are
    // created back to back (they then land on different hardware queues)
    if (!e->cap_stream) ENG_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
    ENG_HIP(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue(e->cap_stream);
    hipError_t ee = hipStreamEndCapture(e->cap_stream, &g.graph);
    if (rc != BOFI_OK) { if (g.graph) (void)hipGraphDestroy(g.graph); return rc; }
    if (ee != hipSuccess) return fail(BOFI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
    ENG_HIP(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    if (e->graphs.size() >= 16) {                      // small cache: drop the oldest capture
        ENG_HIP(hipDeviceSynchronize());               // (a launch of it may still be in flight on any of the caller's streams: rare path, wait it out)
        (void)hipGraphExecDestroy(e->graphs.front().exec);
        (void)hipGraphDestroy(e->graphs.front().graph);
        e->graphs.erase(e->graphs.begin());
    }
    e->graphs.push_back(g);
    ENG_HIP(hipGraphLaunch(g.exec, s));
    return BOFI_OK;
}

// ================================================================================================
extern "C" {

int bofi_abi_version(void) { return 4; }
const char* bofi_last_error(void) { return g_err.c_str(); }

int bofi_engine_create(const bofi_config_t* c, bofi_engine_t** out) {
    if (!c || !out) return fail(BOFI_ERR_ARG, "null argument");
    if (c->dtype != BOFI_DT_F32 && c->dtype != BOFI_DT_BF16) return fail(BOFI_ERR_ARG, "dtype must be 0 (f32) or 1 (bf16)");
    if (c->heads <= 0 || c->d_model != c->heads * 64) return fail(BOFI_ERR_ARG, "d_model / heads must be 64");
    if (c->d_model % 128 || c->d_ff % 64 || c->feat % 64) return fail(BOFI_ERR_ARG, "d_model must be a multiple of 128, d_ff and feat of 64");
    if (c->seq_length <= 0 || c->seq_length + 2 > 64) return fail(BOFI_ERR_ARG, "seq_length must be in 1..62");
    if (c->max_batch <= 0 || c->max_regions <= 0 || c->max_regions > 128) return fail(BOFI_ERR_ARG, "max_batch > 0, 0 < max_regions <= 128");
    if (c->vocab <= 0 || c->n_enc < 0 || c->n_dec < 0 || c->head_hidden <= 0) return fail(BOFI_ERR_ARG, "bad layer/vocab counts");
    if (c->n_len < 1 || c->n_len > 8) return fail(BOFI_ERR_ARG, "n_len must be in 1..8");
    auto* e = new bofi_engine();
    e->cfg = *c;
    e->n_len = c->n_len;
    { const char* v = getenv("BOFI_BOUND_DENSE"); e->bound_dense = c->n_len > 1 || (v && atoi(v) != 0); }
    { const char* v = getenv("BOFI_SAIC_CACHE"); e->saic_cache = !v || atoi(v) != 0; }     // 0: every row through the decoder in every iteration
    { const char* v = getenv("BOFI_SAIC_LEAN"); e->saic_lean = !v || atoi(v) != 0; }       // 0: the row-list iterations on the general GEMM / attention kernels
    e->L = c->seq_length + 2;
    e->tsz = c->dtype == BOFI_DT_F32 ? 4 : 2;
    *out = e;
    return BOFI_OK;
}

void bofi_engine_destroy(bofi_engine_t* e) {
    if (!e) return;
    for (auto& g : e->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->run_stream) (void)hipStreamDestroy(e->run_stream);
    for (void* p : e->allocs) (void)hipFree(p);
    if (!e->is_fork) { if (e->rt_dev) (void)hipFree(e->rt_dev); if (e->rt_pin) (void)hipHostFree(e->rt_pin); }
    delete e;
}

int bofi_engine_fork_sized(bofi_engine_t* parent, int max_batch, bofi_engine_t** out) {
    g_err.clear();
    if (!parent || !out) return fail(BOFI_ERR_ARG, "null argument");
    if (!parent->finalized) return fail(BOFI_ERR_STATE, "fork needs a finalized engine");
    if (max_batch < 0) return fail(BOFI_ERR_ARG, "fork: max_batch >= 0 (0 = the parent's)");
    auto* e = new bofi_engine(*parent);          // copies config and every weight pointer
    e->host.clear();
    e->allocs.clear();                           // owns nothing of the parent's
    e->graphs.clear();
    e->cap_stream = nullptr;
    e->run_stream = nullptr;
    e->is_fork = true;
    e->rt_dev = nullptr; e->rt_pin = nullptr; e->rt_bytes = 0; e->rt_cache.clear();
    e->dbg_part = nullptr;
    e->q1_group = 0;                             // per-call knobs are not inherited (the Python handle starts from the defaults)
    e->sample_temperature = 1.0f;
    e->sample_seed = 0;
    e->bound_iter_cap = 0;                       // (a capped parent must not truncate the fork's bounding loop)
    e->row_plogp_out = nullptr; e->row_chosen_out = nullptr;
    e->in_flight = 0;
    e->live_max = nullptr;                       // (a raw device pointer the fork's handle does not keep alive)
    e->sat_out = nullptr;
    e->loop_mode = -1;
    e->saic_it_begin = 1;
    e->saic_it_end = 0;
    e->st = bofi::BoundState{};
    if (max_batch > 0) e->cfg.max_batch = max_batch;          // (the weights do not depend on it: only the workspace is sized by it)
    int rc = e->alloc_workspace();
    if (rc == BOFI_OK && hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(BOFI_ERR_HIP, "hipStreamCreate");
    if (rc != BOFI_OK) { bofi_engine_destroy(e); return rc; }
    *out = e;
    return BOFI_OK;
}

int bofi_engine_fork(bofi_engine_t* parent, bofi_engine_t** out) { return bofi_engine_fork_sized(parent, 0, out); }

int bofi_engine_refresh_device(bofi_engine_t* e, int n, const char* const* names, const float* const* ptrs, const int64_t* numels, void* stream) {
    if (!e || !names || !ptrs || !numels || n <= 0) return fail(BOFI_ERR_ARG, "null argument");
    g_err.clear();
    if (e->is_fork) return fail(BOFI_ERR_STATE, "refresh the parent engine (forks share its weights)");
    if (!e->finalized) return fail(BOFI_ERR_STATE, "engine not finalized: the first load goes through set_weight + finalize");
    hipStream_t s = (hipStream_t)stream;
    const bofi_config_t& c = e->cfg;
    const int d = c.d_model, L = e->L, hh = c.head_hidden;
    std::map<std::string, std::pair<const float*, int64_t>> src;
    for (int i = 0; i < n; ++i) src[names[i]] = {ptrs[i], numels[i]};
    auto get = [&](const std::string& name, int64_t numel) -> const float* {
        auto it = src.find(name);
        if (it == src.end() || !it->second.first) { g_err = "missing weight " + name; return nullptr; }
        if (it->second.second != numel) { g_err = "weight " + name + " has " + std::to_string(it->second.second) + " elements, expected " + std::to_string(numel); return nullptr; }
        return it->second.first;
    };
    // Every Linear's repack (stack, fold the pre-norm, cast, column sums), the fragment-major copies and the small float32 vectors as THREE launches
    // over descriptor tables in device memory (they were ~230 launches and copies: 1.5 ms of a training step that decodes with its current weights).
    // The tables hold the caller's pointers: uploaded again only when one of them changed.
    std::vector<bofi::PackLinDesc> lt;
    std::vector<bofi::CopyDesc> ct;
    int rows = 0, fblocks = 0, cblocks = 0;
    auto add_copy = [&](const float* src, float* dst, int n) { ct.push_back(bofi::CopyDesc{src, dst, n, cblocks}); cblocks += (n + 255) / 256; };
    for (const auto& r : e->lin_recipes) {
        bofi::PackLinDesc dsc{};
        bofi::PackLinArgs& a = dsc.a;
        a.nsrc = (int)r.prefixes.size(); a.n_each = r.n_each; a.K = r.K;
        if (a.nsrc > 16) return fail(BOFI_ERR_STATE, "too many stacked matrices");
        for (int i = 0; i < a.nsrc; ++i) {
            a.w[i] = get(r.prefixes[i] + ".weight", (int64_t)r.n_each * r.K);
            a.b[i] = get(r.prefixes[i] + ".bias", r.n_each);
            if (!a.w[i] || !a.b[i]) return BOFI_ERR_STATE;
        }
        if (!r.fold.empty()) {
            a.gain = get(r.fold + ".a_2", r.K);
            a.bln = get(r.fold + ".b_2", r.K);
            if (!a.gain || !a.bln) return BOFI_ERR_STATE;
        }
        a.bout = r.out->b; a.cs = r.out->cs;
        if (!r.out->w || !a.bout || (a.gain && !a.cs)) return fail(BOFI_ERR_STATE, "refresh: a packed weight is missing");
        dsc.wout = r.out->w; dsc.wp = r.out->wp; dsc.Npad = r.out->Npad;
        dsc.row0 = rows; rows += (a.n_each * a.nsrc + 3) / 4 * 4;
        dsc.blk0 = fblocks; if (dsc.wp) fblocks += (int)(((size_t)dsc.Npad * r.K / 8 + 255) / 256);
        lt.push_back(dsc);
    }
    for (const auto& r : e->norm_recipes) {
        const float *g = get(r.prefix + ".a_2", r.d), *b = get(r.prefix + ".b_2", r.d);
        if (!g || !b) return BOFI_ERR_STATE;
        add_copy(g, r.out->g, r.d);
        add_copy(b, r.out->b, r.d);
    }
    const float *ls = get("model.syn_embed.lut.weight", (int64_t)10 * d), *lt_w = get("model.tgt_embed.lut.weight", (int64_t)c.vocab * d);
    if (!ls || !lt_w) return BOFI_ERR_STATE;
    add_copy(ls, e->lut_syn, 10 * d);
    const float *lw1, *lb1, *sw1, *sb1;
    {
        const std::string lp = "model.length_predictor";
        lw1 = get(lp + ".Length_classifier1.weight", (int64_t)hh * d); lb1 = get(lp + ".Length_classifier1.bias", hh);
        sw1 = get(lp + ".Syntactic_classifier1.weight", (int64_t)hh * d); sb1 = get(lp + ".Syntactic_classifier1.bias", hh);
        const float *lw2 = get(lp + ".Length_classifier2.weight", (int64_t)20 * hh), *lb2 = get(lp + ".Length_classifier2.bias", 20);
        const float *sw2 = get(lp + ".Syntactic_classifier2.weight", (int64_t)10 * hh), *sb2 = get(lp + ".Syntactic_classifier2.bias", 10);
        if (!lw1 || !lb1 || !sw1 || !sb1 || !lw2 || !lb2 || !sw2 || !sb2) return BOFI_ERR_STATE;
        auto& h = e->heads;
        add_copy(lw2, const_cast<float*>(h.len_w2), 20 * hh); add_copy(lb2, const_cast<float*>(h.len_b2), 20);
        add_copy(sw2, const_cast<float*>(h.syn_w2), 10 * hh); add_copy(sb2, const_cast<float*>(h.syn_b2), 10);
    }
    {   // [PackLinDesc x n][CopyDesc x m] in one buffer
        const size_t lb = lt.size() * sizeof(bofi::PackLinDesc), cb = ct.size() * sizeof(bofi::CopyDesc), bytes = lb + cb;
        std::vector<char> img(bytes);
        memcpy(img.data(), lt.data(), lb);
        memcpy(img.data() + lb, ct.data(), cb);
        if (img != e->rt_cache) {
            // (rare path: the pack kernels of a previous refresh -- possibly issued on ANOTHER stream -- may still be reading the table, and its upload the
            // pinned buffer: the whole device is waited out before either is touched)
            ENG_HIP(hipDeviceSynchronize());
            if (bytes > e->rt_bytes) {
                if (e->rt_dev) (void)hipFree(e->rt_dev);
                if (e->rt_pin) (void)hipHostFree(e->rt_pin);
                e->rt_dev = e->rt_pin = nullptr; e->rt_bytes = 0;
                ENG_HIP(hipMalloc(&e->rt_dev, bytes));
                ENG_HIP(hipHostMalloc(&e->rt_pin, bytes, hipHostMallocDefault));
                e->rt_bytes = bytes;
            }
            memcpy(e->rt_pin, img.data(), bytes);
            ENG_HIP(hipMemcpyAsync(e->rt_dev, e->rt_pin, bytes, hipMemcpyHostToDevice, s));
            e->rt_cache.swap(img);
        }
        const auto* ltab = static_cast<const bofi::PackLinDesc*>(e->rt_dev);
        const auto* ctab = reinterpret_cast<const bofi::CopyDesc*>(static_cast<const char*>(e->rt_dev) + lb);
        ENG_OK(bofi::launch_pack_lin_multi(ltab, (int)lt.size(), rows, c.dtype, s));
        ENG_OK(bofi::launch_pack_frag_multi(ltab, (int)lt.size(), fblocks, s));
        ENG_OK(bofi::launch_copy_multi(ctab, (int)ct.size(), cblocks, s));
    }
    ENG_HIP(hipMemcpyAsync(e->lut_tok, lt_w, (size_t)c.vocab * d * 4, hipMemcpyDeviceToDevice, s));
    ENG_OK(bofi::launch_pack_heads(lw1, sw1, lb1, sb1, const_cast<float*>(e->heads.w1t), const_cast<float*>(e->heads.b1), d, hh, s));
    // the bound layer's input-independent tables (as at the end of finalize)
    ENG_OK(bofi::launch_bound_table(e->lut_syn, e->lut_tok, e->pe, e->d_xt, e->b_x0, e->b_x0_sa, L, d, c.len_idx, s));
    bofi_engine::LinOpt o; o.ln = &e->b_n0;
    ENG_OK(e->linear(e->d_xt, BOFI_DT_F32, d, e->t_kvself, e->b_kvtab, c.dtype, 2 * d, L * 10, o, s));
    ENG_OK(e->linear(e->b_x0, BOFI_DT_F32, d, e->t_qself, e->b_q0, c.dtype, d, 1, o, s));
    ENG_OK(e->linear(e->b_x0_sa, BOFI_DT_F32, d, e->t_qself, e->b_q0_sa, c.dtype, d, 1, o, s));
    ENG_OK(e->derive_bound_tables(s));
    if (e->loop_config_ok()) {
        const std::string bl = "model.length_predictor.LengthPredictor.0", lp = "model.length_predictor";
        const int dff = c.d_ff;
        bofi_engine::BoundSrc b{};
        b.wq_self = get(bl + ".self_attn.linears.0.weight", (int64_t)d * d); b.bq_self = get(bl + ".self_attn.linears.0.bias", d);
        b.wk_self = get(bl + ".self_attn.linears.1.weight", (int64_t)d * d); b.bk_self = get(bl + ".self_attn.linears.1.bias", d);
        b.wv_self = get(bl + ".self_attn.linears.2.weight", (int64_t)d * d); b.bv_self = get(bl + ".self_attn.linears.2.bias", d);
        b.n0g = get(bl + ".sublayer.0.norm.a_2", d); b.n0b = get(bl + ".sublayer.0.norm.b_2", d);
        b.wo_self = get(bl + ".self_attn.linears.3.weight", (int64_t)d * d);
        b.wq_src = get(bl + ".src_attn.linears.0.weight", (int64_t)d * d); b.n1g = get(bl + ".sublayer.1.norm.a_2", d);
        b.wo_src = get(bl + ".src_attn.linears.3.weight", (int64_t)d * d);
        b.w1 = get(bl + ".ff.w_1.weight", (int64_t)dff * d); b.n2g = get(bl + ".sublayer.2.norm.a_2", d);
        b.w2 = get(bl + ".ff.w_2.weight", (int64_t)d * dff);
        b.lw1 = lw1; b.sw1 = sw1; b.hng = get(lp + ".norm.a_2", d);
        if (!b.wq_self || !b.bq_self || !b.wk_self || !b.bk_self || !b.wv_self || !b.bv_self || !b.n0g || !b.n0b || !b.wo_self || !b.wq_src || !b.n1g || !b.wo_src ||
            !b.w1 || !b.n2g || !b.w2 || !b.hng)
            return BOFI_ERR_STATE;
        ENG_OK(e->derive_bound_f16(b, s));
    }
    e->host.clear();                          // the host copies are stale now; a later finalize needs set_weight again
    return BOFI_OK;
}

// Developer aid: copy a workspace buffer of the bounding iteration into user memory (device to device, on `stream`).
int bofi_engine_debug_copy(bofi_engine_t* e, const char* name, void* dst, int64_t bytes, void* stream) {
    if (!e || !name || !dst) return fail(BOFI_ERR_ARG, "null argument");
    const std::string n = name;
    const void* src = n == "by1" ? (const void*)e->by1 : n == "byb" ? (const void*)e->byb : n == "st_b" ? (const void*)e->st_b :
                      n == "bq2" ? (const void*)e->bq2 : n == "bctx2" ? (const void*)e->bctx2 : n == "by2" ? (const void*)e->by2 :
                      n == "bh" ? (const void*)e->bh : n == "by3" ? (const void*)e->by3 : n == "dbg_part" ? (const void*)e->dbg_part :
                      n == "counters" ? (const void*)e->st.counters : nullptr;      // ("counters": int32 [8] of the last decode: [5] = groups whose bounding loop ran as a PAIR of workgroups)
    if (!src) return fail(BOFI_ERR_ARG, "unknown buffer");
    ENG_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BOFI_OK;
}

int bofi_engine_set_q1_group(bofi_engine_t* e, int group) {
    if (!e || group < 0) return fail(BOFI_ERR_ARG, "group must be >= 0");
    e->q1_group = group;
    return BOFI_OK;
}

int bofi_engine_set_decodes_in_flight(bofi_engine_t* e, int n) {
    g_err.clear();
    if (!e || n < 0) return fail(BOFI_ERR_ARG, "decodes in flight: >= 0 (0 = unknown)");
    e->in_flight = n;
    return BOFI_OK;
}

int bofi_engine_bound_loop_active(bofi_engine_t* e, int R) { return e && e->finalized && e->bound_loop_ok(R) ? 1 : 0; }

int bofi_engine_set_bound_iter_cap(bofi_engine_t* e, int cap) {
    g_err.clear();
    if (!e) return fail(BOFI_ERR_ARG, "null engine");
    if (cap < 0) return fail(BOFI_ERR_ARG, "bound iteration cap: >= 0 (0 = seq_length)");
    e->bound_iter_cap = cap;
    return BOFI_OK;
}

int bofi_engine_set_live_iterations_max(bofi_engine_t* e, int* live_max) {
    g_err.clear();
    if (!e) return fail(BOFI_ERR_ARG, "null engine");
    e->live_max = live_max;
    return BOFI_OK;
}

int bofi_engine_set_saturation_out(bofi_engine_t* e, int* word) {
    g_err.clear();
    if (!e) return fail(BOFI_ERR_ARG, "null engine");
    e->sat_out = word;
    return BOFI_OK;
}

int bofi_engine_set_bound_loop(bofi_engine_t* e, int mode) {
    g_err.clear();
    if (!e || (mode != -1 && mode != 0 && mode != 2)) return fail(BOFI_ERR_ARG, "bound loop mode: -1 (environment / hint), 0 (five-launch iterations) or 2 (persistent loop kernel)");
    e->loop_mode = mode;
    return BOFI_OK;
}

int bofi_engine_set_saic_range(bofi_engine_t* e, int it_begin, int it_end) {
    g_err.clear();
    if (!e) return fail(BOFI_ERR_ARG, "null engine");
    if (it_begin < 1 || it_begin > e->cfg.seq_length || it_end < 0 || (it_end > 0 && it_end < it_begin)) return fail(BOFI_ERR_ARG, "saic range: 1 <= begin <= end <= seq_length, or end 0");
    e->saic_it_begin = it_begin;
    e->saic_it_end = it_end;
    return BOFI_OK;
}

int bofi_engine_saic_put_words(bofi_engine_t* e, const int64_t* seq, int B, void* stream) {
    g_err.clear();
    if (!e || !seq) return fail(BOFI_ERR_ARG, "null argument");
    if (B < 1 || B > e->cfg.max_batch || B != e->cur_B) return fail(BOFI_ERR_ARG, "saic_put_words: B must be the batch of the decode it continues");
    return bofi::launch_saic_put_words(e->st, e->sa, seq, B, e->L, e->cfg.seq_length, (hipStream_t)stream);
}

int bofi_engine_set_sampling(bofi_engine_t* e, float temperature, uint64_t seed) {
    if (!e || !(temperature > 0.f)) return fail(BOFI_ERR_ARG, "temperature must be positive");
    e->sample_temperature = temperature;
    e->sample_seed = seed;
    return BOFI_OK;
}

int bofi_engine_set_row_stats_out(bofi_engine_t* e, float* row_plogp, float* row_chosen) {
    if (!e || (!row_plogp != !row_chosen)) return fail(BOFI_ERR_ARG, "row statistics: both buffers or none");
    e->row_plogp_out = row_plogp;
    e->row_chosen_out = row_chosen;
    return BOFI_OK;
}

const float* bofi_engine_logprob(bofi_engine_t* e) {
    return e ? e->logits : nullptr;
}

void* bofi_engine_stream(bofi_engine_t* e) {
    if (!e) return nullptr;
    if (!e->run_stream && hipStreamCreateWithFlags(&e->run_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return (void*)e->run_stream;
}

int bofi_engine_set_weight(bofi_engine_t* e, const char* name, const float* data, int64_t numel) {
    if (!e || !name || !data || numel < 0) return fail(BOFI_ERR_ARG, "null argument");
    if (e->is_fork) return fail(BOFI_ERR_STATE, "weights belong to the parent engine");
    e->host[name].assign(data, data + numel);
    e->finalized = false;
    return BOFI_OK;
}

int bofi_engine_finalize(bofi_engine_t* e) {
    if (!e) return fail(BOFI_ERR_ARG, "null engine");
    g_err.clear();
    if (e->is_fork) return fail(BOFI_ERR_STATE, "finalize the parent engine, then fork again");
    const bofi_config_t& c = e->cfg;
    const int d = c.d_model, dff = c.d_ff, L = e->L, hh = c.head_hidden;
    // drop everything from a previous finalize (weights may have changed)
    for (auto& g : e->graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
    e->graphs.clear();
    ENG_HIP(hipDeviceSynchronize());
    for (void* p : e->allocs) (void)hipFree(p);
    e->allocs.clear();
    e->enc.assign(c.n_enc, EncLayer());
    e->dec.assign(c.n_dec, DecLayer());
    e->lin_recipes.clear();
    e->norm_recipes.clear();

    auto S = [](const char* fmt, int a, int b = 0) { char buf[256]; std::snprintf(buf, sizeof buf, fmt, a, b); return std::string(buf); };
    ENG_OK(e->make_lin(&e->att_embed, {"att_embed.0"}, d, c.feat));
    for (int l = 0; l < c.n_enc; ++l) {
        auto& E = e->enc[l];
        const std::string p = S("model.encoder.layers.%d", l);
        ENG_OK(e->make_lin(&E.qkv, {p + ".self_attn.linears.0", p + ".self_attn.linears.1", p + ".self_attn.linears.2"}, d, d, p + ".sublayer.0.norm", 0, true));
        ENG_OK(e->make_lin(&E.o, {p + ".self_attn.linears.3"}, d, d, "", 0, true));
        ENG_OK(e->make_lin(&E.w1, {p + ".feed_forward.w_1"}, dff, d, p + ".sublayer.1.norm", 0, true));
        ENG_OK(e->make_lin(&E.w2, {p + ".feed_forward.w_2"}, d, dff, "", 0, true));
        ENG_OK(e->make_norm(&E.n0, p + ".sublayer.0.norm", d));
        ENG_OK(e->make_norm(&E.n1, p + ".sublayer.1.norm", d));
    }
    ENG_OK(e->make_norm(&e->enc_norm, "model.encoder.norm", d));
    const std::string bl = "model.length_predictor.LengthPredictor.0";
    std::vector<std::string> kvs;                 // cross-attention K|V: the bound layers first, then the decoder layers
    for (int l = 0; l < e->n_len; ++l) {
        const std::string p = S("model.length_predictor.LengthPredictor.%d", l);
        kvs.push_back(p + ".src_attn.linears.1");
        kvs.push_back(p + ".src_attn.linears.2");
    }
    e->blay.assign(e->bound_dense ? e->n_len : 0, DecLayer());
    for (size_t l = 0; l < e->blay.size(); ++l) {
        auto& D = e->blay[l];
        const std::string p = S("model.length_predictor.LengthPredictor.%d", (int)l);
        ENG_OK(e->make_lin(&D.qkv, {p + ".self_attn.linears.0", p + ".self_attn.linears.1", p + ".self_attn.linears.2"}, d, d, p + ".sublayer.0.norm"));
        ENG_OK(e->make_lin(&D.o, {p + ".self_attn.linears.3"}, d, d));
        ENG_OK(e->make_lin(&D.q_src, {p + ".src_attn.linears.0"}, d, d, p + ".sublayer.1.norm"));
        ENG_OK(e->make_lin(&D.o_src, {p + ".src_attn.linears.3"}, d, d));
        ENG_OK(e->make_lin(&D.w1, {p + ".ff.w_1"}, dff, d, p + ".sublayer.2.norm"));
        ENG_OK(e->make_lin(&D.w2, {p + ".ff.w_2"}, d, dff));
    }
    for (int l = 0; l < c.n_dec; ++l) {
        auto& D = e->dec[l];
        const std::string p = S("model.decoder.layers.%d", l);
        ENG_OK(e->make_lin(&D.qkv, {p + ".self_attn.linears.0", p + ".self_attn.linears.1", p + ".self_attn.linears.2"}, d, d, p + ".sublayer.0.norm", 0, true));
        ENG_OK(e->make_lin(&D.o, {p + ".self_attn.linears.3"}, d, d, "", 0, true));
        ENG_OK(e->make_lin(&D.q_src, {p + ".src_attn.linears.0"}, d, d, p + ".sublayer.1.norm", 0, true));
        ENG_OK(e->make_lin(&D.o_src, {p + ".src_attn.linears.3"}, d, d, "", 0, true));
        ENG_OK(e->make_lin(&D.w1, {p + ".feed_forward.w_1"}, dff, d, p + ".sublayer.2.norm", 0, true));
        ENG_OK(e->make_lin(&D.w2, {p + ".feed_forward.w_2"}, d, dff, "", 0, true));
        ENG_OK(e->make_norm(&D.n0, p + ".sublayer.0.norm", d));
        ENG_OK(e->make_norm(&D.n1, p + ".sublayer.1.norm", d));
        ENG_OK(e->make_norm(&D.n2, p + ".sublayer.2.norm", d));
        kvs.push_back(p + ".src_attn.linears.1");
        kvs.push_back(p + ".src_attn.linears.2");
    }
    ENG_OK(e->make_norm(&e->dec_norm, "model.decoder.norm", d));
    ENG_OK(e->make_lin(&e->kv_all, kvs, d, d, "model.encoder.norm", 0, true));
    ENG_OK(e->make_lin(&e->gen, {"model.generator.proj"}, c.vocab, d, "model.decoder.norm", 128, true));      // padded to whole 128-column tiles (the persistent GEMM)
    {
        const auto* ls = e->get("model.syn_embed.lut.weight", (size_t)10 * d);
        const auto* lt = e->get("model.tgt_embed.lut.weight", (size_t)c.vocab * d);
        auto it = e->host.find("model.pos_embed.pe");
        if (!ls || !lt) return BOFI_ERR_STATE;
        if (it == e->host.end() || it->second.size() < (size_t)L * d) return fail(BOFI_ERR_STATE, "missing weight model.pos_embed.pe");
        ENG_OK(e->upload_f32(&e->lut_syn, *ls));
        ENG_OK(e->upload_f32(&e->lut_tok, *lt));
        std::vector<float> pe(it->second.begin(), it->second.begin() + (size_t)L * d);
        ENG_OK(e->upload_f32(&e->pe, pe));
    }
    // bound layer
    ENG_OK(e->make_lin(&e->b_o_self, {bl + ".self_attn.linears.3"}, d, d));
    ENG_OK(e->make_lin(&e->b_q_src, {bl + ".src_attn.linears.0"}, d, d, bl + ".sublayer.1.norm"));
    ENG_OK(e->make_lin(&e->b_o_src, {bl + ".src_attn.linears.3"}, d, d));
    ENG_OK(e->make_lin(&e->b_w1, {bl + ".ff.w_1"}, dff, d, bl + ".sublayer.2.norm"));
    ENG_OK(e->make_lin(&e->b_w2, {bl + ".ff.w_2"}, d, dff));
    ENG_OK(e->make_norm(&e->b_n0, bl + ".sublayer.0.norm", d));
    ENG_OK(e->make_norm(&e->b_n1, bl + ".sublayer.1.norm", d));
    ENG_OK(e->make_norm(&e->b_n2, bl + ".sublayer.2.norm", d));
    {
        const std::string lp = "model.length_predictor";
        Norm& nf = e->head_norm;
        ENG_OK(e->make_norm(&nf, lp + ".norm", d));
        const auto *lw1 = e->get(lp + ".Length_classifier1.weight", (size_t)hh * d), *lb1 = e->get(lp + ".Length_classifier1.bias", hh);
        const auto *sw1 = e->get(lp + ".Syntactic_classifier1.weight", (size_t)hh * d), *sb1 = e->get(lp + ".Syntactic_classifier1.bias", hh);
        const auto *lw2 = e->get(lp + ".Length_classifier2.weight", (size_t)20 * hh), *lb2 = e->get(lp + ".Length_classifier2.bias", 20);
        const auto *sw2 = e->get(lp + ".Syntactic_classifier2.weight", (size_t)10 * hh), *sb2 = e->get(lp + ".Syntactic_classifier2.bias", 10);
        if (!lw1 || !lb1 || !sw1 || !sb1 || !lw2 || !lb2 || !sw2 || !sb2) return BOFI_ERR_STATE;
        std::vector<float> w1((size_t)d * 2 * hh), b1(*lb1);          // transposed: [d][2*hh]
        for (int j = 0; j < hh; ++j)
            for (int k = 0; k < d; ++k) {
                w1[(size_t)k * 2 * hh + j] = (*lw1)[(size_t)j * d + k];
                w1[(size_t)k * 2 * hh + hh + j] = (*sw1)[(size_t)j * d + k];
            }
        b1.insert(b1.end(), sb1->begin(), sb1->end());
        float *p_w1, *p_b1, *p_lw2, *p_lb2, *p_sw2, *p_sb2;
        ENG_OK(e->upload_f32(&p_w1, w1)); ENG_OK(e->upload_f32(&p_b1, b1));
        ENG_OK(e->upload_f32(&p_lw2, *lw2)); ENG_OK(e->upload_f32(&p_lb2, *lb2));
        ENG_OK(e->upload_f32(&p_sw2, *sw2)); ENG_OK(e->upload_f32(&p_sb2, *sb2));
        ENG_OK(e->dalloc((char**)&e->b_w1p, (size_t)d * 2 * hh, e->tsz));
        e->heads = bofi::BoundHeadWeights{nf.g, nf.b, p_w1, p_b1, p_lw2, p_lb2, p_sw2, p_sb2, e->b_w1p};
        e->loop_ready = false;
        if (e->loop_config_ok()) {      // the persistent loop kernel's operands (filled by derive_bound_f16 below)
            ENG_OK(e->make_lin(&e->b_heads, {lp + ".Length_classifier1", lp + ".Syntactic_classifier1"}, hh, d, lp + ".norm", 256));
            for (Lin* l : {&e->b_o_self, &e->b_q_src, &e->b_o_src, &e->b_w1, &e->b_w2, &e->b_heads}) ENG_OK(e->dalloc((char**)&l->wp16, (size_t)l->Npad * l->K, 2));
            ENG_OK(e->dalloc(&e->b_q0_32, (size_t)d)); ENG_OK(e->dalloc(&e->b_sctab, (size_t)L * 10 * c.heads)); ENG_OK(e->dalloc(&e->b_vtab, (size_t)L * 10 * d));
            ENG_OK(e->dalloc(&e->b_wsat, 4));
        }
        if (e->rb_ok()) {               // the filling pass's layer-0 q|k|v by (label, position) (derive_fill_table)
            ENG_OK(e->dalloc((char**)&e->f_qkv0, (size_t)10 * c.seq_length * 3 * d, e->tsz));
            ENG_OK(e->dalloc(&e->f_x0, (size_t)10 * c.seq_length * d));
            ENG_OK(e->dalloc((char**)&e->f_syn, (size_t)10 * L, sizeof(int)));
            std::vector<int> pat((size_t)10 * L);
            for (int b = 0; b < 10; ++b) for (int t = 0; t < L; ++t) pat[(size_t)b * L + t] = b;
            ENG_HIP(hipMemcpy(e->f_syn, pat.data(), pat.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }

    e->n_weight_allocs = e->allocs.size();      // everything allocated so far is weights (shared with forks)
    ENG_OK(e->alloc_workspace());

    // input-independent tables of the bound layer: layer input at (position p, label s) is
    // lut_syn[s]*sqrt(d) + pe[p]; K|V of all L*10 rows and the query of row 0 ([LEN] at position 0)
    {
        const auto& ls = e->host["model.syn_embed.lut.weight"];
        const auto& pe = e->host["model.pos_embed.pe"];
        const float sq = (float)std::sqrt((double)d);
        std::vector<float> xt((size_t)L * 10 * d);
        for (int p = 0; p < L; ++p)
            for (int s = 0; s < 10; ++s)
                for (int k = 0; k < d; ++k) xt[((size_t)p * 10 + s) * d + k] = ls[(size_t)s * d + k] * sq + pe[(size_t)p * d + k];
        std::vector<float> x0(xt.begin() + (size_t)c.len_idx * d, xt.begin() + (size_t)(c.len_idx + 1) * d);
        float*& d_xt = e->d_xt;
        Lin &kvself = e->t_kvself, &qself = e->t_qself;
        ENG_OK(e->upload_f32(&d_xt, xt));
        ENG_OK(e->upload_f32(&e->b_x0, x0));
        ENG_OK(e->make_lin(&kvself, {bl + ".self_attn.linears.1", bl + ".self_attn.linears.2"}, d, d));
        ENG_OK(e->make_lin(&qself, {bl + ".self_attn.linears.0"}, d, d));
        ENG_OK(e->dalloc((char**)&e->b_kvtab, (size_t)L * 10 * 2 * d, e->tsz));
        ENG_OK(e->dalloc((char**)&e->b_q0, (size_t)d, e->tsz));
        bofi_engine::LinOpt o; o.ln = &e->b_n0;
        ENG_OK(e->linear(d_xt, BOFI_DT_F32, d, kvself, e->b_kvtab, c.dtype, 2 * d, L * 10, o, nullptr));
        ENG_OK(e->linear(e->b_x0, BOFI_DT_F32, d, qself, e->b_q0, c.dtype, d, 1, o, nullptr));
        // SAIC: position 0 of the bound input is tgt_embed([LEN]) (TransformerModel.py:1903, 515-518)
        const auto& lt = e->host["model.tgt_embed.lut.weight"];
        std::vector<float> x0s(d);
        for (int k = 0; k < d; ++k) x0s[k] = lt[(size_t)c.len_idx * d + k] * sq + pe[k];
        ENG_OK(e->upload_f32(&e->b_x0_sa, x0s));
        ENG_OK(e->dalloc((char**)&e->b_q0_sa, (size_t)d, e->tsz));
        ENG_OK(e->linear(e->b_x0_sa, BOFI_DT_F32, d, qself, e->b_q0_sa, c.dtype, d, 1, o, nullptr));
        ENG_OK(e->make_lin(&e->b_kv_self, {bl + ".self_attn.linears.1", bl + ".self_attn.linears.2"}, d, d, bl + ".sublayer.0.norm"));
        ENG_OK(e->dalloc((char**)&e->b_votab, (size_t)L * 10 * c.heads * d, e->tsz));
        ENG_OK(e->dalloc(&e->b_x0b, (size_t)d));
        ENG_OK(e->derive_bound_tables(nullptr));
        ENG_HIP(hipDeviceSynchronize());
    }
    if (e->loop_config_ok()) {      // fp16 copies and float32 tables of the bounding layer from the float32 parameters (temporaries on the device)
        std::vector<void*> tmp;
        auto up = [&](const std::string& name, const float** out) -> int {
            auto it = e->host.find(name);
            if (it == e->host.end()) return fail(BOFI_ERR_STATE, "missing weight " + name);
            void* q = nullptr;
            ENG_HIP(hipMalloc(&q, it->second.size() * 4));
            tmp.push_back(q);
            ENG_HIP(hipMemcpy(q, it->second.data(), it->second.size() * 4, hipMemcpyHostToDevice));
            *out = (const float*)q;
            return BOFI_OK;
        };
        bofi_engine::BoundSrc b{};
        const std::string lp = "model.length_predictor";
        int rc = BOFI_OK;
        const std::pair<std::string, const float**> items[] = {
            {bl + ".self_attn.linears.0.weight", &b.wq_self}, {bl + ".self_attn.linears.0.bias", &b.bq_self}, {bl + ".self_attn.linears.1.weight", &b.wk_self},
            {bl + ".self_attn.linears.1.bias", &b.bk_self}, {bl + ".self_attn.linears.2.weight", &b.wv_self}, {bl + ".self_attn.linears.2.bias", &b.bv_self},
            {bl + ".sublayer.0.norm.a_2", &b.n0g}, {bl + ".sublayer.0.norm.b_2", &b.n0b}, {bl + ".self_attn.linears.3.weight", &b.wo_self},
            {bl + ".src_attn.linears.0.weight", &b.wq_src}, {bl + ".sublayer.1.norm.a_2", &b.n1g}, {bl + ".src_attn.linears.3.weight", &b.wo_src},
            {bl + ".ff.w_1.weight", &b.w1}, {bl + ".sublayer.2.norm.a_2", &b.n2g}, {bl + ".ff.w_2.weight", &b.w2},
            {lp + ".Length_classifier1.weight", &b.lw1}, {lp + ".Syntactic_classifier1.weight", &b.sw1}, {lp + ".norm.a_2", &b.hng}};
        for (const auto& it : items) if (rc == BOFI_OK) rc = up(it.first, it.second);
        if (rc == BOFI_OK) rc = e->derive_bound_f16(b, nullptr);
        (void)hipDeviceSynchronize();
        for (void* q : tmp) (void)hipFree(q);
        if (rc != BOFI_OK) return rc;
    }
    // NB: hardware-queue assignment follows stream creation order; the capture stream is created here (and in
    // fork) because that order measured best with 4 decodes in flight on torch's pooled streams
    if (!e->cap_stream) ENG_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
    e->finalized = true;
    return BOFI_OK;
}

static int check_call(bofi_engine_t* e, int B, int R) {
    if (!e) return fail(BOFI_ERR_ARG, "null engine");
    if (!e->finalized) return fail(BOFI_ERR_STATE, "engine not finalized");
    if (B <= 0 || B > e->cfg.max_batch) return fail(BOFI_ERR_ARG, "batch outside 1..max_batch");
    if (R <= 0 || R > e->cfg.max_regions) return fail(BOFI_ERR_ARG, "regions outside 1..max_regions");
    return BOFI_OK;
}
static int check_feats(bofi_engine_t* e, const void* feats, int feats_dtype) {
    if (!feats) return fail(BOFI_ERR_ARG, "null att_feats");
    if (feats_dtype != BOFI_DT_F32 && feats_dtype != e->cfg.dtype) return fail(BOFI_ERR_ARG, "att_feats must be float32 or the engine's compute dtype");
    return BOFI_OK;
}

int bofi_engine_encode(bofi_engine_t* e, const void* feats, int feats_dtype, const int* att_len, int B, int R,
                       float* memory_out, void* stream) {
    g_err.clear();
    ENG_OK(check_call(e, B, R));
    ENG_OK(check_feats(e, feats, feats_dtype));
    return e->enqueue_encode(feats, feats_dtype, att_len, B, R, memory_out, (hipStream_t)stream);
}

int bofi_engine_bound_step(bofi_engine_t* e, const int* ext_syn, const int* last, int B, int R, const int* att_len,
                           float* len_logp, float* syn_logp, void* stream) {
    g_err.clear();
    ENG_OK(check_call(e, B, R));
    if (!ext_syn || !last || !len_logp || !syn_logp) return fail(BOFI_ERR_ARG, "null argument");
    if (e->bound_dense) return fail(BOFI_ERR_STATE, "bound_step is the incremental (N_len = 1) form's stage; the dense bounding pass runs inside decode_naic");
    e->cur_B = B;
    if (e->bound_loop_ok(R)) return e->bound_loop(B, R, att_len, ext_syn, last, 0, len_logp, syn_logp, (hipStream_t)stream);
    ENG_OK(e->bound_tail(nullptr, 1, ext_syn, last, B, BOUND_ATTN, nullptr, nullptr, (hipStream_t)stream));
    return e->enqueue_bound_iter(B, R, att_len, ext_syn, last, 0, len_logp, syn_logp, false, (hipStream_t)stream);
}

int bofi_engine_fill_naic(bofi_engine_t* e, const int* ext_syn, const int* last, int B, int R, const int* att_len, int flags,
                          int64_t* seq, float* seq_logprob, void* stream) {
    g_err.clear();
    ENG_OK(check_call(e, B, R));
    if (!ext_syn || !last || !seq) return fail(BOFI_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    ENG_HIP(hipMemcpyAsync(e->st.ext_syn, ext_syn, (size_t)B * e->L * sizeof(int), hipMemcpyDeviceToDevice, s));
    ENG_HIP(hipMemcpyAsync(e->st.last, last, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
    e->cur_B = B;
    return e->enqueue_fill(att_len, B, R, flags & ~BOFI_FLAG_GRAPH, seq, seq_logprob, s);
}

int bofi_engine_decode_saic(bofi_engine_t* e, const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags,
                            int64_t* seq, float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn,
                            int* bound_iters, void* stream) {
    g_err.clear();
    ENG_OK(check_call(e, B, R));
    ENG_OK(check_feats(e, feats, feats_dtype));
    if (!seq) return fail(BOFI_ERR_ARG, "null seq");
    if (e->n_len != 1) return fail(BOFI_ERR_STATE, "the semi-autoregressive decode is built for a one-layer bounding network (N_len = 1)");
    hipStream_t s = (hipStream_t)stream;
    for (auto& q : e->qkv_dec)                              // the per-layer q|k|v buffers of this mode (never inside a capture)
        if (!q) ENG_OK(e->dalloc((char**)&q, (size_t)e->cfg.max_batch * e->cfg.seq_length * 3 * e->cfg.d_model, e->tsz));
    if (flags & BOFI_FLAG_SAMPLE) ENG_OK(bofi::launch_set_u64(e->d_seed, e->sample_seed, s));     // outside the graph: fresh draws per call
    if (!(flags & BOFI_FLAG_GRAPH))
        return e->enqueue_decode_saic(feats, feats_dtype, att_len, B, R, flags, seq, seq_logprob, phrase_num, phrase_length, phrase_syn,
                                      bound_iters, s);
    uint32_t tbits; std::memcpy(&tbits, &e->sample_temperature, 4);
    std::vector<uintptr_t> key = {(uintptr_t)1, (uintptr_t)feats, (uintptr_t)feats_dtype, (uintptr_t)att_len, (uintptr_t)B, (uintptr_t)R,
                                  (uintptr_t)flags, (uintptr_t)seq, (uintptr_t)seq_logprob, (uintptr_t)phrase_num,
                                  (uintptr_t)phrase_length, (uintptr_t)phrase_syn, (uintptr_t)bound_iters, (uintptr_t)tbits,
                                  (uintptr_t)e->saic_it_begin, (uintptr_t)e->saic_it_end, (uintptr_t)e->in_flight, (uintptr_t)bofi::g_env_generation};
    return run_graphed(e, key, s, [&](hipStream_t cs) {
        return e->enqueue_decode_saic(feats, feats_dtype, att_len, B, R, flags, seq, seq_logprob, phrase_num, phrase_length, phrase_syn,
                                      bound_iters, cs);
    });
}

int bofi_engine_decode_naic(bofi_engine_t* e, const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags,
                            int64_t* seq, float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn,
                            float* memory_out, int* bound_iters, void* stream) {
    g_err.clear();
    ENG_OK(check_call(e, B, R));
    ENG_OK(check_feats(e, feats, feats_dtype));
    if (!seq) return fail(BOFI_ERR_ARG, "null seq");
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & BOFI_FLAG_GRAPH))
        return e->enqueue_decode(feats, feats_dtype, att_len, B, R, flags, seq, seq_logprob, phrase_num, phrase_length,
                                 phrase_syn, memory_out, bound_iters, s);
    // graph path: the captured launch sequence is keyed by every argument that is baked into it
    std::vector<uintptr_t> key = {(uintptr_t)0, (uintptr_t)feats, (uintptr_t)feats_dtype, (uintptr_t)att_len, (uintptr_t)B, (uintptr_t)R,
                                  (uintptr_t)flags, (uintptr_t)seq, (uintptr_t)seq_logprob, (uintptr_t)phrase_num,
                                  (uintptr_t)phrase_length, (uintptr_t)phrase_syn, (uintptr_t)memory_out, (uintptr_t)bound_iters,
                                  (uintptr_t)e->q1_group, (uintptr_t)e->bound_iter_cap, (uintptr_t)e->live_max, (uintptr_t)e->in_flight,
                                  (uintptr_t)bofi::g_env_generation, (uintptr_t)e->row_plogp_out, (uintptr_t)e->row_chosen_out,
                                  (uintptr_t)e->sat_out, (uintptr_t)(e->loop_mode + 1)};
    return run_graphed(e, key, s, [&](hipStream_t cs) {
        return e->enqueue_decode(feats, feats_dtype, att_len, B, R, flags, seq, seq_logprob, phrase_num, phrase_length,
                                 phrase_syn, memory_out, bound_iters, cs);
    });
}

}  // extern "C"
