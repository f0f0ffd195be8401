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

struct Lin { void* w = nullptr; float* b = nullptr; int N = 0, K = 0; };
struct Norm { float* g = nullptr; float* b = nullptr; };
struct EncLayer { Lin qkv, o, w1, w2; Norm n0, n1; };
struct DecLayer { Lin qkv, o, q_src, o_src, w1, w2; Norm n0, n1, n2; };

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
    Lin kv_all;                      // stacked cross-attention K|V: bound layer, then decoder layers
    Lin gen;
    float *lut_syn = nullptr, *lut_tok = nullptr, *pe = nullptr;
    // bound layer
    Lin b_o_self, b_q_src, b_o_src, b_w1, b_w2;
    Norm b_n0, b_n1, b_n2;
    bofi::BoundHeadWeights heads{};
    void *b_q0 = nullptr, *b_kvtab = nullptr;     // compute dtype: [d], [L*10, 2d]
    float* b_x0 = nullptr;                        // [d] residual input of row 0

    // workspace
    float *x_enc = nullptr, *x_fill = nullptr, *logits = nullptr;
    void *qkv = nullptr, *ctx = nullptr, *hdn = nullptr, *mem = nullptr, *kv = nullptr, *qs = nullptr, *xn = nullptr;
    float *by1 = nullptr, *by2 = nullptr, *by3 = nullptr;
    void *bctx = nullptr, *bq2 = nullptr, *bctx2 = nullptr, *bh = nullptr;
    bofi::BoundState st{};

    hipStream_t cap_stream = nullptr;
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
    // stack several [n_i, K] matrices (and their biases) into one Lin
    int make_lin(Lin* out, const std::vector<std::string>& prefixes, int n_each, int K) {
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
        ENG_OK(upload_t(&out->w, w));
        ENG_OK(upload_f32(&out->b, b));
        return BOFI_OK;
    }
    int make_norm(Norm* out, const std::string& prefix, int d) {
        const auto* g = get(prefix + ".a_2", d);
        const auto* b = get(prefix + ".b_2", d);
        if (!g || !b) return BOFI_ERR_STATE;
        ENG_OK(upload_f32(&out->g, *g));
        ENG_OK(upload_f32(&out->b, *b));
        return BOFI_OK;
    }

    // ---- kernels -------------------------------------------------------------------------------
    int linear(const void* x, int x_dtype, int ldx, const Lin& l, const float* residual, int ldr, void* y, int y_dtype,
               int ldy, int M, int relu, const Norm* ln, const int* row_len, int rpg, bool early, hipStream_t s) {
        if (ln) {
            // pre-norm of SublayerConnection (TransformerModel.py:1361-1363): LayerNorm kernel into the
            // compute-dtype scratch, then the LDS-DMA GEMM reads it
            if (x_dtype != BOFI_DT_F32 || ldx != l.K) return BOFI_ERR_ARG;
            int rc = bofi::launch_layernorm((const float*)x, ln->g, ln->b, xn, cfg.dtype, M, l.K, s,
                                            early ? st.counters : nullptr, cur_B);
            if (rc != BOFI_OK) return rc;
            x = xn; x_dtype = cfg.dtype; ln = nullptr;
        }
        bofi::LinearArgs a{};
        a.x = x; a.x_dtype = x_dtype; a.ldx = ldx; a.w = l.w; a.w_dtype = cfg.dtype; a.bias = l.b;
        a.residual = residual; a.ldr = ldr; a.y = y; a.y_dtype = y_dtype; a.ldy = ldy;
        a.M = M; a.N = l.N; a.K = l.K; a.relu = relu; a.row_len = row_len; a.rows_per_group = rpg;
        if (ln) { a.ln_gain = ln->g; a.ln_bias = ln->b; }
        if (early) { a.skip_if_ge = st.counters; a.skip_threshold = cur_B; }
        return bofi::launch_linear(a, s);
    }
    int cur_B = 0;

    int enqueue_encode(const void* feats, int feats_dtype, const int* att_len, int B, int R, float* memory_out, hipStream_t s);
    int enqueue_bound_iter(int B, int R, const int* att_len, const int* ext_syn, const int* last, int update, float* len_logp,
                           float* syn_logp, bool early, hipStream_t s);
    int enqueue_decode(const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags, int64_t* seq,
                       float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn, float* memory_out,
                       int* bound_iters, hipStream_t s);
};

// ================================================================================================
int bofi_engine::enqueue_encode(const void* feats, int feats_dtype, const int* att_len, int B, int R, float* memory_out,
                                hipStream_t s) {
    const int d = cfg.d_model, dt = cfg.dtype, M = B * R;
    cur_B = B;
    // att_embed: Linear + ReLU, rows past an image's region count forced to 0 (AttModel.py:46-51)
    ENG_OK(linear(feats, feats_dtype, cfg.feat, att_embed, nullptr, 0, x_enc, BOFI_DT_F32, d, M, 1, nullptr, att_len, R, false, s));
    for (auto& l : enc) {
        ENG_OK(linear(x_enc, BOFI_DT_F32, d, l.qkv, nullptr, 0, qkv, dt, 3 * d, M, 0, &l.n0, nullptr, 0, false, s));
        bofi::AttnArgs a{};
        a.q = qkv; a.k = (char*)qkv + (size_t)d * tsz; a.v = (char*)qkv + (size_t)2 * d * tsz;
        a.ldq = a.ldk = a.ldv = 3 * d; a.out = ctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = R; a.Lk = R;
        a.klen = att_len; a.klen_sb = 1; a.klen_sq = 0;
        ENG_OK(bofi::launch_attention(a, s));
        ENG_OK(linear(ctx, dt, d, l.o, x_enc, d, x_enc, BOFI_DT_F32, d, M, 0, nullptr, nullptr, 0, false, s));
        ENG_OK(linear(x_enc, BOFI_DT_F32, d, l.w1, nullptr, 0, hdn, dt, cfg.d_ff, M, 1, &l.n1, nullptr, 0, false, s));
        ENG_OK(linear(hdn, dt, cfg.d_ff, l.w2, x_enc, d, x_enc, BOFI_DT_F32, d, M, 0, nullptr, nullptr, 0, false, s));
    }
    ENG_OK(bofi::launch_layernorm(x_enc, enc_norm.g, enc_norm.b, mem, dt, M, d, s));
    if (memory_out) ENG_OK(bofi::launch_layernorm(x_enc, enc_norm.g, enc_norm.b, memory_out, BOFI_DT_F32, M, d, s));
    // cross-attention K|V of the bound layer and of every decoder layer in one GEMM
    ENG_OK(linear(mem, dt, d, kv_all, nullptr, 0, kv, dt, kv_all.N, M, 0, nullptr, nullptr, 0, false, s));
    return BOFI_OK;
}

int bofi_engine::enqueue_bound_iter(int B, int R, const int* att_len, const int* ext_syn, const int* last, int update,
                                    float* len_logp, float* syn_logp, bool early, hipStream_t s) {
    // One bounding iteration AFTER the row-0 self-attention context (bctx) has been produced by the
    // previous launch_bound_tail(..., BOUND_ATTN).
    const int d = cfg.d_model, dt = cfg.dtype;
    cur_B = B;
    // y1 = x0 + (Wo ctx + bo): the row-0 residual input is the same vector for every image (ldr = 0)
    ENG_OK(linear(bctx, dt, d, b_o_self, b_x0, 0, by1, BOFI_DT_F32, d, B, 0, nullptr, nullptr, 0, early, s));
    ENG_OK(linear(by1, BOFI_DT_F32, d, b_q_src, nullptr, 0, bq2, dt, d, B, 0, &b_n1, nullptr, 0, early, s));
    bofi::AttnArgs a{};
    a.q = bq2; a.ldq = d; a.k = kv; a.v = (char*)kv + (size_t)d * tsz; a.ldk = a.ldv = kv_all.N;
    a.out = bctx2; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = 1; a.Lk = R;
    a.klen = att_len; a.klen_sb = 1; a.klen_sq = 0;
    if (early) { a.skip_if_ge = st.counters; a.skip_threshold = B; }
    ENG_OK(bofi::launch_attention(a, s));
    ENG_OK(linear(bctx2, dt, d, b_o_src, by1, d, by2, BOFI_DT_F32, d, B, 0, nullptr, nullptr, 0, early, s));
    ENG_OK(linear(by2, BOFI_DT_F32, d, b_w1, nullptr, 0, bh, dt, cfg.d_ff, B, 1, &b_n2, nullptr, 0, early, s));
    ENG_OK(linear(bh, dt, cfg.d_ff, b_w2, by2, d, by3, BOFI_DT_F32, d, B, 0, nullptr, nullptr, 0, early, s));
    // heads + bookkeeping, fused with the next iteration's row-0 self-attention
    const int flags = BOUND_HEADS | (update ? (BOUND_UPDATE | BOUND_ATTN) : 0) | (early ? BOUND_EARLY : 0);
    ENG_OK(bofi::launch_bound_tail(by3, heads, st, update ? nullptr : ext_syn, update ? nullptr : last, b_q0, b_kvtab, bctx, dt, B, L,
                                   cfg.seq_length, d, cfg.head_hidden, cfg.heads, flags, len_logp, syn_logp, s));
    return BOFI_OK;
}

int bofi_engine::enqueue_decode(const void* feats, int feats_dtype, const int* att_len, int B, int R, int flags, int64_t* seq,
                                float* seq_logprob, int* phrase_num, int* phrase_length, int64_t* phrase_syn,
                                float* memory_out, int* bound_iters, hipStream_t s) {
    const int d = cfg.d_model, dt = cfg.dtype, S = cfg.seq_length, M = B * S;
    ENG_OK(enqueue_encode(feats, feats_dtype, att_len, B, R, memory_out, s));
    // ---- bounding pass (core_NAIC TransformerModel.py:1833-1870)
    ENG_OK(bofi::launch_bound_init(st, B, L, cfg.pad_idx, cfg.len_idx, s));
    ENG_OK(bofi::launch_bound_tail(nullptr, heads, st, nullptr, nullptr, b_q0, b_kvtab, bctx, dt, B, L, S, d, cfg.head_hidden,
                                   cfg.heads, BOUND_ATTN, nullptr, nullptr, s));
    for (int it = 0; it < S; ++it)
        ENG_OK(enqueue_bound_iter(B, R, att_len, st.ext_syn, st.last, 1, nullptr, nullptr, true, s));
    // ---- filling pass (decode_NA :570-587)
    ENG_OK(bofi::launch_embed_fill(lut_tok, lut_syn, pe, st.ext_syn, nullptr, B, S, L, d, cfg.bos_idx, x_fill, s));
    for (size_t li = 0; li < dec.size(); ++li) {
        auto& l = dec[li];
        ENG_OK(linear(x_fill, BOFI_DT_F32, d, l.qkv, nullptr, 0, qkv, dt, 3 * d, M, 0, &l.n0, nullptr, 0, false, s));
        bofi::AttnArgs a{};
        a.q = qkv; a.k = (char*)qkv + (size_t)d * tsz; a.v = (char*)qkv + (size_t)2 * d * tsz;
        a.ldq = a.ldk = a.ldv = 3 * d; a.out = ctx; a.ldo = d; a.dtype = dt; a.B = B; a.H = cfg.heads; a.Lq = S; a.Lk = S;
        // syn_mask[i, :, :last-1] = True; strict mode reproduces the stale index of :1872-1873 (quirk Q1)
        a.klen = st.last; a.klen_sb = 1; a.klen_sq = 0; a.klen_bias = -1;
        a.klen_shared_last = (flags & BOFI_FLAG_STRICT_Q1) ? 1 : 0;
        ENG_OK(bofi::launch_attention(a, s));
        ENG_OK(linear(ctx, dt, d, l.o, x_fill, d, x_fill, BOFI_DT_F32, d, M, 0, nullptr, nullptr, 0, false, s));
        ENG_OK(linear(x_fill, BOFI_DT_F32, d, l.q_src, nullptr, 0, qs, dt, d, M, 0, &l.n1, nullptr, 0, false, s));
        bofi::AttnArgs c{};
        c.q = qs; c.ldq = d;
        c.k = (char*)kv + (size_t)(1 + li) * 2 * d * tsz; c.v = (char*)kv + ((size_t)(1 + li) * 2 * d + d) * tsz;
        c.ldk = c.ldv = kv_all.N; c.out = ctx; c.ldo = d; c.dtype = dt; c.B = B; c.H = cfg.heads; c.Lq = S; c.Lk = R;
        c.klen = att_len; c.klen_sb = 1; c.klen_sq = 0;
        ENG_OK(bofi::launch_attention(c, s));
        ENG_OK(linear(ctx, dt, d, l.o_src, x_fill, d, x_fill, BOFI_DT_F32, d, M, 0, nullptr, nullptr, 0, false, s));
        ENG_OK(linear(x_fill, BOFI_DT_F32, d, l.w1, nullptr, 0, hdn, dt, cfg.d_ff, M, 1, &l.n2, nullptr, 0, false, s));
        ENG_OK(linear(hdn, dt, cfg.d_ff, l.w2, x_fill, d, x_fill, BOFI_DT_F32, d, M, 0, nullptr, nullptr, 0, false, s));
    }
    // ---- vocabulary projection, log-softmax, greedy pick, pad tail
    float* lg = seq_logprob ? seq_logprob : logits;
    ENG_OK(linear(x_fill, BOFI_DT_F32, d, gen, nullptr, 0, lg, BOFI_DT_F32, cfg.vocab, M, 0, &dec_norm, nullptr, 0, false, s));
    ENG_OK(bofi::launch_vocab_finalize(lg, M, cfg.vocab, S, (flags & BOFI_FLAG_RAW_LOGITS) ? 0 : 1, st.last, -1, cfg.pad_idx, seq, s));
    ENG_OK(bofi::launch_bound_export(st, B, L, S, phrase_num, phrase_length, phrase_syn, bound_iters, s));
    return BOFI_OK;
}

// ================================================================================================
extern "C" {

int bofi_abi_version(void) { return 1; }
const char* bofi_last_error(void) { return g_err.c_str(); }

int bofi_engine_create(const bofi_config_t* c, bofi_engine_t** out) {
    if (!c || !out) return fail(BOFI_ERR_ARG, "null argument");
    if (c->dtype != BOFI_DT_F32 && c->dtype != BOFI_DT_BF16) return fail(BOFI_ERR_ARG, "dtype must be 0 (f32) or 1 (bf16)");
    if (c->heads <= 0 || c->d_model != c->heads * 64) return fail(BOFI_ERR_ARG, "d_model / heads must be 64");
    if (c->d_model % 64 || c->d_ff % 64 || c->feat % 64) return fail(BOFI_ERR_ARG, "d_model, d_ff, feat must be multiples of 64");
    if (c->seq_length <= 0 || c->seq_length + 2 > 64) return fail(BOFI_ERR_ARG, "seq_length must be in 1..62");
    if (c->max_batch <= 0 || c->max_regions <= 0 || c->max_regions > 128) return fail(BOFI_ERR_ARG, "max_batch > 0, 0 < max_regions <= 128");
    if (c->vocab <= 0 || c->n_enc < 0 || c->n_dec < 0 || c->head_hidden <= 0) return fail(BOFI_ERR_ARG, "bad layer/vocab counts");
    auto* e = new bofi_engine();
    e->cfg = *c;
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
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}

int bofi_engine_set_weight(bofi_engine_t* e, const char* name, const float* data, int64_t numel) {
    if (!e || !name || !data || numel < 0) return fail(BOFI_ERR_ARG, "null argument");
    e->host[name].assign(data, data + numel);
    e->finalized = false;
    return BOFI_OK;
}

int bofi_engine_finalize(bofi_engine_t* e) {
    if (!e) return fail(BOFI_ERR_ARG, "null engine");
    g_err.clear();
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

    auto S = [](const char* fmt, int a, int b = 0) { char buf[256]; std::snprintf(buf, sizeof buf, fmt, a, b); return std::string(buf); };
    ENG_OK(e->make_lin(&e->att_embed, {"att_embed.0"}, d, c.feat));
    for (int l = 0; l < c.n_enc; ++l) {
        auto& E = e->enc[l];
        const std::string p = S("model.encoder.layers.%d", l);
        ENG_OK(e->make_lin(&E.qkv, {p + ".self_attn.linears.0", p + ".self_attn.linears.1", p + ".self_attn.linears.2"}, d, d));
        ENG_OK(e->make_lin(&E.o, {p + ".self_attn.linears.3"}, d, d));
        ENG_OK(e->make_lin(&E.w1, {p + ".feed_forward.w_1"}, dff, d));
        ENG_OK(e->make_lin(&E.w2, {p + ".feed_forward.w_2"}, d, dff));
        ENG_OK(e->make_norm(&E.n0, p + ".sublayer.0.norm", d));
        ENG_OK(e->make_norm(&E.n1, p + ".sublayer.1.norm", d));
    }
    ENG_OK(e->make_norm(&e->enc_norm, "model.encoder.norm", d));
    const std::string bl = "model.length_predictor.LengthPredictor.0";
    std::vector<std::string> kvs = {bl + ".src_attn.linears.1", bl + ".src_attn.linears.2"};
    for (int l = 0; l < c.n_dec; ++l) {
        auto& D = e->dec[l];
        const std::string p = S("model.decoder.layers.%d", l);
        ENG_OK(e->make_lin(&D.qkv, {p + ".self_attn.linears.0", p + ".self_attn.linears.1", p + ".self_attn.linears.2"}, d, d));
        ENG_OK(e->make_lin(&D.o, {p + ".self_attn.linears.3"}, d, d));
        ENG_OK(e->make_lin(&D.q_src, {p + ".src_attn.linears.0"}, d, d));
        ENG_OK(e->make_lin(&D.o_src, {p + ".src_attn.linears.3"}, d, d));
        ENG_OK(e->make_lin(&D.w1, {p + ".feed_forward.w_1"}, dff, d));
        ENG_OK(e->make_lin(&D.w2, {p + ".feed_forward.w_2"}, d, dff));
        ENG_OK(e->make_norm(&D.n0, p + ".sublayer.0.norm", d));
        ENG_OK(e->make_norm(&D.n1, p + ".sublayer.1.norm", d));
        ENG_OK(e->make_norm(&D.n2, p + ".sublayer.2.norm", d));
        kvs.push_back(p + ".src_attn.linears.1");
        kvs.push_back(p + ".src_attn.linears.2");
    }
    ENG_OK(e->make_norm(&e->dec_norm, "model.decoder.norm", d));
    ENG_OK(e->make_lin(&e->kv_all, kvs, d, d));
    ENG_OK(e->make_lin(&e->gen, {"model.generator.proj"}, c.vocab, d));
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
    ENG_OK(e->make_lin(&e->b_q_src, {bl + ".src_attn.linears.0"}, d, d));
    ENG_OK(e->make_lin(&e->b_o_src, {bl + ".src_attn.linears.3"}, d, d));
    ENG_OK(e->make_lin(&e->b_w1, {bl + ".ff.w_1"}, dff, d));
    ENG_OK(e->make_lin(&e->b_w2, {bl + ".ff.w_2"}, d, dff));
    ENG_OK(e->make_norm(&e->b_n0, bl + ".sublayer.0.norm", d));
    ENG_OK(e->make_norm(&e->b_n1, bl + ".sublayer.1.norm", d));
    ENG_OK(e->make_norm(&e->b_n2, bl + ".sublayer.2.norm", d));
    {
        const std::string lp = "model.length_predictor";
        Norm nf;
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
        e->heads = bofi::BoundHeadWeights{nf.g, nf.b, p_w1, p_b1, p_lw2, p_lb2, p_sw2, p_sb2};
    }

    // workspace
    const size_t Bm = c.max_batch, Rm = c.max_regions, Sq = c.seq_length;
    const size_t rows = Bm * (Rm > Sq ? Rm : Sq);
    ENG_OK(e->dalloc(&e->x_enc, Bm * Rm * d));
    ENG_OK(e->dalloc(&e->x_fill, Bm * Sq * d));
    ENG_OK(e->dalloc(&e->logits, Bm * Sq * c.vocab));
    ENG_OK(e->dalloc((char**)&e->qkv, rows * 3 * d, e->tsz));
    ENG_OK(e->dalloc((char**)&e->ctx, rows * d, e->tsz));
    ENG_OK(e->dalloc((char**)&e->hdn, rows * dff, e->tsz));
    ENG_OK(e->dalloc((char**)&e->mem, Bm * Rm * d, e->tsz));
    ENG_OK(e->dalloc((char**)&e->kv, Bm * Rm * (size_t)e->kv_all.N, e->tsz));
    ENG_OK(e->dalloc((char**)&e->qs, Bm * Sq * d, e->tsz));
    ENG_OK(e->dalloc((char**)&e->xn, (rows > (size_t)L * 10 ? rows : (size_t)L * 10) * d, e->tsz));
    ENG_OK(e->dalloc(&e->by1, Bm * d)); ENG_OK(e->dalloc(&e->by2, Bm * d)); ENG_OK(e->dalloc(&e->by3, Bm * d));
    ENG_OK(e->dalloc((char**)&e->bctx, Bm * d, e->tsz)); ENG_OK(e->dalloc((char**)&e->bq2, Bm * d, e->tsz));
    ENG_OK(e->dalloc((char**)&e->bctx2, Bm * d, e->tsz)); ENG_OK(e->dalloc((char**)&e->bh, Bm * dff, e->tsz));
    ENG_OK(e->dalloc(&e->st.last, Bm)); ENG_OK(e->dalloc(&e->st.finished, Bm)); ENG_OK(e->dalloc(&e->st.phrase_num, Bm));
    ENG_OK(e->dalloc(&e->st.phrase_length, Bm * L)); ENG_OK(e->dalloc(&e->st.phrase_syn, Bm * L));
    ENG_OK(e->dalloc(&e->st.ext_syn, Bm * L)); ENG_OK(e->dalloc(&e->st.counters, 4));

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
        float* d_xt;
        ENG_OK(e->upload_f32(&d_xt, xt));
        ENG_OK(e->upload_f32(&e->b_x0, x0));
        Lin kvself, qself;
        ENG_OK(e->make_lin(&kvself, {bl + ".self_attn.linears.1", bl + ".self_attn.linears.2"}, d, d));
        ENG_OK(e->make_lin(&qself, {bl + ".self_attn.linears.0"}, d, d));
        ENG_OK(e->dalloc((char**)&e->b_kvtab, (size_t)L * 10 * 2 * d, e->tsz));
        ENG_OK(e->dalloc((char**)&e->b_q0, (size_t)d, e->tsz));
        ENG_OK(e->linear(d_xt, BOFI_DT_F32, d, kvself, nullptr, 0, e->b_kvtab, c.dtype, 2 * d, L * 10, 0, &e->b_n0, nullptr, 0, false, nullptr));
        ENG_OK(e->linear(e->b_x0, BOFI_DT_F32, d, qself, nullptr, 0, e->b_q0, c.dtype, d, 1, 0, &e->b_n0, nullptr, 0, false, nullptr));
        ENG_HIP(hipDeviceSynchronize());
    }
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
    ENG_OK(bofi::launch_bound_tail(nullptr, e->heads, e->st, ext_syn, last, e->b_q0, e->b_kvtab, e->bctx, e->cfg.dtype, B, e->L,
                                   e->cfg.seq_length, e->cfg.d_model, e->cfg.head_hidden, e->cfg.heads, BOUND_ATTN, nullptr, nullptr,
                                   (hipStream_t)stream));
    return e->enqueue_bound_iter(B, R, att_len, ext_syn, last, 0, len_logp, syn_logp, false, (hipStream_t)stream);
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
    std::vector<uintptr_t> key = {(uintptr_t)feats, (uintptr_t)feats_dtype, (uintptr_t)att_len, (uintptr_t)B, (uintptr_t)R,
                                  (uintptr_t)flags, (uintptr_t)seq, (uintptr_t)seq_logprob, (uintptr_t)phrase_num,
                                  (uintptr_t)phrase_length, (uintptr_t)phrase_syn, (uintptr_t)memory_out, (uintptr_t)bound_iters};
    for (auto& g : e->graphs)
        if (g.key == key) { ENG_HIP(hipGraphLaunch(g.exec, s)); return BOFI_OK; }
    GraphEntry g;
    g.key = key;
    ENG_HIP(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = e->enqueue_decode(feats, feats_dtype, att_len, B, R, flags, seq, seq_logprob, phrase_num, phrase_length,
                                     phrase_syn, memory_out, bound_iters, e->cap_stream);
    hipError_t ee = hipStreamEndCapture(e->cap_stream, &g.graph);
    if (rc != BOFI_OK) { if (g.graph) (void)hipGraphDestroy(g.graph); return rc; }
    if (ee != hipSuccess) return fail(BOFI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
    ENG_HIP(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    if (e->graphs.size() >= 8) {                       // small cache: drop the oldest capture
        (void)hipGraphExecDestroy(e->graphs.front().exec);
        (void)hipGraphDestroy(e->graphs.front().graph);
        e->graphs.erase(e->graphs.begin());
    }
    e->graphs.push_back(g);
    ENG_HIP(hipGraphLaunch(g.exec, s));
    return BOFI_OK;
}

}  // extern "C"
