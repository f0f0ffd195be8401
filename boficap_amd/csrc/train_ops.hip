// Backward kernels for the XE training step (reference tools/train.py:212-227 runs autograd over the torch ops of
// TransformerModel._forward; here each forward kernel gets its hand-written backward).  All float32: the training
// path is built for parity first.  Gradients w.r.t. parameters that are sums over rows are accumulated with float
// atomics into zero-initialised buffers (order-dependent in the last bits, as any parallel reduction).
#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

// ------------------------------------------------------------------------------------------------
// LayerNorm backward (forward: y = g * t / s + b, t = x - mean, s = sqrt(sum t^2 / (d-1)) + eps; TransformerModel.py:1346-1349)
//   dt_i = dyh_i / s - t_i * (sum_j dyh_j t_j) / (s^2 * sigma * (d-1)),  dyh = dy * g,  dx = dt - mean(dt)
// One wavefront per row.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gain,
                                                     const float* __restrict__ dy, float* __restrict__ dx, float* dgain,
                                                     float* dbias, int rows, int d, const float* __restrict__ add) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * d;
    const float* dyr = dy + (size_t)row * d;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) s += xr[k];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f, c = 0.f;
    for (int k = lane; k < d; k += 64) { const float t = xr[k] - mean; q += t * t; c += dyr[k] * gain[k] * t; }
    q = wave_sum(q); c = wave_sum(c);
    const float sigma = sqrtf(q / (float)(d - 1)), sd = sigma + 1e-6f;
    const float coef = sigma > 0.f ? c / (sd * sd * sigma * (float)(d - 1)) : 0.f;
    float m = 0.f;
    for (int k = lane; k < d; k += 64) m += dyr[k] * gain[k] / sd - (xr[k] - mean) * coef;
    m = wave_sum(m) / (float)d;
    for (int k = lane; k < d; k += 64) {
        const float t = xr[k] - mean;
        dx[(size_t)row * d + k] = dyr[k] * gain[k] / sd - t * coef - m + (add ? add[(size_t)row * d + k] : 0.f);
        atomicAdd(&dgain[k], dyr[k] * t / sd);
        atomicAdd(&dbias[k], dyr[k]);
    }
}

// Fast path for d = 64 * NJ: a wavefront keeps its row in registers, accumulates the gain/bias gradients of its rows in
// registers, the four wavefronts of the workgroup combine through LDS, and the workgroup issues ONE atomic per column.
template <int NJ>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ gain,
                                                          const float* __restrict__ dy, float* __restrict__ dx, float* dgain,
                                                          float* dbias, int rows, int rows_per_block, const float* __restrict__ add,
                                                          bf16_t* __restrict__ dz, uint32_t drop_thresh, float drop_scale, uint64_t drop_seed0,
                                                          const uint64_t* drop_step) {
    constexpr int d = NJ * 64;
    __shared__ float red[2][4][d];
    const uint64_t drop_seed = drop_seed0 + ((dz && drop_thresh && drop_step) ? *drop_step : 0ull);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NJ], dg[NJ], db[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { g[j] = gain[lane + 64 * j]; dg[j] = 0.f; db[j] = 0.f; }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    // a wavefront walks its rows one after the other and every row is three dependent wave reductions behind its loads: the NEXT
    // row's three streams are requested before the current row is reduced (second register set), or each row pays a full
    // memory latency of its own
    float xn[NJ], dn[NJ], an[NJ];
    auto request = [&](int row) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            xn[j] = x[(size_t)row * d + lane + 64 * j];
            dn[j] = dy[(size_t)row * d + lane + 64 * j];
            an[j] = add ? add[(size_t)row * d + lane + 64 * j] : 0.f;
        }
    };
    if (r0 + wave < r1) request(r0 + wave);
    for (int row = r0 + wave; row < r1; row += 4) {
        float xv[NJ], dv[NJ], av[NJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { xv[j] = xn[j]; dv[j] = dn[j]; av[j] = an[j]; }
        if (row + 4 < r1) request(row + 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) s += xv[j];
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f, c = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { xv[j] -= mean; q += xv[j] * xv[j]; c += dv[j] * g[j] * xv[j]; }
        q = wave_sum(q); c = wave_sum(c);
        const float sigma = sqrtf(q / (float)(d - 1)), sd = sigma + 1e-6f, inv = 1.f / sd;
        const float coef = sigma > 0.f ? c / (sd * sd * sigma * (float)(d - 1)) : 0.f;
        float m = 0.f, t[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { t[j] = dv[j] * g[j] * inv - xv[j] * coef; m += t[j]; }
        m = wave_sum(m) / (float)d;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const size_t o = (size_t)row * d + lane + 64 * j;
            const float v = t[j] - m + av[j];
            dx[o] = v;
            // dz: the same gradient with the dropout mask of the linear that PRODUCED x applied, in bf16 -- what that linear's
            // backward GEMMs read (x = residual + dropout(h W^T + b): d/dh = mask o d/dx)
            if (dz) dz[o] = f32_to_bf16((!drop_thresh || drop_hash(drop_seed, o) >= drop_thresh) ? v * drop_scale : 0.f);
            dg[j] += dv[j] * xv[j] * inv;
            db[j] += dv[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { red[0][wave][lane + 64 * j] = dg[j]; red[1][wave][lane + 64 * j] = db[j]; }
    __syncthreads();
    for (int k = threadIdx.x; k < d; k += 256) {
        atomicAdd(&dgain[k], (red[0][0][k] + red[0][1][k]) + (red[0][2][k] + red[0][3][k]));
        atomicAdd(&dbias[k], (red[1][0][k] + red[1][1][k]) + (red[1][2][k] + red[1][3][k]));
    }
}

// ------------------------------------------------------------------------------------------------
// Attention backward for the short sequences of this model (Lk <= 128 keys: up to max_boxes = 100 regions; d_k = 64), float32, one
// workgroup per (batch item, head, chunk of LQ query rows): recompute P, then dV = P^T dO, dP = dO V^T,
// dS = P (dP - rowsum(dP P)), dQ = dS K / 8, dK = dS^T Q / 8.  With more than one query chunk (or shared keys, kdiv > 1) the key-side
// gradients of the chunks meet in float atomics: the caller passes zeroed dk / dv.
struct AttnBwdParams {
    const float* q; int ldq; const float* k; int ldk; const float* v; int ldv;
    const float* dout; int ldo;
    float* dq; float* dk; float* dv;          // same layouts as q / k / v; dk, dv accumulate with atomics when kdiv > 1
    int B, H, Lq, Lk, kdiv;
    const int* klen; int klen_sb, klen_sq, klen_bias;
    const int* q_start; const int* q_count; int k_ragged;     // unpadded layout (bofi_kernels.h: AttnArgs)
};

template <int LQ, int LK>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBwdParams p) {
    constexpr int DS = 65, PS = LK + 1;                        // padded strides: conflict-free column walks
    __shared__ float sq[LQ * DS], sdo[LQ * DS], sk[LK * DS], sv[LK * DS], sp[LQ * PS], sds[LQ * PS];
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H, tid = threadIdx.x;
    const int bk = b / p.kdiv;
    const int Lq_all = p.q_start ? p.q_count[b] : p.Lq;         // the item's query rows; this workgroup takes rows qc0 .. qc0 + LQ
    const size_t qrow_item = p.q_start ? (size_t)p.q_start[b] : (size_t)b * p.Lq;
    const int qc0 = blockIdx.y * LQ;
    const bool kr = p.q_start && p.k_ragged;
    const int Lk = kr ? Lq_all : p.Lk;
    const size_t krow0 = kr ? qrow_item : (size_t)bk * p.Lk;
    if (qc0 >= Lq_all) return;
    const int Lq = min(LQ, Lq_all - qc0);
    const size_t qrow0 = qrow_item + qc0;
    const bool shared = p.kdiv > 1 || gridDim.y > 1;            // several workgroups add into the same dk / dv rows
    for (int i = tid; i < Lq * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        sq[r * DS + c] = p.q[(qrow0 + r) * p.ldq + h * 64 + c];
        sdo[r * DS + c] = p.dout[(qrow0 + r) * p.ldo + h * 64 + c];
    }
    for (int i = tid; i < Lk * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        sk[r * DS + c] = p.k[(krow0 + r) * p.ldk + h * 64 + c];
        sv[r * DS + c] = p.v[(krow0 + r) * p.ldv + h * 64 + c];
    }
    __syncthreads();
    // scores and dP = dO V^T over the real Lq x Lk rectangle
    for (int e = tid; e < Lq * Lk; e += 256) {
        const int i = e / Lk, j = e - i * Lk;
        float s = 0.f, dp = 0.f;
#pragma unroll 16
        for (int c = 0; c < 64; ++c) { s = fmaf(sq[i * DS + c], sk[j * DS + c], s); dp = fmaf(sdo[i * DS + c], sv[j * DS + c], dp); }
        sp[i * PS + j] = s * 0.125f;
        sds[i * PS + j] = dp;
    }
    __syncthreads();
    // row softmax + dS: 4 lanes per query row
    {
        const int i = tid >> 2, sub = tid & 3;
        if (i < Lq) {
            int kl = Lk;
            if (p.klen) { kl = (p.q_start ? p.klen[qrow0 + i] : p.klen[b * p.klen_sb + (qc0 + i) * p.klen_sq]) + p.klen_bias; kl = max(0, min(kl, Lk)); }
            float m = -INFINITY;
            for (int j = sub; j < kl; j += 4) m = fmaxf(m, sp[i * PS + j]);
            m = quad_max(m);
            float sum = 0.f;
            for (int j = sub; j < kl; j += 4) { const float e = expf(sp[i * PS + j] - m); sp[i * PS + j] = e; sum += e; }
            sum = quad_sum(sum);
            float dot = 0.f;
            for (int j = sub; j < Lk; j += 4) { const float pv = j < kl ? sp[i * PS + j] / sum : 0.f; sp[i * PS + j] = pv; dot += pv * sds[i * PS + j]; }
            dot = quad_sum(dot);
            for (int j = sub; j < Lk; j += 4) sds[i * PS + j] = sp[i * PS + j] * (sds[i * PS + j] - dot) * 0.125f;
        }
    }
    __syncthreads();
    for (int e = tid; e < Lq * 64; e += 256) {                 // dQ[r][c] = sum_j dS[r][j] K[j][c]
        const int r = e >> 6, c = e & 63;
        float a = 0.f;
        for (int j = 0; j < Lk; ++j) a = fmaf(sds[r * PS + j], sk[j * DS + c], a);
        p.dq[(qrow0 + r) * p.ldq + h * 64 + c] = a;
    }
    for (int e = tid; e < Lk * 64; e += 256) {                 // dK[r][c] = sum_i dS[i][r] Q[i][c];  dV[r][c] = sum_i P[i][r] dO[i][c]
        const int r = e >> 6, c = e & 63;
        float a = 0.f, g = 0.f;
        for (int i = 0; i < Lq; ++i) { a = fmaf(sds[i * PS + r], sq[i * DS + c], a); g = fmaf(sp[i * PS + r], sdo[i * DS + c], g); }
        float* dkp = p.dk + (krow0 + r) * p.ldk + h * 64 + c;
        float* dvp = p.dv + (krow0 + r) * p.ldv + h * 64 + c;
        if (shared) { atomicAdd(dkp, a); atomicAdd(dvp, g); } else { *dkp = a; *dvp = g; }
    }
}

// ------------------------------------------------------------------------------------------------
// log_softmax backward: dx = dy - exp(y) * rowsum(dy)   (y = log-probabilities)
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                             float* __restrict__ dx, int V) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* yr = y + (size_t)row * V;
    const float* dyr = dy + (size_t)row * V;
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += dyr[i];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = tid; i < V; i += 256) dx[(size_t)row * V + i] = dyr[i] - expf(yr[i]) * s;
}

// backward of picked[r] = y[r][label[r]] through log_softmax, y = log-probabilities: dx[r][c] = dpicked[r] * ((c == label[r]) - exp(y[r][c]))
// (the token NLL of the criterion: no dense dL/dy tensor is built, zero-filled and scattered into)
// (dx float32 [rows, V], or -- dx_bf16 -- bf16 [rows, ldb] with the columns V..ldb-1 written as zeros: the operand the
// vocabulary projection's backward GEMMs read)
__global__ __launch_bounds__(256) void nll_bwd_kernel(const float* __restrict__ y, const int64_t* __restrict__ label, const float* __restrict__ dpicked,
                                                      float* __restrict__ dx, bf16_t* __restrict__ dx_bf16, int ldb, int V) {
    const int row = blockIdx.x;
    const float g = dpicked[row];
    const int lab = (int)label[row];
    const float* yr = y + (size_t)row * V;
    if (dx_bf16) {
        bf16_t* dr = dx_bf16 + (size_t)row * ldb;
        for (int i = threadIdx.x; i < ldb; i += 256) dr[i] = (g == 0.f || i >= V) ? (bf16_t)0 : f32_to_bf16(g * ((i == lab ? 1.f : 0.f) - expf(yr[i])));
        return;
    }
    float* dr = dx + (size_t)row * V;
    for (int i = threadIdx.x; i < V; i += 256) dr[i] = g == 0.f ? 0.f : g * ((i == lab ? 1.f : 0.f) - expf(yr[i]));
}


// ------------------------------------------------------------------------------------------------
// LanguageModelCriterion_UIC (captioning/modules/losses.py:319-369, reduction 'mean') for the paired training forward, in one
// launch: the four slot terms read their labels / mask straight from the loader's phrase tensors
//   part[0] = -sum_{n, p < phrase_num[n]} sa_len[n][p][phrase_length[n][p+1]] / denom      (2: sa_syn / phrase_syn, 3: na_len, 5: na_syn)
//   part[1] = -sum_r picked[r] w_sa[r] / denom,  part[4] likewise with w_na,  denom = sum_r w_sa[r],  out[6] = sum of the six.
// One workgroup: a few thousand elements.  The reference does this with ~40 elementwise launches and as many in the backward.
struct UicCritArgs {
    const float* lp[4]; int C[4];                 // sa_len, sa_syn, na_len, na_syn: [N, Pm, C]
    const int64_t* phrase_num; const int64_t* phrase_length; const int64_t* phrase_syn; int N, Pm, L;
    const float* picked; const float* w_sa; const float* w_na; int T;
};

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    return s;
}

__global__ __launch_bounds__(1024) void uic_criterion_kernel(UicCritArgs a, float* __restrict__ out) {
    __shared__ float red[16];
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // 0..5 parts (unnormalised, negated later), 6 denom
    for (int i = threadIdx.x; i < a.N * a.Pm; i += 1024) {
        const int n = i / a.Pm, p = i - n * a.Pm;
        if (p >= a.phrase_num[n]) continue;
        const int ll = (int)a.phrase_length[(size_t)n * a.L + p + 1], sl = (int)a.phrase_syn[(size_t)n * a.L + p + 1];
        acc[0] += a.lp[0][(size_t)i * a.C[0] + ll];
        acc[2] += a.lp[1][(size_t)i * a.C[1] + sl];
        acc[3] += a.lp[2][(size_t)i * a.C[2] + ll];
        acc[5] += a.lp[3][(size_t)i * a.C[3] + sl];
    }
    for (int r = threadIdx.x; r < a.T; r += 1024) {
        const float pk = a.picked[r], ws = a.w_sa[r], wn = a.w_na[r];
        if (ws != 0.f) acc[1] += pk * ws;
        if (wn != 0.f) acc[4] += pk * wn;
        acc[6] += ws;
    }
    float tot[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) tot[k] = block_sum_1024(acc[k], red);
    if (threadIdx.x == 0) {
        float loss = 0.f;
        for (int k = 0; k < 6; ++k) { out[k] = -tot[k] / tot[6]; loss += out[k]; }
        out[6] = loss;
        out[7] = tot[6];
    }
}

// gradients for upstream g = dL/d(out[6]) (the six parts are reported, not differentiated): dense [N, Pm, C] tensors for the four
// slot outputs (zero except at the label of a counted slot) and dpicked [T]
__global__ __launch_bounds__(256) void uic_criterion_bwd_kernel(UicCritArgs a, const float* __restrict__ g, const float* __restrict__ fwd,
                                                                float* d0, float* d1, float* d2, float* d3, float* __restrict__ dpicked) {
    const float scale = -g[0] / fwd[7];
    float* d[4] = {d0, d1, d2, d3};
    const int slots = a.N * a.Pm;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < slots; i += gridDim.x * 256) {
        const int n = i / a.Pm, p = i - n * a.Pm;
        const bool on = p < a.phrase_num[n];
        const int ll = (int)a.phrase_length[(size_t)n * a.L + p + 1], sl = (int)a.phrase_syn[(size_t)n * a.L + p + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lab = (k & 1) ? sl : ll;
            for (int c = 0; c < a.C[k]; ++c) d[k][(size_t)i * a.C[k] + c] = (on && c == lab) ? scale : 0.f;
        }
    }
    for (int r = blockIdx.x * 256 + threadIdx.x; r < a.T; r += gridDim.x * 256) dpicked[r] = scale * (a.w_sa[r] + a.w_na[r]);
}

// column sums (bias gradients): out[n] += sum_m x[m][n]
__global__ void colsum_kernel(const float* __restrict__ x, float* out, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * 64, m1 = min(M, m0 + 64);
    float s = 0.f;
    for (int m = m0; m < m1; ++m) s += x[(size_t)m * N + n];
    atomicAdd(&out[n], s);
}

// embedding backward: d_lut[id[r]] += scale * dx[r].  A workgroup owns 32 consecutive rows and one thread a column; runs of
// equal ids (the NA decoder input is [BOS] everywhere, pad tails are all 0) are summed in registers and flushed with ONE
// atomic per run, so the hot rows of the table are not hammered once per position.
__global__ __launch_bounds__(128) void embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids, float* dlut, int rows,
                                                        int d, float scale) {
    const int r0 = blockIdx.x * 32, r1 = min(rows, r0 + 32);
    // blockIdx.y: 128-column slice (a few thousand rows x d = 512 is 160 row blocks only: the column slices fill the CUs)
    for (int k = blockIdx.y * 128 + threadIdx.x; k < d; k += 128 * gridDim.y) {
        int64_t cur = ids[r0];
        float s = 0.f;
        for (int r = r0; r < r1; ++r) {
            const int64_t id = ids[r];
            if (id != cur) { if (cur >= 0) atomicAdd(&dlut[(size_t)cur * d + k], scale * s); cur = id; s = 0.f; }
            s += dx[(size_t)r * d + k];
        }
        if (cur >= 0) atomicAdd(&dlut[(size_t)cur * d + k], scale * s);                 // negative id: the row has no such term
    }
}

// embedding forward for teacher-forced rows: x[r] = (tok ? lut_tok[tok[r]] * sqrt(d) : 0) (+ syn likewise) + pe[r % L]
// (Embeddings TransformerModel.py:1484-1492, PositionalEncoding :1494-1511; (tok + syn) + pe keeps the reference's order)
__global__ __launch_bounds__(128) void embed_fwd_kernel(const float* __restrict__ lut_tok, const float* __restrict__ lut_syn,
                                                        const float* __restrict__ pe, const int64_t* tok, const int64_t* syn, int L,
                                                        int d, float sqrt_d, float* __restrict__ x, const int64_t* __restrict__ pos) {
    const int r = blockIdx.x;
    const float* tr = (tok && tok[r] >= 0) ? lut_tok + (size_t)tok[r] * d : nullptr;      // negative id: the row has no such term
    const float* sr = (syn && syn[r] >= 0) ? lut_syn + (size_t)syn[r] * d : nullptr;
    const float* pr = pe + (size_t)(pos ? pos[r] : r % L) * d;
    for (int k = threadIdx.x; k < d; k += 128) {
        float v;
        if (tr && sr) v = (tr[k] * sqrt_d + sr[k] * sqrt_d) + pr[k];
        else if (tr || sr) v = (tr ? tr[k] : sr[k]) * sqrt_d + pr[k];
        else v = pr[k];
        x[(size_t)r * d + k] = v;
    }
}

// xt[n][m] = x[m][n] for m < M, 0 for M <= m < Mpad   (operands of the weight-gradient GEMM, whose inner dimension is M);
// the output is written in the GEMM's compute dtype
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ x, int ldx, T* __restrict__ xt, int M, int N, int Mpad,
                                                            float* colsum) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, n = n0 + tx;
        tile[i][tx] = (m < M && n < N) ? x[(size_t)m * ldx + n] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, m = m0 + tx;
        if (n < N && m < Mpad) ElemOps<T>::store(xt + (size_t)n * Mpad + m, tile[tx][i]);
    }
    if (colsum && threadIdx.x < 32 && n0 + tx < N && m0 < M) {   // the tile is in LDS anyway: its column sums are the bias gradient
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += tile[i][tx];
        atomicAdd(&colsum[n0 + tx], s);
    }
}

// y[m][n] = bf16(x'[m][n]) for n < N, 0 for N <= n < ldy, where x' = x with the optional ReLU mask (relu_y: forward output of a
// ReLU layer; x is then dy and passes only where relu_y > 0) and the optional dropout mask of bofi_linear_dropout applied;
// colsum (may be NULL): colsum[n] += sum_m x'[m][n], the bias gradient when x is dz.
// Workgroup = 64 column quads x 4 row lanes over a 256-column x 32-row tile: float4 loads, 8-byte bf16 stores, the four
// row lanes combine their column sums through LDS and the workgroup issues one atomic per column.
template <bool VEC>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int M, int N,
                                                       float* colsum, const float* __restrict__ relu_y, uint32_t drop_thresh,
                                                       float drop_scale, uint64_t drop_seed0, const uint64_t* drop_step) {
    __shared__ float red[4][256];
    const uint64_t drop_seed = drop_seed0 + ((drop_thresh && drop_step) ? *drop_step : 0ull);
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + cq * 4;
    const int m0 = blockIdx.y * 32, m1 = min(M, m0 + 32);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    auto finish = [&](int m, float (&v)[4]) {
        if (drop_thresh) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (n + c < N) v[c] = drop_hash(drop_seed, (uint64_t)m * N + n + c) >= drop_thresh ? v[c] * drop_scale : 0.f;
        }
        uint2 o;
        o.x = pack_bf16(v[0], v[1]);
        o.y = pack_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(y + (size_t)m * ldy + n) = o;              // ldy % 4 == 0 and n % 4 == 0
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += v[c];
    };
    if (n < ldy) {
        if (VEC && n + 3 < N) {
            // the eight rows of this thread are requested before the first one is converted
            float4 t[8], r[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = m0 + rl + 4 * it;
                t[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                r[it] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (m < m1) {
                    t[it] = *reinterpret_cast<const float4*>(x + (size_t)m * ldx + n);
                    if (relu_y) r[it] = *reinterpret_cast<const float4*>(relu_y + (size_t)m * ldx + n);
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = m0 + rl + 4 * it;
                if (m >= m1) break;
                float v[4] = {r[it].x > 0.f ? t[it].x : 0.f, r[it].y > 0.f ? t[it].y : 0.f, r[it].z > 0.f ? t[it].z : 0.f,
                              r[it].w > 0.f ? t[it].w : 0.f};
                finish(m, v);
            }
        } else {
            for (int m = m0 + rl; m < m1; m += 4) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (n + c < N) {
                        v[c] = x[(size_t)m * ldx + n + c];
                        if (relu_y && !(relu_y[(size_t)m * ldx + n + c] > 0.f)) v[c] = 0.f;
                    }
                finish(m, v);
            }
        }
    }
    if (!colsum) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rl][cq * 4 + c] = s[c];
    __syncthreads();
    const int nn = blockIdx.x * 256 + threadIdx.x;
    if (nn < N) atomicAdd(&colsum[nn], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// y = (residual ? residual : 0) + keep(x) / (1 - p)
__global__ void dropout_kernel(const float* __restrict__ x, const float* __restrict__ residual, float* __restrict__ y, size_t n,
                               uint32_t thresh, float scale, uint64_t seed0, const uint64_t* drop_step) {
    const uint64_t seed = seed0 + (drop_step ? *drop_step : 0ull);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = drop_hash(seed, i) >= thresh ? x[i] * scale : 0.f;
        y[i] = residual ? residual[i] + v : v;
    }
}

// dx = dy where y > 0 else 0
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// Adam step over the flat parameter bucket (torch.optim.Adam semantics, misc.py:245-251: betas (0.9, 0.98), eps 1e-9)
// with the averaging of the all-reduced gradient and clip_grad_value_ (train.py:225-226) folded in, and an optional
// bf16 copy of the new parameters for the next step's GEMMs.  7 float streams per element: HBM-bound.
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ shadow, size_t n4, float step_size,
                                                        float b1, float b2, float eps, float inv_sqrt_bc2, float clip, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gi = gp[k] * gscale;
            if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
            mp[k] = mp[k] + (1.f - b1) * (gi - mp[k]);
            vp[k] = b2 * vp[k] + (1.f - b2) * gi * gi;
            pp[k] -= step_size * (mp[k] / (sqrtf(vp[k]) * inv_sqrt_bc2 + eps));
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (shadow) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ElemOps<bf16_t>::store(shadow + i * 4 + k, pp[k]);
        }
    }
}


// Every GEMM weight of the model transposed in ONE launch: entry e of ``table`` = (src offset, dst offset, N, K, Np) says that
// the bf16 matrix [N, K] at src + src_off goes to [K, Np] at dst + dst_off (columns N..Np-1 are never written: the caller
// cleared them once).  tile_first[e] is the index of e's first 64 x 64 tile in the grid.  The backward's dX = dZ W GEMMs take
// their weight operand contraction-major; per-weight transposes were 81 launches a step.
__global__ __launch_bounds__(256) void transpose_many_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const int64_t* __restrict__ table, const int* __restrict__ tile_first, int n) {
    __shared__ bf16_t tile[64][66];
    int lo = 0, hi = n - 1;                                     // last entry whose first tile is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = table + (size_t)lo * 5;
    const int N = (int)e[2], K = (int)e[3], Np = (int)e[4];
    const bf16_t* s = src + e[0];
    bf16_t* d = dst + e[1];
    const int t = (int)blockIdx.x - tile_first[lo], tk = (K + 63) >> 6;
    const int n0 = (t / tk) * 64, k0 = (t % tk) * 64;
    const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
    for (int r = r4; r < 64; r += 4) {
        const int nn = n0 + r, kk = k0 + c;
        tile[r][c] = (nn < N && kk < K) ? s[(size_t)nn * K + kk] : (bf16_t)0;
    }
    __syncthreads();
    for (int r = r4; r < 64; r += 4) {
        const int kk = k0 + r, nn = n0 + c;
        if (kk < K && nn < N) d[(size_t)kk * Np + nn] = tile[c][r];
    }
}

}  // namespace bofi

using namespace bofi;

extern "C" int bofi_transpose_many(const void* src_bf16, void* dst_bf16, const int64_t* table, const int* tile_first, int n, int total_tiles,
                                   void* stream) {
    if (!src_bf16 || !dst_bf16 || !table || !tile_first || n < 0 || total_tiles < 0) return BOFI_ERR_ARG;
    if (n == 0 || total_tiles == 0) return BOFI_OK;
    hipLaunchKernelGGL(transpose_many_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src_bf16, (bf16_t*)dst_bf16,
                       table, tile_first, n);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                              float beta2, float eps, int step, float clip_value, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || n < 0 || n % 4 || step < 1 || ((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) return BOFI_ERR_ARG;
    if (n == 0) return BOFI_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const size_t n4 = (size_t)n / 4;
    const int blocks = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16, n4,
                       (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), clip_value, grad_scale);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_embed_rows(const float* lut_tok, const float* lut_syn, const float* pe, const int64_t* tok, const int64_t* syn,
                               const int64_t* pos, int rows, int L, int d, float* x, void* stream) {
    if (!pe || !x || (!tok && !syn) || (tok && !lut_tok) || (syn && !lut_syn) || rows < 0 || L <= 0 || d <= 0) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(rows), dim3(128), 0, (hipStream_t)stream, lut_tok, lut_syn, pe, tok, syn, L, d,
                       (float)sqrt((double)d), x, pos);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_transpose_pad(const float* x, int ldx, void* xt, int out_dtype, int M, int N, int Mpad, float* colsum, void* stream) {
    if (!x || !xt || M <= 0 || N <= 0 || Mpad < M || ldx < N || (out_dtype != BOFI_DT_F32 && out_dtype != BOFI_DT_BF16)) return BOFI_ERR_ARG;
    const dim3 grid((Mpad + 31) / 32, (N + 31) / 32);
    if (out_dtype == BOFI_DT_F32) hipLaunchKernelGGL((transpose_pad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (float*)xt, M, N, Mpad, colsum);
    else hipLaunchKernelGGL((transpose_pad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16_t*)xt, M, N, Mpad, colsum);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_cast_bf16(const float* x, int ldx, void* y, int ldy, int M, int N, float* colsum, const float* relu_y, float drop_p,
                              uint64_t drop_seed, const uint64_t* drop_step, void* stream) {
    if (!x || !y || M < 0 || N <= 0 || ldx < N || ldy < N || ldy % 4 || ((uintptr_t)y % 8) || !(drop_p >= 0.f && drop_p < 1.f)) return BOFI_ERR_ARG;
    if (M == 0) return BOFI_OK;
    const uint32_t drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
    const float drop_scale = 1.0f / (1.0f - drop_p);
    const dim3 grid((ldy + 255) / 256, (M + 31) / 32);
    const bool vec = (ldx % 4 == 0) && ((uintptr_t)x % 16 == 0) && (!relu_y || (uintptr_t)relu_y % 16 == 0);
    if (vec) hipLaunchKernelGGL((cast_pad_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16_t*)y, ldy, M, N, colsum, relu_y,
                                drop_thresh, drop_scale, drop_seed, drop_step);
    else hipLaunchKernelGGL((cast_pad_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16_t*)y, ldy, M, N, colsum, relu_y,
                            drop_thresh, drop_scale, drop_seed, drop_step);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_dropout(const float* x, const float* residual, float* y, int64_t n, float p, uint64_t seed, const uint64_t* drop_step,
                            void* stream) {
    if (!x || !y || n < 0 || !(p >= 0.f && p < 1.f)) return BOFI_ERR_ARG;
    if (n == 0) return BOFI_OK;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, residual, y, (size_t)n, thresh, 1.0f / (1.0f - p), seed, drop_step);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) {
    if (!y || !dy || !dx || n < 0) return BOFI_ERR_ARG;
    if (n == 0) return BOFI_OK;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, dy, dx, (size_t)n);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_layernorm_bwd_ex(const float* x, const float* gain, const float* dy, const float* add, float* dx, float* dgain, float* dbias,
                                     int rows, int d, void* dz_bf16, float drop_p, uint64_t drop_seed, const uint64_t* drop_step, void* stream);

extern "C" int bofi_layernorm_bwd(const float* x, const float* gain, const float* dy, const float* add, float* dx, float* dgain, float* dbias,
                                  int rows, int d, void* stream) {
    if (!x || !gain || !dy || !dx || !dgain || !dbias || rows < 0 || d <= 1) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    return bofi_layernorm_bwd_ex(x, gain, dy, add, dx, dgain, dbias, rows, d, nullptr, 0.f, 0, nullptr, stream);
}

extern "C" int bofi_layernorm_bwd_ex(const float* x, const float* gain, const float* dy, const float* add, float* dx, float* dgain, float* dbias,
                                     int rows, int d, void* dz_bf16, float drop_p, uint64_t drop_seed, const uint64_t* drop_step, void* stream) {
    if (!x || !gain || !dy || !dx || !dgain || !dbias || rows < 0 || d <= 1 || !(drop_p >= 0.f && drop_p < 1.f)) return BOFI_ERR_ARG;
    if (dz_bf16 && d != 512 && d != 128) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    const uint32_t thresh = (uint32_t)((double)drop_p * 4294967296.0);
    const float scale = 1.0f / (1.0f - drop_p);
    const int rpb = rows >= 4096 ? 32 : 8;                 // the per-column atomics of a workgroup cost more than the lost occupancy
    if (d == 512) hipLaunchKernelGGL((ln_bwd_rows_kernel<8>), dim3((rows + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, x, gain, dy, dx, dgain, dbias, rows, rpb, add, (bf16_t*)dz_bf16, thresh, scale, drop_seed, drop_step);
    else if (d == 128) hipLaunchKernelGGL((ln_bwd_rows_kernel<2>), dim3((rows + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, x, gain, dy, dx, dgain, dbias, rows, rpb, add, (bf16_t*)dz_bf16, thresh, scale, drop_seed, drop_step);
    else hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gain, dy, dx, dgain, dbias, rows, d, add);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* dout, int ldo,
                                  float* dq, float* dk, float* dv, int B, int H, int Lq, int Lk, int kdiv, const int* klen, int klen_sb,
                                  int klen_sq, int klen_bias, const int* q_start, const int* q_count, int k_ragged, void* stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || B < 0 || H <= 0 || Lq <= 0 || Lk <= 0 || Lk > 128 || kdiv <= 0) return BOFI_ERR_ARG;
    if (k_ragged && q_start && Lq > 128) return BOFI_ERR_ARG;          // keys = the item's own rows: at most 128 of them
    if (B == 0) return BOFI_OK;
    if ((q_start != nullptr) != (q_count != nullptr)) return BOFI_ERR_ARG;
    AttnBwdParams p{q, ldq, k, ldk, v, ldv, dout, ldo, dq, dk, dv, B, H, Lq, Lk, kdiv, klen, klen_sb, klen_sq, klen_bias, q_start, q_count, k_ragged};
    const dim3 block(256);
    const int keys = (k_ragged && q_start) ? (Lq > Lk ? Lq : Lk) : Lk;    // the LDS tiles are sized for the most keys an item can have
    // query rows in chunks of 32 or 64 (gridDim.y): more than one chunk -> dk / dv by atomics (the caller zeroed them)
    if (keys > 64) hipLaunchKernelGGL((attn_bwd_kernel<32, 128>), dim3(B * H, (Lq + 31) / 32), block, 0, (hipStream_t)stream, p);
    else if (Lq <= 32 && keys <= 32) hipLaunchKernelGGL((attn_bwd_kernel<32, 32>), dim3(B * H), block, 0, (hipStream_t)stream, p);
    else if (Lq <= 32) hipLaunchKernelGGL((attn_bwd_kernel<32, 64>), dim3(B * H), block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_bwd_kernel<64, 64>), dim3(B * H, (Lq + 63) / 64), block, 0, (hipStream_t)stream, p);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_logsoftmax_bwd(const float* y, const float* dy, float* dx, int rows, int V, void* stream) {
    if (!y || !dy || !dx || rows < 0 || V <= 0) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, y, dy, dx, V);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

static int fill_crit(UicCritArgs& a, const float* sa_len, const float* sa_syn, const float* na_len, const float* na_syn, int N, int Pm, int c_len,
                     int c_syn, const int64_t* phrase_num, const int64_t* phrase_length, const int64_t* phrase_syn, int L, const float* picked,
                     const float* w_sa, const float* w_na, int T) {
    if (!sa_len || !sa_syn || !na_len || !na_syn || !phrase_num || !phrase_length || !phrase_syn || !picked || !w_sa || !w_na) return BOFI_ERR_ARG;
    if (N <= 0 || Pm <= 0 || Pm + 1 > L || c_len <= 0 || c_syn <= 0 || T < 0) return BOFI_ERR_ARG;
    a.lp[0] = sa_len; a.lp[1] = sa_syn; a.lp[2] = na_len; a.lp[3] = na_syn;
    a.C[0] = c_len; a.C[1] = c_syn; a.C[2] = c_len; a.C[3] = c_syn;
    a.phrase_num = phrase_num; a.phrase_length = phrase_length; a.phrase_syn = phrase_syn; a.N = N; a.Pm = Pm; a.L = L;
    a.picked = picked; a.w_sa = w_sa; a.w_na = w_na; a.T = T;
    return BOFI_OK;
}

// The loader's phrase-aware collate of SAMPLED captions (captioning/data/dataloader.py:343-428, the semi-autoregressive half: boficap_amd.collate.phrase_collate's
// index arithmetic) as one launch: thread (n, t) -> the decoder input token, label and key count of position t of caption n.  What the self-critical step's per-phrase
// forwards read (xe.rl_prepare_saic_device, whose tensor-op form is the test's reference: 35 launches inside every replayed forward).
__global__ __launch_bounds__(256) void saic_collate_kernel(const int64_t* __restrict__ seq, const int* __restrict__ plen, const int64_t* __restrict__ psyn, int N, int S,
                                                           int bos_idx, int64_t* __restrict__ sa_syn, int64_t* __restrict__ sa_seq, int* __restrict__ sa_klen) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * S) return;
    const int n = i / S, t = i - n * S, L = S + 2;
    const int* pl = plen + (size_t)n * S;
    int ntok = 0, pid = 0;
    for (int j = 0; j < S; ++j) { ntok += pl[j]; if (ntok <= t) ++pid; }          // phrases that end at or before t
    pid = min(pid, S - 1);
    int start = 0;
    for (int j = 0; j < pid; ++j) start += pl[j];
    const int cur = pl[pid], end = start + cur, k = t - start;
    const bool valid = t < ntok, first = pid == 0;
    const int prev = first ? 1 : pl[pid - 1];
    const int prev_start = first ? 0 : 1 + (start - pl[pid - 1]);
    const int cur_s = max(cur, 1), prev_s = max(prev, 1), times = cur_s / prev_s, pre_less = prev_s - cur_s % prev_s;
    const int stretched = k < pre_less * times ? k / max(times, 1) : pre_less + (k - pre_less * times) / (times + 1);
    const int src = cur <= prev ? prev - cur + k : stretched;
    const int idx = min(max(prev_start + src, 0), L - 1);                           // position in [BOS, the S sampled tokens, 0]
    const int64_t word = idx == 0 ? (int64_t)bos_idx : (idx <= S ? seq[(size_t)n * S + idx - 1] : 0);
    sa_seq[i] = valid ? word : 0;
    sa_syn[i] = valid && cur > 0 ? psyn[(size_t)n * S + pid] : 0;
    sa_klen[i] = valid ? end : ntok;
}

extern "C" int bofi_saic_collate(const int64_t* seq, const int* phrase_length, const int64_t* phrase_syn, int N, int S, int bos_idx, int64_t* sa_syn, int64_t* sa_seq,
                                 int* sa_klen, void* stream) {
    if (!seq || !phrase_length || !phrase_syn || !sa_syn || !sa_seq || !sa_klen || N < 0 || S < 1) return BOFI_ERR_ARG;
    if (N == 0) return BOFI_OK;
    hipLaunchKernelGGL(saic_collate_kernel, dim3((N * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, seq, phrase_length, phrase_syn, N, S, bos_idx, sa_syn, sa_seq, sa_klen);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// The self-critical step's bookkeeping behind a draw (loss_wrapper.py:193-209 around core_SAIC's phrase loop): positions of phrases [p0, p1) of every caption take the
// drawn token, remember its log-prob under the row it was drawn from, and are marked as sampled.  One launch instead of a dozen tensor operations per phrase.
__global__ __launch_bounds__(256) void rl_take_draws_kernel(const float* __restrict__ lp, const int64_t* __restrict__ tok, const int* __restrict__ plen, int N, int S, int V,
                                                            int p0, int p1, int64_t* __restrict__ seq, float* __restrict__ drawn, bool* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * S) return;
    const int n = i / S, t = i - n * S;
    const int* pl = plen + (size_t)n * S;
    int lo = 0, hi = 0;
    for (int j = 0; j < p1 && j < S; ++j) { if (j < p0) lo += pl[j]; hi += pl[j]; }
    if (t < lo || t >= hi) return;
    const int64_t w = tok[i];
    seq[i] = w;
    drawn[i] = (w >= 0 && w < V) ? lp[(size_t)i * V + w] : __builtin_nanf("");
    if (mask) mask[i] = true;
}

extern "C" int bofi_rl_take_draws(const float* lp, const int64_t* tok, const int* phrase_length, int N, int S, int V, int p0, int p1, int64_t* seq, float* drawn, void* mask,
                                  void* stream) {
    if (!lp || !tok || !phrase_length || !seq || !drawn || N < 0 || S < 1 || V < 1 || p0 < 0 || p1 < p0) return BOFI_ERR_ARG;
    if (N == 0) return BOFI_OK;
    hipLaunchKernelGGL(rl_take_draws_kernel, dim3((N * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, lp, tok, phrase_length, N, S, V, p0, p1, seq, drawn, (bool*)mask);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_uic_criterion(const float* sa_len, const float* sa_syn, const float* na_len, const float* na_syn, int N, int Pm, int c_len,
                                  int c_syn, const int64_t* phrase_num, const int64_t* phrase_length, const int64_t* phrase_syn, int L,
                                  const float* picked, const float* w_sa, const float* w_na, int T, float* out8, void* stream) {
    UicCritArgs a;
    if (int rc = fill_crit(a, sa_len, sa_syn, na_len, na_syn, N, Pm, c_len, c_syn, phrase_num, phrase_length, phrase_syn, L, picked, w_sa, w_na, T)) return rc;
    if (!out8) return BOFI_ERR_ARG;
    hipLaunchKernelGGL(uic_criterion_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, out8);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_uic_criterion_bwd(const float* sa_len, const float* sa_syn, const float* na_len, const float* na_syn, int N, int Pm, int c_len,
                                      int c_syn, const int64_t* phrase_num, const int64_t* phrase_length, const int64_t* phrase_syn, int L,
                                      const float* picked, const float* w_sa, const float* w_na, int T, const float* g_loss, const float* out8,
                                      float* d_sa_len, float* d_sa_syn, float* d_na_len, float* d_na_syn, float* d_picked, void* stream) {
    UicCritArgs a;
    if (int rc = fill_crit(a, sa_len, sa_syn, na_len, na_syn, N, Pm, c_len, c_syn, phrase_num, phrase_length, phrase_syn, L, picked, w_sa, w_na, T)) return rc;
    if (!g_loss || !out8 || !d_sa_len || !d_sa_syn || !d_na_len || !d_na_syn || !d_picked) return BOFI_ERR_ARG;
    const int work = N * Pm > T ? N * Pm : T;
    hipLaunchKernelGGL(uic_criterion_bwd_kernel, dim3((work + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, g_loss, out8, d_sa_len, d_sa_syn,
                       d_na_len, d_na_syn, d_picked);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_nll_bwd(const float* y, const int64_t* labels, const float* dpicked, void* dx, int dx_dtype, int lddx, int rows, int V,
                            void* stream) {
    if (!y || !labels || !dpicked || !dx || rows < 0 || V <= 0 || lddx < V) return BOFI_ERR_ARG;
    if (dx_dtype != BOFI_DT_F32 && dx_dtype != BOFI_DT_BF16) return BOFI_ERR_ARG;
    if (dx_dtype == BOFI_DT_F32 && lddx != V) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    const bool b16 = dx_dtype == BOFI_DT_BF16;
    hipLaunchKernelGGL(nll_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, y, labels, dpicked, b16 ? nullptr : (float*)dx,
                       b16 ? (bf16_t*)dx : nullptr, lddx, V);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_colsum_add(const float* x, float* out, int M, int N, void* stream) {
    if (!x || !out || M < 0 || N <= 0) return BOFI_ERR_ARG;
    if (M == 0) return BOFI_OK;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, out, M, N);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

extern "C" int bofi_embed_bwd(const float* dx, const int64_t* ids, float* dlut, int rows, int d, float scale, void* stream) {
    if (!dx || !ids || !dlut || rows < 0 || d <= 0) return BOFI_ERR_ARG;
    if (rows == 0) return BOFI_OK;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((rows + 31) / 32, (d + 127) / 128), dim3(128), 0, (hipStream_t)stream, dx, ids, dlut, rows, d, scale);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}
