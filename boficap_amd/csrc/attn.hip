// Multi-head attention core for the short sequences of the bound+fill captioner
// (reference attention() captioning/models/TransformerModel.py:1421-1432, called from
// MultiHeadedAttention.forward :1446-1467):   softmax(q k^T / sqrt(d_k), masked) v,   d_k = 64.
//
// Shapes on this path: 36x36 region self-attention, 20x20 / 22x22 slot self-attention, 20x36 and
// 1x36 slot->region cross-attention.  Every mask the reference builds here is a per-query-row key
// PREFIX, so the kernel takes an int length per (image, row) instead of a dense bool mask.
//
// One wavefront per (image, head, 48-query block).  Q, K and V^T of the head are staged in LDS,
// QK^T and PV run on MFMA (16x16x32 bf16, or exact-f32 16x16x4), the softmax runs in float32 with
// 4 lanes per row and wave shuffles.  A row whose length is 0 produces NaN, as softmax over all
// -inf does in the reference (this is what quirk Q1's empty-last-row batch relies on).
#include <cstdlib>

#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

template <typename T> struct AMma;
template <> struct AMma<bf16_t> {
    static constexpr int KSTEP = 32;
    typedef bf16x8 Frag;
    static __device__ __forceinline__ Frag load(const bf16_t* row, int s, int lane) {
        return *reinterpret_cast<const bf16x8*>(row + s * 32 + (lane >> 4) * 8);
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct AMma<float> {
    static constexpr int KSTEP = 16;      // one float4 fragment = four 16x16x4 steps (see gemm.hip)
    typedef float4 Frag;
    static __device__ __forceinline__ Frag load(const float* row, int s, int lane) {
        return *reinterpret_cast<const float4*>(row + s * 16 + (lane >> 4) * 4);
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
        return c;
    }
};

struct AttnParams {
    const void* q; int ldq;
    const void* k; int ldk;
    const void* v; int ldv;
    void* out; int ldo;
    int B, H, Lq, Lk;
    const int* klen; int klen_sb, klen_sq, klen_bias, klen_shared_last;
    const int* skip_if_ge; int skip_threshold;
    int kdiv;
    const int* q_start; const int* q_count; int k_ragged;     // unpadded layout (bofi_kernels.h)
};

constexpr int DK = 64;

// LQ: query rows per block (multiple of 16); LK: key capacity (multiple of 32)
template <typename T, int LQ, int LK>
__global__ __launch_bounds__(64) void attn_kernel(AttnParams p) {
    constexpr int EPC = 16 / sizeof(T);               // elements per 16-byte chunk
    constexpr int QS = DK + EPC;                       // padded row strides (elements)
    constexpr int VS = LK + EPC;
    constexpr int SS = LK + 1;
    __shared__ __attribute__((aligned(16))) T sq[LQ * QS];
    __shared__ __attribute__((aligned(16))) T sk[LK * QS];
    __shared__ __attribute__((aligned(16))) T svt[DK * VS];      // V transposed: [d][key]
    __shared__ __attribute__((aligned(16))) T sp[LQ * VS];       // probabilities [q][key]
    __shared__ float ss[LQ * SS];                                 // scores

    if (p.skip_if_ge && *p.skip_if_ge >= p.skip_threshold) return;

    const int lane = threadIdx.x;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.y * LQ;
    const int qrow0 = p.q_start ? p.q_start[b] : b * p.Lq, lq = p.q_start ? p.q_count[b] : p.Lq;
    const int nq = min(LQ, lq - q0);
    if (nq <= 0) return;
    const int Lk = (p.q_start && p.k_ragged) ? lq : p.Lk;
    const int nkt = (Lk + 15) >> 4;                    // 16-key tiles that hold real keys
    const int lkr = ((Lk + 31) >> 5) << 5;             // keys rounded up for the PV k-steps

    // ---- stage Q (rows >= nq zero), K (rows >= Lk zero), V^T (keys >= Lk zero)
    constexpr int CPR = DK / EPC;                      // 16-byte chunks per row
    const T* qg = static_cast<const T*>(p.q) + ((size_t)qrow0 + q0) * p.ldq + h * DK;
    for (int c = lane; c < LQ * CPR; c += 64) {
        const int r = c / CPR, ch = c - r * CPR;
        u32x4 val = u32x4{0u, 0u, 0u, 0u};
        if (r < nq) val = *reinterpret_cast<const u32x4*>(qg + (size_t)r * p.ldq + ch * EPC);
        *reinterpret_cast<u32x4*>(&sq[r * QS + ch * EPC]) = val;
    }
    const int bk = b / p.kdiv;                         // captions of one image share its keys (training)
    const size_t krow0 = (p.q_start && p.k_ragged) ? (size_t)qrow0 : (size_t)bk * p.Lk;
    const T* kg = static_cast<const T*>(p.k) + krow0 * p.ldk + h * DK;
    for (int c = lane; c < nkt * 16 * CPR; c += 64) {
        const int r = c / CPR, ch = c - r * CPR;
        u32x4 val = u32x4{0u, 0u, 0u, 0u};
        if (r < Lk) val = *reinterpret_cast<const u32x4*>(kg + (size_t)r * p.ldk + ch * EPC);
        *reinterpret_cast<u32x4*>(&sk[r * QS + ch * EPC]) = val;
    }
    const T* vg = static_cast<const T*>(p.v) + krow0 * p.ldv + h * DK;
    for (int c = lane; c < lkr * CPR; c += 64) {
        const int r = c / CPR, ch = c - r * CPR;
        union { u32x4 v; T e[EPC]; } u;
        u.v = u32x4{0u, 0u, 0u, 0u};
        if (r < Lk) u.v = *reinterpret_cast<const u32x4*>(vg + (size_t)r * p.ldv + ch * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) svt[(ch * EPC + e) * VS + r] = u.e[e];
    }
    __syncthreads();

    // ---- S = Q K^T / sqrt(d_k), masked to the row's key prefix
    const float inv = 1.0f / 8.0f;                     // d_k = 64: the division by sqrt(64) is exact
    for (int qi = 0; qi < LQ / 16; ++qi) {
        if (qi * 16 >= nq) break;
        typename AMma<T>::Frag fa[DK / AMma<T>::KSTEP];
#pragma unroll
        for (int s = 0; s < DK / AMma<T>::KSTEP; ++s) fa[s] = AMma<T>::load(&sq[(qi * 16 + (lane & 15)) * QS], s, lane);
        for (int kj = 0; kj < nkt; ++kj) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DK / AMma<T>::KSTEP; ++s)
                acc = AMma<T>::mma(fa[s], AMma<T>::load(&sk[(kj * 16 + (lane & 15)) * QS], s, lane), acc);
            const int col = kj * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) ss[(qi * 16 + (lane >> 4) * 4 + r) * SS + col] = acc[r] * inv;
        }
    }
    __syncthreads();

    // ---- softmax rows: 4 lanes per row, 16 rows per pass
    for (int r0 = 0; r0 < LQ; r0 += 16) {
        if (r0 >= nq) break;
        const int r = r0 + (lane >> 2), sub = lane & 3;
        int kl = Lk;
        if (p.klen && r < nq) {
            // quirk Q1 per group of klen_shared_last images: every image uses the key count of its group's LAST image
            const int bi = p.klen_shared_last ? min(p.B, (b / p.klen_shared_last + 1) * p.klen_shared_last) - 1 : b;
            kl = (p.q_start ? p.klen[qrow0 + q0 + r] : p.klen[bi * p.klen_sb + (q0 + r) * p.klen_sq]) + p.klen_bias;
            kl = max(0, min(kl, Lk));
        }
        float m = -INFINITY;
        for (int c = sub; c < kl; c += 4) m = fmaxf(m, ss[r * SS + c]);
        m = quad_max(m);
        float sum = 0.f;
        for (int c = sub; c < kl; c += 4) {
            const float e = expf(ss[r * SS + c] - m);
            ss[r * SS + c] = e;
            sum += e;
        }
        sum = quad_sum(sum);
        const bool empty = (kl == 0) && (r < nq);      // softmax over all -inf -> NaN in the reference
        for (int c = sub; c < lkr; c += 4) {
            float pv = 0.f;
            if (c < kl) pv = ss[r * SS + c] / sum;
            if (empty && c < Lk) pv = __builtin_nanf("");
            sp[r * VS + c] = ElemOps<T>::from_f32(pv);
        }
    }
    __syncthreads();

    // ---- O = P V
    T* og = static_cast<T*>(p.out) + ((size_t)qrow0 + q0) * p.ldo + h * DK;
    for (int qi = 0; qi < LQ / 16; ++qi) {
        if (qi * 16 >= nq) break;
        f32x4 acc[DK / 16];
#pragma unroll
        for (int j = 0; j < DK / 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < lkr / AMma<T>::KSTEP; ++s) {
            const typename AMma<T>::Frag fa = AMma<T>::load(&sp[(qi * 16 + (lane & 15)) * VS], s, lane);
#pragma unroll
            for (int j = 0; j < DK / 16; ++j)
                acc[j] = AMma<T>::mma(fa, AMma<T>::load(&svt[(j * 16 + (lane & 15)) * VS], s, lane), acc[j]);
        }
#pragma unroll
        for (int j = 0; j < DK / 16; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qi * 16 + (lane >> 4) * 4 + r;
                if (row < nq) ElemOps<T>::store(og + (size_t)row * p.ldo + j * 16 + (lane & 15), acc[j][r]);
            }
    }
}

template <typename T>
static int launch_attn_t(const AttnParams& p, hipStream_t st) {
    const dim3 block(64);
    if (p.Lk <= 64) {
        const dim3 grid(p.B * p.H, (p.Lq + 47) / 48);
        hipLaunchKernelGGL((attn_kernel<T, 48, 64>), grid, block, 0, st, p);
    } else {
        const dim3 grid(p.B * p.H, (p.Lq + 31) / 32);
        hipLaunchKernelGGL((attn_kernel<T, 32, 128>), grid, block, 0, st, p);
    }
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

int launch_attention(const AttnArgs& a, hipStream_t st) {
    if (!a.q || !a.k || !a.v || !a.out || a.B < 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0 || a.Lk > 128) return BOFI_ERR_ARG;
    if (a.B == 0) return BOFI_OK;
    const int el = a.dtype == BOFI_DT_F32 ? 4 : 2;
    if (a.dtype != BOFI_DT_F32 && a.dtype != BOFI_DT_BF16) return BOFI_ERR_ARG;
    if ((a.ldq * el) % 16 || (a.ldk * el) % 16 || (a.ldv * el) % 16) return BOFI_ERR_ARG;
    if (((uintptr_t)a.q % 16) || ((uintptr_t)a.k % 16) || ((uintptr_t)a.v % 16)) return BOFI_ERR_ARG;
    if (!getenv("BOFI_ATTN_GENERIC")) {                 // bf16, <= 64 keys: the register-resident kernel
        const int rc = launch_attention_bf16(a, st);
        if (rc >= 0) return rc;
    }
    if (a.drop_thresh) return BOFI_ERR_ARG;             // attention dropout lives in the bf16 kernel only
    AttnParams p;
    p.q = a.q; p.ldq = a.ldq; p.k = a.k; p.ldk = a.ldk; p.v = a.v; p.ldv = a.ldv; p.out = a.out; p.ldo = a.ldo;
    p.B = a.B; p.H = a.H; p.Lq = a.Lq; p.Lk = a.Lk;
    p.klen = a.klen; p.klen_sb = a.klen_sb; p.klen_sq = a.klen_sq; p.klen_bias = a.klen_bias;
    p.kdiv = a.kdiv > 0 ? a.kdiv : 1;
    if ((a.q_start != nullptr) != (a.q_count != nullptr)) return BOFI_ERR_ARG;
    p.q_start = a.q_start; p.q_count = a.q_count; p.k_ragged = a.k_ragged;
    p.klen_shared_last = a.klen_shared_last; p.skip_if_ge = a.skip_if_ge; p.skip_threshold = a.skip_threshold;
    return a.dtype == BOFI_DT_F32 ? launch_attn_t<float>(p, st) : launch_attn_t<bf16_t>(p, st);
}

}  // namespace bofi

extern "C" int bofi_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                              int dtype, int B, int H, int Lq, int Lk, const int* klen, int klen_sb, int klen_sq,
                              void* stream) {
    bofi::AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo; a.dtype = dtype;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.klen = klen; a.klen_sb = klen_sb; a.klen_sq = klen_sq;
    return bofi::launch_attention(a, (hipStream_t)stream);
}

extern "C" int bofi_attention_ex(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                                 int dtype, int B, int H, int Lq, int Lk, int kdiv, const int* klen, int klen_sb, int klen_sq,
                                 int klen_bias, float drop_p, uint64_t drop_seed, const uint64_t* drop_step, const int* q_start,
                                 const int* q_count, int k_ragged, int q_rows, void* stream) {
    if (kdiv <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return BOFI_ERR_ARG;
    bofi::AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo; a.dtype = dtype;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.klen = klen; a.klen_sb = klen_sb; a.klen_sq = klen_sq;
    a.klen_bias = klen_bias; a.kdiv = kdiv; a.q_start = q_start; a.q_count = q_count; a.k_ragged = k_ragged;
    a.q_rows = (q_start && dtype == BOFI_DT_BF16) ? q_rows : 0;       // (only the bf16 kernel clears the trailing rows)
    if (drop_p > 0.f) { a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_scale = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.drop_step = drop_step; }
    return bofi::launch_attention(a, (hipStream_t)stream);
}
