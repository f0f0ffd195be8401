// Device-side refresh of the decode engine's packed weights from float32 parameters that already live in HBM (the
// trainer's flat bucket): what bofi_engine_finalize does on the host -- stack q|k|v, fold the pre-norm LayerNorm into the
// consumer GEMM (w' = w * gain, c[n] = bias[n] + sum_k b_ln[k] w[n][k], colsum[n] = sum_k round(w'[n][k])), cast to the
// compute dtype, transpose the bound heads' first layers, rebuild the bound layer's input table -- as a handful of
// kernels, so that a training loop can decode with its current weights without a host round trip.
#include "bofi_common.h"
#include "bofi_kernels.h"
#include "bofi_naic.h"

namespace bofi {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_lin_kernel(PackLinArgs a, T* __restrict__ wout) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.n_each * a.nsrc) return;
    const int src = n / a.n_each, r = n - src * a.n_each;
    const float* row = a.w[src] + (size_t)r * a.K;
    T* out = wout + (size_t)n * a.K;
    double c = 0.0, s = 0.0;
    for (int k = lane; k < a.K; k += 64) {
        float wv = row[k];
        if (a.gain) {
            c += (double)a.bln[k] * (double)wv;
            wv = wv * a.gain[k];
        }
        ElemOps<T>::store(out + k, wv);
        float rv = wv;
        if constexpr (sizeof(T) == 2) rv = bf16_to_f32(f32_to_bf16(wv));
        s += (double)rv;
    }
    if (a.gain) { c = wave_sum_f64(c); s = wave_sum_f64(s); }
    if (lane == 0) {
        a.bout[n] = (float)((double)a.b[src][r] + c);
        if (a.gain) a.cs[n] = (float)s;
    }
}

// ---- the batched forms (bofi_engine_refresh_device: ~230 launches and copies of a refresh as three): a workgroup finds its entry by bisection
// over the table's first rows / blocks
template <typename F>
__device__ __forceinline__ int find_entry(int n_ent, int x, F first) {      // largest e with first(e) <= x
    int lo = 0, hi = n_ent - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first(mid) <= x) lo = mid; else hi = mid - 1; }
    return lo;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_lin_multi_kernel(const PackLinDesc* __restrict__ tab, int n_ent) {
    const int lane = threadIdx.x & 63, grow = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int e = find_entry(n_ent, grow, [&](int i) { return tab[i].row0; });
    const PackLinArgs& a = tab[e].a;
    const int n = grow - tab[e].row0;
    if (n >= a.n_each * a.nsrc) return;
    const int src = n / a.n_each, r = n - src * a.n_each;
    const float* row = a.w[src] + (size_t)r * a.K;
    T* out = static_cast<T*>(tab[e].wout) + (size_t)n * a.K;
    double c = 0.0, s = 0.0;
    for (int k = lane; k < a.K; k += 64) {            // (the same sums in the same order as pack_lin_kernel)
        float wv = row[k];
        if (a.gain) {
            c += (double)a.bln[k] * (double)wv;
            wv = wv * a.gain[k];
        }
        ElemOps<T>::store(out + k, wv);
        float rv = wv;
        if constexpr (sizeof(T) == 2) rv = bf16_to_f32(f32_to_bf16(wv));
        s += (double)rv;
    }
    if (a.gain) { c = wave_sum_f64(c); s = wave_sum_f64(s); }
    if (lane == 0) {
        a.bout[n] = (float)((double)a.b[src][r] + c);
        if (a.gain) a.cs[n] = (float)s;
    }
}

int launch_pack_lin_multi(const PackLinDesc* tab_dev, int n_ent, int total_rows, int dtype, hipStream_t st) {
    if (!tab_dev || n_ent <= 0 || total_rows <= 0 || total_rows % 4) return BOFI_ERR_ARG;
    if (dtype == BOFI_DT_F32) hipLaunchKernelGGL((pack_lin_multi_kernel<float>), dim3(total_rows / 4), dim3(256), 0, st, tab_dev, n_ent);
    else hipLaunchKernelGGL((pack_lin_multi_kernel<bf16_t>), dim3(total_rows / 4), dim3(256), 0, st, tab_dev, n_ent);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// the fragment-major copies of every entry that has one (the layout of rowblock.hip's rb_pack_frag_kernel: [N/64 chunks][K/32 steps][4 tiles][64 lanes][8 bf16])
typedef __attribute__((ext_vector_type(4))) uint32_t rp_u32x4;
__global__ __launch_bounds__(256) void pack_frag_multi_kernel(const PackLinDesc* __restrict__ tab, int n_ent) {
    const int e = find_entry(n_ent, (int)blockIdx.x, [&](int i) { return tab[i].blk0; });
    const int N = tab[e].Npad, K = tab[e].a.K;
    const size_t i = (size_t)((int)blockIdx.x - tab[e].blk0) * 256 + threadIdx.x;
    if (!tab[e].wp || i >= (size_t)N * K / 8) return;
    const int lane = (int)(i & 63), nt = (int)((i >> 6) & 3);
    const size_t t = i >> 8;
    const int kb = (int)(t % (size_t)(K >> 5)), chunk = (int)(t / (size_t)(K >> 5));
    const int n = chunk * 64 + nt * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8;
    static_cast<rp_u32x4*>(tab[e].wp)[i] = *reinterpret_cast<const rp_u32x4*>(static_cast<const bf16_t*>(tab[e].wout) + (size_t)n * K + k);
}

int launch_pack_frag_multi(const PackLinDesc* tab_dev, int n_ent, int total_blocks, hipStream_t st) {
    if (!tab_dev || n_ent <= 0) return BOFI_ERR_ARG;
    if (total_blocks <= 0) return BOFI_OK;
    hipLaunchKernelGGL(pack_frag_multi_kernel, dim3(total_blocks), dim3(256), 0, st, tab_dev, n_ent);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyDesc* __restrict__ tab, int n_ent) {
    const int e = find_entry(n_ent, (int)blockIdx.x, [&](int i) { return tab[i].blk0; });
    const int i = ((int)blockIdx.x - tab[e].blk0) * 256 + threadIdx.x;
    if (i < tab[e].n) tab[e].dst[i] = tab[e].src[i];
}

int launch_copy_multi(const CopyDesc* tab_dev, int n_ent, int total_blocks, hipStream_t st) {
    if (!tab_dev || n_ent <= 0 || total_blocks <= 0) return BOFI_ERR_ARG;
    hipLaunchKernelGGL(copy_multi_kernel, dim3(total_blocks), dim3(256), 0, st, tab_dev, n_ent);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

int launch_pack_lin(const PackLinArgs& a, void* wout, int dtype, hipStream_t st) {
    const int N = a.n_each * a.nsrc;
    if (N <= 0 || a.K <= 0 || a.nsrc > 16 || !wout || !a.bout || (a.gain && (!a.bln || !a.cs))) return BOFI_ERR_ARG;
    if (dtype == BOFI_DT_F32) hipLaunchKernelGGL((pack_lin_kernel<float>), dim3((N + 3) / 4), dim3(256), 0, st, a, (float*)wout);
    else hipLaunchKernelGGL((pack_lin_kernel<bf16_t>), dim3((N + 3) / 4), dim3(256), 0, st, a, (bf16_t*)wout);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// bound heads: first layers of both heads transposed side by side [d][2*hh], biases concatenated
__global__ void pack_heads_kernel(const float* __restrict__ lw1, const float* __restrict__ sw1, const float* __restrict__ lb1,
                                  const float* __restrict__ sb1, float* __restrict__ w1t, float* __restrict__ b1, int d, int hh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hh * d) {
        const int j = i / d, k = i - j * d;
        w1t[(size_t)k * 2 * hh + j] = lw1[i];
        w1t[(size_t)k * 2 * hh + hh + j] = sw1[i];
    }
    if (i < hh) { b1[i] = lb1[i]; b1[hh + i] = sb1[i]; }
}

int launch_pack_heads(const float* lw1, const float* sw1, const float* lb1, const float* sb1, float* w1t, float* b1, int d, int hh, hipStream_t st) {
    hipLaunchKernelGGL(pack_heads_kernel, dim3((hh * d + 255) / 256), dim3(256), 0, st, lw1, sw1, lb1, sb1, w1t, b1, d, hh);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// bound-layer input table: xt[(p*10 + s)*d + k] = lut_syn[s][k]*sqrt(d) + pe[p][k]; x0 = row (0, len_idx); x0_sa from the word table
__global__ void bound_table_kernel(const float* __restrict__ lut_syn, const float* __restrict__ lut_tok, const float* __restrict__ pe,
                                   float* __restrict__ xt, float* __restrict__ x0, float* __restrict__ x0_sa, int L, int d, int len_idx, float sq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L * 10 * d) {
        const int k = i % d, ps = i / d, s = ps % 10, p = ps / 10;
        const float v = lut_syn[(size_t)s * d + k] * sq + pe[(size_t)p * d + k];
        xt[i] = v;
        if (p == 0 && s == len_idx) x0[k] = v;
    }
    if (i < d) x0_sa[i] = lut_tok[(size_t)len_idx * d + i] * sq + pe[i];
}

int launch_bound_table(const float* lut_syn, const float* lut_tok, const float* pe, float* xt, float* x0, float* x0_sa, int L, int d, int len_idx,
                       hipStream_t st) {
    hipLaunchKernelGGL(bound_table_kernel, dim3((L * 10 * d + 255) / 256), dim3(256), 0, st, lut_syn, lut_tok, pe, xt, x0, x0_sa, L, d, len_idx,
                       (float)sqrt((double)d));
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// hidden weights of the bound heads for the tail kernel.  Thread (slice, grp) of that kernel owns outputs grp*4 .. +3 over the k values
// slice*kps .. +kps and reads them as 16-byte pieces of KPL k values x 4 outputs; piece u of all groups of a slice is contiguous
// (a wave-instruction's pieces are adjacent): w1p[((slice*nld + u)*ng + grp)*EPL + kk*4 + o] = w1t[(slice*kps + u*KPL + kk)*nh + grp*4 + o]
template <typename T>
__global__ void pack_w1p_kernel(const float* __restrict__ w1t, T* __restrict__ w1p, int d, int nh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d * nh) return;
    constexpr int EPL = 16 / sizeof(T), KPL = EPL / 4;
    const int ng = nh / 4, kps = d / 8, nld = kps / KPL;
    const int o = i & 3, kk = (i >> 2) % KPL, piece = i / EPL, grp = piece % ng, su = piece / ng, u = su % nld, slice = su / nld;
    ElemOps<T>::store(w1p + i, w1t[(size_t)(slice * kps + u * KPL + kk) * nh + grp * 4 + o]);
}

int launch_pack_w1p(const float* w1t, void* w1p, int dtype, int d, int nh, hipStream_t st) {
    if (d % 8 || nh % 4) return BOFI_ERR_ARG;
    const int n = d * nh;
    if (dtype == BOFI_DT_F32) hipLaunchKernelGGL((pack_w1p_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, st, w1t, (float*)w1p, d, nh);
    else hipLaunchKernelGGL((pack_w1p_kernel<bf16_t>), dim3((n + 255) / 256), dim3(256), 0, st, w1t, (bf16_t*)w1p, d, nh);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

// votab[row][h][c] = sum_e wo[c][h*64 + e] * V[row][h*64 + e] (V = second half of a kvtab row); x0b = x0 + bo.  Operands and
// result in the compute dtype, float32 accumulation.  Grid (rows, H), one thread per output column.
template <typename T>
__global__ void votab_kernel(const T* __restrict__ kvtab, const T* __restrict__ wo, const float* __restrict__ x0, const float* __restrict__ bo,
                             T* __restrict__ votab, float* __restrict__ x0b, int d, int H) {
    __shared__ float vs[64];
    const int row = blockIdx.x, h = blockIdx.y;
    if (threadIdx.x < 64) vs[threadIdx.x] = ElemOps<T>::to_f32(kvtab[(size_t)row * 2 * d + d + h * 64 + threadIdx.x]);
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const T* wr = wo + (size_t)c * d + h * 64;
        float acc = 0.f;
        for (int e = 0; e < 64; ++e) acc = fmaf(ElemOps<T>::to_f32(wr[e]), vs[e], acc);
        ElemOps<T>::store(votab + ((size_t)row * H + h) * d + c, acc);
        if (row == 0 && h == 0) x0b[c] = x0[c] + bo[c];
    }
}

int launch_votab(const void* kvtab, const void* wo, const float* x0, const float* bo, void* votab, float* x0b, int dtype, int rows, int d, int H,
                 hipStream_t st) {
    if (H * 64 != d || rows <= 0) return BOFI_ERR_ARG;
    if (dtype == BOFI_DT_F32)
        hipLaunchKernelGGL((votab_kernel<float>), dim3(rows, H), dim3(256), 0, st, (const float*)kvtab, (const float*)wo, x0, bo, (float*)votab, x0b, d, H);
    else
        hipLaunchKernelGGL((votab_kernel<bf16_t>), dim3(rows, H), dim3(256), 0, st, (const bf16_t*)kvtab, (const bf16_t*)wo, x0, bo, (bf16_t*)votab, x0b, d, H);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi
