// Attention backward on the matrix cores (bf16 operands, fp32 accumulation) for the sequence lengths of this model
// (Lq, Lk <= 64, head dim 64).  One wavefront per (caption, head); everything it needs lives in LDS as bf16:
//
//   S  = Q K^T / 8, dP = dO V^T                     operands read row-wise (contraction along the head dim)
//   P  = softmax(S | key < klen), dS = P (dP - rowsum(P dP)) / 8        in registers, rows reduced over 16-lane groups
//   dQ = dS K                                       A = dS rows (two 8-byte reads in the transposed-read k order),
//                                                   B = K gathered with the transposing LDS read (ds_read_b64_tr_b16)
//   dK = dS^T Q,  dV = P^T dO                       both operands gathered with the transposing read
//
// so no transposed copy of anything is ever written.  q, k, v may be float32 (rounded to bf16 on the way into LDS) or
// bf16; dO, dQ, dK, dV are float32.  Captions of one image share its keys (kdiv): the wavefront owning (image, head) walks
// them and sums dK / dV in registers -- plain stores, no atomics.  With NW > 1 the workgroup has NW wavefronts that take the
// captions (or, for unpadded rows, 16*QT-row chunks of the image's contiguous query rows) round robin and add their dK / dV
// through LDS at the end: a wavefront per image and head alone leaves half the SIMDs idle and serialises ten captions.
#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct AttnBwdMfmaParams {
    const void* q; int ldq; const void* k; int ldk; const void* v; int ldv;
    const float* dout; int ldo;
    void* dq; int lddq; void* dk; void* dv; int lddk;          // float32, or bf16 where dq_bf16 / dkv_bf16 (then they feed a GEMM directly)
    int dq_bf16, dkv_bf16;
    int B, H, Lq, Lk, kdiv;
    const int* klen; int klen_sb, klen_sq, klen_bias;
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed; const uint64_t* drop_step;     // dropout(p_attn) of the forward
    const int* q_start; const int* q_count; int k_ragged;     // unpadded layout (bofi_kernels.h: AttnArgs)
    int q_rows;                                               // > 0: rows of the dq buffer; rows behind the last item are cleared
};

// Head slice [rows_real, 64] -> LDS as bf16, rows past the data zero, in two phases: load() requests every 16-byte piece of
// the slice into registers (ROWS is a compile-time constant, the loops unroll: one memory round trip per operand), store()
// converts and writes them to LDS.  Keeping the phases apart lets the next caption's slices travel while the current one is
// being multiplied.
template <typename TIN, int ROWS>
struct RowStage {
    static constexpr int IT = ROWS * 8 / 64;
    static constexpr int NV = sizeof(TIN) == 4 ? 2 * IT : IT;
    u32x4 v[NV];
    __device__ __forceinline__ void load(const TIN* __restrict__ src, int ld, int rows_real, int lane) {
#pragma unroll
        for (int t = 0; t < IT; ++t) {
            const int i = lane + 64 * t, r = i >> 3, c = (i & 7) * 8;
            if constexpr (sizeof(TIN) == 4) {
                v[2 * t] = u32x4{0u, 0u, 0u, 0u};
                v[2 * t + 1] = v[2 * t];
                if (r < rows_real) {
                    v[2 * t] = *reinterpret_cast<const u32x4*>(src + (size_t)r * ld + c);
                    v[2 * t + 1] = *reinterpret_cast<const u32x4*>(src + (size_t)r * ld + c + 4);
                }
            } else {
                v[t] = u32x4{0u, 0u, 0u, 0u};
                if (r < rows_real) v[t] = *reinterpret_cast<const u32x4*>(src + (size_t)r * ld + c);
            }
        }
    }
    __device__ __forceinline__ void store(bf16_t* dst, int stride, int lane) const {
#pragma unroll
        for (int t = 0; t < IT; ++t) {
            const int i = lane + 64 * t, r = i >> 3, c = (i & 7) * 8;
            if constexpr (sizeof(TIN) == 4) {
                const u32x4 a = v[2 * t], b = v[2 * t + 1];
                u32x4 o;
                o[0] = pack_bf16(__uint_as_float(a[0]), __uint_as_float(a[1]));
                o[1] = pack_bf16(__uint_as_float(a[2]), __uint_as_float(a[3]));
                o[2] = pack_bf16(__uint_as_float(b[0]), __uint_as_float(b[1]));
                o[3] = pack_bf16(__uint_as_float(b[2]), __uint_as_float(b[3]));
                *reinterpret_cast<u32x4*>(dst + r * stride + c) = o;
            } else {
                *reinterpret_cast<u32x4*>(dst + r * stride + c) = v[t];
            }
        }
    }
};

// MFMA operand gathered from a row-major LDS tile whose ROWS are the contraction index: rows row_base .. row_base + 31,
// operand index (A row / B column) = col_base + (lane & 15)
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* tile, int stride, int row_base, int col_base, int g, int tq, int tp) {
    const bf16_t* p0 = tile + (row_base + 4 * g + tq) * stride + col_base + 4 * tp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * stride));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// the matching operand when the contraction index runs along the COLUMNS of the tile: same k order as frag_tr
// (lane group g holds k = 4g..4g+3 and 16+4g..16+4g+3 of the 32-wide step)
__device__ __forceinline__ bf16x8 frag_row_trorder(const bf16_t* tile, int stride, int row, int col_base, int g) {
    const s16x4 lo = *reinterpret_cast<const s16x4*>(tile + row * stride + col_base + 4 * g);
    const s16x4 hi = *reinterpret_cast<const s16x4*>(tile + row * stride + col_base + 16 + 4 * g);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// LDS traffic of one wavefront is in order, so between its own writes and reads of its private staging tiles a wavefront
// only needs the compiler not to reorder them (the waits on lgkmcnt are inserted per access)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (the 32 x 32 instantiation is asked to fit two wavefronts per SIMD: self-attention launches one single-wavefront workgroup per
// caption and head, thousands of them, and at one wavefront per SIMD they run in five rounds)
template <typename TIN, int QT, int KT, int NW>
__global__ __launch_bounds__(64 * NW, (QT == 2 && KT == 2) ? 2 : 1) void attn_bwd_mfma_kernel(AttnBwdMfmaParams p) {
    // padded strides (elements): 160 B for the head-dim tiles, 96 / 160 B for the [q][k] tiles -- odd multiples of 32 B, so that the
    // 8 rows x 32 B a transposing read touches per 32-lane half tile the 256-byte bank row (144 B / 80 B strides cost 30 % of the
    // LDS cycles in bank conflicts, SQ_LDS_BANK_CONFLICT); the b128 row reads are conflict-free at 160 B too
    constexpr int DS = 80, PS = KT == 2 ? 48 : 80;
    constexpr int LQ = QT * 16, LK = KT * 16;
    constexpr int STAGE = 2 * LQ * DS + 2 * LQ * PS;             // one wavefront's private tiles: Q, dO, P, dS (elements)
    __shared__ __attribute__((aligned(16))) bf16_t sk[LK * DS], sv[LK * DS];
    __shared__ __attribute__((aligned(16))) bf16_t stage[NW * STAGE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16_t* const sq = stage + wave * STAGE;
    bf16_t* const sdo = sq + LQ * DS;
    bf16_t* const sp = sdo + LQ * DS;
    bf16_t* const sds = sp + LQ * PS;
    const int lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3;
    // one workgroup per (key owner, head).  Its work items are the kdiv captions that share these keys -- or, when those
    // captions' query rows are unpadded (and therefore one contiguous run of rows), 16*QT-row chunks of that run: chunks are
    // fuller than captions.  Wavefront w takes items w, w + NW, ...; dK / dV stay in registers across them.
    const int bh = blockIdx.x, bk = bh / p.H, h = bh - bk * p.H;
    if (p.q_rows > 0 && bk == p.B / p.kdiv) {                   // the extra workgroups: gradient rows behind the last item are zero
        const int c = (threadIdx.x & 15) * 4;                   // 16 threads x 4 columns per row
        for (int r = p.q_start[p.B - 1] + p.q_count[p.B - 1] + (int)(threadIdx.x >> 4); r < p.q_rows; r += 4 * NW) {
            const size_t oq = (size_t)r * p.lddq + h * 64 + c, ok = (size_t)r * p.lddk + h * 64 + c;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p.dq_bf16) static_cast<bf16_t*>(p.dq)[oq + u] = 0; else static_cast<float*>(p.dq)[oq + u] = 0.f;
                if (p.k_ragged) {
                    if (p.dkv_bf16) { static_cast<bf16_t*>(p.dk)[ok + u] = 0; static_cast<bf16_t*>(p.dv)[ok + u] = 0; }
                    else { static_cast<float*>(p.dk)[ok + u] = 0.f; static_cast<float*>(p.dv)[ok + u] = 0.f; }
                }
            }
        }
        return;
    }
    const bool kr = p.q_start && p.k_ragged;                   // self-attention over unpadded rows (kdiv == 1 then)
    const bool run = p.q_start && !p.k_ragged;                 // cross-attention over unpadded rows: walk the key owner's run in chunks
    const int b_first = bk * p.kdiv;
    int run0 = 0, run_rows = 0, n_items = p.kdiv;
    if (run) {
        run0 = p.q_start[b_first];
        run_rows = p.q_start[b_first + p.kdiv - 1] + p.q_count[b_first + p.kdiv - 1] - run0;
        n_items = (run_rows + LQ - 1) / LQ;
    }
    auto item_rows = [&](int c, size_t& row0) -> int {         // first query row and row count of item c
        if (run) { row0 = (size_t)run0 + (size_t)c * LQ; return min(LQ, run_rows - c * LQ); }
        if (p.q_start) { row0 = (size_t)p.q_start[b_first + c]; return p.q_count[b_first + c]; }
        row0 = (size_t)(b_first + c) * p.Lq;
        return p.Lq;
    };
    size_t qrow0 = 0;
    int Lq = wave < n_items ? item_rows(wave, qrow0) : 0;
    const int Lk = kr ? Lq : p.Lk;
    const size_t krow0 = kr ? qrow0 : (size_t)bk * p.Lk;
    const uint64_t dseed = p.drop_seed + ((p.drop_thresh && p.drop_step) ? *p.drop_step : 0ull);

    RowStage<TIN, LQ> rq;
    RowStage<float, LQ> rdo;
    {   // keys, values and this wavefront's first queries / output gradients: all requested before anything is converted
        // (every wavefront of the workgroup stages the same K / V tile: identical bytes, L2 hits, no hand-off to wait for)
        RowStage<TIN, LK> rk, rv;
        rk.load(static_cast<const TIN*>(p.k) + krow0 * p.ldk + h * 64, p.ldk, Lk, lane);
        rv.load(static_cast<const TIN*>(p.v) + krow0 * p.ldv + h * 64, p.ldv, Lk, lane);
        if (wave < n_items) {
            rq.load(static_cast<const TIN*>(p.q) + qrow0 * p.ldq + h * 64, p.ldq, Lq, lane);
            rdo.load(p.dout + qrow0 * p.ldo + h * 64, p.ldo, Lq, lane);
        }
        rk.store(sk, DS, lane);
        rv.store(sv, DS, lane);
    }
    __syncthreads();

    f32x4 ak[KT][4], av[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { ak[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; av[kt][dt] = ak[kt][dt]; }

    for (int c = wave; c < n_items; c += NW) {
        const int b = b_first + c;                               // (caption index; unused for unpadded rows)
        wave_lds_sync();                                         // the previous item's operands are still being read
        rq.store(sq, DS, lane);
        rdo.store(sdo, DS, lane);
        const int Lq_c = Lq;                                     // this item's rows (the prefetch below moves Lq / qrow0 on)
        const size_t qrow_c = qrow0;
        if (c + NW < n_items) {                                  // the next item's slices travel while this one is multiplied
            Lq = item_rows(c + NW, qrow0);
            rq.load(static_cast<const TIN*>(p.q) + qrow0 * p.ldq + h * 64, p.ldq, Lq, lane);
            rdo.load(p.dout + qrow0 * p.ldo + h * 64, p.ldo, Lq, lane);
        }
        wave_lds_sync();

        // ---- S = Q K^T, dP = dO V^T  (lane holds rows q = qt*16 + 4g + r, column k = kt*16 + l15)
        f32x4 S[QT][KT], dP[QT][KT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            bf16x8 aq[2], ao[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                aq[s] = *reinterpret_cast<const bf16x8*>(&sq[(qt * 16 + l15) * DS + s * 32 + g * 8]);
                ao[s] = *reinterpret_cast<const bf16x8*>(&sdo[(qt * 16 + l15) * DS + s * 32 + g * 8]);
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, d = a;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 bkf = *reinterpret_cast<const bf16x8*>(&sk[(kt * 16 + l15) * DS + s * 32 + g * 8]);
                    const bf16x8 bvf = *reinterpret_cast<const bf16x8*>(&sv[(kt * 16 + l15) * DS + s * 32 + g * 8]);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[s], bkf, a, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao[s], bvf, d, 0, 0, 0);
                }
                S[qt][kt] = a;
                dP[qt][kt] = d;
            }
        }

        // ---- softmax rows and dS; P and dS go to LDS as bf16 [q][k]
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qrow = qt * 16 + 4 * g + r;
                int kl = 0;
                if (qrow < Lq_c) {
                    kl = Lk;
                    if (p.klen) { kl = (p.q_start ? p.klen[qrow_c + qrow] : p.klen[b * p.klen_sb + qrow * p.klen_sq]) + p.klen_bias; kl = max(0, min(kl, Lk)); }
                }
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    if (kt * 16 + l15 < kl) m = fmaxf(m, S[qt][kt][r] * 0.125f);
                m = row16_max(m);
                float e[KT], sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    e[kt] = (kt * 16 + l15 < kl) ? expf(S[qt][kt][r] * 0.125f - m) : 0.f;
                    sum += e[kt];
                }
                sum = row16_sum(sum);
                const float inv = sum > 0.f ? 1.f / sum : 0.f;
                // forward: O = (P o M) V with M = keep / (1 - p); so dV uses P o M, and dP = (dO V^T) o M
                float dot = 0.f, mk[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    mk[kt] = 1.f;
                    if (p.drop_thresh) {
                        const uint64_t ei = (p.q_start ? (uint64_t)(qrow_c + qrow) * p.H + h : (uint64_t)(b * p.H + h) * p.Lq + qrow) * Lk + kt * 16 + l15;
                        mk[kt] = drop_hash(dseed, ei) >= p.drop_thresh ? p.drop_scale : 0.f;
                    }
                    e[kt] *= inv;
                    dot += e[kt] * dP[qt][kt][r] * mk[kt];
                }
                dot = row16_sum(dot);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    sp[qrow * PS + kt * 16 + l15] = f32_to_bf16(e[kt] * mk[kt]);
                    sds[qrow * PS + kt * 16 + l15] = f32_to_bf16(e[kt] * (dP[qt][kt][r] * mk[kt] - dot) * 0.125f);
                }
            }
        wave_lds_sync();

        // ---- dQ = dS K
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 acc[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KT / 2; ++s) {
                const bf16x8 a = frag_row_trorder(sds, PS, qt * 16 + l15, s * 32, g);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, frag_tr(sk, DS, s * 32, dt * 16, g, tq, tp), acc[dt], 0, 0, 0);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qrow = qt * 16 + 4 * g + r;
                    if (qrow < Lq_c) {
                        const size_t o = (qrow_c + qrow) * p.lddq + h * 64 + dt * 16 + l15;
                        if (p.dq_bf16) static_cast<bf16_t*>(p.dq)[o] = f32_to_bf16(acc[dt][r]);
                        else static_cast<float*>(p.dq)[o] = acc[dt][r];
                    }
                }
        }
        // ---- dK += dS^T Q, dV += P^T dO
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < QT / 2; ++s) {
                const bf16x8 a_ds = frag_tr(sds, PS, s * 32, kt * 16, g, tq, tp);
                const bf16x8 a_p = frag_tr(sp, PS, s * 32, kt * 16, g, tq, tp);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    ak[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_ds, frag_tr(sq, DS, s * 32, dt * 16, g, tq, tp), ak[kt][dt], 0, 0, 0);
                    av[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_p, frag_tr(sdo, DS, s * 32, dt * 16, g, tq, tp), av[kt][dt], 0, 0, 0);
                }
            }
    }
    if constexpr (NW > 1) {                                      // dK / dV of the other wavefronts join wavefront 0's through LDS
        static_assert(NW * STAGE * 2 >= KT * 4 * 2 * 256 * 4, "staging tiles too small for the reduction");
        float* red = reinterpret_cast<float*>(stage);            // the staging tiles are no longer needed
        for (int w = 1; w < NW; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        *reinterpret_cast<f32x4*>(&red[((kt * 4 + dt) * 2 + 0) * 256 + lane * 4]) = ak[kt][dt];
                        *reinterpret_cast<f32x4*>(&red[((kt * 4 + dt) * 2 + 1) * 256 + lane * 4]) = av[kt][dt];
                    }
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(&red[((kt * 4 + dt) * 2 + 0) * 256 + lane * 4]);
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&red[((kt * 4 + dt) * 2 + 1) * 256 + lane * 4]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ak[kt][dt][r] += a[r]; av[kt][dt][r] += v[r]; }
                    }
            }
        }
        if (wave != 0) return;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int krow = kt * 16 + 4 * g + r;
                if (krow >= Lk) continue;
                const size_t o = (krow0 + krow) * p.lddk + h * 64 + dt * 16 + l15;
                if (p.dkv_bf16) {
                    static_cast<bf16_t*>(p.dk)[o] = f32_to_bf16(ak[kt][dt][r]);
                    static_cast<bf16_t*>(p.dv)[o] = f32_to_bf16(av[kt][dt][r]);
                } else {
                    static_cast<float*>(p.dk)[o] = ak[kt][dt][r];
                    static_cast<float*>(p.dv)[o] = av[kt][dt][r];
                }
            }
}

template <typename TIN>
static int launch_t(const AttnBwdMfmaParams& p, hipStream_t st) {
    const dim3 grid((p.B / p.kdiv + ((p.q_start && p.q_rows > 0) ? 1 : 0)) * p.H);
    // several captions per key owner: two wavefronts share them
    if (p.Lq <= 32 && p.Lk <= 32) hipLaunchKernelGGL((attn_bwd_mfma_kernel<TIN, 2, 2, 1>), grid, dim3(64), 0, st, p);
    else if ((p.Lq <= 32 && p.kdiv > 1) || (p.q_start && !p.k_ragged)) hipLaunchKernelGGL((attn_bwd_mfma_kernel<TIN, 2, 4, 2>), grid, dim3(128), 0, st, p);
    else if (p.Lq <= 32) hipLaunchKernelGGL((attn_bwd_mfma_kernel<TIN, 2, 4, 1>), grid, dim3(64), 0, st, p);
    else hipLaunchKernelGGL((attn_bwd_mfma_kernel<TIN, 4, 4, 1>), grid, dim3(64), 0, st, p);
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi

extern "C" int bofi_attention_bwd_mfma(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int in_dtype,
                                       const float* dout, int ldo, void* dq, int lddq, void* dk, void* dv, int lddk, int dq_dtype, int dkv_dtype,
                                       int B, int H, int Lq, int Lk, int kdiv, const int* klen, int klen_sb, int klen_sq, int klen_bias,
                                       float drop_p, uint64_t drop_seed, const uint64_t* drop_step, const int* q_start, const int* q_count,
                                       int k_ragged, int q_rows, void* stream) {
    using namespace bofi;
    if (!q || !k || !v || !dout || !dq || !dk || !dv || B < 0 || H <= 0 || Lq <= 0 || Lk <= 0 || (Lq > 64 && !(q_start && !k_ragged)) || Lk > 64 || kdiv <= 0 || B % kdiv || !(drop_p >= 0.f && drop_p < 1.f)) return BOFI_ERR_ARG;
    if (in_dtype != BOFI_DT_F32 && in_dtype != BOFI_DT_BF16) return BOFI_ERR_ARG;
    if ((dq_dtype != BOFI_DT_F32 && dq_dtype != BOFI_DT_BF16) || (dkv_dtype != BOFI_DT_F32 && dkv_dtype != BOFI_DT_BF16)) return BOFI_ERR_ARG;
    const int el = in_dtype == BOFI_DT_F32 ? 4 : 2;
    if ((ldq * el) % 16 || (ldk * el) % 16 || (ldv * el) % 16 || ldo % 4 || ((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)v % 16) ||
        ((uintptr_t)dout % 16))
        return BOFI_ERR_ARG;
    if (B == 0) return BOFI_OK;
    AttnBwdMfmaParams p{q, ldq, k, ldk, v, ldv, dout, ldo, dq, lddq, dk, dv, lddk, dq_dtype == BOFI_DT_BF16, dkv_dtype == BOFI_DT_BF16,
                        B, H, Lq, Lk, kdiv, klen, klen_sb, klen_sq, klen_bias,
                        0u, 1.f, drop_seed, drop_step, q_start, q_count, k_ragged, q_start ? q_rows : 0};
    if ((q_start != nullptr) != (q_count != nullptr) || (k_ragged && kdiv != 1)) return BOFI_ERR_ARG;
    if (drop_p > 0.f) { p.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); p.drop_scale = 1.0f / (1.0f - drop_p); }
    return in_dtype == BOFI_DT_F32 ? launch_t<float>(p, (hipStream_t)stream) : launch_t<bf16_t>(p, (hipStream_t)stream);
}
