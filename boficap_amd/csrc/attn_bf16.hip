// bf16 attention core for the short sequences of the bound+fill path -- the throughput kernel
// (attn.hip is the generic one: f32 engine, > 128 keys).  Same contract as attn.hip:
//   softmax(q k^T / 8, key-prefix mask per query row) v,  d_k = 64,  NaN for a fully masked row
// (reference attention() captioning/models/TransformerModel.py:1421-1432).
//
// One wavefront per (image, head, <=48-query block); everything between the staging loads and the
// output store stays in registers:
//   * S^T = K Q^T on MFMA with the KEY on the accumulator rows ("swapped QK^T"): a lane then holds,
//     for its query column, 4 keys per 16-key tile, so the row softmax is an in-register reduction
//     plus two cross-lane steps (xor 16, 32) -- no score matrix in LDS.
//   * The probabilities feed the second MFMA straight from those registers: O^T = V^T P^T, where the
//     k index of the instruction is mapped to keys exactly as the accumulator holds them
//     (slot j of lane group g = key 16*(2s + j/4) + 4g + j%4), and the V^T fragments come from the
//     row-major V tile with the transposing LDS read ds_read_b64_tr_b16 (no transposed staging).
//   * Output: a lane holds 4 consecutive d of one query row -> one 8-byte store.
#include "bofi_common.h"
#include "bofi_kernels.h"

namespace bofi {

struct AttnParamsB {
    const bf16_t* q; int ldq;
    const bf16_t* k; int ldk;
    const bf16_t* v; int ldv;
    bf16_t* out; int ldo;
    int B, H, Lq, Lk;
    const int* klen; int klen_sb, klen_sq, klen_bias, klen_shared_last;
    const int* skip_if_ge; int skip_threshold;
    int kdiv;
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed; const uint64_t* drop_step;
    const int* q_start; const int* q_count; int k_ragged; int q_rows;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

// LDS row stride of the V tile in elements.  It is read with the transposing read, which is banked per 32-lane half = 8 rows x 32 B
// (MI355X_MICROARCH.md, LDS): the 8 segments tile the 256-byte bank row only if the stride is an odd multiple of 32 B -> 64 + 16 pad
// = 160 B (with 144 B SQ_LDS_BANK_CONFLICT was 25 % of this kernel's LDS cycles).
constexpr int VROW = 80;

template <int NQT, int NKT, bool RAGGED = false>       // 16-row query tiles per block, 16-key tiles (even); RAGGED: per-item row ranges
__global__ __launch_bounds__(64) void attn_bf16_kernel(AttnParamsB p) {
    static_assert(NKT % 2 == 0, "keys are consumed 32 at a time");
    __shared__ __attribute__((aligned(16))) bf16_t sv[NKT * 16 * VROW];     // only V goes through LDS (transposing reads)

    if (p.skip_if_ge && *p.skip_if_ge >= p.skip_threshold) return;
    const int lane = threadIdx.x;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.y * (NQT * 16);
    if constexpr (RAGGED) {
        if (b == p.B) {                             // the extra "item": rows behind the last one.  Nobody attends them, but the
            if (blockIdx.y != 0) return;            // row-wise consumers of the output read them: they get zeros
            for (int r = p.q_start[p.B - 1] + p.q_count[p.B - 1] + (lane >> 3); r < p.q_rows; r += 8)
                *reinterpret_cast<u32x4*>(p.out + (size_t)r * p.ldo + h * 64 + (lane & 7) * 8) = u32x4{0u, 0u, 0u, 0u};
            return;
        }
    }
    int qrow0 = b * p.Lq, lq = p.Lq;                // first query row of this item and how many it has
    if constexpr (RAGGED) { qrow0 = p.q_start[b]; lq = p.q_count[b]; }
    const int nq = min(NQT * 16, lq - q0);
    if constexpr (RAGGED) { if (nq <= 0) return; }
    const int bk = b / p.kdiv;                      // captions of one image share its keys (training)
    int krow0 = bk * p.Lk, Lk = p.Lk;
    if constexpr (RAGGED) { if (p.k_ragged) { krow0 = qrow0; Lk = lq; } }

    // ---- operands.  The Q and K fragments of the first product are rows of the head slices as they lie in memory (fragment row =
    // lane & 15, eight consecutive d per lane quarter = one 16-byte load): they go straight into registers.  Only V is staged, for
    // the transposing reads of the second product.  (With Q and K tiles in LDS as well a workgroup held 26 KB and six of them
    // filled a CU; the kernel is a latency chain per wavefront, so the wavefronts in flight are what its throughput is.)
    const bf16_t* qg = p.q + ((size_t)qrow0 + q0) * p.ldq + h * 64;
    const bf16_t* kg = p.k + (size_t)krow0 * p.ldk + h * 64;
    const bf16_t* vg = p.v + (size_t)krow0 * p.ldv + h * 64;
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
    const int l15 = lane & 15, g = lane >> 4;
    const bf16x8 zero8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 bq[NQT][2], ak[NKT][2];
#pragma unroll
    for (int qi = 0; qi < NQT; ++qi) {
        const int r = qi * 16 + l15;
#pragma unroll
        for (int s = 0; s < 2; ++s) bq[qi][s] = r < nq ? *reinterpret_cast<const bf16x8*>(qg + (size_t)r * p.ldq + s * 32 + g * 8) : zero8;
    }
#pragma unroll
    for (int kj = 0; kj < NKT; ++kj) {
        const int r = kj * 16 + l15;
#pragma unroll
        for (int s = 0; s < 2; ++s) ak[kj][s] = r < Lk ? *reinterpret_cast<const bf16x8*>(kg + (size_t)r * p.ldk + s * 32 + g * 8) : zero8;
    }
#pragma unroll
    for (int it = 0; it < NKT * 2; ++it) {             // NKT * 16 rows x 8 sixteen-byte pieces, 64 per pass
        const int c = lane + it * 64;
        const int r = c >> 3, ch = c & 7;
        *reinterpret_cast<u32x4*>(&sv[r * VROW + ch * 8]) = r < Lk ? *reinterpret_cast<const u32x4*>(vg + (size_t)r * p.ldv + ch * 8) : zero4;
    }

    // ---- S^T[key][q] = K Q^T: A = K rows (key on the MFMA row), B = Q rows (query on the column)
    f32x4 st[NKT][NQT];
#pragma unroll
    for (int kj = 0; kj < NKT; ++kj) {
        const bf16x8 a0 = ak[kj][0], a1 = ak[kj][1];
#pragma unroll
        for (int qi = 0; qi < NQT; ++qi) {
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bq[qi][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq[qi][1], c, 0, 0, 0);
            st[kj][qi] = c;
        }
    }

    // ---- softmax over keys per query column; lane holds keys kj*16 + g*4 + r
    const uint64_t dseed = p.drop_seed + ((p.drop_thresh && p.drop_step) ? *p.drop_step : 0ull);
    bf16x8 bp[NQT][NKT / 2];
#pragma unroll
    for (int qi = 0; qi < NQT; ++qi) {
        const int qrow = qi * 16 + l15;
        int kl = Lk;
        if (p.klen && qrow < nq) {
            // quirk Q1 per group of klen_shared_last images: every image uses the key count of its group's LAST image
            const int bi = p.klen_shared_last ? min(p.B, (b / p.klen_shared_last + 1) * p.klen_shared_last) - 1 : b;
            if constexpr (RAGGED) kl = p.klen[qrow0 + q0 + qrow] + p.klen_bias;
            else kl = p.klen[bi * p.klen_sb + (q0 + qrow) * p.klen_sq] + p.klen_bias;
            kl = max(0, min(kl, Lk));
        }
        float m = -INFINITY;
#pragma unroll
        for (int kj = 0; kj < NKT; ++kj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sv_ = (kj * 16 + g * 4 + r < kl) ? st[kj][qi][r] * 0.125f : -INFINITY;
                st[kj][qi][r] = sv_;
                m = fmaxf(m, sv_);
            }
        m = xor32_max(xor16_max(m));
        float sum = 0.f;
#pragma unroll
        for (int kj = 0; kj < NKT; ++kj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (kj * 16 + g * 4 + r < kl) ? __expf(st[kj][qi][r] - m) : 0.f;      // v_exp_f32 (the probabilities end as bf16)
                st[kj][qi][r] = e;
                sum += e;
            }
        sum = xor32_sum(xor16_sum(sum));
        // an empty row gives 0/0 = NaN for every key, as softmax over all -inf does in the reference
        const float inv_sum = 1.0f / sum;                 // (one division per row: 0 * (1/0) is NaN as 0/0 is)
#pragma unroll
        for (int s = 0; s < NKT / 2; ++s) {
            bf16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kj = 2 * s + (j >> 2), r = j & 3;
                float pv = st[kj][qi][r] * inv_sum;
                if (kl == 0 && kj * 16 + g * 4 + r >= Lk) pv = 0.f;      // padded keys of an empty row: V is 0 there
                if (p.drop_thresh) {                                     // training: dropout(p_attn)
                    // element id of the mask: (item, head, query, key), or (global query row, head, key) for unpadded rows --
                    // the backward walks those in chunks that ignore item boundaries
                    const uint64_t e = (RAGGED ? (uint64_t)(qrow0 + q0 + qrow) * p.H + h : (uint64_t)(b * p.H + h) * p.Lq + q0 + qrow) * Lk +
                                       kj * 16 + g * 4 + r;
                    pv = drop_hash(dseed, e) >= p.drop_thresh ? pv * p.drop_scale : 0.f;
                }
                f[j] = (short)f32_to_bf16(pv);
            }
            bp[qi][s] = f;
        }
    }

    // ---- O^T[d][q] = V^T P^T: A = V^T via the transposing read of the row-major V tile
    __syncthreads();                                  // one wavefront: orders the V tile's writes before the transposing reads
    f32x4 ot[4][NQT];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qi = 0; qi < NQT; ++qi) ot[dt][qi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = l15 >> 2, tp = l15 & 3;            // lane 4q'+p' of its 16-lane group addresses row q', columns 4p'..4p'+3
#pragma unroll
    for (int s = 0; s < NKT / 2; ++s) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16_t* a0p = &sv[(32 * s + 4 * g + tq) * VROW + dt * 16 + 4 * tp];
            const bf16_t* a1p = a0p + 16 * VROW;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0p);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1p);
            bf16x8 av;
            av[0] = lo[0]; av[1] = lo[1]; av[2] = lo[2]; av[3] = lo[3];
            av[4] = hi[0]; av[5] = hi[1]; av[6] = hi[2]; av[7] = hi[3];
#pragma unroll
            for (int qi = 0; qi < NQT; ++qi) ot[dt][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bp[qi][s], ot[dt][qi], 0, 0, 0);
        }
    }

    // ---- store: lane holds O[q = qi*16 + l15][d = dt*16 + g*4 + 0..3]
    bf16_t* og = p.out + ((size_t)qrow0 + q0) * p.ldo + h * 64;
#pragma unroll
    for (int qi = 0; qi < NQT; ++qi) {
        const int qrow = qi * 16 + l15;
        if (qrow >= nq) continue;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 o;
            o.x = pack_bf16(ot[dt][qi][0], ot[dt][qi][1]);
            o.y = pack_bf16(ot[dt][qi][2], ot[dt][qi][3]);
            *reinterpret_cast<uint2*>(og + (size_t)qrow * p.ldo + dt * 16 + g * 4) = o;
        }
    }
}

template <int NQT, int NKT>
static void launch_ab(const AttnParamsB& p, hipStream_t st) {
    const dim3 grid((p.B + ((p.q_start && p.q_rows > 0) ? 1 : 0)) * p.H, (p.Lq + NQT * 16 - 1) / (NQT * 16));
    if (p.q_start) hipLaunchKernelGGL((attn_bf16_kernel<NQT, NKT, true>), grid, dim3(64), 0, st, p);
    else hipLaunchKernelGGL((attn_bf16_kernel<NQT, NKT, false>), grid, dim3(64), 0, st, p);
}

// returns -1 when the call is not eligible (then attn.hip handles it)
int launch_attention_bf16(const AttnArgs& a, hipStream_t st) {
    if (a.dtype != BOFI_DT_BF16 || a.Lk > 128 || a.ldo % 4 || ((uintptr_t)a.out % 8)) return -1;
    AttnParamsB p;
    p.q = (const bf16_t*)a.q; p.ldq = a.ldq; p.k = (const bf16_t*)a.k; p.ldk = a.ldk; p.v = (const bf16_t*)a.v; p.ldv = a.ldv;
    p.out = (bf16_t*)a.out; p.ldo = a.ldo; p.B = a.B; p.H = a.H; p.Lq = a.Lq; p.Lk = a.Lk;
    p.klen = a.klen; p.klen_sb = a.klen_sb; p.klen_sq = a.klen_sq; p.klen_bias = a.klen_bias;
    p.klen_shared_last = a.klen_shared_last; p.skip_if_ge = a.skip_if_ge; p.skip_threshold = a.skip_threshold;
    p.kdiv = a.kdiv > 0 ? a.kdiv : 1;
    p.drop_thresh = a.drop_thresh; p.drop_scale = a.drop_scale; p.drop_seed = a.drop_seed; p.drop_step = a.drop_step;
    if ((a.q_start != nullptr) != (a.q_count != nullptr)) return BOFI_ERR_ARG;
    p.q_start = a.q_start; p.q_count = a.q_count; p.k_ragged = a.k_ragged; p.q_rows = a.q_rows;
    if (a.q_rows > 0 && (a.ldo % 8 || ((uintptr_t)a.out % 16))) return BOFI_ERR_ARG;
    // 65 .. 128 keys (round 5: real bottom-up features have up to 100 regions, captioning/utils/opts.py:84): six or eight key tiles, at most two query
    // tiles per wavefront (the scores alone are 8 x 2 accumulator tiles); longer query blocks go over blockIdx.y
    const int nkt = a.Lk <= 32 ? 2 : a.Lk <= 64 ? 4 : a.Lk <= 96 ? 6 : 8;
    const int nqt = a.Lq <= 16 ? 1 : ((a.Lq <= 32 || nkt > 4) ? 2 : 3);
    switch (nqt * 10 + nkt) {
        case 16: launch_ab<1, 6>(p, st); break;
        case 18: launch_ab<1, 8>(p, st); break;
        case 26: launch_ab<2, 6>(p, st); break;
        case 28: launch_ab<2, 8>(p, st); break;
        case 12: launch_ab<1, 2>(p, st); break;
        case 14: launch_ab<1, 4>(p, st); break;
        case 22: launch_ab<2, 2>(p, st); break;
        case 24: launch_ab<2, 4>(p, st); break;
        case 32: launch_ab<3, 2>(p, st); break;
        default: launch_ab<3, 4>(p, st); break;
    }
    BOFI_CHECK_LAUNCH();
    return BOFI_OK;
}

}  // namespace bofi
