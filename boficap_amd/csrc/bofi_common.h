// Shared device helpers for the BoFiCap gfx950 kernels (wave64, MFMA fragment types, bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BOFI_OK 0
#define BOFI_ERR_ARG 1
#define BOFI_ERR_HIP 2
#define BOFI_ERR_STATE 3

#define BOFI_DT_F32 0
#define BOFI_DT_BF16 1

namespace bofi {

constexpr int WAVE = 64;

typedef uint16_t bf16_t;                                       // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;       // MFMA A/B fragment, 16x16x32 bf16
typedef __attribute__((ext_vector_type(4))) float f32x4;        // MFMA C/D fragment, 16x16 tiles
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN stays NaN with the quiet bit set: gfx950's v_cvt_pk_bf16_f32.  It agrees with the integer form
//   u > 0x7f800000 (NaN) ? (u >> 16) | 0x40 : (u + 0x7fff + ((u >> 16) & 1)) >> 16
// on every one of the 2^32 float32 bit patterns (tools/exp/cvt_bf16_check.hip, run on the MI355X) at a seventh of the vector
// instructions -- which is what the GEMM epilogues' time goes to at K = 512.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two at once: lo in bits 0..15, hi in bits 16..31 (one instruction)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    const f2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
}

template <typename T> struct ElemOps;
template <> struct ElemOps<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
    static __device__ __forceinline__ float to_f32(float v) { return v; }
};
template <> struct ElemOps<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
    static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
};

// Cross-lane moves inside a row of 16 lanes as DPP modifiers (folded into the VALU op that consumes them) instead of
// ds_bpermute round trips through the LDS pipeline (which is what __shfl_xor compiles to): quad_perm [1,0,3,2] and
// [2,3,0,1] are lane ^ 1 and lane ^ 2, row_half_mirror pairs the two quads of a half row, row_mirror the two halves.
// All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

// sums / maxima over groups of 4, 8 or 16 consecutive lanes; every lane of the group ends with the result
__device__ __forceinline__ float quad_sum(float v) { v += dpp_f32<DPP_XOR1>(v); v += dpp_f32<DPP_XOR2>(v); return v; }
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, dpp_f32<DPP_XOR1>(v)); return fmaxf(v, dpp_f32<DPP_XOR2>(v)); }
__device__ __forceinline__ float oct_sum(float v) { v = quad_sum(v); return v + dpp_f32<DPP_HALF_MIRROR>(v); }
__device__ __forceinline__ float row16_sum(float v) { v = oct_sum(v); return v + dpp_f32<DPP_MIRROR>(v); }
__device__ __forceinline__ float row16_max(float v) {
    v = quad_max(v);
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    return fmaxf(v, dpp_f32<DPP_MIRROR>(v));
}
// Across rows: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16- (32-) lane rows of one register with the even
// rows of another.  With both operands = v the two results hold, in every lane, the lane's own row pair: (r0,r0,r2,r2) and
// (r1,r1,r3,r3) for the 16-lane form, (lo,lo) and (hi,hi) for the 32-lane form -- so a commutative combine of the two results is
// the lane ^ 16 (lane ^ 32) step of a butterfly, in one VALU instruction instead of a ds_bpermute (probe: tools/exp/permlane_probe.hip).
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// 64-bit counter hash (splitmix64 finaliser), upper half: the uniform numbers of the sampling kernels
__device__ __forceinline__ uint32_t hash64_hi(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

// counter-based dropout mask of the training path: element i of stream `seed` is kept iff drop_hash(seed, i) >= p * 2^32.
// Forward and backward regenerate the same mask from (seed, i); nothing is stored.  The stream key is a full 64-bit mix of
// the seed -- wave-uniform, so it is scalar-ALU work done once per kernel -- and the per-element part is a 32-bit
// multiply-xorshift hash (two v_mul_lo_u32): 64-bit multiplies are 4 quarter-rate VALU multiplies each on CDNA and a
// splitmix64 per element was a third of the attention-backward time.
__device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t i) {
    uint64_t k = (seed + 0x632BE59BD9B4E019ull) * 0x9E3779B97F4A7C15ull;
    k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
    k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
    k ^= k >> 31;
    uint32_t x = (uint32_t)i + __builtin_rotateleft32((uint32_t)(i >> 32), 19);
    x ^= (uint32_t)k;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x += (uint32_t)(k >> 32); x *= 0x846CA68Bu;
    return x ^ (x >> 16);
}

}  // namespace bofi

// host-side: an integer developer knob from the environment, read again after bofi_reload_env (no knob is latched for the life of the process: the
// value a launch sees is the value of the last reload -- VERDICT r4 item 9b).  Every use site has its own cache (statics of its own lambda).
#include <cstdlib>
namespace bofi { extern int g_env_generation; }      // bumped by bofi_reload_env (gemm_glds.hip)
#define BOFI_ENV_INT(name, dflt)                                                                                       \
    ([]() -> int {                                                                                                     \
        static int gen__ = -1, v__ = (dflt);                                                                           \
        if (gen__ != bofi::g_env_generation) { const char* e__ = getenv(name); v__ = e__ ? atoi(e__) : (dflt); gen__ = bofi::g_env_generation; } \
        return v__;                                                                                                    \
    }())

// host-side launch check: kernels are enqueued on a stream, so this only catches launch errors
#define BOFI_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return BOFI_ERR_HIP;           \
    } while (0)
#define BOFI_HIP(call)                                        \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return BOFI_ERR_HIP;           \
    } while (0)
