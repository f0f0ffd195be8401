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
// round-to-nearest-even, NaN stays NaN (plain integer rounding would turn some NaNs into inf/0)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// counter-based dropout mask of the training path: element i of stream `seed` is kept iff drop_hash(seed, i) >= p * 2^32
// (splitmix64 finaliser).  Forward and backward regenerate the same mask from (seed, i); nothing is stored.
__device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}

}  // namespace bofi

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
