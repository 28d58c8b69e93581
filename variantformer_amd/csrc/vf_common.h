// Shared device/host helpers for libvf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vf_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define VF_WAVE 64

void vf_set_error(const char* fmt, ...);

// Per-device launch state (dynamic-LDS attributes) is indexed by the current HIP device; -1 = out of range / error,
// in which case the caller simply redoes the (idempotent) setup.
#define VF_MAX_DEVICES 64
static inline int vf_current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (d >= 0 && d < VF_MAX_DEVICES) ? d : -1;
}

#define VF_REQUIRE(cond, ...)                        \
    do {                                             \
        if (!(cond)) {                               \
            vf_set_error(__VA_ARGS__);               \
            return VF_ERR_INVALID_ARG;               \
        }                                            \
    } while (0)

#define VF_CHECK_LAUNCH(name)                                                          \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            vf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
            return VF_ERR_LAUNCH;                                                      \
        }                                                                              \
    } while (0)

// fp32 -> bf16 bits, round to nearest even (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950,
// NaN stays NaN).
__device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 b = (__bf16)x;
    return *reinterpret_cast<unsigned short*>(&b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16 pair (lo in bits 0..15): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned int, b);
}
__device__ __forceinline__ float bf2f(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// two fp32 -> packed IEEE half pair, round to nearest even
__device__ __forceinline__ unsigned int pack2h(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    return __builtin_bit_cast(unsigned int, h);
}
__device__ __forceinline__ float h2f(unsigned short b) {
    return (float)__builtin_bit_cast(_Float16, b);
}

// The 16-bit operand type of the MFMA kernels: VF_BF16 (the reference's bf16-mixed path) or VF_F16 (its 16-mixed /
// fp16 flash-attn path, seq2gene/modules/layers.py:102-125, utils/functions.py:12-32); fp32 accumulation in both,
// same MFMA shape and rate.  ONE / NEG_BIG are the bit patterns of 1.0 and -32768.0 (attention key masking).
template <int DT> struct Op16;
template <> struct Op16<VF_BF16> {
    using frag = bf16x8_t;
    static __device__ __forceinline__ f32x4_t mfma(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned int pack2(float lo, float hi) { return pack2bf(lo, hi); }
    static __device__ __forceinline__ float to_f32(unsigned short b) { return bf2f(b); }
    static constexpr unsigned int ONE = 0x3F80u, NEG_BIG = 0xC700u;
};
template <> struct Op16<VF_F16> {
    using frag = f16x8_t;
    static __device__ __forceinline__ f32x4_t mfma(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned int pack2(float lo, float hi) { return pack2h(lo, hi); }
    static __device__ __forceinline__ float to_f32(unsigned short b) { return h2f(b); }
    static constexpr unsigned int ONE = 0x3C00u, NEG_BIG = 0xF800u;
};

// exact (erf) GELU, as nn.GELU() / F.gelu default
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
