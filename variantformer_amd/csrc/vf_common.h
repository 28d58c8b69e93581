// Shared device/host helpers for libvf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vf_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define VF_WAVE 64

void vf_set_error(const char* fmt, ...);
// vf_last_kernel(): the launchers name the kernel they dispatched to (0 = GEMM, 1 = attention; static strings)
void vf_note_kernel(int which, const char* name);

// Per-device launch state (dynamic-LDS attributes) is indexed by the current HIP device; -1 = out of range / error,
// in which case the caller simply redoes the (idempotent) setup.
#define VF_MAX_DEVICES 64
static inline int vf_current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (d >= 0 && d < VF_MAX_DEVICES) ? d : -1;
}

// A/B switches between kernel forms exist in the tuning library only (libvf_hip_tuning.so, -DVF_TUNING: scripts/ and the
// tuning tests); the product library always takes the measured default, so no environment variable reaches a kernel choice.
#ifdef VF_TUNING
#include <stdlib.h>
static inline int vf_tuning_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
#define vf_tuning_env(name, dflt) (dflt)
#endif

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
    static __device__ __forceinline__ f32x16_t mfma32(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
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
    static __device__ __forceinline__ f32x16_t mfma32(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned int pack2(float lo, float hi) { return pack2h(lo, hi); }
    static __device__ __forceinline__ float to_f32(unsigned short b) { return h2f(b); }
    static constexpr unsigned int ONE = 0x3C00u, NEG_BIG = 0xF800u;
};

// erf GELU (nn.GELU() / F.gelu default): x * Phi(x) with Phi from erfc(|x| / sqrt 2) = poly(t) * exp(-x^2 / 2),
// t = 1 / (1 + p |x| / sqrt 2) (Abramowitz & Stegun 7.1.26, |error of erf| <= 1.5e-7).  Phi(x) = erfc/2 for x < 0 and
// 1 - erfc/2 otherwise, so the negative tail has no 1 + erf cancellation: against float64 erf, over [-12, 12], the fp32
// result differs by at most 4.2e-7 absolute -- the same as 0.5 x (1 + erff(x / sqrt 2)) evaluated in fp32 (4.5e-7) --
// at a third of the instructions of the libdevice erff (46 VALU per 4 values, packed fp32 math; the GeGLU epilogue of
// the 256 x 256 GEMM tile was 2600 VALU instructions per wave, longer than the MFMA loop of a K = 512 tile).
__device__ __forceinline__ f32x4_t gelu_erf4(f32x4_t x) {
    f32x4_t ax, t, e;
#pragma unroll
    for (int i = 0; i < 4; ++i) ax[i] = __builtin_fabsf(x[i]);
    const f32x4_t d = ax * 0.23164190f + 1.0f;                 // p / sqrt 2, p = 0.3275911
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_rcpf(d[i]);
    f32x4_t p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const f32x4_t w = x * x * -0.72134752044448170368f;        // -x^2 / 2 in base 2
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(w[i]);
    const f32x4_t hq = p * t * e * 0.5f;                       // erfc(|x| / sqrt 2) / 2
    f32x4_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = x[i] * (x[i] >= 0.f ? 1.0f - hq[i] : hq[i]);
    return r;
}
__device__ __forceinline__ float gelu_erf(float x) {
    return gelu_erf4((f32x4_t){x, x, x, x})[0];
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
