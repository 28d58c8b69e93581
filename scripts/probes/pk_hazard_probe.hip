// Round 6: isolate the cross-wave hazard probe 4 found (a wave's v_pk_fma_f32 loses the product of its low half in lanes 48..63
// when an attention kernel's waves share the SIMD).  Victim kernels execute ONE packed-fp32 (or, as control, plain) instruction
// form in a loop and compare every result bit for bit with the two scalar operations it stands for; co-runner kernels on a
// second stream execute one instruction class in a loop.  Both grids are 2 x 256 CUs x 256 threads so every SIMD holds waves of
// both.  Output: wrong results per (victim form, co-runner class), with the (quarter-wave, half) distribution of the errors.
//   hipcc --offload-arch=gfx950 -O2 -o pk_hazard_probe pk_hazard_probe.hip && ./pk_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float sfma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float smul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float sadd(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// err[0..3][0..1] = wrong results by quarter-wave and half; err[8] = of those, "product lost" (result == addend)
template <int FORM> __global__ __launch_bounds__(256) void victim(unsigned long long* err, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x2 a = {seed + 0.001f * threadIdx.x, seed * 0.5f - 0.002f * threadIdx.x};
    f32x2 b = {1.25f + 0.01f * lane, 0.75f - 0.003f * lane};
    f32x2 c = {0.5f * blockIdx.x * 1e-3f + 0.1f, -0.25f + 0.004f * lane};
    unsigned bad0 = 0, bad1 = 0, lost = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 d, e;
        if (FORM == 0) {          // the LayerNorm-consumer form: both halves times the HIGH half of src1
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e[0] = sfma(a[0], b[1], c[0]); e[1] = sfma(a[1], b[1], c[1]);
        } else if (FORM == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e[0] = sfma(a[0], b[0], c[0]); e[1] = sfma(a[1], b[1], c[1]);
        } else if (FORM == 2) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
            e[0] = smul(a[0], b[0]); e[1] = smul(a[1], b[1]);
        } else if (FORM == 3) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
            e[0] = sadd(a[0], b[0]); e[1] = sadd(a[1], b[1]);
        } else if (FORM == 5) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e[0] = sfma(a[1], b[0], c[0]); e[1] = sfma(a[1], b[1], c[1]);
        } else if (FORM == 6) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e[0] = sfma(a[0], b[0], c[1]); e[1] = sfma(a[1], b[1], c[1]);
        } else if (FORM == 7) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e[0] = sfma(a[0], b[0], c[0]); e[1] = sfma(a[1], b[0], c[1]);
        } else if (FORM == 8) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            e[0] = sfma(a[0], b[0], c[0]); e[1] = sfma(a[0], b[1], c[1]);
        } else if (FORM == 9) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            e[0] = smul(a[0], b[1]); e[1] = smul(a[1], b[1]);
        } else if (FORM == 10) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            e[0] = smul(a[0], b[0]); e[1] = smul(a[1], b[0]);
        } else if (FORM == 11) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            e[0] = sadd(a[0], b[1]); e[1] = sadd(a[1], b[1]);
        } else if (FORM == 12) {  // v_fma_mix_f32 with src1 = the f16 in the HIGH half of a register (what hipcc emits for fp32 += float(half) * s:
                                  // 1 296 of them in the product library, e.g. the fp16 trunk residual of the producer epilogues)
            const unsigned hb = __builtin_bit_cast(unsigned, b[1]) & 0xffff0000u;          // a half in the high 16 bits
            float lo;
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(lo) : "v"(a[0]), "v"(hb), "v"(c[0]));
            float hf;
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hf) : "v"(hb >> 16));
            d[0] = lo; d[1] = lo;
            e[0] = sfma(a[0], hf, c[0]); e[1] = e[0];
        } else if (FORM == 13) {  // the same with the half in the LOW 16 bits
            const unsigned hb = (__builtin_bit_cast(unsigned, b[1]) >> 16);
            float lo;
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(lo) : "v"(a[0]), "v"(hb), "v"(c[0]));
            float hf;
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hf) : "v"(hb));
            d[0] = lo; d[1] = lo;
            e[0] = sfma(a[0], hf, c[0]); e[1] = e[0];
        } else if (FORM == 14) {  // packed f16: low result from the HIGH half of src1
            const unsigned ah = 0x3c003800u ^ (threadIdx.x & 0xff), bh = 0x40004100u + ((it & 7) << 4), ch = 0x34003200u;
            unsigned dd, e0, e1;
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(dd) : "v"(ah), "v"(bh), "v"(ch));
            asm volatile("v_fma_f16 %0, %1, %2, %3" : "=v"(e0) : "v"(ah), "v"(bh >> 16), "v"(ch));
            asm volatile("v_fma_f16 %0, %1, %2, %3" : "=v"(e1) : "v"(ah >> 16), "v"(bh >> 16), "v"(ch >> 16));
            d[0] = __builtin_bit_cast(float, dd & 0xffffu); d[1] = __builtin_bit_cast(float, dd >> 16);
            e[0] = __builtin_bit_cast(float, e0 & 0xffffu); e[1] = __builtin_bit_cast(float, e1 & 0xffffu);
        } else {                  // control: two scalar fused multiply-adds
            asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %5, %3, %6" : "=&v"(d[0]), "=&v"(d[1]) : "v"(a[0]), "v"(b[1]), "v"(c[0]), "v"(a[1]), "v"(c[1]));
            e[0] = sfma(a[0], b[1], c[0]); e[1] = sfma(a[1], b[1], c[1]);
        }
        asm volatile("" : "+v"(e));                                   // keep the scalar pair scalar
        if (__builtin_bit_cast(unsigned, d[0]) != __builtin_bit_cast(unsigned, e[0])) { ++bad0; lost += d[0] == c[0]; }
        if (__builtin_bit_cast(unsigned, d[1]) != __builtin_bit_cast(unsigned, e[1])) { ++bad1; lost += d[1] == c[1]; }
        a[0] += 0.0009765625f; a[1] -= 0.00048828125f;
        if ((it & 255) == 255) { a[0] = seed + 0.001f * threadIdx.x; a[1] = seed * 0.5f - 0.002f * threadIdx.x; }
    }
    if (bad0) atomicAdd(&err[(lane >> 4) * 2 + 0], (unsigned long long)bad0);
    if (bad1) atomicAdd(&err[(lane >> 4) * 2 + 1], (unsigned long long)bad1);
    if (lost) atomicAdd(&err[8], (unsigned long long)lost);
}

template <int CLS> __global__ __launch_bounds__(256) void corunner(float* sink, const float* src, int iters) {
    __shared__ float lds[256 * 8];
    const int t = threadIdx.x, lane = t & 63;
    float x = 0.3f + 0.001f * t, y = 1.0f - 0.002f * t, z = 0.f;
    f32x16 acc32 = {};
    f32x4 acc16 = {};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (t + i)); fb[i] = (__bf16)(0.02f * (t - i)); }
    lds[t] = x;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (CLS == 0) { asm volatile("v_nop\n\tv_nop\n\tv_nop\n\tv_nop"); }
        if (CLS == 1) { asm volatile("v_exp_f32 %0, %1\n\tv_exp_f32 %2, %3" : "=v"(z), "+v"(x), "=v"(y) : "v"(z)); x = z * 0.5f; }
        if (CLS == 2) { asm volatile("v_rcp_f32 %0, %1\n\tv_rsq_f32 %2, %1" : "=&v"(z), "+v"(x), "=&v"(y)); x = z + y; }
        if (CLS == 3) { auto p = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
                        auto q = __builtin_amdgcn_permlane16_swap(p[0], p[1], false, false);
                        x = __builtin_bit_cast(float, q[0]) + 0.5f; y = __builtin_bit_cast(float, q[1]); }
        if (CLS == 4) { x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x111, 0xf, 0xf, true));     // row_shr:1
                        y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xf, 0xf, true)); }   // row_bcast15
        if (CLS == 5) { unsigned p; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y)); x = __builtin_bit_cast(float, p << 16) + 0.1f; }
        if (CLS == 6) { acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc32, 0, 0, 0); }
        if (CLS == 7) { acc16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc16, 0, 0, 0); }
        if (CLS == 8) { z = lds[(t * 5 + it) & 2047]; lds[(t * 3 + it) & 2047] = z + x; x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 17) * 4, __builtin_bit_cast(int, y))); }
        if (CLS == 9) { f32x2 p = {x, y}, q = {y, z}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q)); x = p[0]; y = p[1] + 0.5f; }
        if (CLS == 10) { asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %0, %2, %1" : "+v"(x), "+v"(y) : "v"(z)); }
        if (CLS == 11) { z += src[(size_t)((blockIdx.x * 256 + t) * 16 + (it & 15) * 262144) & 0xffffff]; }
        if (CLS == 12) { acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc32, 0, 0, 0);                      // an attention-like mix
                         asm volatile("v_exp_f32 %0, %1" : "=v"(z) : "v"(x)); x = z * 0.25f;
                         { auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
                           x = __builtin_bit_cast(float, q[0]); y = __builtin_bit_cast(float, q[1]); }
                         unsigned p; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y)); fa[0] = __builtin_bit_cast(__bf16, (unsigned short)p); }
        if (CLS == 13) { int v = __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), it & 63); y += __builtin_bit_cast(float, v) * 1e-3f;
                         asm volatile("s_nop 0"); }
    }
    float r = x + y + z;
    for (int i = 0; i < 16; ++i) r += acc32[i];
    for (int i = 0; i < 4; ++i) r += acc16[i];
    if (r == 12345.678f) sink[t] = r + lds[t];
}

typedef void (*vk_t)(unsigned long long*, int, float);
typedef void (*ck_t)(float*, const float*, int);

int main(int argc, char** argv) {
    const int viters = argc > 1 ? atoi(argv[1]) : 400000, citers = argc > 2 ? atoi(argv[2]) : 400000;
    unsigned long long* err;
    float *sink, *src;
    CK(hipMalloc(&err, 16 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&src, (size_t)16 << 20 << 2));
    CK(hipMemset(src, 0, (size_t)16 << 20 << 2));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    vk_t vks[] = {victim<0>, victim<1>, victim<2>, victim<3>, victim<4>, victim<5>, victim<6>, victim<7>, victim<8>, victim<9>, victim<10>, victim<11>, victim<12>, victim<13>, victim<14>};
    const char* vn[] = {"v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "2 x v_fma_f32 (control)", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1]",
                        "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_mul_f32 op_sel_hi:[1,0]",
                        "v_pk_add_f32 op_sel:[0,1]", "v_fma_mix_f32 (f16 src1 in the high half)", "v_fma_mix_f32 (f16 src1 in the low half)",
                        "v_pk_fma_f16 op_sel:[0,1,0]"};
    ck_t cks[] = {corunner<0>, corunner<1>, corunner<2>, corunner<3>, corunner<4>, corunner<5>, corunner<6>, corunner<7>, corunner<8>,
                  corunner<9>, corunner<10>, corunner<11>, corunner<12>, corunner<13>};
    const char* cn[] = {"v_nop", "v_exp_f32", "v_rcp_f32 + v_rsq_f32", "v_permlane32_swap + v_permlane16_swap", "DPP row_shr + row_bcast",
                        "v_cvt_pk_bf16_f32", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16", "LDS read/write + ds_bpermute",
                        "v_pk_mul_f32", "v_max3_f32 + v_fma_f32", "global loads", "mix: mfma + exp + permlane + cvt_pk", "v_readlane + s_nop"};
    hipEvent_t e0, e1, e2, e3;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    for (int v = 0; v < 15; ++v) {
        for (int c = -1; c < 14; ++c) {
            if (v >= 5 && c >= 0 && c != 6 && c != 7) continue;
            CK(hipMemset(err, 0, 16 * sizeof(unsigned long long)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0));
            hipLaunchKernelGGL(vks[v], dim3(512), dim3(256), 0, s0, err, viters, 0.37f);
            CK(hipEventRecord(e1, s0));
            CK(hipEventRecord(e2, s1));
            if (c >= 0) hipLaunchKernelGGL(cks[c], dim3(512), dim3(256), 0, s1, sink, src, citers);
            CK(hipEventRecord(e3, s1));
            CK(hipDeviceSynchronize());
            unsigned long long h[16];
            CK(hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost));
            float tv, tc;
            CK(hipEventElapsedTime(&tv, e0, e1));
            CK(hipEventElapsedTime(&tc, e2, e3));
            unsigned long long tot = 0;
            for (int i = 0; i < 8; ++i) tot += h[i];
            printf("victim %-42s co-runner %-40s: %12llu wrong of %.3g  (victim %.1f ms, co-runner %.1f ms)", vn[v], c < 0 ? "(none)" : cn[c], tot,
                   (double)viters * 512 * 256 * 2, tv, tc);
            if (tot) {
                printf("  by quarter-wave (low, high half):");
                for (int q = 0; q < 4; ++q) printf(" [%llu, %llu]", h[2 * q], h[2 * q + 1]);
                printf("  product lost: %llu", h[8]);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
