// hw_probe: checks, with exact integer data, every gfx950 lane map the kernels in this
// directory rely on (MFMA operand / accumulator maps, ds_read_b64_tr_b16 gather, LDS-DMA
// destination order, permlane swaps) and times the bf16 MFMA shapes.
// Build:  hipcc --offload-arch=gfx950 -O3 hw_probe.hip -o hw_probe      Run on a MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

__device__ inline __bf16 i2bf(int v) { return (__bf16)(float)v; }

// ---- 1. mfma 16x16x32 bf16 ------------------------------------------------------------
__global__ void k_mfma16(const int* A /*16x32*/, const int* B /*32x16*/, float* C /*16x16*/) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = i2bf(A[(l & 15) * 32 + 8 * (l >> 4) + j]);
        b[j] = i2bf(B[(8 * (l >> 4) + j) * 16 + (l & 15)]);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// ---- 2. mfma 32x32x16 bf16 ------------------------------------------------------------
__global__ void k_mfma32(const int* A /*32x16*/, const int* B /*16x32*/, float* C /*32x32*/) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = i2bf(A[(l & 31) * 16 + 8 * (l >> 5) + j]);
        b[j] = i2bf(B[(8 * (l >> 5) + j) * 32 + (l & 31)]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// ---- 3. legacy mfma 16x16x16 bf16_1k --------------------------------------------------
__global__ void k_mfma16k16(const int* A /*16x16*/, const int* B /*16x16*/, float* C) {
    int l = threadIdx.x;
    s16x4 a, b;
    for (int j = 0; j < 4; ++j) {
        __bf16 x = i2bf(A[(l & 15) * 16 + 4 * (l >> 4) + j]);
        __bf16 y = i2bf(B[(4 * (l >> 4) + j) * 16 + (l & 15)]);
        a[j] = *(short*)&x; b[j] = *(short*)&y;
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// ---- 4. ds_read_b64_tr_b16 ------------------------------------------------------------
// LDS image: rows of RS shorts, value = row*1000 + col. Group g (16 lanes) reads rows 4g..4g+3,
// columns c0..c0+15; lane 4q+p supplies &img[4g+q][c0+4p].
template <int RS>
__global__ void k_tr(short* out /*64x4*/, int c0) {
    __shared__ __attribute__((aligned(16))) short img[32 * RS];
    int l = threadIdx.x;
    for (int i = l; i < 32 * RS; i += 64) img[i] = (short)((i / RS) * 1000 + (i % RS));
    __syncthreads();
    int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const short* addr = &img[(4 * g + q) * RS + c0 + 4 * p];
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = t[e];
}
// ---- 5. global_load_lds dwordx4: destination = base + lane*16 ---------------------------
__global__ void k_glds(const int* src /*64*4 ints*/, int* out /*64*4*/) {
    __shared__ __attribute__((aligned(16))) int buf[64 * 4 * 2];
    int l = threadIdx.x;
    int perm = (l * 7 + 3) & 63;   // per-lane SOURCE permutation
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)(src + perm * 4),
                                     (__attribute__((address_space(3))) void*)(buf + 64 * 4), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = buf[64 * 4 + l * 4 + j];
}
// ---- 6. permlane swaps ------------------------------------------------------------------
__global__ void k_perm(unsigned* out /*64*4*/) {
    unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    auto s = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    out[l * 4 + 0] = r[0]; out[l * 4 + 1] = r[1]; out[l * 4 + 2] = s[0]; out[l * 4 + 3] = s[1];
}
// ---- 7. MFMA issue rates ------------------------------------------------------------------
template <int KIND>
__global__ void k_rate(float* out, long long* cyc, int iters) {
    int l = threadIdx.x;
    bf16x8 a, b;
    s16x4 a4, b4;
    for (int j = 0; j < 8; ++j) { a[j] = i2bf((l + j) & 3); b[j] = i2bf((l * 3 + j) & 3); }
    for (int j = 0; j < 4; ++j) { a4[j] = (short)0x3f80; b4[j] = (short)0x3f80; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0; for (int r = 0; r < 16; ++r) d0[r] = 0; f32x16 d1 = d0;
    long long t0 = wall_clock64();
    long long s0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c3, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
        }
    }
    long long s1 = clock64();
    long long t1 = wall_clock64();
    if (l == 0) { cyc[0] = s1 - s0; cyc[1] = t1 - t0; }
    out[l] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
}

template <typename T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

int main() {
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    printf("device: %s arch=%s CUs=%d clock=%d kHz lds/block=%zu\n", prop.name, prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, prop.sharedMemPerBlock);
    int fails = 0;
    {   // 1
        std::vector<int> A(16 * 32), B(32 * 16); std::vector<float> C(256), R(256, 0.f);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (i * 3 + k * 5) % 7 - 3;
        for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k * 2 + j * 7 + 1) % 5 - 2;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
        int *dA = dalloc<int>(512), *dB = dalloc<int>(512); float* dC = dalloc<float>(256);
        CK(hipMemcpy(dA, A.data(), 512 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512 * 4, hipMemcpyHostToDevice));
        k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]);
        printf("[1] mfma_f32_16x16x32_bf16 A[l&15][8(l>>4)+j] B[8(l>>4)+j][l&15] C[(l>>4)*4+r][l&15]: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // 2
        std::vector<int> A(32 * 16), B(16 * 32); std::vector<float> C(1024), R(1024, 0.f);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (i * 3 + k * 5) % 7 - 3;
        for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (k * 2 + j * 7 + 1) % 5 - 2;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
        int *dA = dalloc<int>(512), *dB = dalloc<int>(512); float* dC = dalloc<float>(1024);
        CK(hipMemcpy(dA, A.data(), 512 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512 * 4, hipMemcpyHostToDevice));
        k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += (C[i] != R[i]);
        printf("[2] mfma_f32_32x32x16_bf16 maps: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // 3
        std::vector<int> A(256), B(256); std::vector<float> C(256), R(256, 0.f);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (i * 3 + k * 5) % 7 - 3;
        for (int k = 0; k < 16; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k * 2 + j * 7 + 1) % 5 - 2;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 16; ++k) R[i * 16 + j] += A[i * 16 + k] * B[k * 16 + j];
        int *dA = dalloc<int>(256), *dB = dalloc<int>(256); float* dC = dalloc<float>(256);
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        k_mfma16k16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]);
        printf("[3] mfma_f32_16x16x16bf16_1k A[l&15][4(l>>4)+j] B[4(l>>4)+j][l&15]: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    for (int variant = 0; variant < 3; ++variant) {   // 4
        short* d = dalloc<short>(256); std::vector<short> h(256);
        int RS = variant == 0 ? 64 : (variant == 1 ? 48 : 80), c0 = variant == 2 ? 16 : 0;
        if (variant == 0) k_tr<64><<<1, 64>>>(d, c0); else if (variant == 1) k_tr<48><<<1, 64>>>(d, c0); else k_tr<80><<<1, 64>>>(d, c0);
        CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
            int g = l >> 4, i = l & 15; short want = (short)((4 * g + e) * 1000 + c0 + i);
            bad += (h[l * 4 + e] != want);
        }
        printf("[4] ds_read_b64_tr_b16 RS=%d c0=%d: lane i of group g gets img[4g+e][c0+i] in element e: %s (%d bad)  lane5:", RS, c0, bad ? "FAIL" : "PASS", bad);
        for (int e = 0; e < 4; ++e) printf(" %d", h[5 * 4 + e]); printf("  lane21:"); for (int e = 0; e < 4; ++e) printf(" %d", h[21 * 4 + e]); printf("\n");
        fails += bad != 0;
    }
    {   // 5
        std::vector<int> s(256), o(256); for (int i = 0; i < 256; ++i) s[i] = i * 11 + 1;
        int *ds = dalloc<int>(256), *dout = dalloc<int>(256);
        CK(hipMemcpy(ds, s.data(), 1024, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(ds, dout); CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += (o[l * 4 + j] != s[((l * 7 + 3) & 63) * 4 + j]);
        printf("[5] global_load_lds x4: LDS[base+16*lane] = src[perm(lane)]: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // 6
        unsigned* d = dalloc<unsigned>(256); std::vector<unsigned> h(256);
        k_perm<<<1, 64>>>(d); CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
        printf("[6] permlane32_swap(l, l+100): lane0 -> (%u,%u) lane40 -> (%u,%u); permlane16_swap: lane0 (%u,%u) lane20 (%u,%u) lane40 (%u,%u)\n",
               h[0], h[1], h[160], h[161], h[2], h[3], h[82], h[83], h[162], h[163]);
    }
    for (int kind = 0; kind < 3; ++kind) {   // 7
        float* o = dalloc<float>(64 * 4); long long* c = dalloc<long long>(2); long long hc[2];
        int iters = 4096;
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) k_rate<0><<<1, 64>>>(o, c, iters); else if (kind == 1) k_rate<1><<<1, 64>>>(o, c, iters); else k_rate<2><<<1, 64>>>(o, c, iters);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost));
        const char* nm[3] = {"16x16x32_bf16", "16x16x16bf16_1k", "32x32x16_bf16"};
        printf("[7] %s: %.2f shader-cycles per MFMA (clock64), %.2f ns per MFMA (100MHz wall)\n", nm[kind], (double)hc[0] / (4.0 * iters), (double)hc[1] * 10.0 / (4.0 * iters));
    }
    printf("hw_probe: %s\n", fails ? "SOME FAILED" : "ALL PASS");
    return fails ? 1 : 0;
}
