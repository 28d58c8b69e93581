// vf_gemm_bf16: out = epilogue(A[M,K] @ W[N,K]^T + bias), bf16 operands, fp32 accumulate, on gfx950 MFMA.
//
// Roofline: MFMA-bound (2*M*N*K flop; M = 10^4..10^5 tokens, N,K in 512..4608).
// Algorithmic bytes per call = 2*(M*K + N*K) + out bytes (+ residual bytes).
//
// Tile 128(m) x 128(n) x 64(k), 256 threads = 4 waves in a 2x2 grid, each wave owns a 64x64 output
// sub-tile as 4x4 accumulators of v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous
// (nn.Linear keeps W as [N][K]), so both MFMA fragments are 16-byte row reads (ds_read_b128).
// The MFMA is issued with W as the A operand and the activations as the B operand, so a lane ends
// up holding 4 CONSECUTIVE n for one m: epilogues are 8/16-byte vector stores and the GeGLU pair
// (a, gate) sits in the same lane.
//
// Staging: global_load_lds_dwordx4 (LDS-DMA), two LDS stages, one barrier per K-step.  An LDS-DMA
// wave-instruction writes 1 KiB linearly (8 rows x 128 B), so the bank-conflict swizzle
// (16-byte chunk c of row r lives at chunk c ^ ((r>>1)&7)) is applied to the per-lane SOURCE
// address and again on the fragment read (cdna_hip_programming.md rule 21).  With it the
// ds_read_b128 fragment reads are conflict-free (checked against the 4x16-lane group rule).
#include "vf_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int OPERAND_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * OPERAND_BYTES;    // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;        // double buffered: 64 KiB

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// Epilogue for 4 consecutive n (n_base .. n_base+3) of row m.  For GEGLU `v` is the `a` half and
// `gate` the gate half; n_out is the output column of v[0].
template <int EPI>
__device__ __forceinline__ void epilogue_store(f32x4_t v, f32x4_t gate, int64_t m, int n_bias, int n_bias_gate,
                                               int n_out, const float* __restrict__ bias,
                                               const float* __restrict__ res, int64_t ldr, void* out, int64_t ldo) {
    if (bias) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(bias + n_bias);
        v += b;
        if (EPI == VF_EPI_GEGLU_BF16) gate += *reinterpret_cast<const f32x4_t*>(bias + n_bias_gate);
    }
    if (EPI == VF_EPI_RES_F32) v += *reinterpret_cast<const f32x4_t*>(res + m * ldr + n_out);
    if (EPI == VF_EPI_GEGLU_BF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] * gelu_erf(gate[i]);
    }
    if (EPI == VF_EPI_GELU_F32 || EPI == VF_EPI_GELU_BF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
    }
    if (EPI == VF_EPI_F32 || EPI == VF_EPI_RES_F32 || EPI == VF_EPI_GELU_F32) {
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + m * ldo + n_out) = v;
    } else {
        u32x2_t p;
        p[0] = pack2bf(v[0], v[1]);
        p[1] = pack2bf(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out) + m * ldo + n_out) = p;
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(const unsigned short* __restrict__ A, int64_t lda,
                                                          const unsigned short* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ res, int64_t ldr, void* out,
                                                          int64_t ldo, int M, int N, int K, int tiles_n, int n_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    // XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
    // XCD a contiguous run of tiles; consecutive tiles walk n first and share the A row panel in L2.
    const int bid = blockIdx.x;
    const int q8 = n_blocks >> 3, r8 = n_blocks & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, g = lane >> 4;

    // ---- LDS-DMA source pointers: this wave stages pieces p = 4*wave .. 4*wave+3 of each operand
    // (piece = 8 rows x 128 B).  lane -> row 8p + (lane>>3), physical chunk lane&7.
    const unsigned short* srcA[4];
    const unsigned short* srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave * 4 + i;
        const int row = 8 * p + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        srcA[i] = A + (int64_t)gm * lda + c * 8;
        srcW[i] = W + (int64_t)gn * K + c * 8;
    }
    char* const ldsA_piece = smem + wave * 4 * 1024;                     // + stage*STAGE_BYTES + i*1024
    char* const ldsW_piece = smem + OPERAND_BYTES + wave * 4 * 1024;

    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(srcA[i] + kt * BK, ldsA_piece + stage * STAGE_BYTES + i * 1024);
            glds16(srcW[i] + kt * BK, ldsW_piece + stage * STAGE_BYTES + i * 1024);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside an operand tile: row*128 + ((4*ks+g) ^ (r>>1))*16
    const int sw = r >> 1;
    const int offW = (wn * 64 + r) * 128;
    const int offA = (wm * 64 + r) * 128;

    const int nkt = K / BK;
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // tile kt landed for every wave; stage (kt+1)&1 is free
        if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
        const char* sA = smem + (kt & 1) * STAGE_BYTES;
        const char* sW = sA + OPERAND_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ((4 * ks + g) ^ sw) * 16;
            bf16x8_t wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8_t*>(sW + offW + i * 16 * 128 + chunk);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sA + offA + i * 16 * 128 + chunk);
#pragma unroll
            for (int in = 0; in < 4; ++in)
#pragma unroll
                for (int im = 0; im < 4; ++im)
                    acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[in], af[im], acc[in][im], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds out[m = m0+wm*64+im*16+r][n = n0+wn*64+in*16+4g .. +3]
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        const int64_t m = m0 + wm * 64 + im * 16 + r;
        if (m >= M) continue;
        if (EPI == VF_EPI_GEGLU_BF16) {
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int nb = n0 + wn * 64 + ip * 32 + 4 * g;          // bias index of the `a` half
                if (nb >= N) continue;
                const int n_out = (n0 + wn * 64) / 2 + ip * 16 + 4 * g;
                epilogue_store<EPI>(acc[2 * ip][im], acc[2 * ip + 1][im], m, nb, nb + 16, n_out, bias, res, ldr, out, ldo);
            }
        } else {
#pragma unroll
            for (int in = 0; in < 4; ++in) {
                const int nb = n0 + wn * 64 + in * 16 + 4 * g;
                if (nb >= N) continue;
                epilogue_store<EPI>(acc[in][im], acc[in][im], m, nb, nb, nb, bias, res, ldr, out, ldo);
            }
        }
    }
}

// Shape-generic fallback (any K % 8 == 0): 64x64 tile, fp32 FMA out of LDS.  Same lane->output
// ownership as the MFMA kernel so the epilogues are shared.  Only small/odd shapes come here.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const unsigned short* __restrict__ A, int64_t lda,
                                                          const unsigned short* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ res, int64_t ldr, void* out,
                                                          int64_t ldo, int M, int N, int K) {
    __shared__ float sA[64][33];
    __shared__ float sW[64][33];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, ml = tid & 63, ng = tid >> 6;
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        // 64 rows x 32 k per operand = 2048 elements, 8 per thread
        const int row = tid >> 2, kc = (tid & 3) * 8;
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + kc + e;
            sA[row][kc + e] = k < K ? bf2f(A[(int64_t)gm * lda + k]) : 0.f;
            sW[row][kc + e] = k < K ? bf2f(W[(int64_t)gn * K + k]) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float a = sA[ml][k];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(a, sW[16 * j + 4 * ng + e][k], acc[j][e]);
        }
        __syncthreads();
    }
    const int64_t m = m0 + ml;
    if (m >= M) return;
    if (EPI == VF_EPI_GEGLU_BF16) {
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int nb = n0 + ip * 32 + 4 * ng;
            if (nb >= N) continue;
            epilogue_store<EPI>(acc[2 * ip], acc[2 * ip + 1], m, nb, nb + 16, n0 / 2 + ip * 16 + 4 * ng, bias, res, ldr, out, ldo);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + 16 * j + 4 * ng;
            if (nb >= N) continue;
            epilogue_store<EPI>(acc[j], acc[j], m, nb, nb, nb, bias, res, ldr, out, ldo);
        }
    }
}

template <int EPI>
int launch_gemm(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
                int64_t ldo, int M, int N, int K, hipStream_t st) {
    if (K % BK == 0) {
        const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
        const int n_blocks = tiles_m * tiles_n;
        hipLaunchKernelGGL(gemm_mfma_kernel<EPI>, dim3(n_blocks), dim3(256), 0, st, (const unsigned short*)A, lda,
                           (const unsigned short*)W, bias, res, ldr, out, ldo, M, N, K, tiles_n, n_blocks);
    } else {
        dim3 grid((N + 63) / 64, (M + 63) / 64);
        hipLaunchKernelGGL(gemm_generic_kernel<EPI>, grid, dim3(256), 0, st, (const unsigned short*)A, lda,
                           (const unsigned short*)W, bias, res, ldr, out, ldo, M, N, K);
    }
    VF_CHECK_LAUNCH("vf_gemm_bf16");
    return VF_OK;
}

__global__ void pack_geglu_rows_kernel(const unsigned short* __restrict__ W, const float* __restrict__ bias,
                                       unsigned short* __restrict__ Wo, float* __restrict__ bo, int two_f, int K) {
    const int row_out = blockIdx.x;                 // 0 .. 2F-1
    const int F = two_f / 2;
    const int b = row_out >> 5, t = row_out & 31;
    const int row_in = t < 16 ? 16 * b + t : F + 16 * b + (t - 16);
    for (int k = threadIdx.x; k < K; k += blockDim.x) Wo[(int64_t)row_out * K + k] = W[(int64_t)row_in * K + k];
    if (bias && threadIdx.x == 0) bo[row_out] = bias[row_in];
}

}  // namespace

extern "C" int vf_gemm_bf16(const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                            int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int epilogue, void* stream) {
    VF_REQUIRE(A && W && out, "vf_gemm_bf16: null pointer");
    VF_REQUIRE(M >= 0 && N > 0 && K > 0, "vf_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    VF_REQUIRE(K % 8 == 0 && N % 8 == 0, "vf_gemm_bf16: K and N must be multiples of 8 (K=%d N=%d)", K, N);
    VF_REQUIRE(lda % 8 == 0 && lda >= K, "vf_gemm_bf16: lda=%lld must be >= K and a multiple of 8", (long long)lda);
    VF_REQUIRE(ldo % 4 == 0, "vf_gemm_bf16: ldo=%lld must be a multiple of 4", (long long)ldo);
    VF_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)out % 16 == 0),
               "vf_gemm_bf16: pointers must be 16-byte aligned");
    if (M == 0) return VF_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case VF_EPI_BF16: return launch_gemm<VF_EPI_BF16>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st);
        case VF_EPI_F32: return launch_gemm<VF_EPI_F32>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st);
        case VF_EPI_RES_F32:
            VF_REQUIRE(residual && ldr % 4 == 0 && ((uintptr_t)residual % 16 == 0), "vf_gemm_bf16: residual epilogue needs an aligned residual");
            return launch_gemm<VF_EPI_RES_F32>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st);
        case VF_EPI_GEGLU_BF16:
            VF_REQUIRE(N % 32 == 0, "vf_gemm_bf16: GEGLU epilogue needs N %% 32 == 0 (N=%d)", N);
            return launch_gemm<VF_EPI_GEGLU_BF16>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st);
        case VF_EPI_GELU_F32: return launch_gemm<VF_EPI_GELU_F32>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st);
        case VF_EPI_GELU_BF16: return launch_gemm<VF_EPI_GELU_BF16>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st);
        default: VF_REQUIRE(false, "vf_gemm_bf16: unknown epilogue %d", epilogue);
    }
    return VF_OK;
}

extern "C" int vf_pack_geglu_rows(const void* W, const float* bias, void* W_out, float* bias_out, int two_f, int K,
                                  void* stream) {
    VF_REQUIRE(W && W_out && two_f > 0 && two_f % 32 == 0 && K > 0, "vf_pack_geglu_rows: need 2F %% 32 == 0 (2F=%d)", two_f);
    VF_REQUIRE(!bias || bias_out, "vf_pack_geglu_rows: bias given without bias_out");
    hipLaunchKernelGGL(pack_geglu_rows_kernel, dim3(two_f), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)W,
                       bias, (unsigned short*)W_out, bias_out, two_f, K);
    VF_CHECK_LAUNCH("vf_pack_geglu_rows");
    return VF_OK;
}
