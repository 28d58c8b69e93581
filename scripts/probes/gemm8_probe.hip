// Phase timing of the one-shot 256x256 GEMM with a LayerNorm-producer epilogue (scripts/probes: measurement only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DVF_G8_PROF -I../../variantformer_amd/csrc -I../../include gemm8_probe.hip -o gemm8_probe
#include "../../variantformer_amd/csrc/vf_gemm.hip"
#include <cstdio>
#include <cstdarg>
#include <vector>
void vf_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static void run(const char* name, int M, int N, int K, bool r16, bool need_x) {
    unsigned short *A, *W, *x16, *r16p; float *bias, *res, *out, *part; unsigned long long* prof;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&x16, (size_t)M * N * 2));
    CK(hipMalloc(&r16p, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&res, (size_t)M * N * 4));
    CK(hipMalloc(&out, (size_t)M * N * 4)); CK(hipMalloc(&part, (size_t)(N / 32) * M * 8));
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), nrec = tiles / 16 + 1;
    CK(hipMalloc(&prof, nrec * 64));
    CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(W, 0x3c, (size_t)N * K * 2)); CK(hipMemset(r16p, 0x3c, (size_t)M * N * 2));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(res, 0, (size_t)M * N * 4));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(vf_g8_prof), &prof, sizeof(prof)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(prof, 0, nrec * 64));
        CK(hipEventRecord(a));
        int rc = vf_gemm_ln(A, K, W, bias, r16 ? (const void*)r16p : (const void*)res, N, r16 ? VF_BF16 : VF_F32, need_x ? out : nullptr, N,
                            M, N, K, VF_EPI_RES_F32, VF_BF16, nullptr, nullptr, x16, N, part, 1.0f, 1.0f, nullptr);
        if (rc) { printf("rc %d\n", rc); return; }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep < 2) continue;
        std::vector<unsigned long long> hp(nrec * 8);
        CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
        double s[7] = {0}; int n = 0;
        for (int i = 0; i < tiles / 16; ++i) { if (!hp[i * 8 + 7]) continue; for (int k = 0; k < 7; ++k) s[k] += (double)(hp[i * 8 + k + 1] - hp[i * 8 + k]); ++n; }
        const char* nm[7] = {"prologue (K-tile 0 in LDS)", "K loop", "epilogue pass 0", "pass 1", "pass 2", "pass 3", "store drain"};
        double tot = 0; for (int k = 0; k < 7; ++k) tot += s[k] / n;
        printf("%s: %.1f us, %d tiles (%.2f per CU); wave 0 of %d sampled blocks, cycles per tile (total %.0f):\n", name, ms * 1e3, tiles, tiles / 256.0, n, tot);
        for (int k = 0; k < 7; ++k) printf("    %-28s %8.0f  (%4.1f %%)\n", nm[k], s[k] / n, 100 * s[k] / n / tot);
    }
    hipFree(A); hipFree(W); hipFree(x16); hipFree(r16p); hipFree(bias); hipFree(res); hipFree(out); hipFree(part); hipFree(prof);
}

int main() {
    run("gene out_proj (r16 residual, no fp32 store)", 86832, 1536, 1536, true, false);
    run("gene down-projection (fp32 trunk)", 86832, 1536, 1024, false, true);
    run("seq2reg out_proj (r16, no fp32 store)", 769460, 512, 512, true, false);
    return 0;
}
