// Where a tile of the PERSISTENT 256x256 GEMM (gemm8x_kernel) spends its cycles (scripts/probes: measurement only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DVF_G8_PROF -I../../variantformer_amd/csrc -I../../include gemm8x_probe.hip -o gemm8x_probe
// Stamps of wave 0 (group 0) and wave 4 (group 1) of every 8th block, summed over the block's tiles: tile start (K-tile 1
// requested + start barriers), first K-tile, remaining K-tiles, this wave's epilogue, wait for the other waves' epilogues.
#include "../../variantformer_amd/csrc/vf_gemm.hip"
#include <cstdio>
#include <cstdarg>
#include <vector>
#include <algorithm>
void vf_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// kind: 0 = LayerNorm consumer, 16-bit out; 1 = LayerNorm consumer GeGLU; 2 = 16-bit-residual LayerNorm producer (no fp32 rows)
static int g_zero = 0, g_warm = 0;   // clock mode: all-zero operands / launches before the measured ones
static void run(const char* name, int M, int N, int K, int kind) {
    unsigned short *A, *W, *x16, *r16p; float *bias, *colsum, *stats, *part; unsigned long long* prof;
    const int n_out = kind == 1 ? N / 2 : N;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&x16, (size_t)M * n_out * 2));
    CK(hipMalloc(&r16p, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&colsum, N * 4));
    CK(hipMalloc(&stats, (size_t)M * 8)); CK(hipMalloc(&part, (size_t)(N / 32) * M * 8));
    const int nrec = 256 / 8 * 2 + 2;
    CK(hipMalloc(&prof, nrec * 64));
    // random-ish bf16 operands (the clock under load depends on the data: MI355X_MICROARCH.md, DVFS give-back)
    std::vector<unsigned short> h((size_t)M * K);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 16) & 0x3ff) - 0x200 + ((x >> 31) << 15)); }
    if (g_zero) std::fill(h.begin(), h.end(), (unsigned short)0);
    CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)N * K && i < h.size(); ++i) h[i] = g_zero ? (unsigned short)0 : (unsigned short)(h[i] - 0x300);
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(r16p, 0x3c, (size_t)M * N * 2)); CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(colsum, 0, N * 4));
    { std::vector<float> st((size_t)M * 2); for (size_t i = 0; i < st.size(); i += 2) { st[i] = 0.01f; st[i + 1] = 1.0f; } CK(hipMemcpy(stats, st.data(), st.size() * 4, hipMemcpyHostToDevice)); }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(vf_g8_prof), &prof, sizeof(prof)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 3 + g_warm;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemset(prof, 0, nrec * 64));
        CK(hipEventRecord(a));
        int rc;
        if (kind == 2) rc = vf_gemm_ln(A, K, W, bias, r16p, N, VF_BF16, nullptr, N, M, N, K, VF_EPI_RES_F32, VF_BF16, nullptr, nullptr, x16, N, part, 1.0f, 1.0f, nullptr);
        else rc = vf_gemm_ln(A, K, W, bias, nullptr, 0, VF_F32, x16, n_out, M, N, K, kind == 1 ? VF_EPI_GEGLU_BF16 : VF_EPI_BF16, VF_BF16, stats, colsum, nullptr, 0, nullptr, 1.0f, 1.0f, nullptr);
        if (rc) { printf("rc %d\n", rc); return; }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep < reps - 1) continue;
        std::vector<unsigned long long> hp(nrec * 8);
        CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
        const char* nm[5] = {"tile start (request K-tile 1, barriers)", "first K-tile", "K-tiles 1 .. n-1 (+ group barrier)", "this wave's epilogue", "wait for the other waves"};
        const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
        printf("%s: M=%d N=%d K=%d  %.1f us = %.0f TFLOP/s, %d tiles (%.2f per CU); cycles per tile:\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, tiles, tiles / 256.0);
        for (int grp = 0; grp < 2; ++grp) {
            double s[5] = {0}, cnt = 0;
            for (int i = 0; i < 32; ++i) { const unsigned long long* r = &hp[(i * 2 + grp) * 8]; if (!r[5]) continue; for (int k = 0; k < 5; ++k) s[k] += (double)r[k]; cnt += (double)r[5]; }
            double tot = 0; for (int k = 0; k < 5; ++k) tot += s[k] / cnt;
            double mt = 0, rt = 0;
            for (int i = 0; i < 32; ++i) { const unsigned long long* r = &hp[(i * 2 + grp) * 8]; mt += (double)r[6]; rt += (double)r[7]; }
            if (grp == 0 && rt > 0) printf("  clock held during the launch (d s_memtime / d s_memrealtime x 100 MHz): %.0f MHz\n", mt / rt * 100.0);
            printf("  wave %d (group %d), %.0f tiles sampled, %.0f cycles per tile:\n", grp * 4, grp, cnt, tot);
            for (int k = 0; k < 5; ++k) printf("    %-42s %8.0f  (%4.1f %%)\n", nm[k], s[k] / cnt, 100 * s[k] / cnt / tot);
        }
    }
    hipFree(A); hipFree(W); hipFree(x16); hipFree(r16p); hipFree(bias); hipFree(colsum); hipFree(stats); hipFree(part); hipFree(prof);
}

int main(int argc, char** argv) {
    if (argc > 1) {          // ./gemm8x_probe clock : in-kernel clock of the step's largest GEMM on random and on all-zero operands,
        g_warm = 600;        // after ~2 s of back-to-back launches each (MI355X_MICROARCH.md, DVFS give-back item 6)
        g_zero = 0; run("gene Wqkv, 32 genes, RANDOM operands", 347328, 4608, 1536, 0);
        g_zero = 1; run("gene Wqkv, 32 genes, ALL-ZERO operands", 347328, 4608, 1536, 0);
        return 0;
    }
    run("seq2reg Wqkv (LN consumer, 16-bit out)", 769460, 1536, 512, 0);
    run("seq2reg GeGLU (LN consumer)", 769460, 2048, 512, 1);
    run("gene Wqkv (LN consumer, 16-bit out)", 86832, 4608, 1536, 0);
    run("gene GeGLU (LN consumer)", 86832, 2048, 1536, 1);
    run("gene out_proj (r16 producer, no fp32 rows)", 86832, 1536, 1536, 2);
    run("seq2reg out_proj (r16 producer)", 769460, 512, 512, 2);
    return 0;
}
