// Interval timing of attn_x32pp_kernel on the gene -> CRE cross attention shape (scripts/probes: measurement only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1 -DVF_TUNING -DVF_X32PP_PROF -I../../variantformer_amd/csrc -I../../include attn_x32pp_probe.hip -o attn_x32pp_probe
#include "../../variantformer_amd/csrc/vf_attn.hip"
#include <cstdio>
#include <cstdarg>
#include <vector>
#include <algorithm>
void vf_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
    const int genes = argc > 1 ? atoi(argv[1]) : 8;
    const int H = 32, DH = 48, D = H * DH, LQ = 54 * 201, LK = 1024;
    const int64_t tq = (int64_t)genes * LQ, tk = (int64_t)genes * LK;
    unsigned short *q, *kv, *out; int *cuq, *cuk; unsigned long long* prof;
    const int nqb = (LQ + 511) / 512, nblk = genes * H * nqb, nrec = (nblk + 7) / 8 * 2 + 16;
    CK(hipMalloc(&q, tq * D * 2)); CK(hipMalloc(&kv, tk * 2 * D * 2)); CK(hipMalloc(&out, tq * D * 2));
    CK(hipMalloc(&cuq, (genes + 1) * 4)); CK(hipMalloc(&cuk, (genes + 1) * 4)); CK(hipMalloc(&prof, nrec * 64));
    unsigned x = 12345;
    auto fill = [&](unsigned short* d, int64_t n, unsigned short base) {
        std::vector<unsigned short> h(n);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(base + ((x >> 20) & 0x7Fu) + ((x >> 12) & 0x8000u)); }
        CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    };
    fill(q, tq * D, 0x3E80u);        // |q| ~ 0.25 .. 0.5 (bf16): base-2 logits of a few units
    fill(kv, tk * 2 * D, 0x3F00u);
    std::vector<int> hq(genes + 1), hk(genes + 1);
    for (int i = 0; i <= genes; ++i) { hq[i] = i * LQ; hk[i] = i * LK; }
    CK(hipMemcpy(cuq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(cuk, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    AttnParams P{};
    P.q = q; P.k = kv; P.v = kv + D; P.out = out;
    P.q_stride = D; P.k_stride = P.v_stride = 2 * D; P.o_stride = D;
    P.cu_q = cuq; P.cu_k = cuk; P.slopes = nullptr; P.scale_log2 = 1.0f; P.H = H; P.q_at_start = 0; P.q_log2 = 1;
    P.prof = prof;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(prof, 0, nrec * 64));
        AttnParams PP = P;
        const dim3 grid(set_grid(PP, genes, nqb));
        CK(hipEventRecord(a));
        if (launch_x32pp<VF_BF16>(PP, grid, nullptr) != VF_OK) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep < 2) continue;
        std::vector<unsigned long long> hp(nrec * 8);
        CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
        const char* names[6] = {"prologue", "matrix segments", "barrier behind matrix", "vector segments", "barrier behind vector", "vote + epilogue"};
        printf("rep %d: %.1f us for %d blocks of 512 queries (%.1f per CU) = %.0f TFLOP/s\n", rep, ms * 1e3, nblk, nblk / 256.0,
               4.0 * tq * LK * D / (ms * 1e-3) / 1e12);
        for (int grp = 0; grp < 2; ++grp) {
            double s[6] = {0, 0, 0, 0, 0, 0}; int n = 0; int same_simd = 0, pairs = 0;
            for (int i = 0; i < (nblk + 7) / 8; ++i) {
                const unsigned long long* r = &hp[(i * 2 + grp) * 8];
                if (r[7] == 0) continue;
                for (int k = 0; k < 6; ++k) s[k] += r[k];
                ++n;
                const unsigned long long* r0 = &hp[(i * 2) * 8];
                const unsigned long long* r1 = &hp[(i * 2 + 1) * 8];
                if (r0[7] && r1[7]) { ++pairs; same_simd += (r0[6] == r1[6]); }
            }
            double tot = 0; for (int k = 0; k < 6; ++k) tot += s[k] / n;
            printf("  wave %d (group %c), %d blocks sampled, block life %.0f cycles; waves 0 and 4 on the same SIMD in %d of %d blocks\n", grp * 4, "AB"[grp], n, tot, same_simd, pairs);
            for (int k = 0; k < 6; ++k) printf("     %-24s %8.0f cycles  (%4.1f %%)%s\n", names[k], s[k] / n, 100.0 * s[k] / n / tot,
                                               (k >= 1 && k <= 4) ? "   per tile: " : "");
            printf("     per tile (16 tiles): matrix %.0f + barrier %.0f + vector %.0f + barrier %.0f\n", s[1] / n / 16, s[2] / n / 16, s[3] / n / 16, s[4] / n / 16);
        }
    }
    return 0;
}
