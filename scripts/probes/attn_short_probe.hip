// Phase timing of attn_short_kernel on the gene stream's self-attention shape (scripts/probes: measurement only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1 -DVF_SHORT_PROF -I../../variantformer_amd/csrc -I../../include attn_short_probe.hip -o attn_short_probe
#include "../../variantformer_amd/csrc/vf_attn.hip"
#include <cstdio>
#include <cstdarg>
#include <vector>
#include <algorithm>
void vf_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
    const int genes = argc > 1 ? atoi(argv[1]) : 8;
    const int H = 32, DH = 48, D = H * DH, n_seq = 54 * genes, L = 201;
    const int64_t tok = (int64_t)n_seq * L;
    unsigned short* qkv; unsigned short* out; int* cu; float* slopes; unsigned long long* prof;
    const int nrec = (n_seq * H + 63) / 64 + 8;
    CK(hipMalloc(&qkv, tok * 3 * D * 2)); CK(hipMalloc(&out, tok * D * 2));
    CK(hipMalloc(&cu, (n_seq + 1) * 4)); CK(hipMalloc(&slopes, H * 4)); CK(hipMalloc(&prof, nrec * 8 * 8));
    std::vector<unsigned short> h(tok * 3 * D);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3C00u + ((x >> 20) & 0x7Fu) + ((x >> 12) & 0x8000u)); }
    CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<int> hc(n_seq + 1); for (int i = 0; i <= n_seq; ++i) hc[i] = i * L;
    CK(hipMemcpy(cu, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hs(H); for (int i = 0; i < H; ++i) hs[i] = powf(2.f, -8.f * (i + 1) / H);
    CK(hipMemcpy(slopes, hs.data(), H * 4, hipMemcpyHostToDevice));
    AttnParams P{};
    P.q = qkv; P.k = qkv + D; P.v = qkv + 2 * D; P.out = out;
    P.q_stride = P.k_stride = P.v_stride = 3 * D; P.o_stride = D;
    P.cu_q = cu; P.cu_k = cu; P.slopes = slopes; P.scale_log2 = 1.0f; P.H = H; P.q_at_start = 0; P.q_log2 = 1;
    P.prof = prof;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(prof, 0, nrec * 64));
        CK(hipEventRecord(a));
        if (launch_short<48, 4, true, VF_BF16>(P, n_seq, L, nullptr) != VF_OK) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> hp(nrec * 8);
        CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
        const char* names[5] = {"issue loads", "wait loads", "LDS write + barrier", "tiles", "stores"};
        double s[5] = {0, 0, 0, 0, 0}; int n = 0;
        unsigned long long t_min = ~0ull, t_max = 0;
        for (int i = 0; i < (n_seq * H) / 64; ++i) {
            if (hp[i * 8 + 5] == 0) continue;
            for (int k = 0; k < 5; ++k) s[k] += hp[i * 8 + k];
            t_min = std::min(t_min, hp[i * 8 + 6]); t_max = std::max(t_max, hp[i * 8 + 5]);
            ++n;
        }
        printf("rep %d: %.1f us for %d blocks; sampled %d blocks (wave 0), kernel span %llu cycles\n", rep, ms * 1e3, n_seq * H, n, t_max - t_min);
        double tot = 0; for (int k = 0; k < 5; ++k) tot += s[k] / n;
        for (int k = 0; k < 5; ++k) printf("   %-22s %8.0f cycles  (%4.1f %%)\n", names[k], s[k] / n, 100.0 * s[k] / n / tot);
        printf("   block life %.0f cycles; blocks per CU %.1f -> if 2 resident: %.0f cycles per block slot\n", tot, n_seq * H / 256.0, (t_max - t_min) / (n_seq * H / 512.0));
    }
    return 0;
}
