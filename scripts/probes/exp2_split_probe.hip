// Is v_exp_f32(s) == ldexp(v_exp_f32(fract(s)), floor(s)) bit for bit?  (Would make the running-maximum recomputation of the
// no-maximum attention kernels bit-identical to the no-maximum form: p = ldexp(exp2(fract s), floor(s) - m) with integer m.)
//   hipcc --offload-arch=gfx950 -O3 exp2_split_probe.hip -o exp2_split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long* bad, unsigned long long* bad2, float* ex, int n_per) {
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long b = 0, b2 = 0;
    for (int i = 0; i < n_per; ++i) {
        x = x * 1664525u + 1013904223u;
        const float s = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 120.0f;      // [-120, 120)
        const float f = __builtin_amdgcn_fractf(s);
        const float fl = s - f;
        const float a = __builtin_amdgcn_exp2f(s);
        const float c = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)fl);
        if (__float_as_uint(a) != __float_as_uint(c)) { ++b; if (b == 1 && ex) { ex[0] = s; ex[1] = a; ex[2] = c; } }
        // and the form used today: exp2(s - m) * 2^m for an integer m near s
        const int m = (int)fl + (int)((x >> 3) & 31) - 8;
        const float d = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(s - (float)m), m);
        if (__float_as_uint(a) != __float_as_uint(d)) ++b2;
    }
    atomicAdd(bad, b); atomicAdd(bad2, b2);
}
int main() {
    unsigned long long *bad, *bad2; float* ex;
    hipMalloc(&bad, 8); hipMalloc(&bad2, 8); hipMalloc(&ex, 16); hipMemset(bad, 0, 8); hipMemset(bad2, 0, 8); hipMemset(ex, 0, 16);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, bad, bad2, ex, 4096);
    unsigned long long h = 0, h2 = 0; float hx[4];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&h2, bad2, 8, hipMemcpyDeviceToHost); hipMemcpy(hx, ex, 16, hipMemcpyDeviceToHost);
    printf("exp2(s) vs ldexp(exp2(fract s), floor s): %llu of %llu differ (first: s=%g %g %g)\n", h, 1024ull * 256 * 4096, hx[0], hx[1], hx[2]);
    printf("exp2(s) vs ldexp(exp2(s - m), m), integer m:  %llu of %llu differ\n", h2, 1024ull * 256 * 4096);
    return 0;
}
