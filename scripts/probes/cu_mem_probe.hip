// Per-CU store / load ceiling for the access pattern of a 256x256 fp32-epilogue tile (scripts/probes: measurement only).
// Each 512-thread block walks tiles b, b + grid, ...; a wave owns a 128 x 64 sub-block and moves it in 4-row x 256-byte
// dwordx4 instructions (what the GEMM epilogue issues).  mode 0: fp32 stores; 1: fp32 loads; 2: load + store (in place add)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(float* buf, int64_t ld, int tiles_n, int n_tiles, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr = wave >> 2, wc = wave & 3;                       // 2 x 4 waves: 128 rows x 64 cols each
    f4 acc = {1.f, 2.f, 3.f, 4.f};
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        float* base = buf + (int64_t)(tm * 256 + wr * 128) * ld + tn * 256 + wc * 64;
        // one instruction: 4 rows x 16 lanes x 16 B
        for (int r0 = 0; r0 < 128; r0 += 4 * DEPTH) {
            f4 v[DEPTH];
            if (MODE >= 1) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    v[d] = *reinterpret_cast<const f4*>(base + (int64_t)(r0 + 4 * d + (lane >> 4)) * ld + (lane & 15) * 4);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc += v[d];
            }
            if (MODE != 1) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    *reinterpret_cast<f4*>(base + (int64_t)(r0 + 4 * d + (lane >> 4)) * ld + (lane & 15) * 4) = MODE == 2 ? v[d] + acc : acc;
            }
        }
    }
    if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1];
}

template <int MODE, int DEPTH>
void run(const char* name, float* buf, int64_t ld, int tiles_m, int tiles_n, int grid, float* sink) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int per_block = 64;                                     // tiles per block
    const int n_tiles = grid * per_block;
    if (n_tiles > tiles_m * tiles_n) { printf("buffer too small\n"); return; }
    probe<MODE, DEPTH><<<grid, 512>>>(buf, ld, tiles_n, n_tiles, sink);
    CK(hipEventRecord(a));
    probe<MODE, DEPTH><<<grid, 512>>>(buf, ld, tiles_n, n_tiles, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)n_tiles * 256 * 256 * 4 * (MODE == 2 ? 2 : 1);
    printf("%-28s depth %2d grid %3d: %8.1f us/tile  %7.1f GB/s per CU  %7.2f TB/s chip\n", name, DEPTH, grid,
           ms * 1e3 / per_block, bytes / grid / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int tiles_n = 6, ld = 1536;                              // N = 1536 output
    const int tiles_m = 256 * 64 / tiles_n + 1;
    float *buf, *sink;
    CK(hipMalloc(&buf, (size_t)tiles_m * 256 * ld * 4));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(buf, 0, (size_t)tiles_m * 256 * ld * 4));
    for (int grid : {1, 8, 32, 64, 128, 256}) {
        run<0, 4>("fp32 store", buf, ld, tiles_m, tiles_n, grid, sink);
        run<0, 8>("fp32 store", buf, ld, tiles_m, tiles_n, grid, sink);
        run<1, 4>("fp32 load", buf, ld, tiles_m, tiles_n, grid, sink);
        run<1, 8>("fp32 load", buf, ld, tiles_m, tiles_n, grid, sink);
        run<1, 16>("fp32 load", buf, ld, tiles_m, tiles_n, grid, sink);
        run<2, 8>("load + add + store", buf, ld, tiles_m, tiles_n, grid, sink);
    }
    return 0;
}
