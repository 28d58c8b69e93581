// HBM-bound helper kernels of libvf_hip: LayerNorm, token embedding + packing, segment mean pooling,
// row gathers, the expression head's final dot + softplus, casts; plus vf_version / vf_last_error.
// All are streaming kernels: roofline = HBM, algorithmic bytes = (input + output) bytes of one pass.
#include <stdarg.h>
#include "vf_common.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void vf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vf_version(void) { return VF_ABI_VERSION; }
extern "C" const char* vf_last_error(void) { return g_err; }

static thread_local const char* g_last_kernel[2] = {"", ""};
void vf_note_kernel(int which, const char* name) { g_last_kernel[which & 1] = name; }
extern "C" const char* vf_last_kernel(int which) { return g_last_kernel[which & 1]; }

namespace {

// runtime 16-bit output type (VF_BF16 / VF_F16): uniform per launch
__device__ __forceinline__ unsigned int pack2_dt(float lo, float hi, int dt) {
    return dt == VF_F16 ? pack2h(lo, hi) : pack2bf(lo, hi);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (two-pass mean / variance like ATen),
// float4 loads, bf16x4 or float4 stores.
// ---------------------------------------------------------------------------------------------
template <int MAXC>   // MAXC float4 chunks per lane: D <= 256*MAXC
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ out,
                                                        int64_t rows, int D, float eps, int out_dt, int gelu) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n4 = D >> 2;
    const f32x4_t* xr = reinterpret_cast<const f32x4_t*>(x + row * D);
    f32x4_t v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + 64 * c;
        v[c] = i < n4 ? xr[i] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + 64 * c;
        if (i < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[c][e] - mean;
                ss += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(gamma);
    const f32x4_t* b4 = reinterpret_cast<const f32x4_t*>(beta);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + 64 * c;
        if (i < n4) {
            const f32x4_t gg = g4[i], bb = b4[i];
            f32x4_t y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (v[c][e] - mean) * rstd * gg[e] + bb[e];
                if (gelu) y[e] = gelu_erf(y[e]);
            }
            if (out_dt != VF_F32) {
                u32x2_t p;
                p[0] = pack2_dt(y[0], y[1], out_dt);
                p[1] = pack2_dt(y[2], y[3], out_dt);
                reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out) + row * D)[i] = p;
            } else {
                reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + row * D)[i] = y;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm folded into the neighbouring GEMMs (vf_gemm_ln_bf16): row statistics
// ---------------------------------------------------------------------------------------------
// (sum, second moment about the part mean) per 32-column part of a row, written by the producing GEMM's epilogue ->
// (mean, rstd) per row.  Parts are merged with the parallel-variance formula (Chan et al.): M2 = sum_p M2_p +
// 32 * sum_p (mean_p - mean)^2 -- every term is non-negative, so rows whose mean is large against their spread lose no
// digits (E[x^2] - mean^2 would).
// x16_scale: the 16-bit copy of the stream holds x * c (fp16 streams; 1 otherwise) -- the consumer GEMM's accumulator is
// then c * (x . w'), so the pair it needs is (mean * c, rstd / c).  alert (optional): set to 1 when a row's |mean| exceeds
// ratio_limit standard deviations, the regime in which the folded form's 16-bit rounding of the UNCENTRED row costs
// accuracy (tests/test_ops_gpu.py::test_ln_fold_rows_with_large_mean); the model reads the flag back with its outputs.
// NP > 0: n_parts == NP, every part of the row is loaded ONCE, all loads in flight together (one HBM latency per thread; the
// round-2 form swept the parts twice, 8 loads at a time: 15 us for 33 MB on the gene stream).  NP = 0: any part count, two
// sweeps.  Same additions in the same order either way.
// Device flag shared by the statistics kernels (read back with a batch's outputs, ops.ln_fold_alert):
//   bit 0: a row's |mean| exceeds ratio_limit standard deviations -- rounding the UNCENTRED row to 16 bits costs the folded
//          LayerNorm accuracy there;
//   bit 1: a row may hold an element of magnitude >= abs_limit (|x_i - mean| <= sqrt(D * var), so |mean| + sqrt(D * var) bounds
//          every element): its scaled fp16 copies (operand copy of an fp16 stream, the fp16 trunk copy) could overflow to inf.
// Either bit makes the model recompute the batch with the separate-LayerNorm fp32-stream path.  Rare path: atomicOr.
__device__ __forceinline__ void ln_alert(int* alert, float mean, float var, float rstd, int D, float ratio_limit, float abs_limit) {
    if (!alert) return;
    int bits = 0;
    if (fabsf(mean) * rstd > ratio_limit) bits |= 1;
    if (abs_limit > 0.f && !(fabsf(mean) + sqrtf((float)D * var) < abs_limit)) bits |= 2;     // also true for inf / NaN rows
    if (bits) atomicOr(alert, bits);
}

template <int NP>
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, float* __restrict__ row_stats,
                                                         int64_t rows, int n_parts, int D, float eps, float x16_scale,
                                                         float ratio_limit, float abs_limit, int* __restrict__ alert) {
    // part is [n_parts][rows][2] (part-major): one thread per row, consecutive threads read consecutive rows; the parts
    // are added in index order, so the result does not depend on the GEMM tile configuration that wrote them
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s1 = 0.f, m2 = 0.f, mean;
    if constexpr (NP > 0) {
        f32x2_t v[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = *reinterpret_cast<const f32x2_t*>(part + ((int64_t)p * rows + row) * 2);
#pragma unroll
        for (int p = 0; p < NP; ++p) s1 += v[p][0];
        mean = s1 / (float)D;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float d = v[p][0] * (1.0f / 32.0f) - mean;
            m2 += v[p][1] + 32.0f * d * d;
        }
    } else {
#pragma unroll 8
        for (int p = 0; p < n_parts; ++p) s1 += part[((int64_t)p * rows + row) * 2];
        mean = s1 / (float)D;
#pragma unroll 8
        for (int p = 0; p < n_parts; ++p) {                  // second sweep: the same lines, L2-resident
            const f32x2_t v = *reinterpret_cast<const f32x2_t*>(part + ((int64_t)p * rows + row) * 2);
            const float d = v[0] * (1.0f / 32.0f) - mean;
            m2 += v[1] + 32.0f * d * d;
        }
    }
    const float var = m2 / (float)D;
    const float rstd = rsqrtf(var + eps);
    *reinterpret_cast<f32x2_t*>(row_stats + 2 * row) = (f32x2_t){mean * x16_scale, rstd / x16_scale};
    ln_alert(alert, mean, var, rstd, D, ratio_limit, abs_limit);
}

// The same statistics for a stream that no GEMM produced (the first layer's input): one wave per row, two-pass
// mean / variance like layernorm_kernel, plus the 16-bit copy of the row.
template <int MAXC>
__global__ __launch_bounds__(256) void row_stats_cast_kernel(const float* __restrict__ x, void* __restrict__ out16,
                                                            float* __restrict__ row_stats, int64_t rows, int D, float eps,
                                                            int out_dt, float x16_scale, float ratio_limit,
                                                            float abs_limit, int* __restrict__ alert) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n4 = D >> 2;
    const f32x4_t* xr = reinterpret_cast<const f32x4_t*>(x + row * D);
    f32x4_t v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + 64 * c;
        v[c] = i < n4 ? xr[i] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + 64 * c;
        if (i < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[c][e] - mean;
                ss += d * d;
            }
            u32x2_t p;
            p[0] = pack2_dt(v[c][0] * x16_scale, v[c][1] * x16_scale, out_dt);
            p[1] = pack2_dt(v[c][2] * x16_scale, v[c][3] * x16_scale, out_dt);
            reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out16) + row * D)[i] = p;
        }
    }
    const float var = wave_sum(ss) / (float)D;
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) {
        *reinterpret_cast<f32x2_t*>(row_stats + 2 * row) = (f32x2_t){mean * x16_scale, rstd / x16_scale};
        ln_alert(alert, mean, var, rstd, D, ratio_limit, abs_limit);
    }
}

// ---------------------------------------------------------------------------------------------
// valid-token counts (one wave per window, coalesced byte reads) + exclusive scan of the counts (single block)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_count_kernel(const uint8_t* __restrict__ pad, int32_t* __restrict__ cu,
                                                        int W, int L) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= W) return;
    const uint8_t* p = pad + (int64_t)w * L;
    int cnt = 0;
    for (int i = lane; i < L; i += 64) cnt += p[i] == 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) cu[w + 1] = cnt;                          // raw count; mask_scan_kernel turns it into the prefix sum
}

__global__ __launch_bounds__(1024) void mask_scan_kernel(int32_t* __restrict__ cu, int W) {
    __shared__ int part[1024];
    __shared__ int carry_s;
    const int tid = threadIdx.x;
    if (tid == 0) { carry_s = 0; cu[0] = 0; }
    __syncthreads();
    for (int base = 0; base < W; base += 1024) {
        const int w = base + tid;
        part[tid] = w < W ? cu[w + 1] : 0;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
            const int add = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        const int carry = carry_s;
        if (w < W) cu[w + 1] = carry + part[tid];
        __syncthreads();
        if (tid == 1023) carry_s = carry + part[1023];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// embedding + positional table on packed valid tokens: one block per window
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_pack_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ pad,
                                                        const int32_t* __restrict__ cu, const float* __restrict__ table,
                                                        const float* __restrict__ pos_table, float* __restrict__ out,
                                                        int L, int d, int vocab) {
    extern __shared__ int sh[];          // [L] position of the k-th valid token, [L] its token id
    int* vpos = sh;
    int* vid = sh + L;
    __shared__ int nvalid_s;
    const int w = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {                      // wave 0 compacts the valid positions in order
        int base = 0;
        for (int p0 = 0; p0 < L; p0 += 64) {
            const int p = p0 + tid;
            const bool valid = p < L && pad[(int64_t)w * L + p] == 0;
            const unsigned long long bal = __ballot(valid);
            if (valid) {
                const int rank = __popcll(bal & ((1ull << tid) - 1ull));
                long long id = ids[(int64_t)w * L + p];
                id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
                vpos[base + rank] = p;
                vid[base + rank] = (int)id;
            }
            base += __popcll(bal);
        }
        if (tid == 0) nvalid_s = base;
    }
    __syncthreads();
    const int nvalid = nvalid_s;
    const int64_t row0 = cu[w];
    const int n4 = d >> 2;
    for (int i = tid; i < nvalid * n4; i += 256) {
        const int k = i / n4, c = i - k * n4;
        f32x4_t e = reinterpret_cast<const f32x4_t*>(table + (int64_t)vid[k] * d)[c];
        if (pos_table) e += reinterpret_cast<const f32x4_t*>(pos_table + (int64_t)vpos[k] * d)[c];
        reinterpret_cast<f32x4_t*>(out + (row0 + k) * d)[c] = e;
    }
}

// Key of every packed valid token: id * key_L + position (key_L = L for a positional table, 1 without one).  The encoder input
// row x0 = table[id] + pos_table[position] takes at most vocab * L distinct values, so everything the first layer computes per
// ROW from x0 alone (LayerNorm1 -> Wqkv) is a table lookup by this key (seq2reg/model.py, _layer0_qkv_table).  One wave per window,
// same compaction order as embed_pack_kernel.
__global__ __launch_bounds__(64) void token_keys_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ pad,
                                                        const int32_t* __restrict__ cu, int64_t* __restrict__ keys, int L, int vocab,
                                                        int key_L) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int64_t row0 = cu[w];
    int base = 0;
    for (int p0 = 0; p0 < L; p0 += 64) {
        const int p = p0 + tid;
        const bool valid = p < L && pad[(int64_t)w * L + p] == 0;
        const unsigned long long bal = __ballot(valid);
        if (valid) {
            const int rank = __popcll(bal & ((1ull << tid) - 1ull));
            long long id = ids[(int64_t)w * L + p];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            keys[row0 + base + rank] = id * key_L + (key_L > 1 ? p : 0);
        }
        base += __popcll(bal);
    }
}

// The same embedding, handed to the first encoder layer as the LayerNorm-folding GEMMs exchange a stream: per valid token
// row x = table[id] + pos_table[pos] never reaches HBM as fp32 (unless `out` is given) -- the kernel writes its 16-bit
// operand copy, its scaled fp16 trunk copy and its (mean, rstd), with exactly the arithmetic of embed_pack_kernel followed
// by row_stats_cast_kernel (one wave per row, lane i owns the float4 columns i, i + 64, ...).
template <int MAXC>
__global__ __launch_bounds__(256) void embed_stream_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ pad,
                                                          const int32_t* __restrict__ cu, const float* __restrict__ table,
                                                          const float* __restrict__ pos_table, float* __restrict__ out,
                                                          void* __restrict__ out16, unsigned short* __restrict__ t16,
                                                          float* __restrict__ row_stats, int L, int d, int vocab, float eps,
                                                          int out_dt, float x16_scale, float t16_scale, float ratio_limit,
                                                          float abs_limit, int* __restrict__ alert) {
    extern __shared__ int sh[];          // [L] position of the k-th valid token, [L] its token id
    int* vpos = sh;
    int* vid = sh + L;
    __shared__ int nvalid_s;
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 64) {                      // wave 0 compacts the valid positions in order (as embed_pack_kernel)
        int base = 0;
        for (int p0 = 0; p0 < L; p0 += 64) {
            const int p = p0 + tid;
            const bool valid = p < L && pad[(int64_t)w * L + p] == 0;
            const unsigned long long bal = __ballot(valid);
            if (valid) {
                const int rank = __popcll(bal & ((1ull << tid) - 1ull));
                long long id = ids[(int64_t)w * L + p];
                id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
                vpos[base + rank] = p;
                vid[base + rank] = (int)id;
            }
            base += __popcll(bal);
        }
        if (tid == 0) nvalid_s = base;
    }
    __syncthreads();
    const int nvalid = nvalid_s;
    const int64_t row0 = cu[w];
    const int n4 = d >> 2;
    for (int k = wv; k < nvalid; k += 4) {
        const f32x4_t* er = reinterpret_cast<const f32x4_t*>(table + (int64_t)vid[k] * d);
        const f32x4_t* pr = pos_table ? reinterpret_cast<const f32x4_t*>(pos_table + (int64_t)vpos[k] * d) : nullptr;
        const int64_t row = row0 + k;
        f32x4_t v[MAXC];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int i = lane + 64 * c;
            v[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (i < n4) {
                v[c] = er[i];
                if (pr) v[c] += pr[i];
            }
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
        const float mean = wave_sum(s) / (float)d;
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int i = lane + 64 * c;
            if (i < n4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dd = v[c][e] - mean;
                    ss += dd * dd;
                }
                u32x2_t p;
                p[0] = pack2_dt(v[c][0] * x16_scale, v[c][1] * x16_scale, out_dt);
                p[1] = pack2_dt(v[c][2] * x16_scale, v[c][3] * x16_scale, out_dt);
                reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out16) + row * d)[i] = p;
                if (t16) {
                    p[0] = pack2_dt(v[c][0] * t16_scale, v[c][1] * t16_scale, VF_F16);
                    p[1] = pack2_dt(v[c][2] * t16_scale, v[c][3] * t16_scale, VF_F16);
                    reinterpret_cast<u32x2_t*>(t16 + row * d)[i] = p;
                }
                if (out) reinterpret_cast<f32x4_t*>(out + row * d)[i] = v[c];
            }
        }
        const float var = wave_sum(ss) / (float)d;
        const float rstd = rsqrtf(var + eps);
        if (lane == 0) {
            *reinterpret_cast<f32x2_t*>(row_stats + 2 * row) = (f32x2_t){mean * x16_scale, rstd / x16_scale};
            ln_alert(alert, mean, var, rstd, d, ratio_limit, abs_limit);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// segment mean (masked mean pool): one block per window, threads over float4 columns
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x, const int32_t* __restrict__ cu,
                                                          void* __restrict__ out, int d, int out_dt) {
    const int w = blockIdx.x;
    const int a = cu[w], e = cu[w + 1];
    const int n4 = d >> 2;
    const float inv = 1.0f / (float)(e - a);      // e == a -> inf; 0 * inf = NaN like the reference's 0/0
    for (int c = threadIdx.x; c < n4; c += 256) {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int t = a; t < e; ++t) acc += reinterpret_cast<const f32x4_t*>(x + (int64_t)t * d)[c];
        acc *= inv;
        if (out_dt != VF_F32) {
            u32x2_t p;
            p[0] = pack2_dt(acc[0], acc[1], out_dt);
            p[1] = pack2_dt(acc[2], acc[3], out_dt);
            reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out) + (int64_t)w * d)[c] = p;
        } else {
            reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + (int64_t)w * d)[c] = acc;
        }
    }
}

// The same pool over a 16-BIT stream (the GeGLU output / the fp16 trunk copy of the encoder's last layer: the mean is taken
// BEFORE the last down-projection, which is linear -- seq2reg/modules.py).  One block per window; a thread owns one 16-byte
// chunk (8 columns) of every RPI-th row (RPI = 256 / (d / 8) rows in flight per step), partial sums meet in LDS in a fixed order (deterministic).  Results: fp32 means
// (out_f32) and / or the means split into two 16-bit halves hi = rn16(m), lo = rn16(m - hi) stored side by side
// [W, 2 d] = [hi | lo]: a 16-bit MFMA GEMM against [W2 | W2] then carries m to ~2^-17 relative.
template <int DT>
__global__ __launch_bounds__(256) void segment_mean16_kernel(const unsigned short* __restrict__ x, int64_t ldx,
                                                            const int32_t* __restrict__ cu, float in_scale,
                                                            float* __restrict__ out_f32, unsigned short* __restrict__ out_split,
                                                            int d, int split_dt) {
    __shared__ float part[256 * 8];
    const int w = blockIdx.x, tid = threadIdx.x;
    const int a = cu[w], e = cu[w + 1];
    const int nc = d >> 3, rpi = 256 / nc;                        // chunks per row, rows per iteration
    const int c = tid % nc, rl = tid / nc;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < rpi) {
        const unsigned short* xp = x + (int64_t)a * ldx + c * 8;
        int t = a + rl;
        for (; t + 3 * rpi < e; t += 4 * rpi) {                     // four rows in flight per thread
            u32x4_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4_t*>(xp + (int64_t)(t - a + u * rpi) * ldx);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] += Op16<DT>::to_f32((unsigned short)(v[u][j] & 0xFFFFu));
                    acc[2 * j + 1] += Op16<DT>::to_f32((unsigned short)(v[u][j] >> 16));
                }
        }
        for (; t < e; t += rpi) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(xp + (int64_t)(t - a) * ldx);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] += Op16<DT>::to_f32((unsigned short)(v[j] & 0xFFFFu));
                acc[2 * j + 1] += Op16<DT>::to_f32((unsigned short)(v[j] >> 16));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[j * 256 + tid] = acc[j];
    __syncthreads();
    if (tid < nc) {
        const float inv = in_scale / (float)(e - a);               // e == a -> inf; 0 * inf = NaN like the reference's 0/0
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float sum = part[j * 256 + tid];
            for (int r = 1; r < rpi; ++r) sum += part[j * 256 + r * nc + tid];
            m[j] = sum * inv;
        }
        if (out_f32) {
            float* op = out_f32 + (int64_t)w * d + tid * 8;
            *reinterpret_cast<f32x4_t*>(op) = (f32x4_t){m[0], m[1], m[2], m[3]};
            *reinterpret_cast<f32x4_t*>(op + 4) = (f32x4_t){m[4], m[5], m[6], m[7]};
        }
        if (out_split) {
            u32x4_t hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = pack2_dt(m[2 * j], m[2 * j + 1], split_dt);
                const float h0 = split_dt == VF_F16 ? h2f((unsigned short)(hi[j] & 0xFFFFu)) : bf2f((unsigned short)(hi[j] & 0xFFFFu));
                const float h1 = split_dt == VF_F16 ? h2f((unsigned short)(hi[j] >> 16)) : bf2f((unsigned short)(hi[j] >> 16));
                lo[j] = pack2_dt(m[2 * j] - h0, m[2 * j + 1] - h1, split_dt);
            }
            unsigned short* op = out_split + (int64_t)w * 2 * d + tid * 8;
            *reinterpret_cast<u32x4_t*>(op) = hi;
            *reinterpret_cast<u32x4_t*>(op + d) = lo;
        }
    }
}

// "linear" pooling of seq2reg (reference seq2reg/model.py:268-272): out[w, c] = sum_p lin_w[p] * x[w, p, c] * valid(p)
// + lin_b, on the packed stream: the k-th valid token of window w sits at row cu[w] + k and at position vpos[k].
__global__ __launch_bounds__(256) void segment_linear_kernel(const float* __restrict__ x, const int32_t* __restrict__ cu,
                                                            const uint8_t* __restrict__ pad,
                                                            const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                            void* __restrict__ out, int L, int d, int out_dt) {
    extern __shared__ float wsel[];                  // [L] weight of the k-th valid token
    __shared__ int nvalid_s;
    const int w = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {
        int base = 0;
        for (int p0 = 0; p0 < L; p0 += 64) {
            const int p = p0 + tid;
            const bool valid = p < L && pad[(int64_t)w * L + p] == 0;
            const unsigned long long bal = __ballot(valid);
            if (valid) wsel[base + __popcll(bal & ((1ull << tid) - 1ull))] = lin_w[p];
            base += __popcll(bal);
        }
        if (tid == 0) nvalid_s = base;
    }
    __syncthreads();
    const int nvalid = nvalid_s;
    const int64_t row0 = cu[w];
    const int n4 = d >> 2;
    const float bias = lin_b ? lin_b[0] : 0.f;
    for (int c = tid; c < n4; c += 256) {
        f32x4_t acc = (f32x4_t){bias, bias, bias, bias};
        for (int k = 0; k < nvalid; ++k) acc += wsel[k] * reinterpret_cast<const f32x4_t*>(x + (row0 + k) * d)[c];
        if (out_dt != VF_F32) {
            u32x2_t p;
            p[0] = pack2_dt(acc[0], acc[1], out_dt);
            p[1] = pack2_dt(acc[2], acc[3], out_dt);
            reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out) + (int64_t)w * d)[c] = p;
        } else {
            reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + (int64_t)w * d)[c] = acc;
        }
    }
}

// out[i, :] = src[idx[i], :] * (scale ? scale[i] : 1) + (shift ? shift[i] : 0)   (fp32 rows; the per-token context rows of
// a use_context seq2reg tokenizer: embedding row of the window's label, optionally expanded per position)
__global__ __launch_bounds__(256) void affine_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ out, int64_t n, int d) {
    const int n4 = d >> 2;
    const int64_t total = n * n4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / n4;
        const int c = (int)(i - row * n4);
        const float a = scale ? scale[row] : 1.f, b = shift ? shift[row] : 0.f;
        reinterpret_cast<f32x4_t*>(out + row * d)[c] = reinterpret_cast<const f32x4_t*>(src + idx[row] * d)[c] * a + b;
    }
}

// segment max (max pool over the tokens of a sequence): one block per window, threads over float4 columns
__global__ __launch_bounds__(256) void segment_max_kernel(const float* __restrict__ x, const int32_t* __restrict__ cu,
                                                         float* __restrict__ out, int d) {
    const int w = blockIdx.x;
    const int a = cu[w], e = cu[w + 1];
    const int n4 = d >> 2;
    for (int c = threadIdx.x; c < n4; c += 256) {
        f32x4_t acc = (f32x4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};     // empty window -> -inf, as torch.max
        for (int t = a; t < e; ++t) {
            const f32x4_t v = reinterpret_cast<const f32x4_t*>(x + (int64_t)t * d)[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], v[k]);
        }
        reinterpret_cast<f32x4_t*>(out + (int64_t)w * d)[c] = acc;
    }
}

// out[i] = a[ia ? ia[i] : i] + b[ib ? ib[i] : i]   (fp32 rows; gene residual, tissue embedding onto CRE tokens)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ a, const int64_t* __restrict__ ia,
                                                      const float* __restrict__ b, const int64_t* __restrict__ ib,
                                                      float* __restrict__ out, int64_t n, int d) {
    const int n4 = d >> 2;
    const int64_t total = n * n4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / n4;
        const int c = (int)(i - row * n4);
        const int64_t ra = ia ? ia[row] : row, rb = ib ? ib[row] : row;
        reinterpret_cast<f32x4_t*>(out + row * d)[c] =
            reinterpret_cast<const f32x4_t*>(a + ra * d)[c] + reinterpret_cast<const f32x4_t*>(b + rb * d)[c];
    }
}

// ---------------------------------------------------------------------------------------------
// gathers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const int64_t* __restrict__ idx, void* __restrict__ out,
                                                             int64_t n, int d, int out_dt) {
    const int n4 = d >> 2;
    const int64_t total = n * n4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / n4;
        const int c = (int)(i - row * n4);
        const int64_t j = idx[row];
        const float* src = j >= 0 ? a + j * d : b + (-j - 1) * d;
        const f32x4_t v = reinterpret_cast<const f32x4_t*>(src)[c];
        if (out_dt != VF_F32) {
            u32x2_t p;
            p[0] = pack2_dt(v[0], v[1], out_dt);
            p[1] = pack2_dt(v[2], v[3], out_dt);
            reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out) + row * d)[c] = p;
        } else {
            reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + row * d)[c] = v;
        }
    }
}

// narrow fp32 rows (d not a multiple of 4, e.g. the [rows, 2] LayerNorm statistics): one element per thread
__global__ __launch_bounds__(256) void gather_rows_f32_scalar_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                    const int64_t* __restrict__ idx, float* __restrict__ out,
                                                                    int64_t n, int d) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / d;
        const int c = (int)(i - row * d);
        const int64_t j = idx[row];
        out[i] = j >= 0 ? a[j * d + c] : b[(-j - 1) * d + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const unsigned short* __restrict__ src, int64_t ld_src,
                                                              const int64_t* __restrict__ idx,
                                                              unsigned short* __restrict__ out, int64_t ld_out,
                                                              int64_t n, int d) {
    const int n8 = d >> 3;
    const int64_t total = n * n8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / n8;
        const int c = (int)(i - row * n8);
        reinterpret_cast<u32x4_t*>(out + row * ld_out)[c] = reinterpret_cast<const u32x4_t*>(src + idx[row] * ld_src)[c];
    }
}

// ---------------------------------------------------------------------------------------------
// head tail: out[i] = softplus(x[i,:] . w + b)   (one wave per row)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_softplus_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ out,
                                                             int64_t n, int d, int softplus) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int n4 = d >> 2;
    float s = 0.f;
    for (int c = lane; c < n4; c += 64) {
        const f32x4_t xv = reinterpret_cast<const f32x4_t*>(x + row * d)[c];
        const f32x4_t wv = reinterpret_cast<const f32x4_t*>(w)[c];
        s += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
    }
    s = wave_sum(s);
    if (lane == 0) {
        float y = s + (b ? b[0] : 0.f);
        if (softplus) y = y > 20.f ? y : log1pf(expf(y));
        out[row] = y;
    }
}

__global__ __launch_bounds__(256) void cast_f32_16_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                         int64_t n, int out_dt) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4_t v = reinterpret_cast<const f32x4_t*>(x)[i];
        u32x2_t p;
        p[0] = pack2_dt(v[0], v[1], out_dt);
        p[1] = pack2_dt(v[2], v[3], out_dt);
        reinterpret_cast<u32x2_t*>(out)[i] = p;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
        out[(n4 << 2) + threadIdx.x] = (unsigned short)(pack2_dt(x[(n4 << 2) + threadIdx.x], 0.f, out_dt) & 0xFFFFu);
}

inline int stream_grid(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;             // 256 CUs x 8 blocks, grid-stride beyond
    return (int)b;
}

}  // namespace

extern "C" int vf_layernorm(const float* x, const float* gamma, const float* beta, void* out, int64_t rows, int D,
                            float eps, int out_dtype, int gelu, void* stream) {
    VF_REQUIRE(x && gamma && beta && out, "vf_layernorm: null pointer");
    VF_REQUIRE(D > 0 && D % 4 == 0 && D <= 8192, "vf_layernorm: D=%d must be a multiple of 4 and <= 8192", D);
    VF_REQUIRE(out_dtype == VF_F32 || out_dtype == VF_BF16 || out_dtype == VF_F16, "vf_layernorm: bad out_dtype %d", out_dtype);
    if (rows <= 0) return VF_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4));
    const int bf = out_dtype;
    if (D <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, st, x, gamma, beta, out, rows, D, eps, bf, gelu);
    else if (D <= 2048) hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, st, x, gamma, beta, out, rows, D, eps, bf, gelu);
    else hipLaunchKernelGGL(layernorm_kernel<32>, grid, dim3(256), 0, st, x, gamma, beta, out, rows, D, eps, bf, gelu);
    VF_CHECK_LAUNCH("vf_layernorm");
    return VF_OK;
}

extern "C" int vf_ln_finalize2(const float* part_stats, int64_t rows, int n_parts, int D, float eps, float x16_scale,
                               float ratio_limit, float abs_limit, int* alert, float* row_stats, void* stream) {
    VF_REQUIRE(part_stats && row_stats && n_parts > 0 && D > 0 && x16_scale > 0.f, "vf_ln_finalize: bad arguments");
    // the merge assumes whole 32-column parts carrying (sum, M2 about the part mean) -- what the ABI >= 5 producers write
    VF_REQUIRE(D == 32 * n_parts, "vf_ln_finalize: D=%d must be 32 * n_parts (n_parts=%d)", D, n_parts);
    if (rows <= 0) return VF_OK;
    const dim3 grid((unsigned)((rows + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (n_parts == 48)            // D = 1536: the modulator streams
        hipLaunchKernelGGL(ln_finalize_kernel<48>, grid, dim3(256), 0, st, part_stats, row_stats, rows, n_parts, D, eps, x16_scale, ratio_limit, abs_limit, alert);
    else if (n_parts == 16)       // D = 512: seq2reg
        hipLaunchKernelGGL(ln_finalize_kernel<16>, grid, dim3(256), 0, st, part_stats, row_stats, rows, n_parts, D, eps, x16_scale, ratio_limit, abs_limit, alert);
    else
        hipLaunchKernelGGL(ln_finalize_kernel<0>, grid, dim3(256), 0, st, part_stats, row_stats, rows, n_parts, D, eps, x16_scale, ratio_limit, abs_limit, alert);
    VF_CHECK_LAUNCH("vf_ln_finalize");
    return VF_OK;
}

extern "C" int vf_ln_finalize(const float* part_stats, int64_t rows, int n_parts, int D, float eps, float* row_stats,
                              void* stream) {
    return vf_ln_finalize2(part_stats, rows, n_parts, D, eps, 1.0f, 0.f, 0.f, nullptr, row_stats, stream);
}

extern "C" int vf_row_stats_cast2(const float* x, int64_t rows, int D, float eps, void* out16, int out_dtype, float x16_scale,
                                  float ratio_limit, float abs_limit, int* alert, float* row_stats, void* stream) {
    VF_REQUIRE(x && out16 && row_stats, "vf_row_stats_cast: null pointer");
    VF_REQUIRE(D > 0 && D % 4 == 0 && D <= 8192, "vf_row_stats_cast: D=%d must be a multiple of 4 and <= 8192", D);
    VF_REQUIRE(out_dtype == VF_BF16 || out_dtype == VF_F16, "vf_row_stats_cast: bad out_dtype %d", out_dtype);
    VF_REQUIRE(x16_scale > 0.f, "vf_row_stats_cast: x16_scale must be positive");
    if (rows <= 0) return VF_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4));
    if (D <= 512) hipLaunchKernelGGL(row_stats_cast_kernel<2>, grid, dim3(256), 0, st, x, out16, row_stats, rows, D, eps, out_dtype, x16_scale, ratio_limit, abs_limit, alert);
    else if (D <= 2048) hipLaunchKernelGGL(row_stats_cast_kernel<8>, grid, dim3(256), 0, st, x, out16, row_stats, rows, D, eps, out_dtype, x16_scale, ratio_limit, abs_limit, alert);
    else hipLaunchKernelGGL(row_stats_cast_kernel<32>, grid, dim3(256), 0, st, x, out16, row_stats, rows, D, eps, out_dtype, x16_scale, ratio_limit, abs_limit, alert);
    VF_CHECK_LAUNCH("vf_row_stats_cast");
    return VF_OK;
}

extern "C" int vf_row_stats_cast(const float* x, int64_t rows, int D, float eps, void* out16, int out_dtype,
                                 float* row_stats, void* stream) {
    return vf_row_stats_cast2(x, rows, D, eps, out16, out_dtype, 1.0f, 0.f, 0.f, nullptr, row_stats, stream);
}

extern "C" int vf_mask_to_cu_seqlens(const uint8_t* pad, int32_t* cu, int W, int L, void* stream) {
    VF_REQUIRE(pad && cu && W >= 0 && L > 0, "vf_mask_to_cu_seqlens: bad arguments");
    if (W > 0) hipLaunchKernelGGL(mask_count_kernel, dim3((W + 3) / 4), dim3(256), 0, (hipStream_t)stream, pad, cu, W, L);
    hipLaunchKernelGGL(mask_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cu, W);
    VF_CHECK_LAUNCH("vf_mask_to_cu_seqlens");
    return VF_OK;
}

extern "C" int vf_embed_pack(const int64_t* ids, const uint8_t* pad, const int32_t* cu, const float* table,
                             const float* pos_table, float* out, int W, int L, int d, int vocab, void* stream) {
    VF_REQUIRE(ids && pad && cu && table && out, "vf_embed_pack: null pointer");
    VF_REQUIRE(L > 0 && L <= 4096 && d > 0 && d % 4 == 0 && vocab > 0, "vf_embed_pack: bad shape L=%d d=%d vocab=%d", L, d, vocab);
    if (W <= 0) return VF_OK;
    hipLaunchKernelGGL(embed_pack_kernel, dim3(W), dim3(256), 2 * L * sizeof(int), (hipStream_t)stream, ids, pad, cu,
                       table, pos_table, out, L, d, vocab);
    VF_CHECK_LAUNCH("vf_embed_pack");
    return VF_OK;
}

extern "C" int vf_embed_stream(const int64_t* ids, const uint8_t* pad, const int32_t* cu, const float* table,
                               const float* pos_table, float* out, void* out16, int out_dtype, float x16_scale, void* t16,
                               float t16_scale, float* row_stats, float eps, float ratio_limit, float abs_limit, int* alert,
                               int W, int L, int d, int vocab, void* stream) {
    VF_REQUIRE(ids && pad && cu && table && out16 && row_stats, "vf_embed_stream: null pointer");
    VF_REQUIRE(L > 0 && L <= 4096 && d > 0 && d % 4 == 0 && d <= 2048 && vocab > 0, "vf_embed_stream: bad shape L=%d d=%d vocab=%d", L, d, vocab);
    VF_REQUIRE(out_dtype == VF_BF16 || out_dtype == VF_F16, "vf_embed_stream: bad out_dtype %d", out_dtype);
    VF_REQUIRE(x16_scale > 0.f && t16_scale > 0.f, "vf_embed_stream: scales must be positive");
    if (W <= 0) return VF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (d <= 512)
        hipLaunchKernelGGL(embed_stream_kernel<2>, dim3(W), dim3(256), 2 * L * sizeof(int), st, ids, pad, cu, table, pos_table, out,
                           out16, (unsigned short*)t16, row_stats, L, d, vocab, eps, out_dtype, x16_scale, t16_scale, ratio_limit, abs_limit, alert);
    else
        hipLaunchKernelGGL(embed_stream_kernel<8>, dim3(W), dim3(256), 2 * L * sizeof(int), st, ids, pad, cu, table, pos_table, out,
                           out16, (unsigned short*)t16, row_stats, L, d, vocab, eps, out_dtype, x16_scale, t16_scale, ratio_limit, abs_limit, alert);
    VF_CHECK_LAUNCH("vf_embed_stream");
    return VF_OK;
}

extern "C" int vf_segment_mean(const float* x, const int32_t* cu, void* out, int W, int d, int out_dtype, void* stream) {
    VF_REQUIRE(x && cu && out && d > 0 && d % 4 == 0, "vf_segment_mean: bad arguments (d=%d)", d);
    VF_REQUIRE(out_dtype == VF_F32 || out_dtype == VF_BF16 || out_dtype == VF_F16, "vf_segment_mean: bad out_dtype %d", out_dtype);
    if (W <= 0) return VF_OK;
    hipLaunchKernelGGL(segment_mean_kernel, dim3(W), dim3(256), 0, (hipStream_t)stream, x, cu, out, d, out_dtype);
    VF_CHECK_LAUNCH("vf_segment_mean");
    return VF_OK;
}

extern "C" int vf_token_keys(const int64_t* ids, const uint8_t* pad, const int32_t* cu, int64_t* keys, int W, int L, int vocab,
                             int key_L, void* stream) {
    VF_REQUIRE(ids && pad && cu && keys && L > 0 && vocab > 0 && (key_L == 1 || key_L >= L), "vf_token_keys: bad arguments (L=%d key_L=%d)", L, key_L);
    if (W <= 0) return VF_OK;
    hipLaunchKernelGGL(token_keys_kernel, dim3(W), dim3(64), 0, (hipStream_t)stream, ids, pad, cu, keys, L, vocab, key_L);
    VF_CHECK_LAUNCH("vf_token_keys");
    return VF_OK;
}

extern "C" int vf_segment_mean16(const void* x, int64_t ldx, int dtype, const int32_t* cu, float in_scale, float* out_f32,
                                 void* out_split, int W, int d, void* stream) {
    VF_REQUIRE(x && cu && (out_f32 || out_split) && d >= 8 && d <= 2048 && d % 8 == 0 && ldx >= d && ldx % 8 == 0,
               "vf_segment_mean16: bad arguments (d=%d ldx=%ld; d must be a multiple of 8, at most 2048)", d, (long)ldx);
    VF_REQUIRE(dtype == VF_BF16 || dtype == VF_F16, "vf_segment_mean16: bad dtype %d", dtype);
    if (W <= 0) return VF_OK;
    if (dtype == VF_F16)
        hipLaunchKernelGGL(segment_mean16_kernel<VF_F16>, dim3(W), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ldx,
                           cu, in_scale, out_f32, (unsigned short*)out_split, d, dtype);
    else
        hipLaunchKernelGGL(segment_mean16_kernel<VF_BF16>, dim3(W), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ldx,
                           cu, in_scale, out_f32, (unsigned short*)out_split, d, dtype);
    VF_CHECK_LAUNCH("vf_segment_mean16");
    return VF_OK;
}

extern "C" int vf_segment_linear(const float* x, const int32_t* cu, const uint8_t* pad, const float* lin_w,
                                 const float* lin_b, void* out, int W, int L, int d, int out_dtype, void* stream) {
    VF_REQUIRE(x && cu && pad && lin_w && out && d > 0 && d % 4 == 0 && L > 0 && L <= 8192,
               "vf_segment_linear: bad arguments (L=%d d=%d)", L, d);
    VF_REQUIRE(out_dtype == VF_F32 || out_dtype == VF_BF16 || out_dtype == VF_F16, "vf_segment_linear: bad out_dtype %d", out_dtype);
    if (W <= 0) return VF_OK;
    hipLaunchKernelGGL(segment_linear_kernel, dim3(W), dim3(256), L * sizeof(float), (hipStream_t)stream, x, cu, pad, lin_w,
                       lin_b, out, L, d, out_dtype);
    VF_CHECK_LAUNCH("vf_segment_linear");
    return VF_OK;
}

extern "C" int vf_affine_rows_f32(const float* src, const int64_t* idx, const float* scale, const float* shift, float* out,
                                  int64_t n, int d, void* stream) {
    VF_REQUIRE(src && idx && out && d > 0 && d % 4 == 0, "vf_affine_rows_f32: bad arguments (d=%d)", d);
    if (n <= 0) return VF_OK;
    hipLaunchKernelGGL(affine_rows_kernel, dim3(stream_grid(n * (d / 4))), dim3(256), 0, (hipStream_t)stream, src, idx,
                       scale, shift, out, n, d);
    VF_CHECK_LAUNCH("vf_affine_rows_f32");
    return VF_OK;
}

extern "C" int vf_segment_max(const float* x, const int32_t* cu, float* out, int W, int d, void* stream) {
    VF_REQUIRE(x && cu && out && d > 0 && d % 4 == 0, "vf_segment_max: bad arguments (d=%d)", d);
    if (W <= 0) return VF_OK;
    hipLaunchKernelGGL(segment_max_kernel, dim3(W), dim3(256), 0, (hipStream_t)stream, x, cu, out, d);
    VF_CHECK_LAUNCH("vf_segment_max");
    return VF_OK;
}

extern "C" int vf_add_rows_f32(const float* a, const int64_t* idx_a, const float* b, const int64_t* idx_b, float* out,
                               int64_t n, int d, void* stream) {
    VF_REQUIRE(a && b && out && d > 0 && d % 4 == 0, "vf_add_rows_f32: bad arguments (d=%d)", d);
    if (n <= 0) return VF_OK;
    hipLaunchKernelGGL(add_rows_kernel, dim3(stream_grid(n * (d / 4))), dim3(256), 0, (hipStream_t)stream, a, idx_a, b,
                       idx_b, out, n, d);
    VF_CHECK_LAUNCH("vf_add_rows_f32");
    return VF_OK;
}

extern "C" int vf_gather_rows_f32(const float* a, const float* b, const int64_t* idx, void* out, int64_t n, int d,
                                  int out_dtype, void* stream) {
    VF_REQUIRE(a && idx && out && d > 0 && (d % 4 == 0 || out_dtype == VF_F32),
               "vf_gather_rows_f32: 16-bit outputs need d %% 4 == 0 (d=%d)", d);
    VF_REQUIRE(out_dtype == VF_F32 || out_dtype == VF_BF16 || out_dtype == VF_F16, "vf_gather_rows_f32: bad out_dtype %d", out_dtype);
    if (n <= 0) return VF_OK;
    if (d % 4 != 0) {
        hipLaunchKernelGGL(gather_rows_f32_scalar_kernel, dim3(stream_grid(n * d)), dim3(256), 0, (hipStream_t)stream, a,
                           b ? b : a, idx, (float*)out, n, d);
        VF_CHECK_LAUNCH("vf_gather_rows_f32");
        return VF_OK;
    }
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(stream_grid(n * (d / 4))), dim3(256), 0, (hipStream_t)stream, a,
                       b ? b : a, idx, out, n, d, out_dtype);
    VF_CHECK_LAUNCH("vf_gather_rows_f32");
    return VF_OK;
}

extern "C" int vf_gather_rows_bf16(const void* src, int64_t ld_src, const int64_t* idx, void* out, int64_t ld_out,
                                   int64_t n, int d, void* stream) {
    VF_REQUIRE(src && idx && out && d > 0 && d % 8 == 0 && ld_src % 8 == 0 && ld_out % 8 == 0,
               "vf_gather_rows_bf16: d and strides must be multiples of 8 (d=%d)", d);
    if (n <= 0) return VF_OK;
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(stream_grid(n * (d / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, ld_src, idx, (unsigned short*)out, ld_out, n, d);
    VF_CHECK_LAUNCH("vf_gather_rows_bf16");
    return VF_OK;
}

extern "C" int vf_rowdot_softplus(const float* x, const float* w, const float* b, float* out, int64_t n, int d,
                                  int softplus, void* stream) {
    VF_REQUIRE(x && w && out && d > 0 && d % 4 == 0, "vf_rowdot_softplus: bad arguments (d=%d)", d);
    if (n <= 0) return VF_OK;
    hipLaunchKernelGGL(rowdot_softplus_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, w, b,
                       out, n, d, softplus);
    VF_CHECK_LAUNCH("vf_rowdot_softplus");
    return VF_OK;
}

static int cast_f32_16(const float* x, void* out, int64_t n, int dt, void* stream) {
    VF_REQUIRE(x && out && n >= 0, "vf_cast_f32_*: bad arguments");
    VF_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 8 == 0), "vf_cast_f32_*: misaligned pointer");
    if (n == 0) return VF_OK;
    hipLaunchKernelGGL(cast_f32_16_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)out, n, dt);
    VF_CHECK_LAUNCH("vf_cast_f32_*");
    return VF_OK;
}

extern "C" int vf_cast_f32_bf16(const float* x, void* out, int64_t n, void* stream) { return cast_f32_16(x, out, n, VF_BF16, stream); }
extern "C" int vf_cast_f32_f16(const float* x, void* out, int64_t n, void* stream) { return cast_f32_16(x, out, n, VF_F16, stream); }
