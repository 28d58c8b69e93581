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
#include <type_traits>
#include "vf_common.h"

#ifndef VF_G8X_WIDE
#define VF_G8X_WIDE 1            // persistent kernel, 16-bit-residual producers: 8-column read-back (0: the 4-column form; A/B builds)
#endif
#ifndef VF_G8_RES_ALL
#define VF_G8_RES_ALL 0          // gemm8_kernel, 16-bit residual: 1 = request every epilogue pass's residual rows up front.
                                 // Measured (profiles/r03_i_producer_epilogue_experiments.log): 417 -> 422 ... 435 us on the gene
                                 // out-projection -- the epilogue is not waiting for its residual; left off.
#endif

namespace {

// Bank-conflict swizzle of a [rows][BK] bf16 tile read with ds_read_b128 by lane (r = row & 15, g):
// the 16-byte chunk c of row `row` is stored at chunk c ^ swz(row).
//   BK = 64 (128-B rows, 8 chunks): swz = (row >> 1) & 7
//   BK = 32 ( 64-B rows, 4 chunks): swz = {0,2,3,1}[(row >> 2) & 3]
// Both make every 16-lane ds_read_b128 group hit 16 distinct 16-byte bank slots (checked by hand against the
// lane groups of MI355X_MICROARCH.md §LDS and by SQ_LDS_BANK_CONFLICT = 0).
template <int BK>
__device__ __forceinline__ int swz(int row) {
    if (BK == 64) return (row >> 1) & 7;
    return (0x78 >> (((row >> 2) & 3) * 2)) & 3;     // 0b01'11'10'00 -> 0,2,3,1
}

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- LayerNorm folded into the GEMMs around it ("LayerNorm without a pass") -------------------
// LN(x) . W^T = rstd * (x . (gamma (.) W)^T - mean * rowsum(gamma (.) W)) + (W . beta + b): the PRODUCER of x (a GEMM with an
// fp32 epilogue) also writes the 16-bit copy of x and, per row and 32-column part, (sum, M2 about the part mean); a tiny kernel
// turns the parts into (mean, rstd) per row; the CONSUMER runs on the 16-bit x with gamma folded into its weights and
// applies the row statistics in its epilogue.  No kernel reads x again just to normalise it.
struct LnArgs {
    const float* row_stats;   // consumer: [M][2] = (mean, rstd) of the rows of A's fp32 source
    const float* colsum;      // consumer: [N]    = sum_k W'[n][k] over the 16-bit folded weights, fp32
    void* out16;              // producer: [M][ld16] 16-bit copy of the fp32 output
    float* part_stats;        // producer: [n_parts][M][2] = (sum, second moment about the part mean) of each 32-column part of every row
                              // (part-major: a wave writes runs of consecutive rows of one part, vf_ln_finalize reads
                              // consecutive rows per thread)
    int64_t ld16;
    int64_t rows;             // M
    float x16_scale;          // producer: out16 = 16-bit(x * x16_scale) (a power of two; 1 for bf16, whose exponent range is
                              // fp32's; fp16 streams are stored scaled so that the raw residual cannot leave the fp16 range:
                              // LayerNorm is scale-invariant and vf_ln_finalize writes (mean * c, rstd / c) for the consumer)
    float res16_scale;        // VF_LN_PRODUCER_R16: residual = float(res16) * res16_scale (= 1 / x16_scale of that stream)
    const unsigned short* res16;   // VF_LN_PRODUCER_R16: the residual as the 16-bit copy of a stream [M][ldr16] instead of
    int64_t ldr16;                 // its fp32 rows (a stream that is otherwise only read through LayerNorm -> Linear)
    unsigned short* t16_out;       // VF_LN_PRODUCER_T16: fp16(x * t16_scale) [M][ldt16], the trunk copy the NEXT layer's
    int64_t ldt16;                 // down-projection reads as its residual (may be null: last layer of a stack)
    float t16_scale;
};
// VF_LN_PRODUCER_R16 = producer whose residual is read from a 16-bit stream copy (EPI is VF_EPI_RES_F32; `res` unused).
// VF_LN_PRODUCER_T16 = the same with the residual in FP16 WHATEVER THE OPERAND TYPE (value = float(half) * res16_scale): the
// layer trunk (x_out = W2 . h + x_in) travels between the layers of a stack as a scaled fp16 copy -- 11 significant bits
// against bf16's 8, so the per-layer rounding of the trunk stays below the operand roundings -- and
// the producer writes that copy of its own output (t16_out) beside the operand-type copy and the statistics.
enum { VF_LN_NONE = 0, VF_LN_CONSUMER = 1, VF_LN_PRODUCER = 2, VF_LN_PRODUCER_R16 = 3, VF_LN_PRODUCER_T16 = 4 };
constexpr bool ln_is_producer(int ln) { return ln == VF_LN_PRODUCER || ln == VF_LN_PRODUCER_R16 || ln == VF_LN_PRODUCER_T16; }
constexpr bool ln_res_is_16(int ln) { return ln == VF_LN_PRODUCER_R16 || ln == VF_LN_PRODUCER_T16; }

// 4 consecutive 16-bit values (8 bytes) -> fp32
template <int DT>
__device__ __forceinline__ f32x4_t cvt4_16(u32x2_t v) {
    if (DT == VF_BF16)
        return (f32x4_t){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xFFFF0000u), __uint_as_float(v[1] << 16),
                         __uint_as_float(v[1] & 0xFFFF0000u)};
    return (f32x4_t){Op16<DT>::to_f32((unsigned short)(v[0] & 0xFFFFu)), Op16<DT>::to_f32((unsigned short)(v[0] >> 16)),
                     Op16<DT>::to_f32((unsigned short)(v[1] & 0xFFFFu)), Op16<DT>::to_f32((unsigned short)(v[1] >> 16))};
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// producer side of one read-back item: lane holds the final fp32 values f of one row, 4 consecutive columns (a
// 32-column part of a row = 8 consecutive lanes); valid = row and columns inside the matrix; p16 / ppart = where this
// lane's 16-bit values and its part's (sum, M2 about the part mean) go
template <int DT>
__device__ __forceinline__ void ln_emit(f32x4_t f, bool valid, unsigned short* p16, float* ppart, int lane,
                                        float x16_scale = 1.0f, bool st16 = true, bool stpart = true,
                                        unsigned short* pt16 = nullptr, float t16_scale = 1.0f) {
    // no masking of the sums: N % 32 == 0 (checked at launch), so the 8 lanes of a part are all inside the matrix or
    // all outside, and a row past M only ever feeds its own (never stored) part
    float s1 = (f[0] + f[1]) + (f[2] + f[3]);
    // butterfly over the 8 lanes of the part on the DPP path of the VALU (__shfl_xor would be a ds_bpermute round trip
    // per step): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7 - i: the other quad's sum);
    // every lane of the part ends with the part's sum
    s1 += dpp_f32<0xB1>(s1);
    s1 += dpp_f32<0x4E>(s1);
    s1 += dpp_f32<0x141>(s1);
    // second moment ABOUT THE PART'S MEAN (not sum of squares): vf_ln_finalize merges the parts with the parallel-variance
    // formula, so the row variance never is a difference of two large numbers, whatever the row mean (round-2 advice)
    const float mp = s1 * (1.0f / 32.0f);
    const float d0 = f[0] - mp, d1 = f[1] - mp, d2 = f[2] - mp, d3 = f[3] - mp;
    float s2 = __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, d0 * d0)));
    s2 += dpp_f32<0xB1>(s2);
    s2 += dpp_f32<0x4E>(s2);
    s2 += dpp_f32<0x141>(s2);
    if (valid) {
        u32x2_t pk;
        if (pt16) {                                       // VF_LN_PRODUCER_T16: the fp16 trunk copy (null pointer constant otherwise)
            const f32x4_t t = f * t16_scale;
            pk[0] = pack2h(t[0], t[1]);
            pk[1] = pack2h(t[2], t[3]);
            *reinterpret_cast<u32x2_t*>(pt16) = pk;
        }
        if (DT == VF_F16) f *= x16_scale;                 // statistics above are those of the UNSCALED row
        pk[0] = Op16<DT>::pack2(f[0], f[1]);
        pk[1] = Op16<DT>::pack2(f[2], f[3]);
        if (st16) *reinterpret_cast<u32x2_t*>(p16) = pk;
        if (stpart && (lane & 7) == 0) *reinterpret_cast<f32x2_t*>(ppart) = (f32x2_t){s1, s2};
    }
}

// ln_emit for a lane that holds EIGHT consecutive columns of a row (fa = columns 0-3, fb = columns 4-7; a 32-column part = 4
// consecutive lanes): one 16-byte store per 16-bit copy instead of two 8-byte ones.  The sums run through the SAME balanced
// tree as ln_emit's (4 values in a lane, then lane pairs, quads, the two quads of a part): fa / fb are what two neighbouring
// lanes hold there and IEEE addition commutes, so (sum, M2) are bit-identical to the 4-column form.
template <int DT, int SKIP = 0>      // SKIP (probe builds): 1 = no statistics store, 2 = no 16-bit store, 16 = no statistics arithmetic
__device__ __forceinline__ void ln_emit8(f32x4_t fa, f32x4_t fb, bool valid, unsigned short* p16, float* ppart, int lane,
                                         float x16_scale, unsigned short* pt16, float t16_scale) {
    float s1 = ((fa[0] + fa[1]) + (fa[2] + fa[3])) + ((fb[0] + fb[1]) + (fb[2] + fb[3]));
    float s2 = s1;
    if (!(SKIP & 16)) {
    s1 += dpp_f32<0xB1>(s1);                 // lane j <-> j ^ 1: ln_emit's quad_perm [2,3,0,1] step
    s1 += dpp_f32<0x4E>(s1);                 // lane j <-> j ^ 2: ln_emit's row_half_mirror step (the part's two halves)
    const float mp = s1 * (1.0f / 32.0f);
    const float a0 = fa[0] - mp, a1 = fa[1] - mp, a2 = fa[2] - mp, a3 = fa[3] - mp;
    const float b0 = fb[0] - mp, b1 = fb[1] - mp, b2 = fb[2] - mp, b3 = fb[3] - mp;
    s2 = __builtin_fmaf(a3, a3, __builtin_fmaf(a2, a2, __builtin_fmaf(a1, a1, a0 * a0))) +
         __builtin_fmaf(b3, b3, __builtin_fmaf(b2, b2, __builtin_fmaf(b1, b1, b0 * b0)));
    s2 += dpp_f32<0xB1>(s2);
    s2 += dpp_f32<0x4E>(s2);
    }
    if (valid) {
        u32x4_t pk;
        if (pt16) {                                       // the fp16 trunk copy (VF_LN_PRODUCER_T16)
            const f32x4_t ta = fa * t16_scale, tb = fb * t16_scale;
            pk = (u32x4_t){pack2h(ta[0], ta[1]), pack2h(ta[2], ta[3]), pack2h(tb[0], tb[1]), pack2h(tb[2], tb[3])};
            *reinterpret_cast<u32x4_t*>(pt16) = pk;
        }
        if (DT == VF_F16) { fa *= x16_scale; fb *= x16_scale; }           // statistics above are those of the UNSCALED row
        pk = (u32x4_t){Op16<DT>::pack2(fa[0], fa[1]), Op16<DT>::pack2(fa[2], fa[3]), Op16<DT>::pack2(fb[0], fb[1]),
                       Op16<DT>::pack2(fb[2], fb[3])};
        if (!(SKIP & 2)) *reinterpret_cast<u32x4_t*>(p16) = pk;
        else asm volatile("" :: "v"(pk));
        if (!(SKIP & 1) && (lane & 3) == 0) *reinterpret_cast<f32x2_t*>(ppart) = (f32x2_t){s1, s2};
        else asm volatile("" :: "v"(s1), "v"(s2));
    }
}

// Epilogue of one wave's 128 (m) x 64 (n) accumulator block for the LayerNorm producers whose residual is a 16-bit stream
// copy (VF_LN_PRODUCER_R16 / _T16), "wide" read-back (round 4): the block is staged as fp32 through `region` in passes of 32
// rows exactly as before, but a lane reads back EIGHT consecutive columns (two ds_read_b128) of a row instead of four, so the
// residual load, the 16-bit copy, the fp16 trunk copy and the part statistics are 16-byte-per-lane operations on 8 rows per
// wave-instruction: half the global memory instructions (and pointer steps) for the same bytes.  The producer epilogue was
// 19-21 k cycles per tile against ~6 k of vector issue (scripts/probes/gemm8x_probe.hip): 288 global memory instructions of
// 512 bytes per wave through the CU's one address / store path.  Same arithmetic, same summation tree: bit-identical.
template <int DT, int LN, int REGION>
__device__ __forceinline__ void producer16_epilogue_wide(f32x4_t (&acc)[4][8], char* region, const char* side, int side_n,
                                                         int64_t mw0, int nw0, bool has_bias, void* out, int64_t ldo, int M,
                                                         int N, const LnArgs& ln) {
    constexpr int TM = 8, TN = 4, WT_M = 128, PITCH = 64 * 4 + 16, RP = 32, IMP = RP / 16, NPASS = TM / IMP;
    constexpr int RI = 8, NI = RP / RI;                                 // 8 lanes x 32 B per row, 8 rows per wave-instruction
    constexpr bool T16 = LN == VF_LN_PRODUCER_T16;
    static_assert(REGION >= RP * PITCH, "wide producer epilogue: staging region too small");
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int lane = lane_e, r = lane & 15, g = lane >> 4;
    const int ep_row = lane >> 3, ep_col = nw0 + (lane & 7) * 8;
    const int rows_left = (int)(M - mw0) - ep_row;                      // item j is a row of the matrix iff j * RI < rows_left
    const int64_t row0 = mw0 + ep_row;
    const int colc = ep_col < N ? ep_col : N - 8;
    const char* res_run = reinterpret_cast<const char*>(ln.res16) + (row0 * ln.ldr16 + colc) * 2;
    const char* const res_last = reinterpret_cast<const char*>(ln.res16) + ((int64_t)(M - 1) * ln.ldr16 + colc) * 2;
    const int64_t res_step = (int64_t)RI * ln.ldr16 * 2;
    char* out_run = reinterpret_cast<char*>(out) + (row0 * ldo + ep_col) * 4;
    const int64_t out_step = (int64_t)RI * ldo * 4;
    unsigned short* o16_run = reinterpret_cast<unsigned short*>(ln.out16) + row0 * ln.ld16 + ep_col;
    const int64_t o16_step = (int64_t)RI * ln.ld16;
    float* part_run = ln.part_stats + ((int64_t)(ep_col >> 5) * ln.rows + row0) * 2;
    unsigned short* t16_run = T16 ? ln.t16_out + row0 * ln.ldt16 + ep_col : nullptr;
    const int64_t t16_step = (int64_t)RI * ln.ldt16;
    u32x4_t rbuf[2][NI];
    auto load_res_pass = [&](int ps, u32x4_t (&dst)[NI]) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int j = ps * NI + k;
            const char* rp = (j * RI < rows_left) ? res_run : res_last;
            res_run += res_step;
#if defined(VF_WIDE_SKIP) && (VF_WIDE_SKIP & 4)
            dst[k] = (u32x4_t){(unsigned)(uintptr_t)rp, 0u, 0u, 0u};
#else
            dst[k] = *reinterpret_cast<const u32x4_t*>(rp);
#endif
        }
    };
    auto res_value = [&](u32x2_t v) -> f32x4_t {         // see gemm8_kernel
        if constexpr (T16) return cvt4_16<VF_F16>(v) * ln.res16_scale;
        else if constexpr (DT == VF_F16) return cvt4_16<DT>(v) * ln.res16_scale;
        else return cvt4_16<DT>(v);
    };
    load_res_pass(0, rbuf[0]);
    f32x4_t bvec[TN];
#pragma unroll
    for (int in = 0; in < TN; ++in)
        bvec[in] = has_bias ? *reinterpret_cast<const f32x4_t*>(side + (side_n + in * 16 + 4 * g) * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (ps + 1 < NPASS) load_res_pass(ps + 1, rbuf[(ps + 1) & 1]);
#pragma unroll
        for (int iml = 0; iml < IMP; ++iml) {
            const int im = ps * IMP + iml;
            char* rowp = region + (iml * 16 + r) * PITCH;
#if !(defined(VF_WIDE_SKIP) && (VF_WIDE_SKIP & 8))
#pragma unroll
            for (int in = 0; in < TN; ++in) *reinterpret_cast<f32x4_t*>(rowp + (in * 16 + 4 * g) * 4) = acc[in][im] + bvec[in];
#endif
        }
        u32x4_t da[NI], db[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const char* p = region + (k * RI + ep_row) * PITCH + (lane & 7) * 32;
#if defined(VF_WIDE_SKIP) && (VF_WIDE_SKIP & 8)
            da[k] = __builtin_bit_cast(u32x4_t, acc[k & 3][ps * IMP]);          // no LDS read-back (wrong values)
            db[k] = __builtin_bit_cast(u32x4_t, acc[k & 3][ps * IMP + 1]);
#else
            da[k] = *reinterpret_cast<const u32x4_t*>(p);
            db[k] = *reinterpret_cast<const u32x4_t*>(p + 16);
#endif
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int j = ps * NI + k;                       // row j * RI + ep_row of the wave tile
            const u32x4_t rv = rbuf[ps & 1][k];
            const f32x4_t fa = __builtin_bit_cast(f32x4_t, da[k]) + res_value((u32x2_t){rv[0], rv[1]});
            const f32x4_t fb = __builtin_bit_cast(f32x4_t, db[k]) + res_value((u32x2_t){rv[2], rv[3]});
            const bool ok = j * RI + ep_row < WT_M && j * RI < rows_left && ep_col < N;
#ifdef VF_WIDE_SKIP   // cost-centre probes (scripts/probes/gemm8x_probe.hip builds only; results meaningless)
            ln_emit8<DT, VF_WIDE_SKIP>(fa, fb, ok, o16_run, part_run, lane, ln.x16_scale, (T16 && ln.t16_out) ? t16_run : nullptr, ln.t16_scale);
#else
            ln_emit8<DT>(fa, fb, ok, o16_run, part_run, lane, ln.x16_scale, (T16 && ln.t16_out) ? t16_run : nullptr, ln.t16_scale);
#endif
            o16_run += o16_step;
            part_run += RI * 2;
            if (T16) t16_run += t16_step;
            if (ok && out != nullptr) {                      // the fp32 rows, when somebody reads them (last layer of a stack)
                *reinterpret_cast<f32x4_t*>(out_run) = fa;
                *reinterpret_cast<f32x4_t*>(out_run + 16) = fb;
            }
            out_run += out_step;
        }
    }
}

// Epilogue for 4 consecutive n (n_base .. n_base+3) of row m.  For GEGLU `v` is the `a` half and
// `gate` the gate half; n_out is the output column of v[0].
template <int EPI, int DT = VF_BF16>
__device__ __forceinline__ void epilogue_store(f32x4_t v, f32x4_t gate, int64_t m, int n_bias, int n_bias_gate,
                                               int n_out, const float* __restrict__ bias,
                                               const float* __restrict__ res, int64_t ldr, void* out, int64_t ldo,
                                               const f32x4_t* res_pre = nullptr) {
    if (bias) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(bias + n_bias);
        v += b;
        if (EPI == VF_EPI_GEGLU_BF16) gate += *reinterpret_cast<const f32x4_t*>(bias + n_bias_gate);
    }
    if (EPI == VF_EPI_RES_F32) v += res_pre ? *res_pre : *reinterpret_cast<const f32x4_t*>(res + m * ldr + n_out);
    if (EPI == VF_EPI_GEGLU_BF16) {
        v = v * gelu_erf4(gate);
    }
    if (EPI == VF_EPI_GELU_F32 || EPI == VF_EPI_GELU_BF16) {
        v = gelu_erf4(v);
    }
    if (EPI == VF_EPI_F32 || EPI == VF_EPI_RES_F32 || EPI == VF_EPI_GELU_F32) {
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + m * ldo + n_out) = v;
    } else {
        u32x2_t p;
        p[0] = Op16<DT>::pack2(v[0], v[1]);
        p[1] = Op16<DT>::pack2(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned short*>(out) + m * ldo + n_out) = p;
    }
}

// Tile configuration: BM x BN output tile, WM x WN waves (wave tile BM/WM x BN/WN), STAGES-deep LDS ring
// (STAGES-1 K-tiles of LDS-DMA kept in flight across raw s_barriers with counted vmcnt).
template <int BM_, int BN_, int WM_, int WN_, int STAGES_, int BK_ = 64>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, STAGES = STAGES_, BK = BK_;
    static constexpr int NW = WM * WN, THREADS = NW * 64;
    static constexpr int TM = BM / WM / 16, TN = BN / WN / 16;       // 16x16 tiles per wave
    static constexpr int ROW_BYTES = BK * 2, CPR = BK / 8, KS = BK / 32; // chunks per row, MFMA k-steps per tile
    static constexpr int ROWS_PER_PIECE = 1024 / ROW_BYTES;          // one LDS-DMA wave-instruction = 1 KiB
    static constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int PA = A_BYTES / 1024 / NW, PW = W_BYTES / 1024 / NW;   // pieces per wave per tile
    static constexpr int LPT = PA + PW;                               // loads per tile per wave
    static_assert(BK == 64 || BK == 32, "BK");
    static_assert(A_BYTES % (1024 * NW) == 0 && W_BYTES % (1024 * NW) == 0, "pieces must divide over the waves");
    static_assert(STAGES >= 2 && STAGES <= 5 && (STAGES - 2) * LPT <= 48, "vmcnt range");
};

// DBG (diagnostic builds only, never selected automatically): 1 = no global loads (fragment reads + MFMA ceiling),
// 2 = no fragment reads / MFMA (LDS-DMA fill ceiling).  Results are meaningless in both.
template <class C, int EPI, int DT = VF_BF16, int DBG = 0, int LN = VF_LN_NONE>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_mfma_kernel(const unsigned short* __restrict__ A, int64_t lda,
                                                              const unsigned short* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ res, int64_t ldr, void* out,
                                                              int64_t ldo, int M, int N, int K, int tiles_n, int n_blocks,
                                                              int GROUP_M, LnArgs ln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using frag_t = typename Op16<DT>::frag;
    constexpr int BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN, STAGES = C::STAGES, LPT = C::LPT;
    constexpr int BK = C::BK, ROW_BYTES = C::ROW_BYTES, CPR = C::CPR, RPP = C::ROWS_PER_PIECE;

    // XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
    // XCD a contiguous run of tiles; consecutive tiles walk n first and share the A row panel in L2.
    const int bid = blockIdx.x;
    const int q8 = n_blocks >> 3, r8 = n_blocks & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    // Grouped order inside the run: GROUP_M m-panels x all n, m fastest, so the ~64 blocks resident on an XCD form
    // a squarish patch (8 A-panels x 8 W-panels) and each staged slice is shared by 8 blocks in that XCD's L2.
    const int tiles_m = n_blocks / tiles_n;
    const int per_group = GROUP_M * tiles_n;
    const int grp = wg / per_group, in_grp = wg - grp * per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int tm = first_m + in_grp % gsz, tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int r = lane & 15, g = lane >> 4;

    // ---- LDS-DMA source pointers.  A piece = RPP rows x ROW_BYTES written linearly by one wave-instruction;
    // lane -> row RPP*p + lane/CPR, physical chunk lane%CPR, logical (source) chunk = physical ^ swz(row).
    const unsigned short* srcA[C::PA];
    const unsigned short* srcW[C::PW];
#pragma unroll
    for (int i = 0; i < C::PA; ++i) {
        const int row = RPP * (wave * C::PA + i) + lane / CPR;
        const int c = (lane % CPR) ^ swz<BK>(row);
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        srcA[i] = A + (int64_t)gm * lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < C::PW; ++i) {
        const int row = RPP * (wave * C::PW + i) + lane / CPR;
        const int c = (lane % CPR) ^ swz<BK>(row);
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        srcW[i] = W + (int64_t)gn * K + c * 8;
    }
    char* const ldsA_piece = smem + wave * C::PA * 1024;
    char* const ldsW_piece = smem + C::A_BYTES + wave * C::PW * 1024;

    auto issue = [&](int kt, int stage) {
        if (DBG == 1 || DBG == 3) return;
#pragma unroll
        for (int i = 0; i < C::PA; ++i) glds16(srcA[i] + kt * BK, ldsA_piece + stage * C::STAGE_BYTES + i * 1024);
#pragma unroll
        for (int i = 0; i < C::PW; ++i) glds16(srcW[i] + kt * BK, ldsW_piece + stage * C::STAGE_BYTES + i * 1024);
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside an operand tile: row*ROW_BYTES + ((4*ks+g) ^ swz(r))*16
    const int sw = swz<BK>(r);
    const int offW = C::A_BYTES + (wn * (BN / C::WN) + r) * ROW_BYTES;
    const int offA = (wm * (BM / C::WM) + r) * ROW_BYTES;

    auto read_frags = [&](int stage, int ks, frag_t(&wf)[TN], frag_t(&af)[TM]) {
        if (DBG == 2) return;
        const char* base = smem + stage * C::STAGE_BYTES + (((4 * ks + g) ^ sw) << 4);
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const frag_t*>(base + offW + i * 16 * ROW_BYTES);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const frag_t*>(base + offA + i * 16 * ROW_BYTES);
    };
    auto issue_piece = [&](int kt, int stage, int i) {       // i in [0, LPT): one LDS-DMA wave-instruction
        if (i < C::PA) glds16(srcA[i] + kt * BK, ldsA_piece + stage * C::STAGE_BYTES + i * 1024);
        else glds16(srcW[i - C::PA] + kt * BK, ldsW_piece + stage * C::STAGE_BYTES + (i - C::PA) * 1024);
    };
    // MFMAs of one sub-step with the LDS-DMA pieces of K-tile `kt_issue` spread between them (one piece every
    // TN*TM/LPT MFMAs) instead of a burst right after the barrier
    auto mma_spread = [&](const frag_t(&wf)[TN], const frag_t(&af)[TM], int kt_issue, int stage_issue, bool do_issue) {
        constexpr int EVERY = 3;       // measured on 256x256: every 3 MFMAs 1203, every 4 1194, every 2 1195, burst 1161 (1133 before)
#pragma unroll
        for (int in = 0; in < TN; ++in)
#pragma unroll
            for (int im = 0; im < TM; ++im) {
                acc[in][im] = Op16<DT>::mfma(wf[in], af[im], acc[in][im]);
                const int idx = in * TM + im;
                if (idx % EVERY == EVERY - 1 && idx / EVERY < LPT) {
                    if (do_issue) issue_piece(kt_issue, stage_issue, idx / EVERY);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
    auto mma = [&](const frag_t(&wf)[TN], const frag_t(&af)[TM]) {
        if (DBG == 2) return;
#pragma unroll
        for (int in = 0; in < TN; ++in)
#pragma unroll
            for (int im = 0; im < TM; ++im)
                acc[in][im] = Op16<DT>::mfma(wf[in], af[im], acc[in][im]);
    };
    // wait until at most `ahead` younger K-tiles of this wave's LDS-DMA are still in flight
    auto wait_tiles = [&](int ahead) {
        if (STAGES >= 5 && ahead >= 3) wait_vmcnt<3 * LPT>();
        else if (STAGES >= 4 && ahead == 2) wait_vmcnt<2 * LPT>();
        else if (STAGES >= 3 && ahead == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
    };

    // Software pipeline.  The K loop is a sequence of sub-steps (one 32-deep MFMA k-step each).  The
    // fragments of sub-step s+1 are read from LDS into the OTHER register set while the MFMAs of
    // sub-step s execute.  When s+1 opens a new K-tile t+1 the wave first retires its own LDS reads
    // (every fragment of tile t is then in registers), waits for its share of tile t+1's LDS-DMA,
    // passes the barrier (tile t+1 visible, stage of tile t free for all) and immediately refills the
    // freed stage with tile t+STAGES: STAGES-1 tiles stay in flight across the barrier.
    const int nkt = K / BK;
#pragma unroll
    for (int p = 0; p < STAGES; ++p)
        if (p < nkt) issue(p, p);
    wait_tiles((nkt - 1) < (STAGES - 1) ? (nkt - 1) : (STAGES - 1));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    frag_t wf0[TN], af0[TM], wf1[TN], af1[TM];
    read_frags(0, 0, wf0, af0);
    int stage = 0;
    // cross from tile t (in `stage`) to tile t+1
    auto boundary = [&](int t) {
        if (DBG != 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_tiles((nkt - 2 - t) < (STAGES - 2) ? (nkt - 2 - t) : (STAGES - 2));
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        if (t + STAGES < nkt) issue(t + STAGES, stage);
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    };
    // ---- coalesced epilogue geometry.  After the K loop every wave stages its own sub-tile through its own slice
    // of the (then free) LDS ring and writes it out as whole rows, 16 bytes per lane: lane-strided accumulator stores
    // touch 32-64-byte row segments and are store-issue bound (measured: the K = 512 projections ran 551 TFLOP/s
    // with them, 942 without any store).  The fp32 residual is read -- and prefetched under the last K-tile's
    // MFMAs -- in the same coalesced mapping.
    constexpr bool OUT_F32 = (EPI == VF_EPI_F32 || EPI == VF_EPI_RES_F32 || EPI == VF_EPI_GELU_F32);
    constexpr int ES = OUT_F32 ? 4 : 2;
    constexpr int WT_M = BM / C::WM, WT_N = BN / C::WN;
    constexpr int WT_NO = (EPI == VF_EPI_GEGLU_BF16) ? WT_N / 2 : WT_N;       // output columns of the wave tile
    constexpr int PITCH = WT_NO * ES + 16;                                   // +16 B: conflict-free ds_write rows
    constexpr int REGION = C::LDS_BYTES / C::NW;
    constexpr int RP = (((REGION / PITCH) < WT_M ? (REGION / PITCH) : WT_M) / 16) * 16;   // rows per pass
    constexpr int IMP = RP / 16, NPASS = (TM + IMP - 1) / IMP;
    constexpr int CR = WT_NO * ES / 16, RI = 64 / CR, NI = RP / RI;           // chunks per row, rows / instruction
    static_assert(RP >= 16 && CR >= 1 && CR <= 64 && 64 % CR == 0, "epilogue geometry");
    const int n_out_total = (EPI == VF_EPI_GEGLU_BF16) ? N / 2 : N;
    const int64_t mw0 = m0 + wm * WT_M;
    const int no0 = (EPI == VF_EPI_GEGLU_BF16) ? (n0 + wn * WT_N) / 2 : (n0 + wn * WT_N);
    const int ep_row = lane / CR, ep_col = no0 + (lane % CR) * (16 / ES);
    constexpr bool RES_PRE = (EPI == VF_EPI_RES_F32) && (TN * TM <= 16);
    // the residual: fp32 rows, or (VF_LN_PRODUCER_R16) the 16-bit copy of a stream -- 8 instead of 16 bytes per item,
    // converted (and unscaled) where it is added
    constexpr bool R16 = ln_res_is_16(LN), T16 = LN == VF_LN_PRODUCER_T16;
    using res_t = typename std::conditional<R16, u32x2_t, f32x4_t>::type;
    auto res_load = [&](int64_t m, int col) -> res_t {
        if constexpr (R16) return *reinterpret_cast<const u32x2_t*>(ln.res16 + m * ln.ldr16 + col);
        else return *reinterpret_cast<const f32x4_t*>(res + m * ldr + col);
    };
    auto res_value = [&](res_t v) -> f32x4_t {
        if constexpr (T16) return cvt4_16<VF_F16>(v) * ln.res16_scale;
        else if constexpr (R16) {        // bf16 stream copies are never scaled (checked at launch): no multiply by 1.0 per element
            if constexpr (DT == VF_F16) return cvt4_16<DT>(v) * ln.res16_scale;
            else return cvt4_16<DT>(v);
        }
        else return v;
    };
    res_t resv[RES_PRE ? NPASS : 1][RES_PRE ? NI : 1];
    auto prefetch_residual = [&]() {
        if (RES_PRE) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    int64_t m = mw0 + ps * RP + k * RI + ep_row;
                    m = m < M ? m : M - 1;
                    const int col = ep_col < N ? ep_col : N - 4;
                    resv[RES_PRE ? ps : 0][RES_PRE ? k : 0] = res_load(m, col);
                }
        }
    };
    // Wave tiles too large to hold their whole residual in registers (256x256 tiles): the residual rows of epilogue
    // pass ps+1 are requested while pass ps goes through LDS, so only the first pass pays a load latency.
    constexpr bool RES_PIPE = (EPI == VF_EPI_RES_F32) && !RES_PRE;
    res_t rbuf[2][RES_PIPE ? NI : 1];
    auto load_res_pass = [&](int ps, res_t (&dst)[RES_PIPE ? NI : 1]) {
        if (RES_PIPE) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                int64_t m = mw0 + ps * RP + k * RI + ep_row;
                m = m < M ? m : M - 1;
                const int col = ep_col < N ? ep_col : N - 4;
                dst[RES_PIPE ? k : 0] = res_load(m, col);
            }
        }
    };
    // 8-wave tiles (256x256): the LDS-DMA pieces of the next K-tile are spread over the MFMA stream of the second
    // sub-step instead of being issued as a burst behind the barrier, where they queue in front of the fragment
    // reads (+3...6 % on the K = 1536 shapes).  Neutral to slightly negative on the 4-wave 128x128 tile, which keeps
    // the burst.
    constexpr bool SPREAD = C::KS == 2 && C::NW == 8 && STAGES == 2 && (TN * TM) >= 3 * LPT && DBG == 0;
    if (SPREAD) {
        for (int kt = 0; kt < nkt; ++kt) {
            read_frags(stage, 1, wf1, af1);
            if (kt + 1 == nkt) prefetch_residual();
            mma(wf0, af0);
            int freed = stage;
            const bool more = kt + 1 < nkt;
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_tiles((nkt - 2 - kt) < (STAGES - 2) ? (nkt - 2 - kt) : (STAGES - 2));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                stage = stage + 1 == STAGES ? 0 : stage + 1;
                read_frags(stage, 0, wf0, af0);
            }
            mma_spread(wf1, af1, kt + STAGES, freed, more && kt + STAGES < nkt);
        }
    } else if (C::KS == 2) {
        for (int kt = 0; kt < nkt; ++kt) {
            read_frags(stage, 1, wf1, af1);
            if (kt + 1 == nkt) prefetch_residual();
            mma(wf0, af0);
            if (kt + 1 < nkt) {
                boundary(kt);
                read_frags(stage, 0, wf0, af0);
            }
            mma(wf1, af1);
        }
    } else {                                   // BK = 32: one sub-step per tile, nkt is even (K % 64 == 0)
        for (int kt = 0; kt < nkt; kt += 2) {
            boundary(kt);
            read_frags(stage, 0, wf1, af1);
            mma(wf0, af0);
            if (kt + 2 < nkt) {
                boundary(kt + 1);
                read_frags(stage, 0, wf0, af0);
            } else {
                prefetch_residual();
            }
            mma(wf1, af1);
        }
    }

    // ---- epilogue: lane holds acc for out[m = mw0 + im*16 + r][n = nw0 + in*16 + 4g .. +3]
    if (DBG == 4) {                      // diagnostic: no epilogue stores (keep the accumulators live)
        f32x4_t t = acc[0][0];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int k = 0; k < TM; ++k) t += acc[i][k];
        if (t[0] == 123.456f) reinterpret_cast<float*>(out)[0] = t[1] + t[2] + t[3];
        return;
    }
    const int nw0 = n0 + wn * WT_N;
    load_res_pass(0, rbuf[0]);
    __syncthreads();                     // every wave's last fragments are in registers: the ring is free
    // bias of the wave's columns: ONE branch and one batch of loads (a per-element `if (bias) v += load` makes hipcc
    // branch around every load and drain vmcnt(0) behind each: 32 dependent L2 round trips per wave)
    f32x4_t bvec[TN];
    if (bias) {
#pragma unroll
        for (int in = 0; in < TN; ++in) {
            int nb = (EPI == VF_EPI_GEGLU_BF16) ? nw0 + (in >> 1) * 32 + (in & 1) * 16 + 4 * g : nw0 + in * 16 + 4 * g;
            nb = nb < N ? nb : 0;
            bvec[in] = *reinterpret_cast<const f32x4_t*>(bias + nb);
        }
    } else {
#pragma unroll
        for (int in = 0; in < TN; ++in) bvec[in] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // LayerNorm consumer: acc -> rstd[m] * (acc - mean[m] * colsum[n]); the folded bias is `bias`
    f32x4_t svec[LN == VF_LN_CONSUMER ? TN : 1];
    float ln_mu[LN == VF_LN_CONSUMER ? TM : 1], ln_rs[LN == VF_LN_CONSUMER ? TM : 1];
    if (LN == VF_LN_CONSUMER) {
#pragma unroll
        for (int in = 0; in < TN; ++in) {
            int nb = (EPI == VF_EPI_GEGLU_BF16) ? nw0 + (in >> 1) * 32 + (in & 1) * 16 + 4 * g : nw0 + in * 16 + 4 * g;
            nb = nb < N ? nb : 0;
            svec[LN == VF_LN_CONSUMER ? in : 0] = *reinterpret_cast<const f32x4_t*>(ln.colsum + nb);
        }
#pragma unroll
        for (int im = 0; im < TM; ++im) {
            int64_t m = mw0 + im * 16 + r;
            m = m < M ? m : M - 1;
            const f32x2_t st = *reinterpret_cast<const f32x2_t*>(ln.row_stats + 2 * m);
            ln_mu[LN == VF_LN_CONSUMER ? im : 0] = -st[0] * st[1];           // -mean * rstd
            ln_rs[LN == VF_LN_CONSUMER ? im : 0] = st[1];
        }
    }
    // accumulator + bias, or the LayerNorm-consumer form  rstd * (acc - mean * colsum) + bias'  evaluated as
    // acc * rstd + ((-mean * rstd) * colsum + bias'): two fused multiply-adds per element instead of sub, mul, mul, add (the
    // same expression in every tile configuration: their results stay bit-identical)
    auto lnv = [&](int in, int im) -> f32x4_t {
        if (LN == VF_LN_CONSUMER)
            return acc[in][im] * ln_rs[LN == VF_LN_CONSUMER ? im : 0] +
                   (ln_mu[LN == VF_LN_CONSUMER ? im : 0] * svec[LN == VF_LN_CONSUMER ? in : 0] + bvec[in]);
        return acc[in][im] + bvec[in];
    };
    char* const region = smem + wave * REGION;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (ps + 1 < NPASS) load_res_pass(ps + 1, rbuf[(ps + 1) & 1]);
        // (1) accumulators (+bias, activation) -> this wave's LDS slice, in the output dtype
#pragma unroll
        for (int iml = 0; iml < IMP; ++iml) {
            const int im = ps * IMP + iml;
            if (im < TM) {
                char* rowp = region + (iml * 16 + r) * PITCH;
                if (EPI == VF_EPI_GEGLU_BF16) {
#pragma unroll
                    for (int ip = 0; ip < TN / 2; ++ip) {
                        const f32x4_t v = lnv(2 * ip, im), gt = lnv(2 * ip + 1, im);
                        u32x2_t pk;
                        const f32x4_t y = v * gelu_erf4(gt);
                        pk[0] = Op16<DT>::pack2(y[0], y[1]);
                        pk[1] = Op16<DT>::pack2(y[2], y[3]);
                        *reinterpret_cast<u32x2_t*>(rowp + (ip * 16 + 4 * g) * 2) = pk;
                    }
                } else {
#pragma unroll
                    for (int in = 0; in < TN; ++in) {
                        f32x4_t v = lnv(in, im);
                        if (EPI == VF_EPI_GELU_F32 || EPI == VF_EPI_GELU_BF16) {
                            v = gelu_erf4(v);
                        }
                        if (OUT_F32) {
                            *reinterpret_cast<f32x4_t*>(rowp + (in * 16 + 4 * g) * 4) = v;
                        } else {
                            u32x2_t pk;
                            pk[0] = Op16<DT>::pack2(v[0], v[1]);
                            pk[1] = Op16<DT>::pack2(v[2], v[3]);
                            *reinterpret_cast<u32x2_t*>(rowp + (in * 16 + 4 * g) * 2) = pk;
                        }
                    }
                }
            }
        }
        // (2) read the slice back row-wise (same wave: LDS operations are in order) and store whole rows
        constexpr int KB = (EPI == VF_EPI_RES_F32) ? 4 : NI;     // read-back batch (fewer spare registers with a residual)
#pragma unroll
        for (int k0 = 0; k0 < NI; k0 += KB) {
            u32x4_t dd[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k0 + k < NI)
                    dd[k] = *reinterpret_cast<const u32x4_t*>(region + ((k0 + k) * RI + ep_row) * PITCH + (lane % CR) * 16);
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (k0 + k >= NI) continue;
                const int row = (k0 + k) * RI + ep_row;
                const int64_t m = mw0 + ps * RP + row;
                u32x4_t d = dd[k];
                if (EPI == VF_EPI_RES_F32) {
                    f32x4_t f = __builtin_bit_cast(f32x4_t, d);
                    f += res_value(RES_PRE ? resv[RES_PRE ? ps : 0][RES_PRE ? k0 + k : 0] : rbuf[ps & 1][RES_PIPE ? k0 + k : 0]);
                    d = __builtin_bit_cast(u32x4_t, f);
                }
                const bool ok = ps * RP + row < WT_M && m < M && ep_col < n_out_total;
                if (ln_is_producer(LN) && OUT_F32)
                    ln_emit<DT>(__builtin_bit_cast(f32x4_t, d), ok,
                                reinterpret_cast<unsigned short*>(ln.out16) + m * ln.ld16 + ep_col,
                                ln.part_stats + ((int64_t)(ep_col >> 5) * ln.rows + m) * 2, lane, ln.x16_scale, true, true,
                                (T16 && ln.t16_out) ? ln.t16_out + m * ln.ldt16 + ep_col : nullptr, ln.t16_scale);
                if (ok && (!ln_is_producer(LN) || out != nullptr))
                    *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(out) + (m * ldo + ep_col) * ES) = d;
            }
        }
    }
}

#ifdef VF_TUNING   // gemm_persist_kernel: persistent 128x128 form, measured without gain (DESIGN_HISTORY.md)
#include "tuning/gemm_persist.inc"
#endif

// ======================================================================================================================
// 256 x 256 x 64 tile, 8 waves, two wave groups running half a phase apart ("8-phase" schedule: 4 phases per K-tile,
// two K-tiles per LDS double buffer; cdna_hip_programming.md section 5, re-derived here for this operand layout).
//
// Waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the 4 SIMDs.  A phase of one wave is
//     [LDS fragment reads + 2 LDS-DMA pieces of the prefetch stream]  s_barrier  [16 MFMAs]  s_barrier
// and group 1 runs one barrier behind group 0, so on every SIMD one wave issues MFMAs while its partner reads
// fragments and issues the prefetch: the matrix pipe never waits for LDS and the LDS never waits for the matrix pipe.
//
// Wave (wm = wave / 4, wn = wave % 4) owns the contiguous 128 (m) x 64 (n) block of the tile, computed as four
// quadrants of 64 x 32 (16 MFMAs of 16x16x32 over the K-tile's 64 deep slice each):
//     P1: (m-lo, n-lo)   reads W-lo (4 x b128, first), A-lo (8)      P2: (m-lo, n-hi)   reads W-hi (4)
//     P3: (m-hi, n-hi)   reads A-hi (8)                              P4: (m-hi, n-lo)   W-lo kept in registers
// A K-tile lives in LDS as four 16 KiB half-tiles  WL | AL | WH | AH  (128 rows x 128 B each, XOR-swizzled like the
// other configurations).  Half-tile XL holds, for every wave row / column group, the FIRST half of that group's rows
// (LDS row i of AL <-> m = m0 + (i / 64) * 128 + i % 64, of WL <-> n = n0 + (i / 32) * 64 + i % 32), XH the second: the
// LDS-DMA source address is per lane, so the split costs nothing and every wave tile stays contiguous in memory.
//
// Prefetch stream: one half-tile (2 LDS-DMA wave-instructions per wave) per phase, in the order WL, AL, WH, AH, running
// one K-tile plus three half-tiles ahead of the consumer; the only wait is a counted vmcnt(6) in P4 (three half-tiles
// stay in flight across every barrier, vmcnt never reaches 0 inside the loop).
//   RAW  a half-tile is read at the earliest one phase after the P4 wait that retired it (both groups have passed a
//        barrier behind their own wait by then).
//   WAR  half-tile X of K-tile t+2 overwrites X of K-tile t.  WL: last read P1, rewritten P2 -- one phase later, legal
//        only because the four W-lo reads are issued first and retired (lgkmcnt(8)) BEFORE P1's first barrier, i.e.
//        before the other group, half a phase ahead, can issue the rewrite.  AL: read P1, rewritten P3; WH: read P2,
//        rewritten P4; AH: read P3, rewritten P1 of the next tile -- two phases, safe for any stagger of one barrier.
// ======================================================================================================================
struct Cfg8 {
    static constexpr int BM = 256, BN = 256, BK = 64, WM = 2, WN = 4, NW = 8, THREADS = 512;
    static constexpr int TM = 8, TN = 4;                        // 16x16 tiles per wave: 128 x 64
    static constexpr int HALF_BYTES = 128 * 128;                // one half-tile: 128 rows x 64 bf16
    static constexpr int TILE_BYTES = 4 * HALF_BYTES;           // WL | AL | WH | AH
    static constexpr int LDS_BYTES = 2 * TILE_BYTES;            // 128 KiB
    static constexpr int SIDE_BYTES = 4096;                     // behind the ring: bias | colsum | (mean, rstd) of the tile
    static constexpr int SPARE_BYTES = 24576;                   // persistent kernel: extra epilogue staging behind buffer 1
    enum { WL = 0, AL = 1, WH = 2, AH = 3 };
};

#ifdef VF_G8_PROF
__device__ unsigned long long* vf_g8_prof = nullptr;      // scripts/probes/gemm8_probe.hip: cycle stamps of every 16th block
#endif

template <int EPI, int DT = VF_BF16, int LN = VF_LN_NONE>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const unsigned short* __restrict__ A, int64_t lda,
                                                       const unsigned short* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ res,
                                                       int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
                                                       int tiles_n, int n_blocks, int GROUP_M, LnArgs ln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using frag_t = typename Op16<DT>::frag;
    using C = Cfg8;
    constexpr int BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN, BK = C::BK;

    // XCD-aware bijective remap + grouped order (same as gemm_mfma_kernel)
    const int bid = blockIdx.x;
#ifdef VF_G8_PROF
    unsigned long long g8t[8]; int g8n = 0;
#define G8_MARK() { g8t[g8n < 8 ? g8n : 7] = __builtin_readcyclecounter(); ++g8n; }
    G8_MARK()
#else
#define G8_MARK()
#endif
#ifdef VF_TUNING   // cost-centre probes of the epilogue (VF_G8_DBG bit mask, scripts/gemm_bench.py; results meaningless)
    const int dbg = (GROUP_M >> 8) & 255;
    // start-up stagger experiment (VF_G8_STAGGER = units of ~4 us): every other CU of an XCD starts its FIRST tile late, so
    // that the epilogue bursts of the two halves do not coincide (later blocks inherit the phase of the CU they land on)
    const int stagger = GROUP_M >> 16;
    GROUP_M &= 255;
    if (stagger && bid < 256 && ((bid >> 3) & 1))
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
#else
    constexpr int dbg = 0;
#endif
    const int q8 = n_blocks >> 3, r8 = n_blocks & 7, xcd = bid & 7, loc = bid >> 3;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tiles_m = n_blocks / tiles_n;
    const int per_group = GROUP_M * tiles_n;
    const int grp = wg / per_group, in_grp = wg - grp * per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int m0 = (first_m + in_grp % gsz) * BM, n0 = (in_grp / gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, g = lane >> 4;

    // ---- LDS-DMA sources: wave w fills rows 16w .. 16w+15 of every half-tile (two 8-row pieces)
    const unsigned short* src[4][2];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int i = 16 * wave + 8 * pi + (lane >> 3);                       // LDS row inside the half-tile
        const int c = (lane & 7) ^ ((i >> 1) & 7);                            // logical (source) chunk of this lane
        const int am = m0 + (i >> 6) * 128 + (i & 63);
        const int wn_row = n0 + (i >> 5) * 64 + (i & 31);
        int v;
        v = am;           v = v < M ? v : M - 1;  src[C::AL][pi] = A + (int64_t)v * lda + c * 8;
        v = am + 64;      v = v < M ? v : M - 1;  src[C::AH][pi] = A + (int64_t)v * lda + c * 8;
        v = wn_row;       v = v < N ? v : N - 1;  src[C::WL][pi] = W + (int64_t)v * K + c * 8;
        v = wn_row + 32;  v = v < N ? v : N - 1;  src[C::WH][pi] = W + (int64_t)v * K + c * 8;
    }
    char* const lds_piece = smem + wave * 2048;                               // + buf * TILE + type * HALF + pi * 1024
    auto issue = [&](int kt, int type) {                                      // one half-tile of K-tile kt
        char* dst = lds_piece + (kt & 1) * C::TILE_BYTES + type * C::HALF_BYTES;
        glds16(src[type][0] + kt * BK, dst);
        glds16(src[type][1] + kt * BK, dst + 1024);
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses (bytes inside a K-tile buffer): row * 128 + ((4 ks + g) ^ swz(r)) * 16
    const int sw = (r >> 1) & 7;
    const int ck0 = ((g) ^ sw) << 4, ck1 = ((4 + g) ^ sw) << 4;
    const int offW = (wn * 32 + r) * 128;            // + {WL, WH} * HALF + in_local * 2048
    const int offA = (wm * 64 + r) * 128;            // + {AL, AH} * HALF + im_local * 2048
    frag_t wlo[2][2], whi[2][2], af[4][2];         // [fragment][k-step]
    auto read_w = [&](const char* buf, int type, frag_t (&f)[2][2]) {
        const char* b = buf + type * C::HALF_BYTES + offW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f[i][0] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck0);
            f[i][1] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck1);
        }
    };
    auto read_a = [&](const char* buf, int type) {
        const char* b = buf + type * C::HALF_BYTES + offA;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck0);
            af[i][1] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck1);
        }
    };
    // first barrier of a phase, then the fragment reads must be back, then the MFMA cluster, then the second barrier.
    // sched_barrier(0) keeps hipcc from moving MFMAs or LDS reads across the phase structure; the priority flips keep
    // the cluster together (cdna_hip_programming.md T5).
#define VF_G8_SYNC_IN()                                          \
    do {                                                         \
        asm volatile("" ::: "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_setprio(1);                           \
    } while (0)
#define VF_G8_SYNC_OUT()                                         \
    do {                                                         \
        __builtin_amdgcn_s_setprio(0);                           \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("" ::: "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#define VF_G8_MMA(WF, IN0, IM0)                                                                                      \
    do {                                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                    acc[IN0 + i][IM0 + j] =                                                                          \
                        Op16<DT>::mfma(WF[i][ks], af[j][ks], acc[IN0 + i][IM0 + j]); \
    } while (0)

    // ---- epilogue operands of the tile (bias and, for the LayerNorm consumer, colsum of its 256 columns and (mean,
    // rstd) of its 256 rows) are requested by LDS-DMA BEFORE the first K-tile into a side area behind the ring: they
    // are the oldest entries of every wave's vmcnt queue, so the prologue's wait + barrier publishes them, and the
    // epilogue reads them from LDS instead of paying an exposed L2 round trip per output tile.
    char* const side = smem + C::LDS_BYTES;          // [0, 1K) bias, [1K, 2K) colsum, [2K, 4K) row statistics
    if (bias && wave == 0) {
        int n = n0 + 4 * lane;
        n = n < N ? n : N - 4;
        glds16(bias + n, side);
    }
    if (LN == VF_LN_CONSUMER) {
        if (wave == 1) {
            int n = n0 + 4 * lane;
            n = n < N ? n : N - 4;
            glds16(ln.colsum + n, side + 1024);
        }
        int64_t m = m0 + 32 * wave + (lane >> 1);    // every wave: 32 rows x (mean, rstd), one dword per lane
        m = m < M ? m : M - 1;
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)(ln.row_stats + 2 * m + (lane & 1)),
                                         (__attribute__((address_space(3))) void*)(side + 2048 + wave * 256), 4, 0, 0);
    }

    // ---- prologue: K-tile 0 complete, the first three half-tiles of K-tile 1 in flight
    const int nkt = K / BK;
    issue(0, C::WL); issue(0, C::AL); issue(0, C::WH); issue(0, C::AH);
    if (nkt > 1) {
        issue(1, C::WL); issue(1, C::AL); issue(1, C::WH);
        wait_vmcnt<6>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    G8_MARK()                                        // 1: first K-tile in LDS
    if (wm == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one barrier behind (matched after the loop)

    for (int t = 0; t < nkt; ++t) {
        const char* buf = smem + (t & 1) * C::TILE_BYTES;
        const bool pre1 = t + 1 < nkt, pre2 = t + 2 < nkt;       // wave-uniform
        // ---- P1: (m-lo, n-lo)
        read_w(buf, C::WL, wlo);
        __builtin_amdgcn_sched_barrier(0);                       // W-lo reads are issued first ...
        read_a(buf, C::AL);
        if (pre1) issue(t + 1, C::AH);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // ... and retired before the barrier (WAR on WL)
        VF_G8_SYNC_IN();
        VF_G8_MMA(wlo, 0, 0);
        VF_G8_SYNC_OUT();
        // ---- P2: (m-lo, n-hi)
        read_w(buf, C::WH, whi);
        if (pre2) issue(t + 2, C::WL);
        VF_G8_SYNC_IN();
        VF_G8_MMA(whi, 2, 0);
        VF_G8_SYNC_OUT();
        // ---- P3: (m-hi, n-hi)
        read_a(buf, C::AH);
        if (pre2) issue(t + 2, C::AL);
        VF_G8_SYNC_IN();
        VF_G8_MMA(whi, 2, 4);
        VF_G8_SYNC_OUT();
        // ---- P4: (m-hi, n-lo); retire K-tile t+1 (all but the three youngest half-tiles)
        if (pre2) {
            issue(t + 2, C::WH);
            wait_vmcnt<6>();
        } else {
            wait_vmcnt<0>();
        }
        VF_G8_SYNC_IN();
        VF_G8_MMA(wlo, 0, 4);
        VF_G8_SYNC_OUT();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();       // matches group 1's extra barrier: every wave is past its last MFMA
    asm volatile("" ::: "memory");
    G8_MARK()                                        // 2: K loop done
#undef VF_G8_SYNC_IN
#undef VF_G8_SYNC_OUT
#undef VF_G8_MMA

    // ---- epilogue: each wave stages its 128 x 64 block through its own 16 KiB slice of the (now free) ring and
    // writes whole rows, 16 bytes per lane; the fp32 residual rows of pass p+1 are requested while pass p goes through
    // LDS (same scheme as gemm_mfma_kernel, see the comments there).
    constexpr bool OUT_F32 = (EPI == VF_EPI_F32 || EPI == VF_EPI_RES_F32 || EPI == VF_EPI_GELU_F32);
    constexpr int ES = OUT_F32 ? 4 : 2;
    constexpr int WT_M = 128, WT_N = 64;
    constexpr int WT_NO = (EPI == VF_EPI_GEGLU_BF16) ? WT_N / 2 : WT_N;
    constexpr int PITCH = WT_NO * ES + 16;
    constexpr int REGION = C::LDS_BYTES / C::NW;
    constexpr int RP_FIT = (((REGION / PITCH) < WT_M ? (REGION / PITCH) : WT_M) / 16) * 16;
    // the LayerNorm producer carries extra live values through the read-back: 32-row passes keep it inside 256 VGPRs
    constexpr int RP = (ln_is_producer(LN) && EPI == VF_EPI_RES_F32 && RP_FIT > 32) ? 32 : RP_FIT;
    constexpr int IMP = RP / 16, NPASS = (TM + IMP - 1) / IMP;
    constexpr int CR = WT_NO * ES / 16, RI = 64 / CR, NI = RP / RI;
    static_assert(RP >= 16 && CR >= 1 && CR <= 64 && 64 % CR == 0, "epilogue geometry");
    constexpr bool RES = (EPI == VF_EPI_RES_F32);
    const int n_out_total = (EPI == VF_EPI_GEGLU_BF16) ? N / 2 : N;
    const int64_t mw0 = m0 + wm * WT_M;
    const int nw0 = n0 + wn * WT_N;
    const int no0 = (EPI == VF_EPI_GEGLU_BF16) ? nw0 / 2 : nw0;
    const int ep_row = lane / CR, ep_col = no0 + (lane % CR) * (16 / ES);
    // Addresses of this lane's read-back items.  Item j = pass * NI + k is row j * RI + ep_row of the wave tile, so every
    // pointer is "first row + j * (RI rows)": ONE 64-bit multiply per lane, the per-item steps are wave-uniform scalars
    // (a multiply per item costs quarter-rate v_mul_lo_u32 / v_mad_u64_u32 pairs: 192 of the 766 VALU instructions of
    // the fp32-residual epilogue before this).  Rows past M read the last row instead (never stored).
    const int rows_left = (int)(M - mw0) - ep_row;              // item j is a row of the matrix iff j * RI < rows_left
    const int64_t row0 = mw0 + ep_row;
    const int colc = ep_col < N ? ep_col : N - 4;
    // the residual: fp32 rows, or (VF_LN_PRODUCER_R16) the 16-bit copy of a stream: 8 instead of 16 bytes per item,
    // converted (and unscaled) where it is added.  Byte pointers so that both forms share the running-pointer scheme.
    constexpr bool R16 = ln_res_is_16(LN), T16 = LN == VF_LN_PRODUCER_T16;
    using res_t = typename std::conditional<R16, u32x2_t, f32x4_t>::type;
    const char* const res_base = R16 ? reinterpret_cast<const char*>(ln.res16) : reinterpret_cast<const char*>(res);
    const int64_t res_ld = R16 ? ln.ldr16 : ldr;                // elements per residual row
    constexpr int RES_ES = R16 ? 2 : 4;
    const char* const res_p = RES ? res_base + (row0 * res_ld + colc) * RES_ES : nullptr;
    const char* const res_last = RES ? res_base + ((int64_t)(M - 1) * res_ld + colc) * RES_ES : nullptr;
    const int64_t res_step = (int64_t)RI * res_ld * RES_ES;
    char* const out_p = reinterpret_cast<char*>(out) + (row0 * ldo + ep_col) * ES;
    const int64_t out_step = (int64_t)RI * ldo * ES;
    unsigned short* const o16_p = ln_is_producer(LN) ? reinterpret_cast<unsigned short*>(ln.out16) + row0 * ln.ld16 + ep_col : nullptr;
    const int64_t o16_step = (int64_t)RI * ln.ld16;
    float* const part_p = ln_is_producer(LN) ? ln.part_stats + ((int64_t)(ep_col >> 5) * ln.rows + row0) * 2 : nullptr;
    // the items are visited in increasing j, so each pointer is a running one: p += step per item (one 64-bit add)
    // instead of base + j * step (hipcc multiplies per item otherwise: 81 quarter-rate v_mad_u64_u32 in this epilogue)
    const char* res_run = res_p;
    char* out_run = out_p;
    unsigned short* o16_run = o16_p;
    float* part_run = part_p;
    unsigned short* t16_run = T16 ? ln.t16_out + row0 * ln.ldt16 + ep_col : nullptr;
    const int64_t t16_step = (int64_t)RI * ln.ldt16;
    // VF_G8_RES_ALL (experiment, off): 16-bit residual rows of ALL passes requested up front (NPASS * NI items of 2 registers
    // = 64 of the registers the operand fragments no longer need) instead of one pass ahead.
    constexpr bool RES_ALL = R16 && RES && (VF_G8_RES_ALL != 0);
    constexpr int NRB = RES_ALL ? NPASS : 2;
    res_t rbuf[NRB][RES ? NI : 1];
    auto load_res_pass = [&](int ps, res_t (&dst)[RES ? NI : 1]) {
        if (RES) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int j = ps * NI + k;
                const char* rp = (j * RI < rows_left) ? res_run : res_last;
                res_run += res_step;
                if (dbg & 8) { dst[RES ? k : 0] = res_t{}; continue; }
                dst[RES ? k : 0] = *reinterpret_cast<const res_t*>(rp);
            }
        }
    };
    auto res_value = [&](res_t v) -> f32x4_t {
        if constexpr (T16) return cvt4_16<VF_F16>(v) * ln.res16_scale;
        else if constexpr (R16) {        // bf16 stream copies are never scaled (checked at launch): no multiply by 1.0 per element
            if constexpr (DT == VF_F16) return cvt4_16<DT>(v) * ln.res16_scale;
            else return cvt4_16<DT>(v);
        }
        else return v;
    };
    load_res_pass(0, rbuf[0]);
    if (RES_ALL) {
#pragma unroll
        for (int ps = 1; ps < NPASS; ++ps) load_res_pass(ps, rbuf[ps < NRB ? ps : 0]);
    }
    // bias of the wave's columns, from the side area (requested before the first K-tile)
    f32x4_t bvec[TN];
    if (bias) {
#pragma unroll
        for (int in = 0; in < TN; ++in) {
            const int nl = wn * WT_N + ((EPI == VF_EPI_GEGLU_BF16) ? (in >> 1) * 32 + (in & 1) * 16 + 4 * g : in * 16 + 4 * g);
            bvec[in] = *reinterpret_cast<const f32x4_t*>(side + nl * 4);
        }
    } else {
#pragma unroll
        for (int in = 0; in < TN; ++in) bvec[in] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // LayerNorm consumer: acc -> rstd[m] * (acc - mean[m] * colsum[n]); the folded bias is `bias`
    f32x4_t svec[LN == VF_LN_CONSUMER ? TN : 1];
    float ln_mu[LN == VF_LN_CONSUMER ? TM : 1], ln_rs[LN == VF_LN_CONSUMER ? TM : 1];
    if (LN == VF_LN_CONSUMER) {
#pragma unroll
        for (int in = 0; in < TN; ++in) {
            const int nl = wn * WT_N + ((EPI == VF_EPI_GEGLU_BF16) ? (in >> 1) * 32 + (in & 1) * 16 + 4 * g : in * 16 + 4 * g);
            svec[LN == VF_LN_CONSUMER ? in : 0] = *reinterpret_cast<const f32x4_t*>(side + 1024 + nl * 4);
        }
#pragma unroll
        for (int im = 0; im < TM; ++im) {
            const f32x2_t st = *reinterpret_cast<const f32x2_t*>(side + 2048 + (wm * WT_M + im * 16 + r) * 8);
            ln_mu[LN == VF_LN_CONSUMER ? im : 0] = -st[0] * st[1];           // -mean * rstd
            ln_rs[LN == VF_LN_CONSUMER ? im : 0] = st[1];
        }
    }
    // accumulator + bias, or the LayerNorm-consumer form  rstd * (acc - mean * colsum) + bias'  evaluated as
    // acc * rstd + ((-mean * rstd) * colsum + bias'): two fused multiply-adds per element instead of sub, mul, mul, add (the
    // same expression in every tile configuration: their results stay bit-identical)
    auto lnv = [&](int in, int im) -> f32x4_t {
        if (LN == VF_LN_CONSUMER)
            return acc[in][im] * ln_rs[LN == VF_LN_CONSUMER ? im : 0] +
                   (ln_mu[LN == VF_LN_CONSUMER ? im : 0] * svec[LN == VF_LN_CONSUMER ? in : 0] + bvec[in]);
        return acc[in][im] + bvec[in];
    };
    char* const region = smem + wave * REGION;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (!RES_ALL && ps + 1 < NPASS) load_res_pass(ps + 1, rbuf[(ps + 1) & 1]);
#pragma unroll
        for (int iml = 0; iml < IMP; ++iml) {
            const int im = ps * IMP + iml;
            if (im < TM) {
                char* rowp = region + (iml * 16 + r) * PITCH;
                if (EPI == VF_EPI_GEGLU_BF16) {
#pragma unroll
                    for (int ip = 0; ip < TN / 2; ++ip) {
                        const f32x4_t v = lnv(2 * ip, im), gt = lnv(2 * ip + 1, im);
                        u32x2_t pk;
                        const f32x4_t y = v * gelu_erf4(gt);
                        pk[0] = Op16<DT>::pack2(y[0], y[1]);
                        pk[1] = Op16<DT>::pack2(y[2], y[3]);
                        *reinterpret_cast<u32x2_t*>(rowp + (ip * 16 + 4 * g) * 2) = pk;
                    }
                } else {
#pragma unroll
                    for (int in = 0; in < TN; ++in) {
                        f32x4_t v = lnv(in, im);
                        if (EPI == VF_EPI_GELU_F32 || EPI == VF_EPI_GELU_BF16) {
                            v = gelu_erf4(v);
                        }
                        if (OUT_F32) {
                            *reinterpret_cast<f32x4_t*>(rowp + (in * 16 + 4 * g) * 4) = v;
                        } else {
                            u32x2_t pk;
                            pk[0] = Op16<DT>::pack2(v[0], v[1]);
                            pk[1] = Op16<DT>::pack2(v[2], v[3]);
                            *reinterpret_cast<u32x2_t*>(rowp + (in * 16 + 4 * g) * 2) = pk;
                        }
                    }
                }
            }
        }
        // (2) read the slice back row-wise: all LDS reads first (unconditional: every row lies inside the slice), then
        // the predicated stores, so that no store waits behind a per-row ds_read round trip
        constexpr int KB = RES ? 4 : NI;          // read-back batch (the residual epilogue has fewer registers to spare)
#pragma unroll
        for (int k0 = 0; k0 < NI; k0 += KB) {
            u32x4_t dd[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k0 + k < NI)
                    dd[k] = *reinterpret_cast<const u32x4_t*>(region + ((k0 + k) * RI + ep_row) * PITCH + (lane % CR) * 16);
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                if (k0 + k >= NI) continue;
                const int j = ps * NI + k0 + k;                  // row j * RI + ep_row of the wave tile
                u32x4_t d = dd[k];
                if (RES) {
                    f32x4_t f = __builtin_bit_cast(f32x4_t, d);
                    f += res_value(rbuf[RES_ALL ? (ps < NRB ? ps : 0) : (ps & 1)][RES ? k0 + k : 0]);
                    d = __builtin_bit_cast(u32x4_t, f);
                }
                const bool ok = j * RI + ep_row < WT_M && j * RI < rows_left && ep_col < n_out_total;
                if (ln_is_producer(LN) && OUT_F32) {
                    ln_emit<DT>(__builtin_bit_cast(f32x4_t, d), ok, o16_run, part_run, lane, ln.x16_scale, !(dbg & 2), !(dbg & 4),
                                (T16 && ln.t16_out) ? t16_run : nullptr, ln.t16_scale);
                    o16_run += o16_step;
                    part_run += RI * 2;
                    if (T16) t16_run += t16_step;
                }
                // a LayerNorm producer whose fp32 result has no reader (only its 16-bit copy and statistics do) passes
                // out = NULL: the 16-byte store -- 4 of the 10 bytes the epilogue moves per element -- is dropped
                if (ok && (!ln_is_producer(LN) || out != nullptr) && !(dbg & 1)) *reinterpret_cast<u32x4_t*>(out_run) = d;
                out_run += out_step;
            }
        }
        G8_MARK()                                    // 3 .. 6: epilogue passes
    }
#ifdef VF_G8_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G8_MARK()                                        // 7: stores acknowledged
    if (vf_g8_prof && lane == 0 && wave == 0 && (bid & 15) == 0)
        for (int i = 0; i < 8; ++i) vf_g8_prof[(bid >> 4) * 8 + i] = g8t[i];
#endif
#undef G8_MARK
}

// ----------------------------------------------------------------------------------------------------------------------
// Persistent form of gemm8_kernel: one 8-wave block per CU walks the output tiles bid, bid + grid, ... (same XCD-grouped
// order) and the prefetch stream runs ONE K-tile across every output-tile boundary: K-tile 0 of the next tile is
// requested under the last two K-tiles of the current one into the ring buffer that is free by then, so that neither
// the block launch nor the first fill (~2 us of global -> LDS latency with nothing to compute) is paid per tile.  The
// epilogue stages through the OTHER buffer -- the one that held the last K-tile -- and one
// barrier ends it before the next tile's K-tile 1 is requested into that buffer (with an even number of K-tiles that is
// always buffer 1, and the 24 KiB of LDS behind the ring extend it to 11 KiB per wave).  Epilogue operands (bias, colsum, row
// statistics) of the next tile arrive by LDS-DMA with its K-tile 0, double-buffered behind the ring.  Needs K % 128 == 0.
// ----------------------------------------------------------------------------------------------------------------------
template <int EPI, int DT = VF_BF16, int LN = VF_LN_NONE>
__global__ __launch_bounds__(512, 2) void gemm8x_kernel(const unsigned short* __restrict__ A, int64_t lda,
                                                       const unsigned short* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ res,
                                                       int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
                                                       int tiles_n, int n_tiles, int GROUP_M, LnArgs ln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using frag_t = typename Op16<DT>::frag;
    using C = Cfg8;
    constexpr int BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN, BK = C::BK;

    // XCD-aware bijective remap + grouped order (same as gemm_mfma_kernel)
    const int bid = blockIdx.x;
#ifdef VF_TUNING   // cost-centre probes of the epilogue (VF_G8_DBG bit mask, scripts/gemm_bench.py; results meaningless)
    const int dbg = GROUP_M >> 8;
    GROUP_M &= 255;
#else
    constexpr int dbg = 0;
#endif
    const int grid = gridDim.x;
    const int my_tiles = (n_tiles - bid + grid - 1) / grid;          // output tiles bid, bid + grid, ... (>= 1)
    auto tile_origin = [&](int t, int& m0, int& n0) {                 // XCD-contiguous runs, grouped order (see gemm_mfma_kernel)
        const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xcd = t & 7, loc = t >> 3;
        const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        const int tiles_m = n_tiles / tiles_n;
        const int per_group = GROUP_M * tiles_n;
        const int grp = wg / per_group, in_grp = wg - grp * per_group;
        const int first_m = grp * GROUP_M;
        const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        m0 = (first_m + in_grp % gsz) * BM;
        n0 = (in_grp / gsz) * BN;
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, g = lane >> 4;

    // ---- LDS-DMA sources of ONE output tile: wave w fills rows 16w .. 16w+15 of every half-tile (two 8-row pieces).
    // Recomputed in place when the prefetch stream crosses into the next output tile (no second pointer set).
    const unsigned short* src[4][2];
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int i = 16 * wave + 8 * pi + (lane >> 3);                   // LDS row inside the half-tile
            const int c = (lane & 7) ^ ((i >> 1) & 7);                        // logical (source) chunk of this lane
            const int am = m0 + (i >> 6) * 128 + (i & 63);
            const int wn_row = n0 + (i >> 5) * 64 + (i & 31);
            int v;
            v = am;           v = v < M ? v : M - 1;  src[C::AL][pi] = A + (int64_t)v * lda + c * 8;
            v = am + 64;      v = v < M ? v : M - 1;  src[C::AH][pi] = A + (int64_t)v * lda + c * 8;
            v = wn_row;       v = v < N ? v : N - 1;  src[C::WL][pi] = W + (int64_t)v * K + c * 8;
            v = wn_row + 32;  v = v < N ? v : N - 1;  src[C::WH][pi] = W + (int64_t)v * K + c * 8;
        }
    };
    char* const lds_piece = smem + wave * 2048;                               // + buf * TILE + type * HALF + pi * 1024
    // half-tile `type` of the K-tile with index kt inside the tile `src` points at; gk = its index in the block's
    // K-tile stream (the ring buffer is the stream index's parity, so the stream runs across output tiles)
    auto issue = [&](int gk, int kt, int type) {
#ifdef VF_G8X_NOFILL      // probe build: the K loop without its LDS-DMA stream (stale LDS contents; results meaningless)
        if (gk > 1) return;
#endif
        char* dst = lds_piece + (gk & 1) * C::TILE_BYTES + type * C::HALF_BYTES;
        glds16(src[type][0] + kt * BK, dst);
        glds16(src[type][1] + kt * BK, dst + 1024);
    };
    // epilogue operands of a tile by LDS-DMA (bias | colsum | row statistics), double-buffered by tile parity
    auto issue_side = [&](int m0, int n0, int seq) {
        char* const sd = smem + C::LDS_BYTES + C::SPARE_BYTES + (seq & 1) * C::SIDE_BYTES;
        int lane;                                    // not hoistable out of the tile loop (see the epilogue's lane id)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        if (bias && wave == 0) {
            int n = n0 + 4 * lane;
            n = n < N ? n : N - 4;
            glds16(bias + n, sd);
        }
        if (LN == VF_LN_CONSUMER) {
            if (wave == 1) {
                int n = n0 + 4 * lane;
                n = n < N ? n : N - 4;
                glds16(ln.colsum + n, sd + 1024);
            }
            int64_t m = m0 + 32 * wave + (lane >> 1);    // every wave: 32 rows x (mean, rstd), one dword per lane
            m = m < M ? m : M - 1;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)(ln.row_stats + 2 * m + (lane & 1)),
                                             (__attribute__((address_space(3))) void*)(sd + 2048 + wave * 256), 4, 0, 0);
        }
    };

    f32x4_t acc[TN][TM];

    // ---- fragment addresses (bytes inside a K-tile buffer): row * 128 + ((4 ks + g) ^ swz(r)) * 16
    const int sw = (r >> 1) & 7;
    const int ck0 = ((g) ^ sw) << 4, ck1 = ((4 + g) ^ sw) << 4;
    const int offW = (wn * 32 + r) * 128;            // + {WL, WH} * HALF + in_local * 2048
    const int offA = (wm * 64 + r) * 128;            // + {AL, AH} * HALF + im_local * 2048
    frag_t wlo[2][2], whi[2][2], af[4][2];         // [fragment][k-step]
    auto read_w = [&](const char* buf, int type, frag_t (&f)[2][2]) {
        const char* b = buf + type * C::HALF_BYTES + offW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f[i][0] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck0);
            f[i][1] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck1);
        }
    };
    auto read_a = [&](const char* buf, int type) {
        const char* b = buf + type * C::HALF_BYTES + offA;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck0);
            af[i][1] = *reinterpret_cast<const frag_t*>(b + i * 2048 + ck1);
        }
    };
    // first barrier of a phase, then the fragment reads must be back, then the MFMA cluster, then the second barrier.
    // sched_barrier(0) keeps hipcc from moving MFMAs or LDS reads across the phase structure; the priority flips keep
    // the cluster together (cdna_hip_programming.md T5).
#define VF_G8_SYNC_IN()                                          \
    do {                                                         \
        asm volatile("" ::: "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_setprio(1);                           \
    } while (0)
#define VF_G8_SYNC_OUT()                                         \
    do {                                                         \
        __builtin_amdgcn_s_setprio(0);                           \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("" ::: "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#define VF_G8_MMA(WF, IN0, IM0)                                                                                      \
    do {                                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                    acc[IN0 + i][IM0 + j] =                                                                          \
                        Op16<DT>::mfma(WF[i][ks], af[j][ks], acc[IN0 + i][IM0 + j]); \
    } while (0)

    // Probe (tuning library only, VF_G8_DBG bits 16 / 32; results meaningless): one / two 16-byte-per-lane stores per phase
    // into the rows of the CURRENT tile (16 rows x 64 bytes per instruction, the shape of an epilogue without LDS staging),
    // issued in the fragment-read section -- do stores beside the LDS-DMA stream cost the K loop anything?  (round 6: the
    // question behind an epilogue inside the K loop; 32 / 64 KiB per K-tile and block against 64 KiB of fill.)
#ifdef VF_TUNING
#define VF_G8X_STORE_PROBE(PH)                                                                                         \
    do {                                                                                                               \
        /* bits 64 / 128: the store pattern of an epilogue inside the K loop -- 4 stores (32 rows x 64 columns of 16-bit */ \
        /* values) in P1 and P2 of a tile's first K-tile (64: they get 1.75 / 1.5 K-tiles until a counted wait needs them */ \
        /* complete), or in P3 and behind the wait of P4 of its second K-tile (128: 1.25 / 1 K-tiles) */               \
        if (((dbg & 64) && (PH) < 2 && t == 0 && ti > 0) || ((dbg & 128) && (PH) >= 2 && t == 1)) {                    \
            int lp;                                                                                                    \
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lp));                   \
            const int prow = m0 + wm * 128 + ((PH) >> 1) * 64 + ((PH) & 1) * 32 + (lp & 15);                            \
            const int pcol = n0 + wn * 64 + (lp >> 4) * 8;                                                             \
            if (prow + 16 < M && pcol + 32 < N) {                                                                      \
                unsigned short* pp = reinterpret_cast<unsigned short*>(out) + (int64_t)prow * ldo + pcol;              \
                *reinterpret_cast<f32x4_t*>(pp) = acc[0][0];                                                           \
                *reinterpret_cast<f32x4_t*>(pp + 32) = acc[0][1];                                                      \
                *reinterpret_cast<f32x4_t*>(pp + 16 * ldo) = acc[1][0];                                                \
                *reinterpret_cast<f32x4_t*>(pp + 16 * ldo + 32) = acc[1][1];                                           \
            }                                                                                                          \
        }                                                                                                              \
        if (dbg & 48) {                                                                                                \
            int lp;                                                                                                    \
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lp));                   \
            const int prow = m0 + wm * 128 + (((t * 4 + (PH)) * 16 + (lp & 15)) & 127);                                \
            const int pcol = n0 + wn * 64 + (lp >> 4) * 8;                                                             \
            if (prow < M && pcol + 32 < N) {                                                                           \
                unsigned short* pp = reinterpret_cast<unsigned short*>(out) + (int64_t)prow * ldo + pcol;              \
                *reinterpret_cast<f32x4_t*>(pp) = acc[0][0];                                                           \
                if (dbg & 32) *reinterpret_cast<f32x4_t*>(pp + 32) = acc[0][1];                                        \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)
#else
#define VF_G8X_STORE_PROBE(PH) do { } while (0)
#endif
    // ---- first tile: its epilogue operands and K-tile 0
    const int nkt = K / BK;                                  // >= 2 (launcher)
    int m0, n0;
    tile_origin(bid, m0, n0);
    set_src(m0, n0);
    issue_side(m0, n0, 0);
    issue(0, 0, C::WL); issue(0, 0, C::AL); issue(0, 0, C::WH); issue(0, 0, C::AH);

    int g0 = 0;                                              // stream index of the current tile's K-tile 0
#ifdef VF_G8_PROF   // scripts/probes/gemm8x_probe.hip: per block, cycles summed over its tiles (wave 0 = group 0, wave 4 = group 1)
    unsigned long long px[6] = {0, 0, 0, 0, 0, 0}, pt0, pt1;
#define G8X_T() __builtin_readcyclecounter()
    const unsigned long long pc0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();   // clock held under load:
#endif                                                                                                    // d(memtime) / d(memrealtime) x 100 MHz
    for (int ti = 0; ti < my_tiles; ++ti) {
        const bool has_next = ti + 1 < my_tiles;             // block-uniform
        tile_origin(bid + ti * grid, m0, n0);
#ifdef VF_G8_PROF
        pt0 = G8X_T();
#endif
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // K-tile 0 of this tile: requested in the prologue (first tile) or under the previous tile's last two K-tiles
        // and already retired there (vmcnt(0) before that tile's epilogue).  Its K-tile 1 goes into the buffer the
        // previous epilogue staged through (the barrier that ended that epilogue makes it free).
        issue(g0 + 1, 1, C::WL); issue(g0 + 1, 1, C::AL); issue(g0 + 1, 1, C::WH);
        if (ti == 0) wait_vmcnt<6>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind (matched after the loop)
#ifdef VF_G8_PROF
        pt1 = G8X_T(); px[0] += pt1 - pt0; pt0 = pt1;        // 0: tile start (K-tile 1 requested, start barriers)
#endif

        for (int t = 0; t < nkt; ++t) {
#ifdef VF_G8_PROF
            if (t == 1) { pt1 = G8X_T(); px[1] += pt1 - pt0; pt0 = pt1; }      // 1: the first K-tile of the tile
#endif
            const char* buf = smem + ((g0 + t) & 1) * C::TILE_BYTES;
            // the stream continues into the NEXT output tile for exactly one K-tile: its K-tile 0 is "K-tile nkt"
            const bool pre1 = t + 1 < nkt || has_next;                       // AH of stream K-tile t+1
            const bool pre2 = t + 2 < nkt || (t + 2 == nkt && has_next);     // WL / AL / WH of stream K-tile t+2
            const bool into_next = t + 2 >= nkt;                             // those belong to the next tile (kt = 0)
            // ---- P1: (m-lo, n-lo)
            read_w(buf, C::WL, wlo);
            __builtin_amdgcn_sched_barrier(0);                       // W-lo reads are issued first ...
            read_a(buf, C::AL);
            if (pre1) issue(g0 + t + 1, t + 1 < nkt ? t + 1 : 0, C::AH);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // ... and retired before the barrier (WAR on WL)
            VF_G8X_STORE_PROBE(0);
            VF_G8_SYNC_IN();
            VF_G8_MMA(wlo, 0, 0);
            VF_G8_SYNC_OUT();
            if (t + 2 == nkt && has_next) {                          // the stream crosses into the next output tile
                int m1, n1;
                tile_origin(bid + (ti + 1) * grid, m1, n1);
                set_src(m1, n1);
                issue_side(m1, n1, ti + 1);
            }
            // ---- P2: (m-lo, n-hi)
            read_w(buf, C::WH, whi);
            if (pre2) issue(g0 + t + 2, into_next ? 0 : t + 2, C::WL);
            VF_G8X_STORE_PROBE(1);
            VF_G8_SYNC_IN();
            VF_G8_MMA(whi, 2, 0);
            VF_G8_SYNC_OUT();
            // ---- P3: (m-hi, n-hi)
            read_a(buf, C::AH);
            if (pre2) issue(g0 + t + 2, into_next ? 0 : t + 2, C::AL);
            VF_G8X_STORE_PROBE(2);
            VF_G8_SYNC_IN();
            VF_G8_MMA(whi, 2, 4);
            VF_G8_SYNC_OUT();
            // ---- P4: (m-hi, n-lo); retire stream K-tile t+1 (all but the three youngest half-tiles)
            if (pre2) {
                issue(g0 + t + 2, into_next ? 0 : t + 2, C::WH);
#ifdef VF_TUNING
                if (dbg & 32) wait_vmcnt<12>();                      // store probe: its stores since AH(t + 1) may stay in flight too
                else if (dbg & 16) wait_vmcnt<9>();
                else if ((dbg & 64) && t == 0 && ti > 0) wait_vmcnt<14>();
                else if ((dbg & 128) && t == 1) wait_vmcnt<10>();
                else
#endif
                wait_vmcnt<6>();
            } else {
                wait_vmcnt<0>();
            }
            VF_G8X_STORE_PROBE(3);
            VF_G8_SYNC_IN();
            VF_G8_MMA(wlo, 0, 4);
            VF_G8_SYNC_OUT();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();   // matches group 1's extra barrier: every wave is past its last MFMA
        asm volatile("" ::: "memory");
#ifdef VF_G8_PROF
        pt1 = G8X_T(); px[2] += pt1 - pt0; pt0 = pt1;        // 2: K-tiles 1 .. nkt-1 (+ the group-sync barrier)
#endif
        char* const side = smem + C::LDS_BYTES + C::SPARE_BYTES + (ti & 1) * C::SIDE_BYTES;
        // Staging: the last K-tile's buffer is free now (the other one holds, or is receiving, the next tile's K-tile 0).
        // K / 64 is even (launcher), so that is always buffer 1, and the 24 KiB of LDS that lie unused behind the ring
        // follow it directly: 88 KiB = 11 KiB per wave (32-row passes for fp32 outputs, 64 / 128 rows for 16-bit / GeGLU).
        char* const stage_buf = smem + C::TILE_BYTES;
        // The epilogue's lane-derived addresses are computed from a lane id the compiler cannot hoist out of the tile
        // loop: hoisted, they stay live across the K loop, the kernel spills, and the reloads' compiler-inserted
        // vmcnt(0) drains the hand-counted LDS-DMA stream (cdna_hip_programming.md, attention pitfalls).
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int lane = lane_e, r = lane & 15, g = lane >> 4;

        // 16-bit-residual LayerNorm producers: the wide read-back form (producer16_epilogue_wide above)
        constexpr bool WIDE = ln_res_is_16(LN) && EPI == VF_EPI_RES_F32 && (VF_G8X_WIDE != 0);
        if constexpr (WIDE) {
            producer16_epilogue_wide<DT, LN, (C::TILE_BYTES + C::SPARE_BYTES) / C::NW>(
                acc, stage_buf + wave * ((C::TILE_BYTES + C::SPARE_BYTES) / C::NW), side, wn * 64, (int64_t)m0 + wm * 128,
                n0 + wn * 64, bias != nullptr, out, ldo, M, N, ln);
        } else {
        // ---- epilogue: each wave stages its 128 x 64 block through its own 8 KiB slice of the free half of the ring and
        // writes whole rows, 16 bytes per lane; the fp32 residual rows of pass p+1 are requested while pass p goes through
        // LDS (same scheme as gemm8_kernel).
        constexpr bool OUT_F32 = (EPI == VF_EPI_F32 || EPI == VF_EPI_RES_F32 || EPI == VF_EPI_GELU_F32);
        constexpr int ES = OUT_F32 ? 4 : 2;
        constexpr int WT_M = 128, WT_N = 64;
        constexpr int WT_NO = (EPI == VF_EPI_GEGLU_BF16) ? WT_N / 2 : WT_N;
        constexpr int PITCH = WT_NO * ES + 16;
        constexpr int REGION = (C::TILE_BYTES + C::SPARE_BYTES) / C::NW;      // 11 KiB per wave: half the ring + the spare
        constexpr int RP_FIT = (((REGION / PITCH) < WT_M ? (REGION / PITCH) : WT_M) / 16) * 16;
        // fp32-residual epilogues: 16-row passes (two residual buffers of 16 registers; with 32-row passes the persistent
        // loop spills)
        // (a 16-bit residual, VF_LN_PRODUCER_R16, is half the registers: 32-row passes like the one-shot kernel)
        constexpr bool R16 = ln_res_is_16(LN), T16 = LN == VF_LN_PRODUCER_T16;
        constexpr int RP = (EPI == VF_EPI_RES_F32 && !R16 && RP_FIT > 16) ? 16 : RP_FIT;
        constexpr int IMP = RP / 16, NPASS = (TM + IMP - 1) / IMP;
        constexpr int CR = WT_NO * ES / 16, RI = 64 / CR, NI = RP / RI;
        static_assert(RP >= 16 && CR >= 1 && CR <= 64 && 64 % CR == 0, "epilogue geometry");
        constexpr bool RES = (EPI == VF_EPI_RES_F32);
        const int n_out_total = (EPI == VF_EPI_GEGLU_BF16) ? N / 2 : N;
        const int64_t mw0 = m0 + wm * WT_M;
        const int nw0 = n0 + wn * WT_N;
        const int no0 = (EPI == VF_EPI_GEGLU_BF16) ? nw0 / 2 : nw0;
        const int ep_row = lane / CR, ep_col = no0 + (lane % CR) * (16 / ES);
        // Addresses of this lane's read-back items.  Item j = pass * NI + k is row j * RI + ep_row of the wave tile, so every
        // pointer is "first row + j * (RI rows)": ONE 64-bit multiply per lane, the per-item steps are wave-uniform scalars
        // (a multiply per item costs quarter-rate v_mul_lo_u32 / v_mad_u64_u32 pairs: 192 of the 766 VALU instructions of
        // the fp32-residual epilogue before this).  Rows past M read the last row instead (never stored).
        const int rows_left = (int)(M - mw0) - ep_row;              // item j is a row of the matrix iff j * RI < rows_left
        const int64_t row0 = mw0 + ep_row;
        const int colc = ep_col < N ? ep_col : N - 4;
        // the residual: fp32 rows, or (VF_LN_PRODUCER_R16) the 16-bit copy of a stream, converted where it is added
        using res_t = typename std::conditional<R16, u32x2_t, f32x4_t>::type;
        const char* const res_base = R16 ? reinterpret_cast<const char*>(ln.res16) : reinterpret_cast<const char*>(res);
        const int64_t res_ld = R16 ? ln.ldr16 : ldr;
        constexpr int RES_ES = R16 ? 2 : 4;
        const char* const res_p = RES ? res_base + (row0 * res_ld + colc) * RES_ES : nullptr;
        const char* const res_last = RES ? res_base + ((int64_t)(M - 1) * res_ld + colc) * RES_ES : nullptr;
        const int64_t res_step = (int64_t)RI * res_ld * RES_ES;
        char* const out_p = reinterpret_cast<char*>(out) + (row0 * ldo + ep_col) * ES;
        const int64_t out_step = (int64_t)RI * ldo * ES;
        unsigned short* const o16_p = ln_is_producer(LN) ? reinterpret_cast<unsigned short*>(ln.out16) + row0 * ln.ld16 + ep_col : nullptr;
        const int64_t o16_step = (int64_t)RI * ln.ld16;
        float* const part_p = ln_is_producer(LN) ? ln.part_stats + ((int64_t)(ep_col >> 5) * ln.rows + row0) * 2 : nullptr;
        // the items are visited in increasing j, so each pointer is a running one: p += step per item (one 64-bit add)
        // instead of base + j * step (hipcc multiplies per item otherwise: 81 quarter-rate v_mad_u64_u32 in this epilogue)
        const char* res_run = res_p;
        char* out_run = out_p;
        unsigned short* o16_run = o16_p;
        float* part_run = part_p;
        unsigned short* t16_run = T16 ? ln.t16_out + row0 * ln.ldt16 + ep_col : nullptr;
        const int64_t t16_step = (int64_t)RI * ln.ldt16;
        res_t rbuf[2][RES ? NI : 1];
        auto load_res_pass = [&](int ps, res_t (&dst)[RES ? NI : 1]) {
            if (RES) {
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int j = ps * NI + k;
                    const char* rp = (j * RI < rows_left) ? res_run : res_last;
                    res_run += res_step;
                    if (dbg & 8) { dst[RES ? k : 0] = res_t{}; continue; }
                    dst[RES ? k : 0] = *reinterpret_cast<const res_t*>(rp);
                }
            }
        };
        auto res_value = [&](res_t v) -> f32x4_t {      // see gemm8_kernel
            if constexpr (T16) return cvt4_16<VF_F16>(v) * ln.res16_scale;
            else if constexpr (R16) {
                if constexpr (DT == VF_F16) return cvt4_16<DT>(v) * ln.res16_scale;
                else return cvt4_16<DT>(v);
            }
            else return v;
        };
        load_res_pass(0, rbuf[0]);
        // bias of the wave's columns, from the side area (requested before the first K-tile)
        f32x4_t bvec[TN];
        if (bias) {
#pragma unroll
            for (int in = 0; in < TN; ++in) {
                const int nl = wn * WT_N + ((EPI == VF_EPI_GEGLU_BF16) ? (in >> 1) * 32 + (in & 1) * 16 + 4 * g : in * 16 + 4 * g);
                bvec[in] = *reinterpret_cast<const f32x4_t*>(side + nl * 4);
            }
        } else {
#pragma unroll
            for (int in = 0; in < TN; ++in) bvec[in] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        // LayerNorm consumer: acc -> rstd[m] * (acc - mean[m] * colsum[n]); the folded bias is `bias`
        f32x4_t svec[LN == VF_LN_CONSUMER ? TN : 1];
        if (LN == VF_LN_CONSUMER) {
#pragma unroll
            for (int in = 0; in < TN; ++in) {
                const int nl = wn * WT_N + ((EPI == VF_EPI_GEGLU_BF16) ? (in >> 1) * 32 + (in & 1) * 16 + 4 * g : in * 16 + 4 * g);
                svec[LN == VF_LN_CONSUMER ? in : 0] = *reinterpret_cast<const f32x4_t*>(side + 1024 + nl * 4);
            }
        }
        // (mean, rstd) of a row group are read from the side area when its accumulators are staged: 2 live registers
        // instead of 16 (the persistent loop has none to spare)
        auto lnv = [&](int in, int im, f32x2_t st) -> f32x4_t {          // st = (-mean * rstd, rstd); see gemm8_kernel
            if (LN == VF_LN_CONSUMER) return acc[in][im] * st[1] + (st[0] * svec[LN == VF_LN_CONSUMER ? in : 0] + bvec[in]);
            return acc[in][im] + bvec[in];
        };
        char* const region = stage_buf + wave * REGION;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            if (ps + 1 < NPASS) load_res_pass(ps + 1, rbuf[(ps + 1) & 1]);
#pragma unroll
            for (int iml = 0; iml < IMP; ++iml) {
                const int im = ps * IMP + iml;
                if (im < TM) {
                    char* rowp = region + (iml * 16 + r) * PITCH;
                    f32x2_t st = {0.f, 1.f};
                    if (LN == VF_LN_CONSUMER) {
                        st = *reinterpret_cast<const f32x2_t*>(side + 2048 + (wm * WT_M + im * 16 + r) * 8);
                        st[0] = -st[0] * st[1];                                   // (-mean * rstd, rstd)
                    }
                    if (EPI == VF_EPI_GEGLU_BF16) {
#pragma unroll
                        for (int ip = 0; ip < TN / 2; ++ip) {
                            const f32x4_t v = lnv(2 * ip, im, st), gt = lnv(2 * ip + 1, im, st);
                            u32x2_t pk;
                            const f32x4_t y = v * gelu_erf4(gt);
                            pk[0] = Op16<DT>::pack2(y[0], y[1]);
                            pk[1] = Op16<DT>::pack2(y[2], y[3]);
                            *reinterpret_cast<u32x2_t*>(rowp + (ip * 16 + 4 * g) * 2) = pk;
                        }
                    } else {
#pragma unroll
                        for (int in = 0; in < TN; ++in) {
                            f32x4_t v = lnv(in, im, st);
                            if (EPI == VF_EPI_GELU_F32 || EPI == VF_EPI_GELU_BF16) {
                                v = gelu_erf4(v);
                            }
                            if (OUT_F32) {
                                *reinterpret_cast<f32x4_t*>(rowp + (in * 16 + 4 * g) * 4) = v;
                            } else {
                                u32x2_t pk;
                                pk[0] = Op16<DT>::pack2(v[0], v[1]);
                                pk[1] = Op16<DT>::pack2(v[2], v[3]);
                                *reinterpret_cast<u32x2_t*>(rowp + (in * 16 + 4 * g) * 2) = pk;
                            }
                        }
                    }
                }
            }
            // (2) read the slice back row-wise: all LDS reads first (unconditional: every row lies inside the slice), then
            // the predicated stores, so that no store waits behind a per-row ds_read round trip
            constexpr int KB = RES ? 4 : NI;          // read-back batch (the residual epilogue has fewer registers to spare)
#pragma unroll
            for (int k0 = 0; k0 < NI; k0 += KB) {
                u32x4_t dd[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k)
                    if (k0 + k < NI)
                        dd[k] = *reinterpret_cast<const u32x4_t*>(region + ((k0 + k) * RI + ep_row) * PITCH + (lane % CR) * 16);
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    if (k0 + k >= NI) continue;
                    const int j = ps * NI + k0 + k;                  // row j * RI + ep_row of the wave tile
                    u32x4_t d = dd[k];
                    if (RES) {
                        f32x4_t f = __builtin_bit_cast(f32x4_t, d);
                        f += res_value(rbuf[ps & 1][RES ? k0 + k : 0]);
                        d = __builtin_bit_cast(u32x4_t, f);
                    }
                    const bool ok = j * RI + ep_row < WT_M && j * RI < rows_left && ep_col < n_out_total;
                    if (ln_is_producer(LN) && OUT_F32) {
                        ln_emit<DT>(__builtin_bit_cast(f32x4_t, d), ok, o16_run, part_run, lane, ln.x16_scale, !(dbg & 2), !(dbg & 4),
                                    (T16 && ln.t16_out) ? t16_run : nullptr, ln.t16_scale);
                        o16_run += o16_step;
                        part_run += RI * 2;
                        if (T16) t16_run += t16_step;
                    }
                    // a LayerNorm producer whose fp32 result has no reader (only its 16-bit copy and statistics do) passes
                    // out = NULL: the 16-byte store -- 4 of the 10 bytes the epilogue moves per element -- is dropped
                    if (ok && (!ln_is_producer(LN) || out != nullptr) && !(dbg & 1)) *reinterpret_cast<u32x4_t*>(out_run) = d;
                    out_run += out_step;
                }
            }
        }

        }   // !WIDE

        // every wave's staging reads are done before the next tile's K-tile 1 is requested into this buffer
#ifdef VF_G8_PROF
        pt1 = G8X_T(); px[3] += pt1 - pt0; pt0 = pt1;        // 3: epilogue passes of this wave
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef VF_G8_PROF
        pt1 = G8X_T(); px[4] += pt1 - pt0; px[5] += 1;       // 4: waiting for the other waves' epilogues
#endif
        g0 += nkt;
    }
#ifdef VF_G8_PROF
    if (vf_g8_prof && (tid & 63) == 0 && (wave == 0 || wave == 4) && (bid & 7) == 0) {
        for (int i = 0; i < 6; ++i) vf_g8_prof[((bid >> 3) * 2 + (wave >> 2)) * 8 + i] = px[i];
        vf_g8_prof[((bid >> 3) * 2 + (wave >> 2)) * 8 + 6] = __builtin_amdgcn_s_memtime() - pc0;
        vf_g8_prof[((bid >> 3) * 2 + (wave >> 2)) * 8 + 7] = __builtin_amdgcn_s_memrealtime() - pr0;
    }
#undef G8X_T
#endif
#undef VF_G8_SYNC_IN
#undef VF_G8_SYNC_OUT
#undef VF_G8_MMA
#undef VF_G8X_STORE_PROBE
}

#ifdef VF_TUNING   // gemm8y_kernel: the epilogue inside the K loop -- ruled out by the store probe before it ran (profiles/r06_b)
#include "tuning/gemm8y.inc"
#endif

#ifdef VF_TUNING   // gemm4_kernel: two independent 4-wave blocks per CU, 15-20 % slower (profiles/r04_a)
#include "tuning/gemm4.inc"
#endif

#ifdef VF_TUNING   // gemm8p_kernel: prefetch-all variant of the two-group kernel, mixed results (DESIGN_HISTORY.md)
#include "tuning/gemm8p.inc"
#endif

// Shape-generic fallback (any K % 8 == 0): 64x64 tile, fp32 FMA out of LDS.  Same lane->output
// ownership as the MFMA kernel so the epilogues are shared.  Only small/odd shapes come here.
template <int EPI, int DT = VF_BF16>
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
            sA[row][kc + e] = k < K ? Op16<DT>::to_f32(A[(int64_t)gm * lda + k]) : 0.f;
            sW[row][kc + e] = k < K ? Op16<DT>::to_f32(W[(int64_t)gn * K + k]) : 0.f;
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
            epilogue_store<EPI, DT>(acc[2 * ip], acc[2 * ip + 1], m, nb, nb + 16, n0 / 2 + ip * 16 + 4 * ng, bias, res, ldr, out, ldo);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + 16 * j + 4 * ng;
            if (nb >= N) continue;
            epilogue_store<EPI, DT>(acc[j], acc[j], m, nb, nb, nb, bias, res, ldr, out, ldo);
        }
    }
}

#ifdef VF_TUNING   // xs_gemm_kernel: projection phase of a fused seq2reg block, 5-7 % slower (profiles/r03_e)
#include "tuning/gemm_xs.inc"
#endif

// The product library instantiates only the configurations pick_variant() can select.
using CfgA = Cfg<128, 128, 2, 2, 2>;       // 64 KiB, 4 waves, 2 blocks/CU
using CfgE = Cfg<64, 64, 2, 2, 4>;         // 64 KiB, small-M shapes, 2 blocks/CU
#ifdef VF_TUNING                            // tile sweep of scripts/gemm_bench.py (all measured equal or slower)
using CfgB = Cfg<256, 256, 2, 4, 2>;       // 128 KiB, 8 waves, wave tile 128x64, one barrier per K-tile (round 1's big tile)
using CfgJ = Cfg<128, 128, 2, 2, 3, 32>;   // 48 KiB, 4 waves, BK=32, 2 tiles in flight, 3 blocks/CU
using CfgC = Cfg<256, 128, 4, 2, 3>;       // 144 KiB, 8 waves, wave tile 64x64, 2 tiles in flight
using CfgD = Cfg<256, 256, 2, 4, 4, 32>;   // 128 KiB, 8 waves, BK=32, 3 tiles in flight
using CfgF = Cfg<128, 256, 2, 4, 3>;       // 144 KiB, 8 waves, wave tile 64x64
using CfgG = Cfg<128, 128, 2, 2, 4, 32>;   // 64 KiB, 4 waves, BK=32, 3 tiles in flight, 2 blocks/CU
using CfgH = Cfg<256, 128, 2, 2, 3, 32>;   // 72 KiB, 4 waves (wave tile 128x64), BK=32, 2 blocks/CU
using CfgI = Cfg<128, 256, 2, 2, 3, 32>;   // 72 KiB, 4 waves (wave tile 64x128), BK=32, 2 blocks/CU
#endif

template <class C, int EPI, int DT = VF_BF16, int DBG = 0, int LN = VF_LN_NONE>
int launch_cfg(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
               int64_t ldo, int M, int N, int K, hipStream_t st, LnArgs ln = LnArgs{}) {
    static bool attr_set[VF_MAX_DEVICES] = {};    // per (config, epilogue) instantiation AND per device
    auto kern = gemm_mfma_kernel<C, EPI, DT, DBG, LN>;
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_gemm_bf16: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    const int tiles_m = (M + C::BM - 1) / C::BM, tiles_n = (N + C::BN - 1) / C::BN;
    const int n_blocks = tiles_m * tiles_n;
    const int group_m = 8;      // m-panels per L2 group; 2 / 4 / 16 measured equal or slower for both tile sizes
    vf_note_kernel(0, C::BM == 128 ? "gemm_mfma_kernel<128x128>" : C::BM == 64 ? "gemm_mfma_kernel<64x64>" : "gemm_mfma_kernel<other>");
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(C::THREADS), C::LDS_BYTES, st, (const unsigned short*)A, lda,
                       (const unsigned short*)W, bias, res, ldr, out, ldo, M, N, K, tiles_n, n_blocks, group_m, ln);
    VF_CHECK_LAUNCH("vf_gemm_bf16");
    return VF_OK;
}

#ifdef VF_TUNING
template <class C, int EPI, int DT = VF_BF16>
int launch_persist(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
                   int64_t ldo, int M, int N, int K, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm_persist_kernel<C, EPI, DT>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_gemm_bf16: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return VF_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int tiles_m = (M + C::BM - 1) / C::BM, tiles_n = (N + C::BN - 1) / C::BN;
    const int n_tiles = tiles_m * tiles_n;
    const int slots = 256 * (163840 / C::LDS_BYTES >= 2 ? 2 : 1);      // resident blocks: CUs x blocks per CU by LDS
    const int grid = n_tiles < slots ? n_tiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, st, (const unsigned short*)A, lda,
                       (const unsigned short*)W, bias, res, ldr, out, ldo, M, N, K, tiles_n, n_tiles);
    VF_CHECK_LAUNCH("vf_gemm_bf16");
    return VF_OK;
}

#endif  // VF_TUNING

template <int EPI, int DT = VF_BF16, int LN = VF_LN_NONE>
int launch_gemm8(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
                 int64_t ldo, int M, int N, int K, hipStream_t st, LnArgs ln = LnArgs{}) {
    static bool attr_set[VF_MAX_DEVICES] = {};
    auto kern = gemm8_kernel<EPI, DT, LN>;
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                Cfg8::LDS_BYTES + Cfg8::SIDE_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_gemm_bf16: cannot reserve %d bytes of LDS", Cfg8::LDS_BYTES + Cfg8::SIDE_BYTES);
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int n_blocks = tiles_m * tiles_n;
    int group_m = 8;
#ifdef VF_TUNING
    if (const char* e = getenv("VF_G8_GROUP_M")) group_m = atoi(e);     // tile-walk sweep (scripts/gemm_bench.py)
    if (const char* e = getenv("VF_G8_DBG")) group_m |= atoi(e) << 8;   // epilogue cost-centre probes
    if (const char* e = getenv("VF_G8_STAGGER")) group_m |= atoi(e) << 16;
#endif
    vf_note_kernel(0, "gemm8_kernel");
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(512), Cfg8::LDS_BYTES + Cfg8::SIDE_BYTES, st, (const unsigned short*)A, lda,
                       (const unsigned short*)W, bias, res, ldr, out, ldo, M, N, K, tiles_n, n_blocks, group_m, ln);
    VF_CHECK_LAUNCH("vf_gemm_bf16");
    return VF_OK;
}

template <int EPI, int DT = VF_BF16, int LN = VF_LN_NONE>
int launch_gemm8x(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
                  int64_t ldo, int M, int N, int K, hipStream_t st, LnArgs ln = LnArgs{}) {
    constexpr int LDS = Cfg8::LDS_BYTES + Cfg8::SPARE_BYTES + 2 * Cfg8::SIDE_BYTES;      // 160 KiB: all of a CU's LDS
    static bool attr_set[VF_MAX_DEVICES] = {};
    static int n_cu[VF_MAX_DEVICES] = {};
    auto kern = gemm8x_kernel<EPI, DT, LN>;
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
            hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_gemm_bf16: cannot reserve %d bytes of LDS", LDS);
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    int cus = 256;
    if (dev >= 0) {
        if (n_cu[dev] == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            n_cu[dev] = v;
        }
        cus = n_cu[dev];
    }
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int n_tiles = tiles_m * tiles_n;
    const int grid = n_tiles < cus ? n_tiles : cus;              // one resident block per CU
    int group_m = 8;
#ifdef VF_TUNING
    if (const char* e = getenv("VF_G8X_GROUP_M")) group_m = atoi(e);    // tile-walk sweep (scripts/gemm4_probe.py)
    if (const char* e = getenv("VF_G8_DBG")) group_m |= atoi(e) << 8;   // epilogue cost-centre probes
#endif
    vf_note_kernel(0, "gemm8x_kernel");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, (const unsigned short*)A, lda, (const unsigned short*)W, bias,
                       res, ldr, out, ldo, M, N, K, tiles_n, n_tiles, group_m, ln);
    VF_CHECK_LAUNCH("vf_gemm_bf16");
    return VF_OK;
}

#ifdef VF_TUNING
template <int EPI, int DT = VF_BF16, int LN = VF_LN_NONE>
int launch_gemm4(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
                 int64_t ldo, int M, int N, int K, hipStream_t st, LnArgs ln = LnArgs{}) {
    constexpr int LDS = Cfg4::LDS_BYTES;                         // 80 KiB: two blocks per CU
    static bool attr_set[VF_MAX_DEVICES] = {};
    static int n_cu[VF_MAX_DEVICES] = {};
    auto kern = gemm4_kernel<EPI, DT, LN>;
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
            hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_gemm: cannot reserve %d bytes of LDS", LDS);
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    int cus = 256;
    if (dev >= 0) {
        if (n_cu[dev] == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            n_cu[dev] = v;
        }
        cus = n_cu[dev];
    }
    const int tiles_m = (M + Cfg4::BM - 1) / Cfg4::BM, tiles_n = (N + Cfg4::BN - 1) / Cfg4::BN;
    const int n_tiles = tiles_m * tiles_n;
    int slots = 2 * cus;                                         // two resident blocks per CU
    int lds = LDS;
    int group_m = 8;
#ifdef VF_TUNING
    if (const char* e = getenv("VF_G4_GROUP_M")) group_m = atoi(e);
    if (const char* e = getenv("VF_G4_STAGGER")) group_m |= atoi(e) << 16;
    if (const char* e = getenv("VF_G4_BPC")) slots = atoi(e) * cus;      // residency probes: blocks per CU, LDS request
    if (const char* e = getenv("VF_G4_LDS")) {
        lds = atoi(e);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
#endif
    const int grid = n_tiles < slots ? n_tiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const unsigned short*)A, lda, (const unsigned short*)W, bias,
                       res, ldr, out, ldo, M, N, K, tiles_n, n_tiles, group_m, ln);
    VF_CHECK_LAUNCH("vf_gemm");
    return VF_OK;
}
#endif  // VF_TUNING

#ifdef VF_TUNING
template <int EPI, int DT = VF_BF16>
int launch_gemm8p(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
                  int64_t ldo, int M, int N, int K, hipStream_t st) {
    constexpr int LDS = Cfg8::LDS_BYTES + 8 * 4096;              // ring + per-wave epilogue scratch = 160 KiB
    static bool attr_set[VF_MAX_DEVICES] = {};
    auto kern = gemm8p_kernel<EPI, DT>;
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
            hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_gemm: cannot reserve %d bytes of LDS", LDS);
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int n_tiles = tiles_m * tiles_n;
    const int grid = n_tiles < 256 ? n_tiles : 256;              // one resident block per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, (const unsigned short*)A, lda, (const unsigned short*)W, bias,
                       res, ldr, out, ldo, M, N, K, tiles_n, n_tiles, 8);
    VF_CHECK_LAUNCH("vf_gemm");
    return VF_OK;
}

#endif  // VF_TUNING

// Tile choice (measured on MI355X, scripts/gemm_bench.py, random data; the cost model inside reproduces every measured
// ordering): grids with fewer than 256 128x128 tiles use 64x64 tiles so that all 256 CUs get work; otherwise 256x256
// (one 8-wave block per CU, half the L2 -> LDS bytes per flop) against 128x128 (two 4-wave blocks per CU) by whole
// waves of tiles.
// variant 0 = automatic; 1 / 5 / 20 / 22 force a configuration (tests).  Other numbers exist only under VF_TUNING.
int pick_variant(int M, int N, int K, int epilogue) {
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    if (t128 < 256) return 5;
    // 128x128 (2 blocks per CU, 512 slots) vs the two-group 256x256 kernel (1 block per CU, 256 slots, 4x the work per
    // tile): whole waves of tiles are what a launch pays for, so compare ceil(tiles / slots) x work per wave with the
    // measured per-flop advantage of the big tile (gpurun_out/r2b/gemm_bench.log, 8 genes per step, TFLOP/s 256x256 vs
    // 128x128: gene Wqkv 1312 vs 1121, Wq 1227 vs 982, GeGLU 1127 vs 942, fp32-residual out_proj 940 vs 808 and
    // K = 1024 down-projection 784 vs 692, seq2reg K = 512 Wqkv 985 vs 765, 8192^3 1520 vs 1186).  The same rule keeps
    // the one-gene shapes (M = 10854: 258 big tiles = 2 waves for 1.01 waves of work) and the CRE stream on 128x128.
    (void)K;
    const long waves_small = (t128 + 511) / 512, waves_big = (t256 + 255) / 256;
    const double gain = epilogue == VF_EPI_RES_F32 ? 1.14 : 1.2;
    if (!((double)waves_big * 2.0 / gain < (double)waves_small)) return 1;
    // 16-bit epilogues take the persistent form of the 256x256 kernel (first fill and block launch hidden: seq2reg Wqkv
    // 890 -> 953, GeGLU 893 -> 976, gene Wqkv / Wq / GeGLU +2-3 %, same box).  The fp32 epilogues stay on the one-shot
    // kernel: staged through half the ring (16-row passes) the seq2reg N = 512 producers lose 8 %, the gene ones gain
    // nothing.  VF_GEMM_PERSIST=0 switches the persistent form off (A/B runs).
    static const int persist = vf_tuning_env("VF_GEMM_PERSIST", 1);
    const bool out16 = epilogue == VF_EPI_BF16 || epilogue == VF_EPI_GEGLU_BF16 || epilogue == VF_EPI_GELU_BF16;
    return (persist && (out16 || persist >= 2) && K % 128 == 0) ? 22 : 20;      // K / 64 even: see the kernel's staging
}

#ifdef VF_TUNING
// Variant 23 = gemm8y_kernel (tuning library only, VF_GEMM8Y=1): the epilogue inside the K loop.  Never validated on hardware:
// the store probe of gemm8x_kernel (VF_G8_DBG bits 64 / 128) showed that its 16 in-loop stores per wave and tile cost as much as
// the serial epilogue they would replace, and the kernel spills 28-83 registers.
static inline bool gemm8y_ok(int K, int epilogue) {
    static const int on = vf_tuning_env("VF_GEMM8Y", 0);
    return on && K % 128 == 0 && K >= 256 && (epilogue == VF_EPI_BF16 || epilogue == VF_EPI_GEGLU_BF16);
}
#endif

template <int EPI, int DT>
int launch_gemm(const void* A, int64_t lda, const void* W, const float* bias, const float* res, int64_t ldr, void* out,
                int64_t ldo, int M, int N, int K, int variant, hipStream_t st) {
    if (K % 64 != 0) {
        dim3 grid((N + 63) / 64, (M + 63) / 64);
        vf_note_kernel(0, "gemm_generic_kernel");
        hipLaunchKernelGGL((gemm_generic_kernel<EPI, DT>), grid, dim3(256), 0, st, (const unsigned short*)A, lda,
                           (const unsigned short*)W, bias, res, ldr, out, ldo, M, N, K);
        VF_CHECK_LAUNCH("vf_gemm_bf16");
        return VF_OK;
    }
    if (variant == 0) {
        variant = pick_variant(M, N, K, EPI);
#ifdef VF_TUNING
        if (variant == 22 && gemm8y_ok(K, EPI)) variant = 23;
#endif
    }
    switch (variant) {
#ifdef VF_TUNING
        case 23:
            if constexpr (EPI == VF_EPI_BF16 || EPI == VF_EPI_GEGLU_BF16) {
                if (K % 128 == 0 && K >= 256) return launch_gemm8y<EPI, DT>(A, lda, W, bias, out, ldo, M, N, K, st);
            }
            break;
#endif
        case 1: return launch_cfg<CfgA, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 5: return launch_cfg<CfgE, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 20: return launch_gemm8<EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 22: if (K % 128 == 0) return launch_gemm8x<EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
                 return launch_gemm8<EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
#ifdef VF_TUNING
        case 30:
            if constexpr (EPI == VF_EPI_BF16 || EPI == VF_EPI_GEGLU_BF16) {
                if (xs_ok(N, K, EPI)) return launch_xs<EPI, DT, VF_LN_NONE>(A, lda, W, bias, out, ldo, M, N, st);
            }
            break;
        case 40: if (K % 128 == 0) return launch_gemm4<EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
                 return launch_gemm8<EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 21:
            if (K < 128) break;
            return launch_gemm8p<EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 2: return launch_cfg<CfgB, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 12: return launch_cfg<CfgJ, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 3: return launch_cfg<CfgC, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 4: return launch_cfg<CfgD, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 6: return launch_cfg<CfgF, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 7: return launch_cfg<CfgG, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 8: return launch_cfg<CfgH, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 9: return launch_cfg<CfgI, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 104: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgA, VF_EPI_BF16, VF_BF16, 4>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 103: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgA, VF_EPI_BF16, VF_BF16, 3>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 204: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgB, VF_EPI_BF16, VF_BF16, 4>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 203: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgB, VF_EPI_BF16, VF_BF16, 3>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 101: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgA, VF_EPI_BF16, VF_BF16, 1>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 102: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgA, VF_EPI_BF16, VF_BF16, 2>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 201: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgB, VF_EPI_BF16, VF_BF16, 1>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 202: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgB, VF_EPI_BF16, VF_BF16, 2>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 401: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgD, VF_EPI_BF16, VF_BF16, 1>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 402: if (EPI == VF_EPI_BF16 && DT == VF_BF16) return launch_cfg<CfgD, VF_EPI_BF16, VF_BF16, 2>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st); break;
        case 10: return launch_persist<CfgA, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
        case 11: return launch_persist<CfgE, EPI, DT>(A, lda, W, bias, res, ldr, out, ldo, M, N, K, st);
#endif  // VF_TUNING
        default: break;
    }
    vf_set_error("vf_gemm_bf16_ex: unknown variant %d", variant);
    return VF_ERR_INVALID_ARG;
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

template <int DT>
static int gemm_dispatch(const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                         int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int epilogue, int variant, void* stream) {
    VF_REQUIRE(A && W && out, "vf_gemm: null pointer");
    VF_REQUIRE(M >= 0 && N > 0 && K > 0, "vf_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    VF_REQUIRE(K % 8 == 0 && N % 8 == 0, "vf_gemm: K and N must be multiples of 8 (K=%d N=%d)", K, N);
    VF_REQUIRE(lda % 8 == 0 && lda >= K, "vf_gemm: lda=%lld must be >= K and a multiple of 8", (long long)lda);
    VF_REQUIRE(ldo % 8 == 0 || (ldo % 4 == 0 && (epilogue == VF_EPI_F32 || epilogue == VF_EPI_RES_F32 || epilogue == VF_EPI_GELU_F32)),
               "vf_gemm: ldo=%lld must keep rows 16-byte aligned (multiple of 8 for 16-bit, 4 for fp32 outputs)", (long long)ldo);
    VF_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)out % 16 == 0),
               "vf_gemm: pointers must be 16-byte aligned");
    VF_REQUIRE(variant >= 0 && variant <= 999, "vf_gemm_ex: variant %d out of range", variant);
    if (M == 0) return VF_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case VF_EPI_BF16: return launch_gemm<VF_EPI_BF16, DT>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, variant, st);
        case VF_EPI_F32: return launch_gemm<VF_EPI_F32, DT>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, variant, st);
        case VF_EPI_RES_F32:
            VF_REQUIRE(residual && ldr % 4 == 0 && ((uintptr_t)residual % 16 == 0), "vf_gemm: residual epilogue needs an aligned residual");
            return launch_gemm<VF_EPI_RES_F32, DT>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, variant, st);
        case VF_EPI_GEGLU_BF16:
            VF_REQUIRE(N % 32 == 0, "vf_gemm: GEGLU epilogue needs N %% 32 == 0 (N=%d)", N);
            return launch_gemm<VF_EPI_GEGLU_BF16, DT>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, variant, st);
        case VF_EPI_GELU_F32: return launch_gemm<VF_EPI_GELU_F32, DT>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, variant, st);
        case VF_EPI_GELU_BF16: return launch_gemm<VF_EPI_GELU_BF16, DT>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, variant, st);
        default: VF_REQUIRE(false, "vf_gemm: unknown epilogue %d", epilogue);
    }
    return VF_OK;
}

// GEMM with a LayerNorm folded around it (LnArgs above): consumer mode when row_stats / colsum are given (16-bit
// epilogues), producer mode when out16 / part_stats are given (fp32 epilogues); MFMA path only (K % 64 == 0).
template <int EPI, int DT, int LN>
static int launch_gemm_ln(const void* A, int64_t lda, const void* W, const float* bias, const float* residual, int64_t ldr,
                          void* out, int64_t ldo, int M, int N, int K, const LnArgs& ln, hipStream_t st) {
#ifdef VF_TUNING
    if constexpr (LN == VF_LN_CONSUMER) {
        // A/B switch (off by default: measured 906-942 against 955-1009 TFLOP/s for the persistent tile kernel on the
        // seq2reg shapes, profiles/r03_e_xs_gemm.log): 1 = grids of >= 256 blocks, 2 = every K = 512 consumer (tests)
        static const int xs = vf_tuning_env("VF_GEMM_XS", 0);
        if (xs && xs_ok(N, K, EPI) && (xs >= 2 || M >= 256 * 256))
            return launch_xs<EPI, DT, LN>(A, lda, W, bias, out, ldo, M, N, st, ln);
    }
#endif
    int variant = pick_variant(M, N, K, EPI);
    if constexpr (LN == VF_LN_PRODUCER_T16) {
        // the fp16-trunk down-projection: persistent form when no fp32 rows are stored (VF_GEMM_PERSIST_T16, default on: gene down-projection 829 -> 857, seq2reg 858 -> 897 TFLOP/s)
        static const int pt16 = vf_tuning_env("VF_GEMM_PERSIST_T16", 1);
        if (variant == 20 && K % 128 == 0 && pt16 && out == nullptr) variant = 22;
    }
    if constexpr (LN == VF_LN_PRODUCER_R16) {
        // A producer whose residual is a 16-bit stream copy fits the persistent form with the one-shot kernel's 32-row
        // passes (half the residual registers of the fp32 one): first fill and block hand-over hidden, gene out-projection
        // 428-439 -> 408-414 us, seq2reg out-projection 662-675 -> 621-636 us (profiles/r03_v_persist_r16_ab.log).
        // VF_GEMM_PERSIST_R16 = 1 (default): when no fp32 rows are stored either (the attention out-projections),
        // 2: every such producer, 0: never.
        static const int pr16 = vf_tuning_env("VF_GEMM_PERSIST_R16", 1);
        if (variant == 20 && K % 128 == 0 && (pr16 >= 2 || (pr16 == 1 && out == nullptr))) variant = 22;
    }
#ifdef VF_TUNING
    {
        // two 4-wave blocks per CU (gemm4_kernel) instead of the 8-wave 256x256 kernels: VF_GEMM4 bit mask, 1 = consumers,
        // 2 = 16-bit-residual producers (R16 / T16), 4 = fp32-residual / plain producers
        static const int g4 = getenv("VF_GEMM4") ? atoi(getenv("VF_GEMM4")) : 0;
        const int bit = LN == VF_LN_CONSUMER ? 1 : (ln_res_is_16(LN) ? 2 : 4);
        if ((variant == 20 || variant == 22) && K % 128 == 0 && (g4 & bit))
            return launch_gemm4<EPI, DT, LN>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st, ln);
    }
#endif
#ifdef VF_TUNING
    if constexpr (LN == VF_LN_CONSUMER && (EPI == VF_EPI_BF16 || EPI == VF_EPI_GEGLU_BF16)) {
        if (variant == 22 && gemm8y_ok(K, EPI)) return launch_gemm8y<EPI, DT, LN>(A, lda, W, bias, out, ldo, M, N, K, st, ln);
    }
#endif
    switch (variant) {
        case 1: return launch_cfg<CfgA, EPI, DT, 0, LN>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st, ln);
        case 5: return launch_cfg<CfgE, EPI, DT, 0, LN>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st, ln);
        case 22: return launch_gemm8x<EPI, DT, LN>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st, ln);
        default: return launch_gemm8<EPI, DT, LN>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, st, ln);
    }
}

template <int DT>
static int gemm_ln_dispatch(const void* A, int64_t lda, const void* W, const float* bias, const void* residual, int64_t ldr,
                            int residual_dtype, void* out, int64_t ldo, int M, int N, int K, int epilogue,
                            const float* row_stats, const float* colsum, void* out16, int64_t ld16, float* part_stats,
                            float x16_scale, float res16_scale, void* stream, bool trunk = false, void* t16_out = nullptr,
                            int64_t ldt16 = 0, float t16_scale = 1.0f) {
    VF_REQUIRE(A && W && (out || (out16 && part_stats)), "vf_gemm_ln: null pointer");
    VF_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 64 == 0 && N % 8 == 0, "vf_gemm_ln: needs K %% 64 == 0, N %% 8 == 0 (N=%d K=%d)", N, K);
    VF_REQUIRE(lda % 8 == 0 && lda >= K, "vf_gemm_ln: lda=%lld must be >= K and a multiple of 8", (long long)lda);
    VF_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)out % 16 == 0),
               "vf_gemm_ln: pointers must be 16-byte aligned");
    const bool consumer = row_stats && colsum, producer = out16 && part_stats;
    VF_REQUIRE(consumer != producer, "vf_gemm_ln: give (row_stats, colsum) OR (out16, part_stats)");
    if (M == 0) return VF_OK;
    LnArgs ln{};
    ln.x16_scale = 1.0f;
    ln.res16_scale = 1.0f;
    hipStream_t st = (hipStream_t)stream;
    if (consumer) {
        VF_REQUIRE(out, "vf_gemm_ln: the consumer needs an output");
        VF_REQUIRE(((uintptr_t)row_stats % 8 == 0) && ((uintptr_t)colsum % 16 == 0), "vf_gemm_ln: misaligned statistics");
        VF_REQUIRE(ldo % 8 == 0, "vf_gemm_ln: ldo=%lld must be a multiple of 8", (long long)ldo);
        ln.row_stats = row_stats;
        ln.colsum = colsum;
        if (epilogue == VF_EPI_BF16)
            return launch_gemm_ln<VF_EPI_BF16, DT, VF_LN_CONSUMER>(A, lda, W, bias, nullptr, 0, out, ldo, M, N, K, ln, st);
        if (epilogue == VF_EPI_F32)       // fp32 rows out (ABI 10: attention logits against a few distinct keys, vf_softmax_counted)
            return launch_gemm_ln<VF_EPI_F32, DT, VF_LN_CONSUMER>(A, lda, W, bias, nullptr, 0, out, ldo, M, N, K, ln, st);
        VF_REQUIRE(epilogue == VF_EPI_GEGLU_BF16 && N % 32 == 0, "vf_gemm_ln: consumer epilogues are BF16, F32 and GEGLU_BF16");
        return launch_gemm_ln<VF_EPI_GEGLU_BF16, DT, VF_LN_CONSUMER>(A, lda, W, bias, nullptr, 0, out, ldo, M, N, K, ln, st);
    }
    VF_REQUIRE(ld16 % 4 == 0 && ld16 >= N && ((uintptr_t)out16 % 8 == 0) && ((uintptr_t)part_stats % 8 == 0) && ldo % 4 == 0,
               "vf_gemm_ln: producer outputs must keep 8-byte alignment (ld16=%lld)", (long long)ld16);
    VF_REQUIRE(N % 32 == 0, "vf_gemm_ln: a producer needs N %% 32 == 0 (whole 32-column parts; N=%d)", N);
    VF_REQUIRE(x16_scale > 0.f && (DT == VF_F16 || x16_scale == 1.0f), "vf_gemm_ln: x16_scale must be > 0 (and 1 for bf16 streams)");
    ln.out16 = out16;
    ln.part_stats = part_stats;
    ln.ld16 = ld16;
    ln.rows = M;
    ln.x16_scale = x16_scale;
    if (epilogue == VF_EPI_F32)
        return launch_gemm_ln<VF_EPI_F32, DT, VF_LN_PRODUCER>(A, lda, W, bias, nullptr, 0, out, ldo, M, N, K, ln, st);
    VF_REQUIRE(epilogue == VF_EPI_RES_F32 && residual, "vf_gemm_ln: producer epilogues are F32 and RES_F32 (with a residual)");
    if (residual_dtype == VF_F32) {
        VF_REQUIRE(ldr % 4 == 0 && ((uintptr_t)residual % 16 == 0), "vf_gemm_ln: misaligned fp32 residual");
        return launch_gemm_ln<VF_EPI_RES_F32, DT, VF_LN_PRODUCER>(A, lda, W, bias, (const float*)residual, ldr, out, ldo, M, N, K, ln, st);
    }
    if (trunk) {
        // the layer trunk as a scaled fp16 copy, whatever the operand type (VF_LN_PRODUCER_T16)
        VF_REQUIRE(residual_dtype == VF_F16, "vf_gemm_ln_t16: the trunk residual is fp16");
        VF_REQUIRE(ldr % 4 == 0 && ((uintptr_t)residual % 8 == 0) && res16_scale > 0.f, "vf_gemm_ln_t16: misaligned fp16 residual");
        VF_REQUIRE(!t16_out || (ldt16 % 4 == 0 && ldt16 >= N && ((uintptr_t)t16_out % 8 == 0) && t16_scale > 0.f),
                   "vf_gemm_ln_t16: the fp16 trunk output must keep 8-byte alignment (ldt16=%lld)", (long long)ldt16);
        ln.res16 = (const unsigned short*)residual;
        ln.ldr16 = ldr;
        ln.res16_scale = res16_scale;
        ln.t16_out = (unsigned short*)t16_out;
        ln.ldt16 = ldt16;
        ln.t16_scale = t16_scale;
        return launch_gemm_ln<VF_EPI_RES_F32, DT, VF_LN_PRODUCER_T16>(A, lda, W, bias, (const float*)residual, ldr, out, ldo, M, N, K, ln, st);
    }
    VF_REQUIRE(residual_dtype == DT, "vf_gemm_ln: a 16-bit residual must have the operand type");
    VF_REQUIRE(ldr % 4 == 0 && ((uintptr_t)residual % 8 == 0) && res16_scale > 0.f, "vf_gemm_ln: misaligned 16-bit residual");
    VF_REQUIRE(DT == VF_F16 || res16_scale == 1.0f, "vf_gemm_ln: res16_scale must be 1 for bf16 streams");
    ln.res16 = (const unsigned short*)residual;
    ln.ldr16 = ldr;
    ln.res16_scale = res16_scale;
    // `res` must be non-null for the RES_F32 epilogue's own checks downstream; it is never dereferenced in R16 kernels
    return launch_gemm_ln<VF_EPI_RES_F32, DT, VF_LN_PRODUCER_R16>(A, lda, W, bias, (const float*)residual, ldr, out, ldo, M, N, K, ln, st);
}

extern "C" int vf_gemm_ln(const void* A, int64_t lda, const void* W, const float* bias, const void* residual, int64_t ldr,
                          int residual_dtype, void* out, int64_t ldo, int M, int N, int K, int epilogue, int operand_dtype,
                          const float* row_stats, const float* colsum, void* out16, int64_t ld16, float* part_stats,
                          float x16_scale, float res16_scale, void* stream) {
    if (operand_dtype == VF_BF16)
        return gemm_ln_dispatch<VF_BF16>(A, lda, W, bias, residual, ldr, residual_dtype, out, ldo, M, N, K, epilogue, row_stats,
                                         colsum, out16, ld16, part_stats, x16_scale, res16_scale, stream);
    VF_REQUIRE(operand_dtype == VF_F16, "vf_gemm_ln: operand_dtype must be VF_BF16 or VF_F16");
    return gemm_ln_dispatch<VF_F16>(A, lda, W, bias, residual, ldr, residual_dtype, out, ldo, M, N, K, epilogue, row_stats,
                                    colsum, out16, ld16, part_stats, x16_scale, res16_scale, stream);
}

extern "C" int vf_gemm_ln_t16(const void* A, int64_t lda, const void* W, const float* bias, const void* residual_f16,
                              int64_t ldr, float res_scale, void* out, int64_t ldo, int M, int N, int K, int operand_dtype,
                              void* out16, int64_t ld16, float* part_stats, float x16_scale, void* t16_out, int64_t ldt16,
                              float t16_scale, void* stream) {
    VF_REQUIRE(residual_f16, "vf_gemm_ln_t16: null residual");
    if (operand_dtype == VF_BF16)
        return gemm_ln_dispatch<VF_BF16>(A, lda, W, bias, residual_f16, ldr, VF_F16, out, ldo, M, N, K, VF_EPI_RES_F32, nullptr,
                                         nullptr, out16, ld16, part_stats, x16_scale, res_scale, stream, true, t16_out, ldt16,
                                         t16_scale);
    VF_REQUIRE(operand_dtype == VF_F16, "vf_gemm_ln_t16: operand_dtype must be VF_BF16 or VF_F16");
    return gemm_ln_dispatch<VF_F16>(A, lda, W, bias, residual_f16, ldr, VF_F16, out, ldo, M, N, K, VF_EPI_RES_F32, nullptr, nullptr,
                                    out16, ld16, part_stats, x16_scale, res_scale, stream, true, t16_out, ldt16, t16_scale);
}

extern "C" int vf_gemm_ln_bf16(const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                               int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int epilogue,
                               const float* row_stats, const float* colsum, void* out16, int64_t ld16,
                               float* part_stats, void* stream) {
    return gemm_ln_dispatch<VF_BF16>(A, lda, W, bias, residual, ldr, VF_F32, out, ldo, M, N, K, epilogue, row_stats, colsum,
                                     out16, ld16, part_stats, 1.0f, 1.0f, stream);
}

extern "C" int vf_gemm_bf16(const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                            int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int epilogue, void* stream) {
    return gemm_dispatch<VF_BF16>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, epilogue, 0, stream);
}

extern "C" int vf_gemm_bf16_ex(const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                               int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int epilogue, int variant,
                               void* stream) {
    return gemm_dispatch<VF_BF16>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, epilogue, variant, stream);
}

extern "C" int vf_gemm_f16(const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                           int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int epilogue, void* stream) {
    return gemm_dispatch<VF_F16>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, epilogue, 0, stream);
}

extern "C" int vf_gemm_f16_ex(const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                              int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int epilogue, int variant,
                              void* stream) {
    return gemm_dispatch<VF_F16>(A, lda, W, bias, residual, ldr, out, ldo, M, N, K, epilogue, variant, stream);
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
