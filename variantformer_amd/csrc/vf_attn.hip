// vf_attn_varlen_fwd: variable-length multi-head attention forward for gfx950 (flash-style, online
// softmax, no S x S matrix in memory).  bf16 Q/K/V/O, fp32 scores / softmax / accumulation.
//
// Roofline: HBM/L2-bound for the short sequences of this model (seq2reg windows <= 200 tokens, gene
// stream <= 201: intensity ~ s/2 flop/B), MFMA-bound only for the long shared-K/V cross attention.
// Algorithmic bytes per (sequence, head) = 2*dh*(2*sq + 2*sk); flops = 4*sq*sk*dh.
//
// Work split: block = (sequence, head, q-tile), 256 threads = 4 waves, each wave owns QG groups of
// 16 queries.  K/V tiles of 64 keys are staged through LDS (two stages, register-prefetched while
// the previous tile is being consumed -- T14 issue-early / write-late).
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16, lane = (r = lane&15, g = lane>>4)):
//   S^T[key][q]  = K . Q^T   A = K rows (ds_read_b128 from the swizzled K tile), B = Q (registers);
//                  the accumulator leaves lane (r,g) with keys 16*kt+4g+reg of QUERY r, so the
//                  softmax row reduction is in-lane plus two cross-lane steps (xor 16, xor 32).
//   O^T[d][q]   += V^T . P^T A = V^T gathered by ds_read_b64_tr_b16 from the row-major V tile,
//                  B = P^T straight from the score registers (bf16-packed).  The contraction index
//                  of one 32-key step is permuted (element j of lane group g <-> key 4g+j for j<4,
//                  16+4g+(j-4) for j>=4) identically on both operands, so no lane movement is needed.
// dh = 48 (the modulator's 1536/32) is padded to 64 only along the QK^T contraction (zero chunks in
// the K tile and zero Q fragments); PV uses exactly dh/16 output tiles.
#include <stdlib.h>
#include <type_traits>
#include "vf_common.h"

namespace {

constexpr int BKV = 64;

// K tile: one row per key, the head dimension padded to 64 (dh <= 64: 128-byte rows, 8 chunks of 16 bytes) or to 128
// (dh = 96 / 128: 256-byte rows, 16 chunks).  Chunk c of row `row` is stored at chunk c ^ swz(row), which makes the
// ds_read_b128 of a 16-lane group (rows r = 0..15, same logical chunk) hit 16 distinct 16-byte bank slots:
// 128-byte rows alternate between the two halves of the 256-byte bank row, so 8 XOR values suffice; 256-byte rows all
// start on the same bank and need all 16.
template <int DH>
struct KLayout {
    static constexpr int ROW = DH <= 64 ? 128 : 256;
    static constexpr int TILE = BKV * ROW;
    static constexpr int KS = DH <= 64 ? 2 : DH / 32;            // 32-deep MFMA steps of the QK^T contraction
    static __device__ __forceinline__ int swz(int row) { return ROW == 128 ? (row >> 1) & 7 : row & 15; }
};

template <int DH>
struct VLayout {
    // V row stride in bytes: a multiple of 32 with an odd 32-byte count, so that the 8 rows a
    // 32-lane half touches in one transposed read fall on 8 distinct 32-byte bank windows.
    static constexpr int ROW = (((DH * 2) / 32) & 1) ? DH * 2 : DH * 2 + 32;
    static constexpr int TILE = BKV * ROW;
};

// For dh < 64 the QK^T contraction is padded to 64, so key masking rides on the MFMA: Q carries 1.0 in pad slot
// d = dh, K carries 0 (valid key) or -32768 (row past the sequence end) there.  Valid scores get exactly +0.0, masked
// scores ~ -7e3 after scaling -> exp2 underflows to 0.  dh = 64 has no spare slot and masks on the VALU.
template <int DH> struct HwMask { static constexpr bool value = DH < 64; };
template <int DT> __device__ __forceinline__ u32x4_t mask_chunk(bool valid) { return (u32x4_t){valid ? 0u : Op16<DT>::NEG_BIG, 0u, 0u, 0u}; }
template <int DT> __device__ __forceinline__ u32x4_t q_pad_chunk(int d0, int dh) { return (u32x4_t){d0 == dh ? Op16<DT>::ONE : 0u, 0u, 0u, 0u}; }

// max over the lanes {l, l^16, l^32, l^48} (the four key sub-blocks g of one query) without LDS round trips:
// v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows on the VALU (lane semantics checked by
// tools/hw_probe.hip), replacing two ds_bpermute on the serial softmax chain.
// max(a, b, c): hipcc selects ONE v_max3_f32 for this shape (the file is built with -fno-honor-nans, so no
// canonicalising v_max_f32 x, x is put in front of it).  NOT inline asm: the hazard recognizer does not see inside an
// asm statement, and an MFMA result needs software wait states before a VALU instruction may read it.
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float max2f(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float max_over_g(float x) {
    const unsigned u = __float_as_uint(x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = max2f(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned v = __float_as_uint(y);
    auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max2f(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

struct AttnParams {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    unsigned short* out;
    int64_t q_stride, k_stride, v_stride, o_stride;
    const int32_t* cu_q;
    const int32_t* cu_k;
    const float* slopes;
    float scale_log2;   // scale * log2(e)
    int H;
    int nqb;            // query blocks per (sequence, head) (tiled kernel)
    int chunk;          // blocks per XCD = ceil(n_seq * H * nqb / 8)
    int total;          // n_seq * H * nqb
    int q_at_start;     // ALiBi query positions: 0 = queries are the LAST len_q positions of the key sequence
                        // (flash-attn convention), 1 = query i sits at position i
    int q_log2;         // 1: Q was projected with weights pre-multiplied by scale * log2(e) (VF_ATTN_Q_LOG2): scores are
                        // base-2 logits as they leave the matrix pipe, scale_log2 = 1
    const int64_t* q_rows;   // optional row maps (vf_attn_varlen_fwd_rows): token t's query row is row q_rows[t] of q, its key /
    const int64_t* kv_rows;  // value rows are row kv_rows[t] of k / v -- the operands are TABLES of distinct rows (the first
                             // layers' projections by lookup) and the gather happens in the loads; null: row t itself
#if defined(VF_SHORT_PROF) || defined(VF_X32PP_PROF)
    unsigned long long* prof;   // scripts/probes/attn_*_probe.hip only
#endif
#ifdef VF_TUNING
    int dbg;                    // VF_ATTN_SHORT_DBG (attn_short2_kernel; results meaningless): 1 = loads, LDS staging and stores only
                                // (no tile arithmetic): the streaming ceiling of the one-block-per-(sequence, head) structure
#endif
};


// Block -> (sequence, head, query block).  Workgroups are dealt to the 8 XCDs round-robin by linear id, so XCD x gets
// the logical range [x * chunk, (x + 1) * chunk): all heads and query blocks of a sequence run on one XCD at about the
// same time.  Their K/V re-reads and the cache lines that neighbouring heads share then hit that XCD's L2, and the
// 96/128-byte head slices of one token row reach HBM together instead of one DRAM page activation per slice.
// Logical order: query block fastest, then head, then sequence.  Returns false for the padding blocks.
__device__ __forceinline__ bool block_coords(const AttnParams& P, int& seq, int& h, int& qb) {
    const int id = blockIdx.x;
    const int L = (id & 7) * P.chunk + (id >> 3);
    if ((id >> 3) >= P.chunk || L >= P.total) return false;
    qb = L % P.nqb;
    const int sh = L / P.nqb;
    h = sh % P.H;
    seq = sh / P.H;
    return true;
}

// out[tok0 .. tok0+n)[head h] = 0 (sequences without keys)
template <int DH>
__device__ __forceinline__ void zero_rows(const AttnParams& P, int tok0, int n, int h) {
    constexpr int CPR = DH / 4;                       // 8-byte pieces (the output is only 8-byte aligned)
    for (int i = threadIdx.x; i < n * CPR; i += blockDim.x)
        *reinterpret_cast<u32x2_t*>(P.out + (int64_t)(tok0 + i / CPR) * P.o_stride + h * DH + (i % CPR) * 4) =
            (u32x2_t){0u, 0u};
}

__device__ __forceinline__ s16x4_t lds_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

#ifndef VF_ATTN_PV_INTERLEAVE
#define VF_ATTN_PV_INTERLEAVE 1
#endif

// One 64-key tile for QG query groups of a wave: S^T = K.Q^T, online softmax, O^T += V^T.P^T.
// sK / sV point at the tile's first key row in LDS.  All state is per lane (r = query, g = key sub-block).
// DBGT (diagnostic builds of the long-stream kernel only): 1 = no softmax VALU work (P = S), 2 = additionally no LDS
// fragment reads (operands reused), used to locate the binding ceiling; results are meaningless.
// NKT (4, 2 or 1): number of 16-key sub-tiles that can hold a valid key -- the LAST tile of a sequence whose remainder is
// <= 32 / <= 16 keys runs the 2 / 1 sub-tile form: no K fragments, QK^T MFMAs, bias, maximum, exponentials for the rest, and
// for NKT <= 2 no second 32-key PV block either.  Bit-identical to the full tile (whose masked keys contribute p = 0 exactly,
// and x + 0 = x in the accumulators); a 201-key sequence (3 tiles + 9 keys) saves 3/16 of its arithmetic.
// SM (softmax mode): 0 = online softmax with a running maximum, scale applied here; 1 and 2: q carries the base-2 softmax
// scale (AttnParams::q_log2) -- 1 = NO maximum: p = exp2(s [+ bias]), unnormalised sums; the caller checks the denominators
// and recomputes with mode 2 when one left the safe range (see attn_x32_kernel); 2 = running maximum rounded up to an
// INTEGER: a power-of-two offset does not move the rounding of p, so modes 1 and 2 produce the same bits -- which is what
// lets the kernels a query may be served by differ in mode (a gene's result must not depend on the batch geometry that picks
// the kernel: the tiled kernels run mode 2, the one-block-per-sequence kernel mode 1).
template <int DH, int QG, bool ALIBI, int DT, int DBGT = 0, int NKT = 4, int SM = 0>
__device__ __forceinline__ void attn_tile(const char* sK, const char* sV, int kb0, int len_k, int r, int g, float c,
                                          float slope2, const typename Op16<DT>::frag (&qf)[QG][KLayout<DH>::KS], const float (&q_pos)[QG],
                                          f32x4_t (&o)[QG][DH / 16], float (&m_run)[QG], f32x4_t (&l_acc)[QG]) {
    using frag_t = typename Op16<DT>::frag;
    constexpr int NDT = DH / 16;
    constexpr int VROW = VLayout<DH>::ROW;
    constexpr int KS = KLayout<DH>::KS, K_ROW_BYTES = KLayout<DH>::ROW;
    // ---- S^T = K . Q^T : 4 key tiles x 2 k-steps, K fragments shared by the QG query groups
    f32x4_t s[QG][4];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[qg][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // all 8 K fragments of the tile are requested before the first MFMA and the V fragments of the first 32-key
    // block right after, so LDS latency overlaps the MFMAs and the softmax arithmetic instead of preceding every
    // MFMA pair (diagnostic builds: just-in-time fragment reads cost ~70 of 150 us on the gene->CRE shape)
    frag_t kf[NKT][KS];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            kf[kt][ks] = DBGT >= 2 ? qf[0][ks] : *reinterpret_cast<const frag_t*>(
                sK + (16 * kt + r) * K_ROW_BYTES + (((4 * ks + g) ^ KLayout<DH>::swz(r)) << 4));
    auto read_v = [&](int kb, frag_t(&vf)[NDT]) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const char* vp = sV + (32 * kb + 4 * g + (r >> 2)) * VROW + 32 * dt + 8 * (r & 3);
            if (DBGT >= 2) { vf[dt] = qf[0][0]; continue; }
            const s16x4_t lo = lds_tr_read(vp);
            const s16x4_t hi = lds_tr_read(vp + 16 * VROW);
            vf[dt] = __builtin_bit_cast(frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int qg = 0; qg < QG; ++qg)
                s[qg][kt] = Op16<DT>::mfma(kf[kt][ks], qf[qg][ks], s[qg][kt]);
    frag_t vf0[NDT];
    read_v(0, vf0);

    // ---- online softmax (lane (r,g): query r, keys kb0 + 16kt + 4g + e), in three steps so that the long part is
    // ONE basic block in which matrix and vector instructions can be interleaved:
    //   (1) bias / mask and the running maximum of every query group;
    //   (2) ONE wave-uniform branch: rescale O and l of all groups when any running maximum moved (alpha is exactly 1
    //       for the rows whose maximum did not move, so the result is the same as a per-group test);
    //   (3) per group: p = exp2(...), pack to 16 bit -- followed in program order by the PV / row-sum MFMAs of the
    //       PREVIOUS group, which have no dependence on this group's exponentials: the wave issues in order, so this
    //       is what lets the matrix pipe work under the v_exp stream (before: 250 VALU instructions with an idle
    //       matrix pipe, then 21 MFMAs back to back with an idle VALU).
    frag_t pf[QG][2];
    const bool tail = kb0 + BKV > len_k;           // wave-uniform: only the last (ragged) key tile masks keys
    const int klim = len_k - kb0 - 4 * g;          // key 16kt+e of this lane is valid iff 16kt+e < klim
    const float k_pos0 = (float)(kb0 + 4 * g);
    constexpr unsigned int one2 = Op16<DT>::ONE * 0x10001u;
    const u32x4_t ones_bits = {one2, one2, one2, one2};
    const frag_t ones = __builtin_bit_cast(frag_t, ones_bits);
    auto pack_p = [&](int qg) {                        // sub-tiles kt >= NKT still hold their initial zeros = 16-bit zeros
#pragma unroll
        for (int kb = 0; kb < (NKT > 2 ? 2 : 1); ++kb) {
            u32x4_t pk;
            pk[0] = Op16<DT>::pack2(s[qg][2 * kb][0], s[qg][2 * kb][1]);
            pk[1] = Op16<DT>::pack2(s[qg][2 * kb][2], s[qg][2 * kb][3]);
            pk[2] = Op16<DT>::pack2(s[qg][2 * kb + 1][0], s[qg][2 * kb + 1][1]);
            pk[3] = Op16<DT>::pack2(s[qg][2 * kb + 1][2], s[qg][2 * kb + 1][3]);
            pf[qg][kb] = *reinterpret_cast<frag_t*>(&pk);
        }
    };
    // O^T += V^T . P^T and the softmax denominators of one query group.  The denominators run on the matrix pipe: a
    // V^T fragment of ones gives l[q] += sum_k P[q][k] in every accumulator row, i.e. each lane ends up with the
    // complete row sum of its query (no per-element v_add_f32 -- the softmax is VALU-issue bound -- and no cross-lane
    // reduction at the end).  The sum runs over the 16-bit P the PV product uses, so O / l is an exact convex
    // combination of the V rows.
    frag_t vf1[NDT];
    auto pv = [&](int qg) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[qg][dt] = Op16<DT>::mfma(vf0[dt], pf[qg][0], o[qg][dt]);
        if (NKT > 2) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[qg][dt] = Op16<DT>::mfma(vf1[dt], pf[qg][1], o[qg][dt]);
        }
        if (DBGT == 0) {
            l_acc[qg] = Op16<DT>::mfma(ones, pf[qg][0], l_acc[qg]);
            if (NKT > 2) l_acc[qg] = Op16<DT>::mfma(ones, pf[qg][1], l_acc[qg]);
        }
    };
    if (DBGT >= 1) {                                  // diagnostic: P = S, no softmax arithmetic
        read_v(1, vf1);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            pack_p(qg);
            l_acc[qg] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
            pv(qg);
        }
        return;
    }
    float m_new[QG];
    bool moved = false;
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        if (ALIBI) {
            const float dq = q_pos[qg] - k_pos0;
            // q carries the scale (modes 1, 2, imax): bias and logit meet in ONE fused multiply-add, the same expression in
            // every kernel; otherwise scale and bias are applied together (wave-uniform choice)
            if (SM != 0) {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[qg][kt][e] = fmaf(-slope2, fabsf(dq - (float)(16 * kt + e)), s[qg][kt][e]);
            } else {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[qg][kt][e] = fmaf(s[qg][kt][e], c, -slope2 * fabsf(dq - (float)(16 * kt + e)));
            }
        }
        if (tail && !HwMask<DH>::value) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[qg][kt][e] = (16 * kt + e) < klim ? s[qg][kt][e] : -INFINITY;
        }
        // 16 scores -> 1 maximum in 8 v_max3_f32 + 1 v_max_f32 (two chains of max(max(a, b), c) for instruction-level
        // parallelism; a pairwise tree compiled to 24 v_max / v_max3 per query group: the loop is VALU-issue bound)
        if (SM == 1) continue;
        float mxa = max3f(s[qg][0][0], s[qg][0][1], s[qg][0][2]), mxb;
        if (NKT == 4) {
            mxb = max3f(s[qg][2][0], s[qg][2][1], s[qg][2][2]);
            mxa = max3f(mxa, s[qg][0][3], s[qg][1][0]);
            mxb = max3f(mxb, s[qg][2][3], s[qg][3][0]);
            mxa = max3f(mxa, s[qg][1][1], s[qg][1][2]);
            mxb = max3f(mxb, s[qg][3][1], s[qg][3][2]);
            mxa = max3f(mxa, s[qg][1][3], s[qg][3][3]);
        } else if (NKT == 2) {
            mxb = max3f(s[qg][1][0], s[qg][1][1], s[qg][1][2]);
            mxa = max3f(mxa, s[qg][0][3], s[qg][1][3]);
        } else {
            mxb = s[qg][0][3];
        }
        float mx = max_over_g(max2f(mxa, mxb));
        if (SM == 2) mx = __builtin_ceilf(mx);
        m_new[qg] = max2f(m_run[qg], mx);           // finite: tile 0 always holds a valid key
        moved = moved || (m_new[qg] > m_run[qg]);
    }
    if (SM != 1 && __any(moved)) {                   // wave-uniform; rare after the first tiles of a sequence
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const float alpha = __builtin_amdgcn_exp2f(ALIBI ? (m_run[qg] - m_new[qg]) : (m_run[qg] - m_new[qg]) * c);
            l_acc[qg] *= alpha;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[qg][dt] *= alpha;
        }
    }
    if (NKT > 2) read_v(1, vf1);
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        if (SM == 1) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[qg][kt][e] = __builtin_amdgcn_exp2f(s[qg][kt][e]);
        } else {
            m_run[qg] = m_new[qg];
            // p = exp2(c*s - c*m) (no ALiBi: scale folded into one FMA) or exp2(s - m) (ALiBi: already scaled)
            const float mc = ALIBI ? -m_new[qg] : -m_new[qg] * c;
            const float cc = ALIBI ? 1.0f : c;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[qg][kt][e] = __builtin_amdgcn_exp2f(fmaf(s[qg][kt][e], cc, mc));
        }
        pack_p(qg);
        if (VF_ATTN_PV_INTERLEAVE && qg > 0) pv(qg - 1);
    }
    if (VF_ATTN_PV_INTERLEAVE) {
        pv(QG - 1);
    } else {
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) pv(qg);
    }
}

// The last key tile of a sequence in its short form: `rem` = keys it holds, <= 32 (the caller's tile loop stops one tile
// early for such a remainder; a longer one is an ordinary full tile of that loop -- no second copy of the full tile body).
template <int DH, int QG, bool ALIBI, int DT, int SM = 0>
__device__ __forceinline__ void attn_tile_short(int rem, const char* sK, const char* sV, int kb0, int len_k, int r, int g, float c,
                                                float slope2, const typename Op16<DT>::frag (&qf)[QG][KLayout<DH>::KS],
                                                const float (&q_pos)[QG], f32x4_t (&o)[QG][DH / 16], float (&m_run)[QG],
                                                f32x4_t (&l_acc)[QG]) {
    if (rem <= 16) attn_tile<DH, QG, ALIBI, DT, 0, 1, SM>(sK, sV, kb0, len_k, r, g, c, slope2, qf, q_pos, o, m_run, l_acc);
    else attn_tile<DH, QG, ALIBI, DT, 0, 2, SM>(sK, sV, kb0, len_k, r, g, c, slope2, qf, q_pos, o, m_run, l_acc);
}

// (2 query groups with ALiBi sit at the 170-register edge of three waves per SIMD -- the CRE stream's self attention; the
// integer-maximum form compiles to 172 without the bound)
template <int DH, int QG, bool ALIBI, int DT = VF_BF16, int DBG = 0, bool QL = false>
__global__ __launch_bounds__(256, (QL && QG == 2 && ALIBI && DH <= 48 ? 3 : 1)) void attn_fwd_kernel(AttnParams P) {
    using frag_t = typename Op16<DT>::frag;
    constexpr int CPR = DH / 8;                       // 16-byte chunks per K/V row
    constexpr int NCHUNK = BKV * CPR;                 // chunks per tile
    constexpr int NLD = (NCHUNK + 255) / 256;         // chunk loads per thread per operand
    constexpr int NDT = DH / 16;                       // output d-tiles
    constexpr int VROW = VLayout<DH>::ROW;
    constexpr int KS = KLayout<DH>::KS, K_ROW_BYTES = KLayout<DH>::ROW, K_TILE_BYTES = KLayout<DH>::TILE;
    constexpr int STAGE = K_TILE_BYTES + VLayout<DH>::TILE;
    constexpr int BQ = 4 * QG * 16;

    extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 * STAGE bytes (attn_fwd_lds<DH>())

    int seq, h, qblk;
    if (!block_coords(P, seq, h, qblk)) return;
    const int q_tok0 = P.cu_q[seq], len_q = P.cu_q[seq + 1] - q_tok0;
    const int k_tok0 = P.cu_k[seq], len_k = P.cu_k[seq + 1] - k_tok0;
    const int qb0 = qblk * BQ;
    if (qb0 >= len_q) return;                         // block-uniform
    if (len_k <= 0) {                                 // no keys: the output rows are zeros (flash-attn convention)
        zero_rows<DH>(P, q_tok0 + qb0, (len_q - qb0) < BQ ? (len_q - qb0) : BQ, h);
        return;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;

    // ---- zero the K pad chunks (d >= DH) of both stages once; loads never touch them
    // (chunk CPR is the mask slot and belongs to write_lds alone: zeroing it here as well would race with the mask
    // write of tile 0, which is issued by another wave without a barrier in between)
    if (CPR < 7) {                                    // dh = 32 / 48 only (dh >= 64 has no pad chunks that are read)
        constexpr int PADC = 7 - CPR;
        for (int i = tid; i < 2 * BKV * PADC; i += 256) {
            const int st = i / (BKV * PADC), rem = i % (BKV * PADC);
            const int row = rem / PADC, c = CPR + 1 + rem % PADC;
            *reinterpret_cast<u32x4_t*>(smem + st * STAGE + row * K_ROW_BYTES + ((c ^ ((row >> 1) & 7)) << 4)) =
                (u32x4_t){0u, 0u, 0u, 0u};
        }
    }

    // ---- Q fragments (B operand): lane (r,g) holds Q[q = r][d = 32ks + 8g .. +7]
    frag_t qf[QG][KS];
    int q_abs[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        q_abs[qg] = qb0 + (wave * QG + qg) * 16 + r;
        const int row = q_abs[qg] < len_q ? q_abs[qg] : len_q - 1;
        const unsigned short* qp = P.q + (int64_t)(q_tok0 + row) * P.q_stride + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 32 * ks + 8 * g;
            u32x4_t raw = q_pad_chunk<DT>(d0, DH);
            if (d0 < DH) raw = *reinterpret_cast<const u32x4_t*>(qp + d0);
            qf[qg][ks] = *reinterpret_cast<frag_t*>(&raw);
        }
    }

    // scores live in the log2 domain: p = exp2(c*s + bias - m).  Without ALiBi the running max is kept on the RAW
    // score (c > 0, so the arg-max is the same) and scale and max are folded into one FMA per element.
    const float c = P.scale_log2;
    const float slope2 = ALIBI ? P.slopes[h] * 1.4426950408889634f : 0.f;
    float q_pos[QG];                                   // query position + (sk - sq), as float (exact: < 2^24)
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) q_pos[qg] = (float)(q_abs[qg] + (P.q_at_start ? 0 : len_k - len_q));

    f32x4_t o[QG][NDT];
    float m_run[QG];
    f32x4_t l_acc[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m_run[qg] = -INFINITY;
        l_acc[qg] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[qg][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    // ---- K/V tile staging: global -> registers -> LDS.  The tile's K chunks and V chunks form ONE list of 2*NCHUNK
    // 16-byte items, exactly NLD2 per thread, so the prefetch has no divergent branch (with per-thread `if`s hipcc's
    // wait insertion drained the prefetch -- s_waitcnt vmcnt(0) -- in front of the first MFMA of every tile).
    constexpr int NLD2 = 2 * NCHUNK / 256;
    static_assert(2 * NCHUNK % 256 == 0, "staging items must divide over the threads");
    u32x4_t kvreg[NLD2];
    const unsigned short* kbase = P.k + (int64_t)k_tok0 * P.k_stride + h * DH;
    const unsigned short* vbase = P.v + (int64_t)k_tok0 * P.v_stride + h * DH;
    // per-item constants of the staging list: source pointer at key 0 of the sequence and the row stride; the per-tile
    // address is base + min(key, len_k - 1) * stride with a 24-bit multiply (full rate; max_seqlen_k * stride < 2^31 is
    // checked at launch) -- the 64-bit multiply this replaces cost three quarter-rate VALU instructions per load
    const unsigned short* kv_src[NLD2];
    unsigned kv_stride[NLD2];
    int kv_row[NLD2];
#pragma unroll
    for (int i = 0; i < NLD2; ++i) {
        const int item = tid + 256 * i;
        const bool is_v = item >= NCHUNK;
        const int ci = is_v ? item - NCHUNK : item;
        kv_row[i] = ci / CPR;
        kv_src[i] = (is_v ? vbase : kbase) + (ci % CPR) * 8;
        kv_stride[i] = (unsigned)(is_v ? P.v_stride : P.k_stride);
    }
    auto load_regs = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD2; ++i) {
            int key = t * BKV + kv_row[i];
            key = key < len_k ? key : len_k - 1;                     // finite data for masked keys
            const unsigned off = __umul24((unsigned)key, kv_stride[i]);
            kvreg[i] = *reinterpret_cast<const u32x4_t*>(kv_src[i] + off);
        }
    };
    auto write_lds = [&](int stage, int t) {
        char* sK = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NLD2; ++i) {
            const int item = tid + 256 * i;
            const bool is_v = item >= NCHUNK;
            const int ci = is_v ? item - NCHUNK : item;
            const int row = ci / CPR, c = ci % CPR;
            const int off = is_v ? K_TILE_BYTES + row * VROW + (c << 4)
                                 : row * K_ROW_BYTES + ((c ^ KLayout<DH>::swz(row)) << 4);
            *reinterpret_cast<u32x4_t*>(sK + off) = kvreg[i];
        }
        if (HwMask<DH>::value && tid < BKV)          // pad slot d = DH of every key row: 0 or the mask value
            *reinterpret_cast<u32x4_t*>(sK + tid * K_ROW_BYTES + ((CPR ^ ((tid >> 1) & 7)) << 4)) =
                mask_chunk<DT>(t * BKV + tid < len_k);
    };

    const int nkv = (len_k + BKV - 1) / BKV;
    load_regs(0);
    write_lds(0, 0);
    __syncthreads();

    // The last tile is peeled so that the loop body has no conditional loads / stores: with them hipcc's wait
    // bookkeeping turns conservative and drains the just-issued prefetch (s_waitcnt vmcnt(0)) before the first MFMA
    // of every tile.
    // a wave whose 16 * QG query rows all lie past the sequence (the last query block of a seq2reg window: 98 tokens =
    // one full block + 34 rows, i.e. one idle and one almost idle wave) still stages K/V and meets the barriers, but
    // computes nothing
    // (QG = 1 kernels only: in the 2-group ALiBi instantiation the extra branch costs 4 VGPRs, which is the third wave
    // per SIMD: 172 > 170)
    const bool active = QG > 1 || qb0 + wave * QG * 16 < len_q;       // wave-uniform
    for (int t = 0; t + 1 < nkv; ++t) {
        if (DBG < 3) load_regs(t + 1);
        const char* sK = smem + (t & 1) * STAGE;
        if (active)
            attn_tile<DH, QG, ALIBI, DT, (DBG > 2 ? 0 : DBG), 4, (QL ? 2 : 0)>(sK, sK + K_TILE_BYTES, t * BKV, len_k, r, g, c, slope2, qf,
                                                           q_pos, o, m_run, l_acc);
        if (DBG < 3) write_lds((t + 1) & 1, t + 1);
        if (DBG < 4) __syncthreads();
    }
    if (active) {
        const int t = nkv - 1;
        const char* sK = smem + (t & 1) * STAGE;
        // short sequences (one query group per wave: seq2reg windows of 70-200 tokens, 2-4 tiles): the last tile in its
        // 16 / 32-key form when the remainder allows (a 100-token window: 64 + 36 keys; a 200-token chunk: 3 x 64 + 8)
        if (QG == 1 && DBG == 0 && len_k - t * BKV <= 32)
            attn_tile_short<DH, QG, ALIBI, DT, (QL ? 2 : 0)>(len_k - t * BKV, sK, sK + K_TILE_BYTES, t * BKV, len_k, r, g, c, slope2, qf,
                                               q_pos, o, m_run, l_acc);
        else
            attn_tile<DH, QG, ALIBI, DT, (DBG > 2 ? 0 : DBG), 4, (QL ? 2 : 0)>(sK, sK + K_TILE_BYTES, t * BKV, len_k, r, g, c, slope2, qf, q_pos,
                                                           o, m_run, l_acc);
    }

    // ---- normalise and store: lane (r,g) holds O[q = r][d = 16dt + 4g .. +3]
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const float l = l_acc[qg][0];
        const float inv = 1.0f / l;
        if (q_abs[qg] < len_q) {
            unsigned short* op = P.out + (int64_t)(q_tok0 + q_abs[qg]) * P.o_stride + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                u32x2_t pk;
                pk[0] = Op16<DT>::pack2(o[qg][dt][0] * inv, o[qg][dt][1] * inv);
                pk[1] = Op16<DT>::pack2(o[qg][dt][2] * inv, o[qg][dt][3] * inv);
                *reinterpret_cast<u32x2_t*>(op + 16 * dt) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Long query streams without positional bias at dh = 48 (the gene -> CRE cross attention: the T x G ~ 10^4 query rows of a
// gene against its ~10^3 CRE keys) on v_mfma_f32_32x32x16.  Why a second formulation: with 16x16x32 tiles this shape is
// bound by instruction ISSUE, not by the matrix pipe (per 64-key tile and wave 64 MFMAs x 8 issue cycles + 64 v_exp x 8 +
// ~300 other VALU x 4 = 2000 cycles against 1024 cycles of matrix work), and dh = 48 wastes a quarter of the QK^T MFMAs on
// padding the contraction to 64.  Here
//   S^T[key 32][query 32] = K . Q^T   A = 32 key rows x 16 d (ds_read_b128, three 16-deep steps = 48 exactly),
//                                     B = Q^T from registers; lane (q = lane & 31, h = lane >> 5) ends up with 16 keys of
//                                     ITS query per 32-key block: the row maximum is in-lane plus ONE half swap;
//   O^T[d 32][query 32] += V^T . P^T  A = V^T gathered by ds_read_b64_tr_b16 from the row-major V tile, B = P^T straight
//                                     from the score registers (regs 8s .. 8s+7 of a 32-key block are the 16-deep step s;
//                                     the k order inside a step is permuted identically on both operands);
//   the head dimension is padded to 64 on the OUTPUT side only, and the first padding column of V holds 1.0: row 48 of
//   O^T is then sum_k P = the softmax denominator -- the padding MFMAs do the row sums (no ones-fragment MFMAs).
// Per tile and wave (64 queries x 64 keys): 28 MFMAs of 32 cycles (12 QK^T + 16 PV) instead of 64 of 16, one permlane
// step per 32 queries instead of two per 16.
// LDS: K rows 112 B (96 + 16: the ds_read_b128 lane groups {0-3, 12-15, 20-27} then hit 16 distinct 16-byte slots),
// V rows 192 B (the 4 rows x 64 B a 32-lane half reads transposed fall on four distinct quarters of the 256-B bank row).
// ---------------------------------------------------------------------------------------------------------------------
// FAST (Q pre-scaled, AttnParams::q_log2): the accumulator of K . Q^T already is the base-2 logit, and the softmax runs
// WITHOUT a running maximum: p = exp2(s), O^T and the denominator (row 48) accumulate unnormalised -- per score one v_exp and
// half a v_cvt_pk instead of max3 + fma + exp + cvt (the kernel is bound by instruction issue; -14 % time).  Softmax is
// invariant under the offset, and fp32 / bf16 / fp16-with-offset carry the dynamic range as long as every row's largest
// logit stays inside (-100, +100 - log2(keys)): trained attention logits live within a few tens.  A block in which some
// valid row's denominator is not a normal number of moderate size (overflow, or every key underflowed) is detected at the
// end (one block-wide vote) and recomputed with the running-maximum form -- correctness never depends on the range, only
// the speed of such a block does.  The recomputation uses INTEGER offsets (ceil of the maximum): powers of two commute with
// the roundings, so both forms round the same probabilities.
#ifndef VF_X32_WIDE_STORE
#define VF_X32_WIDE_STORE 1
#endif
template <int DT, int QB, bool FAST = false>
__global__ __launch_bounds__(256) void attn_x32_kernel(AttnParams P) {
    using frag_t = typename Op16<DT>::frag;
    constexpr int DH = 48, KSTEPS = 3, KROW = 112, VROW = 192;
    constexpr int K_TILE = BKV * KROW, V_TILE = BKV * VROW, STAGE = K_TILE + V_TILE;
    constexpr int CPR = DH / 8, NCHUNK = BKV * CPR, NLD2 = 2 * NCHUNK / 256;
    constexpr int BQ = 4 * QB * 32;
    static_assert(2 * NCHUNK % 256 == 0, "staging items must divide over the threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 * STAGE bytes

    int seq, hd, qblk;
    if (!block_coords(P, seq, hd, qblk)) return;
    const int q_tok0 = P.cu_q[seq], len_q = P.cu_q[seq + 1] - q_tok0;
    const int k_tok0 = P.cu_k[seq], len_k = P.cu_k[seq + 1] - k_tok0;
    const int qb0 = qblk * BQ;
    if (qb0 >= len_q) return;                         // block-uniform
    if (len_k <= 0) {
        zero_rows<DH>(P, q_tok0 + qb0, (len_q - qb0) < BQ ? (len_q - qb0) : BQ, hd);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;

    // padding columns 48 .. 63 of every V row, both stages, once: column 48 = 1.0 (the denominator column), the rest 0
    if (tid < 2 * BKV) {
        char* vp = smem + (tid >> 6) * STAGE + K_TILE + (tid & 63) * VROW + DH * 2;
        *reinterpret_cast<u32x4_t*>(vp) = (u32x4_t){Op16<DT>::ONE, 0u, 0u, 0u};
        *reinterpret_cast<u32x4_t*>(vp + 16) = (u32x4_t){0u, 0u, 0u, 0u};
    }

    // ---- Q fragments (B operand): lane (q, h) holds Q[q][16 ks + 8 h .. + 7]
    frag_t qf[QB][KSTEPS];
    int q_abs[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        q_abs[qb] = qb0 + (wave * QB + qb) * 32 + ql;
        const int row = q_abs[qb] < len_q ? q_abs[qb] : len_q - 1;
        const unsigned short* qp = P.q + (int64_t)(q_tok0 + row) * P.q_stride + hd * DH + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qp + 16 * ks);
            qf[qb][ks] = __builtin_bit_cast(frag_t, raw);
        }
    }
    const float c = P.scale_log2;
    f32x16_t o[QB][2];
    float m_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qb][dt][i] = 0.f;
    }

    // ---- K/V tile staging: global -> registers -> LDS, one list of 2 * NCHUNK 16-byte items, NLD2 per thread (see
    // attn_fwd_kernel: no divergent branch around the prefetch)
    u32x4_t kvreg[NLD2];
    const unsigned short* kbase = P.k + (int64_t)k_tok0 * P.k_stride + hd * DH;
    const unsigned short* vbase = P.v + (int64_t)k_tok0 * P.v_stride + hd * DH;
    const unsigned short* kv_src[NLD2];
    unsigned kv_stride[NLD2];
    int kv_row[NLD2], kv_off[NLD2];
#pragma unroll
    for (int i = 0; i < NLD2; ++i) {
        const int item = tid + 256 * i;
        const bool is_v = item >= NCHUNK;
        const int ci = is_v ? item - NCHUNK : item;
        kv_row[i] = ci / CPR;
        kv_src[i] = (is_v ? vbase : kbase) + (ci % CPR) * 8;
        kv_stride[i] = (unsigned)(is_v ? P.v_stride : P.k_stride);
        kv_off[i] = is_v ? K_TILE + kv_row[i] * VROW + (ci % CPR) * 16 : kv_row[i] * KROW + (ci % CPR) * 16;
    }
    auto load_regs = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD2; ++i) {
            int key = t * BKV + kv_row[i];
            key = key < len_k ? key : len_k - 1;                     // finite data for masked keys
            kvreg[i] = *reinterpret_cast<const u32x4_t*>(kv_src[i] + __umul24((unsigned)key, kv_stride[i]));
        }
    };
    auto write_lds = [&](int stage) {
#pragma unroll
        for (int i = 0; i < NLD2; ++i) *reinterpret_cast<u32x4_t*>(smem + stage * STAGE + kv_off[i]) = kvreg[i];
    };

    // fragment addresses inside a stage
    const int k_off = ql * KROW + h * 16;                                           // + kb * 32 * KROW + ks * 32
    const int v_off = K_TILE + (4 * h + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    //                                                                              // + (32 kb + 16 s [+ 8]) * VROW + dt * 64

    // TAIL: the key tile may reach past len_k (only the last tile of a sequence; a compile-time flag, so that the
    // steady-state loop body carries no masking code at all)
    auto tile = [&](const char* st, int kb0, auto tail_c, auto fast_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        constexpr bool NOMAX = decltype(fast_c)::value;
        // ---- S^T = K . Q^T
        frag_t kf[2][KSTEPS];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
                kf[kb][ks] = *reinterpret_cast<const frag_t*>(st + k_off + kb * 32 * KROW + ks * 32);
        f32x16_t s[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[qb][kb][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) s[qb][kb] = Op16<DT>::mfma32(kf[kb][ks], qf[qb][ks], s[qb][kb]);
            }
        // V^T fragments of the whole tile: [kb][s][dt]
        frag_t vf[2][2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vp = st + v_off + (32 * kb + 16 * sx) * VROW + dt * 64;
                    const s16x4_t lo = lds_tr_read(vp);
                    const s16x4_t hi = lds_tr_read(vp + 8 * VROW);
                    vf[kb][sx][dt] = __builtin_bit_cast(frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
        // ---- online softmax: lane (q, h) holds keys kb0 + 32 kb + (i & 3) + 8 (i >> 2) + 4 h of query q
        float m_new[QB];
        bool moved = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (TAIL) {
                const int klim = len_k - kb0 - 4 * h;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        s[qb][kb][i] = (32 * kb + (i & 3) + 8 * (i >> 2)) < klim ? s[qb][kb][i] : -INFINITY;
            }
            if (NOMAX) continue;
            float ma = max3f(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            float mb = max3f(s[qb][1][0], s[qb][1][1], s[qb][1][2]);
#pragma unroll
            for (int i = 3; i + 1 < 16; i += 2) {
                ma = max3f(ma, s[qb][0][i], s[qb][0][i + 1]);
                mb = max3f(mb, s[qb][1][i], s[qb][1][i + 1]);
            }
            float mx = max3f(ma, mb, max2f(s[qb][0][15], s[qb][1][15]));
            if (P.q_log2) mx = __builtin_ceilf(mx);      // integer offsets whenever q carries the scale: the recomputation pass
                                                         // of the FAST kernel and VF_ATTN_NOMAX=0 round what FAST rounds
            const unsigned u = __float_as_uint(mx);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            m_new[qb] = max3f(m_run[qb], __uint_as_float(sw[0]), __uint_as_float(sw[1]));   // finite: tile 0 holds a valid key
            moved = moved || (m_new[qb] > m_run[qb]);
        }
        if (!NOMAX && __any(moved)) {                    // wave-uniform; rare after the first tiles of a sequence
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new[qb]) * c);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[qb][dt] *= alpha;     // row 48 (the denominator) included
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (NOMAX) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[qb][kb][i] = __builtin_amdgcn_exp2f(s[qb][kb][i]);
            } else {
                m_run[qb] = m_new[qb];
                const float mc = -m_new[qb] * c;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[qb][kb][i] = __builtin_amdgcn_exp2f(fmaf(s[qb][kb][i], c, mc));
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    u32x4_t pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = Op16<DT>::pack2(s[qb][kb][8 * sx + 2 * j], s[qb][kb][8 * sx + 2 * j + 1]);
                    const frag_t pf = __builtin_bit_cast(frag_t, pk);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[qb][dt] = Op16<DT>::mfma32(vf[kb][sx][dt], pf, o[qb][dt]);
                }
        }
    };

    const int nkv = (len_k + BKV - 1) / BKV;
    const bool active = qb0 + wave * QB * 32 < len_q;                 // wave-uniform
    auto pass = [&](auto fast_c) {
        load_regs(0);
        write_lds(0);
        __syncthreads();
        for (int t = 0; t + 1 < nkv; ++t) {
            load_regs(t + 1);
            if (active) tile(smem + (t & 1) * STAGE, t * BKV, std::false_type{}, fast_c);
            write_lds((t + 1) & 1);
            __syncthreads();
        }
        if (active) tile(smem + ((nkv - 1) & 1) * STAGE, (nkv - 1) * BKV, std::true_type{}, fast_c);
    };
    if (FAST) {
        pass(std::true_type{});
        // the denominators (row d = 48: element 8 of the dt = 1 tile in the h = 0 lanes) of the block's valid rows
        bool bad = false;
        if (active && h == 0) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                // 2^-100 .. 2^100 (false for NaN).  fp16 P: overflow shows as inf; at the low end the largest p of a row must
                // still be a NORMAL half with room below it (subnormal P would pass a 2^-100 test with 4 significant bits).
                // The denominator bounds the largest p from below only through the key count, l <= len_k * p_max, so the
                // test is on the MEAN p: l > len_k * 2^-11 guarantees p_max > 2^-11 (8 times the smallest normal half)
                // however many keys share the sum -- thousands of keys at logits of -17 ... -20 add up to 2^-6 with every
                // P subnormal (round-3 advice).  Conservative for peaked rows (recomputed although fine): never wrong.
                const float l_min = DT == VF_F16 ? fmaxf(0.015625f, (float)len_k * 4.8828125e-4f) : 7.8886e-31f;
                const float l = o[qb][1][8];
                bad = bad || !(l > l_min && l < 1.2676e30f);
            }
        }
        if (__syncthreads_or(bad)) {                                  // block-uniform (also orders the LDS stages)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                m_run[qb] = -INFINITY;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[qb][dt][i] = 0.f;
            }
            pass(std::false_type{});
            __syncthreads();                                          // the stages become the output staging below
        }
    } else {
        pass(std::false_type{});
        __syncthreads();
    }

    // ---- normalise and store: lane (q, h) holds O[q][d = 32 dt + (i & 3) + 8 (i >> 2) + 4 h]; the denominator is row
    // d = 48 = element 8 of the dt = 1 tile in the h = 0 lane.
    // Wide form (16-byte aligned output rows, VF_X32_WIDE_STORE): the wave's 32 QB x 48 block goes through LDS (the K / V stages
    // are free: every wave is behind the vote / the barrier that ends the pass) and leaves as whole 96-byte row segments, 16 bytes per lane,
    // ~11 rows per store instruction -- instead of 12 QB eight-byte stores per lane that each touch 32 rows (the store path of
    // the CU is shared with the K / V loads of the co-resident block).  Same values, same bits.
    const bool wide = VF_X32_WIDE_STORE && ((reinterpret_cast<uintptr_t>(P.out) | (uintptr_t)(P.o_stride * 2)) & 15) == 0;
    constexpr int OPITCH = DH * 2 + 16;
    char* const oreg = smem + wave * (QB * 32 * OPITCH);
    static_assert(4 * QB * 32 * OPITCH <= 2 * STAGE, "the output staging must fit the K / V stages");
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const unsigned lu = __float_as_uint(o[qb][1][8]);
        auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
        const float inv = 1.0f / __uint_as_float(sw[0]);              // the lower half's value in both halves
        if (wide || q_abs[qb] < len_q) {
            unsigned short* op = wide ? reinterpret_cast<unsigned short*>(oreg + (qb * 32 + ql) * OPITCH) + 4 * h
                                      : P.out + (int64_t)(q_tok0 + q_abs[qb]) * P.o_stride + hd * DH + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int gq = 0; gq < (dt == 0 ? 4 : 2); ++gq) {
                    u32x2_t pk;
                    pk[0] = Op16<DT>::pack2(o[qb][dt][4 * gq] * inv, o[qb][dt][4 * gq + 1] * inv);
                    pk[1] = Op16<DT>::pack2(o[qb][dt][4 * gq + 2] * inv, o[qb][dt][4 * gq + 3] * inv);
                    *reinterpret_cast<u32x2_t*>(op + 32 * dt + 8 * gq) = pk;
                }
        }
    }
    if (wide && active) {                                              // wave-local: a wave reads back only what it wrote
        constexpr int NCH = QB * 32 * (DH / 8), NIT = (NCH + 63) / 64;
        const int q_w0 = qb0 + wave * QB * 32;
        u32x4_t v[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int item = lane + 64 * i, row = item / (DH / 8), cc = item - row * (DH / 8);
            v[i] = *reinterpret_cast<const u32x4_t*>(oreg + (item < NCH ? row * OPITCH + cc * 16 : 0));
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int item = lane + 64 * i, row = item / (DH / 8), cc = item - row * (DH / 8);
            if (item < NCH && q_w0 + row < len_q)
                *reinterpret_cast<u32x4_t*>(P.out + (int64_t)(q_tok0 + q_w0 + row) * P.o_stride + hd * DH + cc * 8) = v[i];
        }
    }
}

#ifdef VF_TUNING   // attn_x32pp_kernel: anti-phase 8-wave cross attention, 12 % slower (profiles/r04_l)
#include "tuning/attn_x32pp.inc"
#endif

// Short sequences (<= 256 queries and keys: seq2reg windows, the gene stream): one block per (sequence, head).
// The whole K/V of the sequence is staged into LDS once (all loads issued before the first store: one load latency,
// no per-tile barriers) and is fetched once per (sequence, head) instead of once per 64-query block.
// Wave w owns the query groups w, w+4, ... (16 queries each), QG = ceil(max_seqlen_q / 64) of them, processed
// together so that every K / V fragment read from LDS feeds QG MFMAs.
// QL: q carries the softmax scale (AttnParams::q_log2) -> attn_tile mode 2 (integer running maxima), the arithmetic of the
// tiled kernels, which serve the same queries in other batch geometries.  (The no-maximum form of attn_x32_kernel was
// measured here too: -2 % on the gene stream's self attention -- this kernel waits for its loads, not for its exponentials --
// and it cannot be bit-identical to mode 2, whose s - m rounds; not kept.)
template <int DH, int QG, bool ALIBI, int DT = VF_BF16, bool QL = false>
__global__ __launch_bounds__(256, (QG >= 2 ? 2 : 4)) void attn_short_kernel(AttnParams P, int k_rows) {
    constexpr int SM = QL ? 2 : 0;
    using frag_t = typename Op16<DT>::frag;
    constexpr int CPR = DH / 8;
    constexpr int NDT = DH / 16;
    constexpr int VROW = VLayout<DH>::ROW;
    constexpr int K_ROW_BYTES = KLayout<DH>::ROW, KS = KLayout<DH>::KS;
    static_assert(K_ROW_BYTES == 128, "the short-sequence kernel is written for dh <= 64");
    constexpr int MAXIT = 8;                                      // 256 rows x 8 chunk slots / 256 threads
    extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
    char* const sK0 = smem_dyn;
    char* const sV0 = smem_dyn + k_rows * K_ROW_BYTES;

    int seq, h, qblk;
    if (!block_coords(P, seq, h, qblk)) return;
    const int q_tok0 = P.cu_q[seq], len_q = P.cu_q[seq + 1] - q_tok0;
    const int k_tok0 = P.cu_k[seq], len_k = P.cu_k[seq + 1] - k_tok0;
    if (len_q <= 0) return;
    if (len_k <= 0) {
        zero_rows<DH>(P, q_tok0, len_q, h);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int nkv = (len_k + BKV - 1) / BKV;
    const int nchunks = nkv * BKV * 8;
#ifdef VF_SHORT_PROF
    unsigned long long pt0 = __builtin_readcyclecounter(), pt1, ptv[5] = {0, 0, 0, 0, 0};
#define SH_MARK(i) { pt1 = __builtin_readcyclecounter(); ptv[i] = pt1 - pt0; pt0 = pt1; }
#else
#define SH_MARK(i)
#endif

    // ---- Q fragments of this wave's query groups (issued before the K/V loads so that both are in flight together)
    frag_t qf[QG][KS];
    int q_abs[QG];
    float q_pos[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        q_abs[qg] = (wave + 4 * qg) * 16 + r;
        const int row = q_abs[qg] < len_q ? q_abs[qg] : len_q - 1;
        const unsigned short* qp = P.q + (int64_t)(q_tok0 + row) * P.q_stride + h * DH;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int d0 = 32 * ks + 8 * g;
            u32x4_t raw = q_pad_chunk<DT>(d0, DH);
            if (d0 < DH) raw = *reinterpret_cast<const u32x4_t*>(qp + d0);
            qf[qg][ks] = *reinterpret_cast<frag_t*>(&raw);
        }
        q_pos[qg] = (float)(q_abs[qg] + (P.q_at_start ? 0 : len_k - len_q));
    }
    // ---- stage K (swizzled, zero pad chunks) and V; rows >= len_k replicate the last key (finite, masked later)
    const unsigned short* kbase = P.k + (int64_t)k_tok0 * P.k_stride + h * DH;
    const unsigned short* vbase = P.v + (int64_t)k_tok0 * P.v_stride + h * DH;
    {
        u32x4_t kbuf[MAXIT], vbuf[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int ci = tid + 256 * it;
            const int row = ci >> 3, cc = ci & 7;
            kbuf[it] = (HwMask<DH>::value && cc == CPR) ? mask_chunk<DT>(row < len_k) : (u32x4_t){0u, 0u, 0u, 0u};
            vbuf[it] = (u32x4_t){0u, 0u, 0u, 0u};
            if (ci < nchunks && cc < CPR) {
                const int key = row < len_k ? row : len_k - 1;
                kbuf[it] = *reinterpret_cast<const u32x4_t*>(kbase + (int64_t)key * P.k_stride + cc * 8);
                vbuf[it] = *reinterpret_cast<const u32x4_t*>(vbase + (int64_t)key * P.v_stride + cc * 8);
            }
        }
        SH_MARK(0)                                   // loads issued
#ifdef VF_SHORT_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SH_MARK(1)                                   // loads landed
#endif
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int ci = tid + 256 * it;
            const int row = ci >> 3, cc = ci & 7;
            if (ci < nchunks) {
                *reinterpret_cast<u32x4_t*>(sK0 + row * K_ROW_BYTES + ((cc ^ ((row >> 1) & 7)) << 4)) = kbuf[it];
                if (cc < CPR) *reinterpret_cast<u32x4_t*>(sV0 + row * VROW + (cc << 4)) = vbuf[it];
            }
        }
    }

    const float c = P.scale_log2;
    const float slope2 = ALIBI ? P.slopes[h] * 1.4426950408889634f : 0.f;
    f32x4_t o[QG][NDT];
    float m_run[QG];
    f32x4_t l_acc[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m_run[qg] = -INFINITY;
        l_acc[qg] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[qg][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    SH_MARK(2)                                       // LDS written, barrier passed
    if (wave * 16 >= len_q) return;                                // this wave owns no valid query (wave-uniform)

    // A wave whose LAST query group lies past the sequence (201 queries = 13 groups of 16 over 4 waves: waves 1-3 own 3
    // valid groups, wave 0 four) runs the tile function for one group fewer instead of computing a group of clamped
    // rows nobody stores: 19 % of the block's matrix and vector work at the gene stream's 201-token sequences.  The
    // arrays of the first QG - 1 groups are prefixes of the QG-group arrays.
    const bool last_group_valid = (wave + 4 * (QG - 1)) * 16 < len_q;             // wave-uniform
    // the last tile in its 16 / 32-key form when the remainder allows (201 keys = 3 tiles + 9 keys: 3/16 of the arithmetic)
    const int tl = nkv - 1, rem = len_k - tl * BKV;
    const int n_full = rem <= 32 ? tl : nkv;
    if (QG == 1 || last_group_valid) {
        for (int t = 0; t < n_full; ++t)
            attn_tile<DH, QG, ALIBI, DT, 0, 4, SM>(sK0 + t * BKV * K_ROW_BYTES, sV0 + t * BKV * VROW, t * BKV, len_k, r, g, c,
                                                   slope2, qf, q_pos, o, m_run, l_acc);
        if (rem <= 32)
            attn_tile_short<DH, QG, ALIBI, DT, SM>(rem, sK0 + tl * BKV * K_ROW_BYTES, sV0 + tl * BKV * VROW, tl * BKV, len_k,
                                                   r, g, c, slope2, qf, q_pos, o, m_run, l_acc);
    } else {
        constexpr int Q1 = QG > 1 ? QG - 1 : 1;
        auto& qf1 = reinterpret_cast<const frag_t(&)[Q1][KS]>(qf);
        auto& q_pos1 = reinterpret_cast<const float(&)[Q1]>(q_pos);
        auto& o1 = reinterpret_cast<f32x4_t(&)[Q1][NDT]>(o);
        auto& m_run1 = reinterpret_cast<float(&)[Q1]>(m_run);
        auto& l_acc1 = reinterpret_cast<f32x4_t(&)[Q1]>(l_acc);
        for (int t = 0; t < n_full; ++t)
            attn_tile<DH, Q1, ALIBI, DT, 0, 4, SM>(sK0 + t * BKV * K_ROW_BYTES, sV0 + t * BKV * VROW, t * BKV, len_k, r, g, c,
                                                   slope2, qf1, q_pos1, o1, m_run1, l_acc1);
        if (rem <= 32)
            attn_tile_short<DH, Q1, ALIBI, DT, SM>(rem, sK0 + tl * BKV * K_ROW_BYTES, sV0 + tl * BKV * VROW, tl * BKV, len_k,
                                                   r, g, c, slope2, qf1, q_pos1, o1, m_run1, l_acc1);
    }

    SH_MARK(3)                                       // tiles computed
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const float l = l_acc[qg][0];
        const float inv = 1.0f / l;
        if (q_abs[qg] < len_q) {
            unsigned short* op = P.out + (int64_t)(q_tok0 + q_abs[qg]) * P.o_stride + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                u32x2_t pk;
                pk[0] = Op16<DT>::pack2(o[qg][dt][0] * inv, o[qg][dt][1] * inv);
                pk[1] = Op16<DT>::pack2(o[qg][dt][2] * inv, o[qg][dt][3] * inv);
                *reinterpret_cast<u32x2_t*>(op + 16 * dt) = pk;
            }
        }
    }
#ifdef VF_SHORT_PROF
    SH_MARK(4)                                       // stores issued
    if (lane == 0 && wave == 0 && (blockIdx.x & 63) == 0) {
        unsigned long long* pp = P.prof + (blockIdx.x >> 6) * 8;
        for (int i = 0; i < 5; ++i) pp[i] = ptv[i];
        pp[5] = pt0;                                 // end stamp
        pp[6] = pt0 - (ptv[0] + ptv[1] + ptv[2] + ptv[3] + ptv[4]);      // start stamp
    }
#endif
#undef SH_MARK
}

// The same kernel for THREE resident blocks per CU (the gene stream's 201-token sequences).  attn_short_kernel at 4 query
// groups per wave needs 232 VGPRs and 57 KB of LDS: two blocks per CU, whose two long phases -- ~5 us waiting for 58 KB of
// Q / K / V at the latency this kernel's own traffic produces, ~6 us of tile arithmetic on the wave that owns 4 of the 13 query
// groups (profiles/r03_o_attn_short_phase_probe.log) -- hide behind each other and nothing else.  Here a wave makes TWO PASSES
// over the keys with 2 query groups each (<= 170 VGPRs: three waves per SIMD; the K / V fragments are read twice from LDS,
// which has the bandwidth), and the LDS image holds only the rows the tiles read -- keys rounded up to the tail tile's 16 /
// 32 / 64 for K and to 32 / 64 for V (201 keys: 208 K rows + 224 V rows = 47 KB) -- so three blocks fit: a third block's loads
// and a second computing wave per SIMD.  Arithmetic per query: attn_tile's, identical to every other kernel.
// NPASS = 1: sequences of <= 128 queries (seq2reg windows at dh = 64: one block per (window, head) instead of two 64-query
// blocks of the tiled kernel that each fetch K / V and wait for it).
// NPASS = 2 at dh = 64 (round 4): 129-256-token chunks (seq2reg's gene chunks), 62 KB image = two blocks per CU.  The register
// budget stays that of three waves per SIMD: at (256, 2) the compiler hoists loads up to 256 VGPRs + 47 spilled and the launch
// takes 2369 us instead of 1329 us (168 VGPRs, 12 spilled) -- profiles/r04_k.
// ROWS: Q / K / V rows are fetched through AttnParams::q_rows / kv_rows (one index load per row in front of the data loads).
template <int DH, bool ALIBI, int DT = VF_BF16, bool QL = false, int NPASS = 2, bool ROWS = false>
__global__ __launch_bounds__(256, 3) void attn_short2_kernel(AttnParams P, int k_rows) {
    constexpr int SM = QL ? 2 : 0;
    using frag_t = typename Op16<DT>::frag;
    constexpr int CPR = DH / 8, NDT = DH / 16, QG = 2;
    constexpr int VROW = VLayout<DH>::ROW;
    constexpr int K_ROW_BYTES = KLayout<DH>::ROW, KS = KLayout<DH>::KS;
    static_assert(K_ROW_BYTES == 128, "the short-sequence kernel is written for dh <= 64");
    constexpr int MAXIT = 8;                                      // 256 rows x 8 chunk slots / 256 threads
    extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
    char* const sK0 = smem_dyn;
    char* const sV0 = smem_dyn + k_rows * K_ROW_BYTES;

    int seq, h, qblk;
    if (!block_coords(P, seq, h, qblk)) return;
    const int q_tok0 = P.cu_q[seq], len_q = P.cu_q[seq + 1] - q_tok0;
    const int k_tok0 = P.cu_k[seq], len_k = P.cu_k[seq + 1] - k_tok0;
    if (len_q <= 0) return;
    if (len_k <= 0) {
        zero_rows<DH>(P, q_tok0, len_q, h);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int nkv = (len_k + BKV - 1) / BKV;
    const int tl = nkv - 1, rem = len_k - tl * BKV;               // the last tile and the keys it holds
    const int n_full = rem <= 32 ? tl : nkv;
    const int rows_k = tl * BKV + (rem <= 16 ? 16 : rem <= 32 ? 32 : BKV);     // K rows the tiles read
    const int rows_v = tl * BKV + (rem <= 32 ? 32 : BKV);                      // V rows (a PV block spans 32 keys)

    // ---- Q fragments of all four query groups of this wave (issued before the K/V loads: everything in flight together)
    frag_t qf[NPASS][QG][KS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const int qa = (wave + 4 * (QG * ps + qg)) * 16 + r;
            const int row = qa < len_q ? qa : len_q - 1;
            const int64_t qrow = (ROWS && P.q_rows) ? P.q_rows[q_tok0 + row] : (int64_t)(q_tok0 + row);
            const unsigned short* qp = P.q + qrow * P.q_stride + h * DH;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int d0 = 32 * ks + 8 * g;
                u32x4_t raw = q_pad_chunk<DT>(d0, DH);
                if (d0 < DH) raw = *reinterpret_cast<const u32x4_t*>(qp + d0);
                qf[ps][qg][ks] = *reinterpret_cast<frag_t*>(&raw);
            }
        }
    // ---- stage K (swizzled, zero pad chunks) and V; rows >= len_k replicate the last key (finite, masked later)
    const unsigned short* kbase = P.k + h * DH;
    const unsigned short* vbase = P.v + h * DH;
    {
        u32x4_t kbuf[MAXIT], vbuf[MAXIT];
        int64_t src[MAXIT];                          // source row of every staged row: all index loads in flight before the data loads
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int row = (tid + 256 * it) >> 3;
            const int key = row < len_k ? row : len_k - 1;
            src[it] = (ROWS && P.kv_rows) ? (row < rows_v ? P.kv_rows[k_tok0 + key] : 0) : (int64_t)(k_tok0 + key);
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int ci = tid + 256 * it;
            const int row = ci >> 3, cc = ci & 7;
            kbuf[it] = (HwMask<DH>::value && cc == CPR) ? mask_chunk<DT>(row < len_k) : (u32x4_t){0u, 0u, 0u, 0u};
            vbuf[it] = (u32x4_t){0u, 0u, 0u, 0u};
            if (row < rows_v && cc < CPR) {
                if (row < rows_k) kbuf[it] = *reinterpret_cast<const u32x4_t*>(kbase + src[it] * P.k_stride + cc * 8);
                vbuf[it] = *reinterpret_cast<const u32x4_t*>(vbase + src[it] * P.v_stride + cc * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int ci = tid + 256 * it;
            const int row = ci >> 3, cc = ci & 7;
            if (row < rows_k) *reinterpret_cast<u32x4_t*>(sK0 + row * K_ROW_BYTES + ((cc ^ ((row >> 1) & 7)) << 4)) = kbuf[it];
            if (row < rows_v && cc < CPR) *reinterpret_cast<u32x4_t*>(sV0 + row * VROW + (cc << 4)) = vbuf[it];
        }
    }
    const float c = P.scale_log2;
    const float slope2 = ALIBI ? P.slopes[h] * 1.4426950408889634f : 0.f;
    __syncthreads();

#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int g0 = wave + 4 * QG * ps;                         // this pass's first query group (wave-uniform)
        if (g0 * 16 >= len_q) break;
        int q_abs[QG];
        float q_pos[QG], m_run[QG];
        f32x4_t o[QG][NDT], l_acc[QG];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            q_abs[qg] = (g0 + 4 * qg) * 16 + r;
            q_pos[qg] = (float)(q_abs[qg] + (P.q_at_start ? 0 : len_k - len_q));
            m_run[qg] = -INFINITY;
            l_acc[qg] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[qg][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#ifdef VF_TUNING
        if (P.dbg & 1) {                                           // ceiling probe: keep the Q fragments alive, compute nothing
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) {
                asm volatile("" ::"v"(qf[ps][qg][0]), "v"(qf[ps][qg][1]));
                l_acc[qg] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
            }
        } else
#endif
        if ((g0 + 4) * 16 < len_q) {                               // both groups hold valid queries
            for (int t = 0; t < n_full; ++t)
                attn_tile<DH, QG, ALIBI, DT, 0, 4, SM>(sK0 + t * BKV * K_ROW_BYTES, sV0 + t * BKV * VROW, t * BKV, len_k, r, g, c,
                                                       slope2, qf[ps], q_pos, o, m_run, l_acc);
            if (rem <= 32)
                attn_tile_short<DH, QG, ALIBI, DT, SM>(rem, sK0 + tl * BKV * K_ROW_BYTES, sV0 + tl * BKV * VROW, tl * BKV, len_k,
                                                       r, g, c, slope2, qf[ps], q_pos, o, m_run, l_acc);
        } else {
            auto& qf1 = reinterpret_cast<const frag_t(&)[1][KS]>(qf[ps]);
            auto& q_pos1 = reinterpret_cast<const float(&)[1]>(q_pos);
            auto& o1 = reinterpret_cast<f32x4_t(&)[1][NDT]>(o);
            auto& m_run1 = reinterpret_cast<float(&)[1]>(m_run);
            auto& l_acc1 = reinterpret_cast<f32x4_t(&)[1]>(l_acc);
            for (int t = 0; t < n_full; ++t)
                attn_tile<DH, 1, ALIBI, DT, 0, 4, SM>(sK0 + t * BKV * K_ROW_BYTES, sV0 + t * BKV * VROW, t * BKV, len_k, r, g, c,
                                                      slope2, qf1, q_pos1, o1, m_run1, l_acc1);
            if (rem <= 32)
                attn_tile_short<DH, 1, ALIBI, DT, SM>(rem, sK0 + tl * BKV * K_ROW_BYTES, sV0 + tl * BKV * VROW, tl * BKV, len_k,
                                                      r, g, c, slope2, qf1, q_pos1, o1, m_run1, l_acc1);
        }
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const float inv = 1.0f / l_acc[qg][0];
            if (q_abs[qg] < len_q) {
                unsigned short* op = P.out + (int64_t)(q_tok0 + q_abs[qg]) * P.o_stride + h * DH + 4 * g;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    u32x2_t pk;
                    pk[0] = Op16<DT>::pack2(o[qg][dt][0] * inv, o[qg][dt][1] * inv);
                    pk[1] = Op16<DT>::pack2(o[qg][dt][2] * inv, o[qg][dt][3] * inv);
                    *reinterpret_cast<u32x2_t*>(op + 16 * dt) = pk;
                }
            }
        }
    }
}

// A/B toggles of the one-block-per-sequence kernels, read ONCE per process and shared by the dispatch (launch_attn) and by
// rows_supported, which must agree on every geometry (round-5 advice: rows_supported re-read them on every call)
static int env_short2() { static const int v = vf_tuning_env("VF_ATTN_SHORT2", 1); return v; }      // 0: the one-pass kernel
static int env_short64() { static const int v = vf_tuning_env("VF_ATTN_SHORT64", 1); return v; }    // 0: the tiled kernel at dh 64

// Only attn_short2_kernel's ROWS instantiations read AttnParams::q_rows / kv_rows: every other launcher refuses a row map
// instead of silently attending over the wrong rows (round-5 advice).
static int no_row_map(const AttnParams& P, const char* kernel) {
    if (!P.q_rows && !P.kv_rows) return VF_OK;
    vf_set_error("vf_attn_varlen_fwd_rows: %s has no row-map form (ask vf_attn_rows_supported first)", kernel);
    return VF_ERR_INVALID_ARG;
}

// fills the grid decomposition of block_coords; returns the (padded) 1-D grid size
static unsigned set_grid(AttnParams& P, int n_seq, int nqb) {
    const long total = (long)n_seq * P.H * nqb;
    P.nqb = nqb;
    P.total = (int)total;
    P.chunk = (int)((total + 7) / 8);
    return 8u * (unsigned)P.chunk;
}

// dynamic LDS of the tiled kernel: two (K tile + V tile) stages
template <int DH> constexpr int attn_fwd_lds() { return 2 * (KLayout<DH>::TILE + VLayout<DH>::TILE); }

template <int DH, int QG, bool ALIBI, int DT, int DBG, bool QL>
int launch_fwd_k(const AttnParams& P, dim3 grid, hipStream_t st) {
    auto kern = attn_fwd_kernel<DH, QG, ALIBI, DT, DBG, QL>;
    constexpr int lds = attn_fwd_lds<DH>();
    if (lds > 65536) {                             // dh = 128 only: above the default dynamic-LDS limit
        static bool attr_set[VF_MAX_DEVICES] = {};
        const int dev = vf_current_device();
        if (dev < 0 || !attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess) {
                (void)hipGetLastError();
                vf_set_error("vf_attn_varlen_fwd: cannot reserve %d bytes of LDS", lds);
                return VF_ERR_LAUNCH;
            }
            if (dev >= 0) attr_set[dev] = true;
        }
    }
    vf_note_kernel(1, QG == 1 ? "attn_fwd_kernel<64-query blocks>" : QG == 2 ? "attn_fwd_kernel<128-query blocks>" : "attn_fwd_kernel<256-query blocks>");
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, P);
    VF_CHECK_LAUNCH("vf_attn_varlen_fwd");
    return VF_OK;
}

// q carrying the softmax scale (AttnParams::q_log2) selects the integer-maximum instantiation (attn_tile mode 2)
template <int DH, int QG, bool ALIBI, int DT, int DBG = 0>
int launch_fwd(const AttnParams& P, dim3 grid, hipStream_t st) {
    if (const int rc = no_row_map(P, "attn_fwd_kernel")) return rc;
    if (DBG == 0 && P.q_log2) return launch_fwd_k<DH, QG, ALIBI, DT, 0, true>(P, grid, st);
    return launch_fwd_k<DH, QG, ALIBI, DT, DBG, false>(P, grid, st);
}

template <int DH, int QG, bool ALIBI, int DT, bool QL>
int launch_short_k(AttnParams P, int n_seq, int max_k, hipStream_t st) {
    const int k_rows = ((max_k + BKV - 1) / BKV) * BKV;
    const int lds = k_rows * (KLayout<DH>::ROW + VLayout<DH>::ROW);
    auto kern = attn_short_kernel<DH, QG, ALIBI, DT, QL>;
    static bool attr_set[VF_MAX_DEVICES] = {};    // the attribute is per device (and per instantiation)
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                256 * (KLayout<DH>::ROW + VLayout<DH>::ROW)) != hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_attn_varlen_fwd: cannot reserve LDS");
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    const unsigned nblk = set_grid(P, n_seq, 1);
    vf_note_kernel(1, "attn_short_kernel");
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, P, k_rows);
    VF_CHECK_LAUNCH("vf_attn_varlen_fwd");
    return VF_OK;
}

// rows of the LDS image for sequences of at most max_k keys (monotone in max_k: a shorter sequence never reads beyond them)
static inline void short2_rows(int max_k, int& kr, int& vr) {
    const int tl = (max_k - 1) / BKV, rem = max_k - tl * BKV;
    kr = tl * BKV + (rem <= 16 ? 16 : rem <= 32 ? 32 : BKV);
    vr = tl * BKV + (rem <= 32 ? 32 : BKV);
}

template <int DH, bool ALIBI, int DT, bool QL, int NPASS, bool ROWS = false>
int launch_short2_k(AttnParams P, int n_seq, int max_k, hipStream_t st) {
    int kr, vr;
    short2_rows(max_k, kr, vr);
    const int lds = kr * KLayout<DH>::ROW + vr * VLayout<DH>::ROW;
    auto kern = attn_short2_kernel<DH, ALIBI, DT, QL, NPASS, ROWS>;
    static bool attr_set[VF_MAX_DEVICES] = {};
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                256 * (KLayout<DH>::ROW + VLayout<DH>::ROW)) != hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_attn_varlen_fwd: cannot reserve LDS");
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    const unsigned nblk = set_grid(P, n_seq, 1);
    vf_note_kernel(1, ROWS ? (NPASS == 1 ? "attn_short2_kernel<1 pass,rows>" : "attn_short2_kernel<2 passes,rows>")
                           : (NPASS == 1 ? "attn_short2_kernel<1 pass>" : "attn_short2_kernel<2 passes>"));
#ifdef VF_TUNING
    P.dbg = vf_tuning_env("VF_ATTN_SHORT_DBG", 0);
    const int lds_probe = vf_tuning_env("VF_ATTN_SHORT_LDS", 0);     // occupancy probe: request this many bytes instead (>= lds)
    if (lds_probe > lds) {
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds_probe, st, P, kr);
        VF_CHECK_LAUNCH("vf_attn_varlen_fwd");
        return VF_OK;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, P, kr);
    VF_CHECK_LAUNCH("vf_attn_varlen_fwd");
    return VF_OK;
}

template <int DH, bool ALIBI, int DT, int NPASS = 2>
int launch_short2(AttnParams P, int n_seq, int max_k, hipStream_t st) {
    if (P.q_rows || P.kv_rows) {                 // only the model's call form (pre-scaled q) has a row-map instantiation
        if constexpr ((DH == 64 && !ALIBI) || (DH == 48 && ALIBI && NPASS == 2)) {
            if (P.q_log2) return launch_short2_k<DH, ALIBI, DT, true, NPASS, true>(P, n_seq, max_k, st);
        }
        vf_set_error("vf_attn_varlen_fwd_rows: no row-map kernel for this geometry (ask vf_attn_rows_supported first)");
        return VF_ERR_INVALID_ARG;
    }
    if (P.q_log2) return launch_short2_k<DH, ALIBI, DT, true, NPASS>(P, n_seq, max_k, st);
    return launch_short2_k<DH, ALIBI, DT, false, NPASS>(P, n_seq, max_k, st);
}

template <int DH, int QG, bool ALIBI, int DT>
int launch_short(AttnParams P, int n_seq, int max_k, hipStream_t st) {
    if (const int rc = no_row_map(P, "attn_short_kernel")) return rc;
    if (P.q_log2) return launch_short_k<DH, QG, ALIBI, DT, true>(P, n_seq, max_k, st);
    return launch_short_k<DH, QG, ALIBI, DT, false>(P, n_seq, max_k, st);
}

template <int DT, int QB>
int launch_x32(const AttnParams& P, dim3 grid, hipStream_t st) {
    if (const int rc = no_row_map(P, "attn_x32_kernel")) return rc;
    constexpr int lds = 2 * BKV * (112 + 192);
    static const int nomax = vf_tuning_env("VF_ATTN_NOMAX", 1);    // 0: running maximum always (A/B)
    vf_note_kernel(1, QB == 2 ? "attn_x32_kernel<64 queries per wave>" : "attn_x32_kernel<32 queries per wave>");
    if (P.q_log2 && nomax) hipLaunchKernelGGL((attn_x32_kernel<DT, QB, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((attn_x32_kernel<DT, QB, false>), grid, dim3(256), lds, st, P);
    VF_CHECK_LAUNCH("vf_attn_varlen_fwd");
    return VF_OK;
}

#ifdef VF_TUNING
template <int DT>
int launch_x32pp(const AttnParams& P, dim3 grid, hipStream_t st) {
    constexpr int lds = 4 * BKV * (112 + 192);                      // 77 824 bytes: above the default dynamic-LDS limit
    auto kern = attn_x32pp_kernel<DT>;
    static bool attr_set[VF_MAX_DEVICES] = {};
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_attn_varlen_fwd: cannot reserve %d bytes of LDS", lds);
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, P);
    VF_CHECK_LAUNCH("vf_attn_varlen_fwd");
    return VF_OK;
}
#endif

// Geometries the row-map form (AttnParams::q_rows / kv_rows) serves: exactly those launch_attn sends to attn_short2_kernel
// in the model's call form -- seq2reg windows / chunks (dh 64, no bias) and the gene stream's self attention (dh 48, ALiBi,
// 129-256 tokens).  Everything else gathers the rows first (vf_gather_rows_bf16) and calls the plain entry: same bits.
static bool rows_supported(int dh, bool alibi, long n_seq, int H, int max_q, int max_k, bool q_log2) {
    if (!q_log2 || max_q <= 0 || max_k <= 0 || max_q > 256 || max_k > 256) return false;
    int kr, vr;
    short2_rows(max_k, kr, vr);
    // (exactly launch_attn's conditions, in its order: a geometry it sends elsewhere has no row map)
    if (dh == 48 && alibi)
        return max_q > 128 && env_short2() && 3 * (kr * KLayout<48>::ROW + vr * VLayout<48>::ROW) <= 160 * 1024;
    if (dh == 64 && !alibi) {
        if (!env_short64() || n_seq * H < 1024) return false;
        const int image = kr * KLayout<64>::ROW + vr * VLayout<64>::ROW;
        if (max_q <= 128 && max_k <= 128) return 3 * image <= 160 * 1024;
        if (max_q > 128) return 2 * image <= 160 * 1024;          // (<= 256 both: checked above)
        return false;                                               // max_q <= 128 < max_k: the tiled kernel serves it
    }
    return false;
}

template <int DH, bool ALIBI, int DT>
int launch_attn(AttnParams P, int n_seq, int max_q, int max_k, hipStream_t st) {
    if ((P.q_rows || P.kv_rows) && !rows_supported(DH, ALIBI, n_seq, P.H, max_q, max_k, P.q_log2 != 0)) {
        vf_set_error("vf_attn_varlen_fwd_rows: no row-map kernel for dh=%d alibi=%d n_seq=%d H=%d max_q=%d max_k=%d "
                     "(vf_attn_rows_supported says which geometries have one; gather the rows first otherwise)",
                     DH, (int)ALIBI, n_seq, P.H, max_q, max_k);
        return VF_ERR_INVALID_ARG;
    }
    // Measured on MI355X (scripts/attn_bench.py, 8 genes): the one-block-per-(sequence, head) kernel wins for the gene
    // stream (201-token sequences, dh 48: 396 vs 485 us); for seq2reg windows / chunks (dh 64, <= 200 tokens) the tiled
    // kernel with 64-query blocks is faster (460 vs 613 us on 200-token chunks: more blocks in flight, fewer registers;
    // the K/V re-reads of its query blocks hit the XCD's L2 thanks to block_coords).
    // dh = 48 without positional bias (every cross attention of the modulator): ONE kernel for every geometry, so that a
    // gene's result cannot depend on the batch it is evaluated in (the query-block size only changes which wave owns a
    // query, never the arithmetic of a query: tests test_cfg3 / test_headline batch independence at 1e-5).  64 queries
    // per wave once that leaves >= 8 blocks per CU (the batched gene -> CRE cross attention), else 32 (3 waves / SIMD).
    if constexpr (DH == 48 && !ALIBI) {
        const int x32 = vf_tuning_env("VF_ATTN_X32", 1);             // 0: the 16x16x32 kernels (A/B; read per launch in the tuning library)
        if (x32) {
#ifdef VF_TUNING   // VF_ATTN_X32PP=1: the anti-phase experiment (attn_x32pp_kernel: slower, see its header)
            static const int pp = vf_tuning_env("VF_ATTN_X32PP", 0);
            static const int nomax = vf_tuning_env("VF_ATTN_NOMAX", 1);
            if (pp != 0 && nomax != 0 && x32 != 2 && P.q_log2 && (long)n_seq * P.H * ((max_q + 511) / 512) >= 1024)
                return launch_x32pp<DT>(P, dim3(set_grid(P, n_seq, (max_q + 511) / 512)), st);
#endif
            if (x32 != 2 && (long)n_seq * P.H * ((max_q + 255) / 256) >= 2048)
                return launch_x32<DT, 2>(P, dim3(set_grid(P, n_seq, (max_q + 255) / 256)), st);
            return launch_x32<DT, 1>(P, dim3(set_grid(P, n_seq, (max_q + 127) / 128)), st);
        }
    }
    if constexpr (DH <= 48) {
        if (max_q > 128 && max_q <= 256 && max_k <= 256) {
            // three resident blocks per CU (two passes of 2 query groups, trimmed LDS image) once the image leaves room
            // for them; VF_ATTN_SHORT2=0: the one-pass kernel (A/B)
            const int short2 = env_short2();
            int kr, vr;
            short2_rows(max_k, kr, vr);
            if (short2 && 3 * (kr * KLayout<DH>::ROW + vr * VLayout<DH>::ROW) <= 160 * 1024)
                return launch_short2<DH, ALIBI, DT>(P, n_seq, max_k, st);
            if (max_q <= 192) return launch_short<DH, 3, ALIBI, DT>(P, n_seq, max_k, st);
            return launch_short<DH, 4, ALIBI, DT>(P, n_seq, max_k, st);
        }
    }
    if constexpr (DH == 64 || DH == 32) {
        // seq2reg windows (<= 128 tokens at dh = 64: a 36 KB image): one block per (window, head) with the whole K / V in
        // LDS instead of two 64-query blocks that each fetch K / V and wait for it (VF_ATTN_SHORT64=0: the tiled kernel).
        // dh = 32 (a tokenizer geometry the real checkpoint might have, scripts/s2r_dims_sweep.py) takes the same kernel
        // since round 6: the tiled kernel ran its windows at 3.2 TB/s against 4.1 for dh = 64.
        const int short64 = env_short64();
        if (short64 && max_q <= 128 && max_k <= 128 && (long)n_seq * P.H >= 1024) {
            int kr, vr;
            short2_rows(max_k, kr, vr);
            if (3 * (kr * KLayout<DH>::ROW + vr * VLayout<DH>::ROW) <= 160 * 1024)
                return launch_short2<DH, ALIBI, DT, 1>(P, n_seq, max_k, st);
        }
        // 129-256-token chunks (seq2reg's 200-token gene chunks): the same kernel in its two-pass form, one block per (chunk,
        // head) with a 62 KB image (two resident blocks per CU) instead of four 64-query blocks of the tiled kernel that each
        // stage all keys: 1544 -> 1329 us per launch at 32 genes, bit-identical (profiles/r04_k; VF_ATTN_SHORT64=0: tiled kernel)
        if (DH == 64 && short64 && max_q > 128 && max_q <= 256 && max_k <= 256 && (long)n_seq * P.H >= 1024) {
            int kr, vr;
            short2_rows(max_k, kr, vr);
            if (2 * (kr * KLayout<DH>::ROW + vr * VLayout<DH>::ROW) <= 160 * 1024)
                return launch_short2<DH, ALIBI, DT, 2>(P, n_seq, max_k, st);
        }
    }
    // long query streams: 2 query groups per wave (halves K/V LDS traffic per MFMA; for seq2reg windows, 70-200 queries,
    // 128-query blocks measured the same as 64-query blocks: 800 / 428 us either way, the kernel runs at 4 TB/s there);
    // short ones (seq2reg windows, gene stream): 64-query blocks to limit tail waste.
    // 2 query groups per wave only when that still leaves >= 4 blocks per CU (measured: CRE stream, 256 blocks, is
    // 15% faster with 64-query blocks; the 10^4-query gene->CRE cross attention is 17% faster with 128-query blocks).
    // dh = 128 (not a shape of the shipped model) always takes one query group per wave; dh = 96: below.
    if constexpr (DH <= 64) {
        if (max_q > 256 && (long)n_seq * P.H * ((max_q + 127) / 128) >= 1024) {
            const dim3 grid(set_grid(P, n_seq, (max_q + 127) / 128));
#ifdef VF_TUNING                                   // libvf_hip_tuning.so only (scripts/): ceiling-finding builds whose results are meaningless
            if constexpr (DH == 48 && !ALIBI && DT == VF_BF16) {
                static const int dbg = vf_tuning_env("VF_ATTN_DBG", 0);
                if (dbg == 1) return launch_fwd<48, 2, false, VF_BF16, 1>(P, grid, st);
                if (dbg == 2) return launch_fwd<48, 2, false, VF_BF16, 2>(P, grid, st);
                if (dbg == 3) return launch_fwd<48, 2, false, VF_BF16, 3>(P, grid, st);
                if (dbg == 4) return launch_fwd<48, 2, false, VF_BF16, 4>(P, grid, st);
            }
#endif
            // 4 query groups per wave (256-query blocks) once that still leaves >= 8 blocks per CU: every K/V fragment
            // read feeds 4 MFMAs (gene->CRE cross attention at 8 genes: 948 vs 993 us; no gain at one gene, 1376 blocks)
            if constexpr (DH == 48 && !ALIBI) {
                if ((long)n_seq * P.H * ((max_q + 255) / 256) >= 2048)
                    return launch_fwd<48, 4, false, DT>(P, dim3(set_grid(P, n_seq, (max_q + 255) / 256)), st);
            }
            return launch_fwd<DH, 2, ALIBI, DT>(P, grid, st);
        }
    }
    if constexpr (DH == 96) {
        // dh = 96 (a tokenizer geometry the real checkpoint might have, scripts/s2r_dims_sweep.py): windows of 65-128 tokens
        // as ONE 128-query block per (window, head) -- two 64-query blocks would each stage all keys: 3.0 -> 3.5 TB/s.  (At
        // dh = 128 the two-group form needs all 256 VGPRs, one wave per SIMD, and is slower: 3.4 -> 3.1 TB/s.)
        if (max_q > 64 && max_q <= 128 && (long)n_seq * P.H >= 1024)
            return launch_fwd<DH, 2, ALIBI, DT>(P, dim3(set_grid(P, n_seq, 1)), st);
    }
    return launch_fwd<DH, 1, ALIBI, DT>(P, dim3(set_grid(P, n_seq, (max_q + 63) / 64)), st);
}

}  // namespace

template <int DT>
static int attn_dispatch(const void* q, const void* k, const void* v, void* out, int64_t q_stride,
                         int64_t k_stride, int64_t v_stride, int64_t o_stride, const int32_t* cu_seqlens_q,
                         const int32_t* cu_seqlens_k, int n_seq, int max_seqlen_q, int max_seqlen_k, int H,
                         int dh, const float* alibi_slopes, float scale, int flags, void* stream,
                         const int64_t* q_rows = nullptr, const int64_t* kv_rows = nullptr) {
    const int q_at_start = flags & VF_ATTN_Q_AT_START, q_log2 = (flags & VF_ATTN_Q_LOG2) ? 1 : 0;
    VF_REQUIRE((flags & ~(VF_ATTN_Q_AT_START | VF_ATTN_Q_LOG2)) == 0, "vf_attn_varlen_fwd: unknown flag bits 0x%x", flags);
    VF_REQUIRE(q && k && v && out && cu_seqlens_q, "vf_attn_varlen_fwd: null pointer");
    VF_REQUIRE(dh == 32 || dh == 48 || dh == 64 || dh == 96 || dh == 128,
               "vf_attn_varlen_fwd: head_dim %d not supported (32/48/64/96/128)", dh);
    VF_REQUIRE(H > 0 && H <= 65535 && n_seq >= 0, "vf_attn_varlen_fwd: H=%d n_seq=%d out of range", H, n_seq);
    VF_REQUIRE((long)n_seq * H * ((max_seqlen_q + 63) / 64) < (1L << 31) - 8,
               "vf_attn_varlen_fwd: n_seq * H * ceil(max_seqlen_q / 64) exceeds the grid limit");
    VF_REQUIRE(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && o_stride % 4 == 0,
               "vf_attn_varlen_fwd: row strides must keep 16-byte alignment");
    VF_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)out % 8 == 0),
               "vf_attn_varlen_fwd: pointers must be 16-byte aligned");
    if (n_seq == 0 || max_seqlen_q <= 0 || max_seqlen_k <= 0) return VF_OK;
    VF_REQUIRE(max_seqlen_k < (1 << 24) && k_stride < (1 << 24) && v_stride < (1 << 24) &&
                   (int64_t)max_seqlen_k * (k_stride > v_stride ? k_stride : v_stride) < (1LL << 31),
               "vf_attn_varlen_fwd: max_seqlen_k * row stride must stay below 2^31 elements");
    AttnParams P;
    P.q = (const unsigned short*)q; P.k = (const unsigned short*)k; P.v = (const unsigned short*)v;
    P.out = (unsigned short*)out;
    P.q_stride = q_stride; P.k_stride = k_stride; P.v_stride = v_stride; P.o_stride = o_stride;
    P.cu_q = cu_seqlens_q; P.cu_k = cu_seqlens_k ? cu_seqlens_k : cu_seqlens_q;
    P.slopes = alibi_slopes; P.scale_log2 = q_log2 ? 1.0f : scale * 1.4426950408889634f; P.H = H;
    P.q_at_start = q_at_start ? 1 : 0; P.q_log2 = q_log2;
    P.q_rows = q_rows; P.kv_rows = kv_rows;
#ifdef VF_TUNING
    P.dbg = 0;
#endif
    hipStream_t st = (hipStream_t)stream;
    const bool alibi = alibi_slopes != nullptr;
    switch (dh) {
        case 32: return alibi ? launch_attn<32, true, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st) : launch_attn<32, false, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st);
        case 48: return alibi ? launch_attn<48, true, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st) : launch_attn<48, false, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st);
        case 64: return alibi ? launch_attn<64, true, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st) : launch_attn<64, false, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st);
        case 96: return alibi ? launch_attn<96, true, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st) : launch_attn<96, false, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st);
        default: return alibi ? launch_attn<128, true, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st) : launch_attn<128, false, DT>(P, n_seq, max_seqlen_q, max_seqlen_k, st);
    }
}

extern "C" int vf_attn_varlen_fwd(const void* q, const void* k, const void* v, void* out, int64_t q_stride,
                                  int64_t k_stride, int64_t v_stride, int64_t o_stride, const int32_t* cu_seqlens_q,
                                  const int32_t* cu_seqlens_k, int n_seq, int max_seqlen_q, int max_seqlen_k, int H,
                                  int dh, const float* alibi_slopes, float scale, void* stream) {
    return attn_dispatch<VF_BF16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                         max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, 0, stream);
}

extern "C" int vf_attn_varlen_fwd_qstart(const void* q, const void* k, const void* v, void* out, int64_t q_stride,
                                         int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                         const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int n_seq,
                                         int max_seqlen_q, int max_seqlen_k, int H, int dh, const float* alibi_slopes,
                                         float scale, void* stream) {
    return attn_dispatch<VF_BF16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                         max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, 1, stream);
}

extern "C" int vf_attn_varlen_fwd_f16(const void* q, const void* k, const void* v, void* out, int64_t q_stride,
                                      int64_t k_stride, int64_t v_stride, int64_t o_stride, const int32_t* cu_seqlens_q,
                                      const int32_t* cu_seqlens_k, int n_seq, int max_seqlen_q, int max_seqlen_k, int H,
                                      int dh, const float* alibi_slopes, float scale, void* stream) {
    return attn_dispatch<VF_F16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                                 max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, 0, stream);
}

extern "C" int vf_attn_varlen_fwd_qstart_f16(const void* q, const void* k, const void* v, void* out, int64_t q_stride,
                                             int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                             const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int n_seq,
                                             int max_seqlen_q, int max_seqlen_k, int H, int dh, const float* alibi_slopes,
                                             float scale, void* stream) {
    return attn_dispatch<VF_F16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                                 max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, 1, stream);
}

extern "C" int vf_attn_varlen_fwd_v2(const void* q, const void* k, const void* v, void* out, int64_t q_stride,
                                     int64_t k_stride, int64_t v_stride, int64_t o_stride, const int32_t* cu_seqlens_q,
                                     const int32_t* cu_seqlens_k, int n_seq, int max_seqlen_q, int max_seqlen_k, int H,
                                     int dh, const float* alibi_slopes, float scale, int operand_dtype, int flags,
                                     void* stream) {
    if (operand_dtype == VF_BF16)
        return attn_dispatch<VF_BF16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                                      max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, flags, stream);
    VF_REQUIRE(operand_dtype == VF_F16, "vf_attn_varlen_fwd_v2: operand_dtype must be VF_BF16 or VF_F16");
    return attn_dispatch<VF_F16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                                 max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, flags, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Cross attention against keys that are COPIES OF A FEW DISTINCT ROWS: the CRE layers' context cross attention reads
// K / V = Wkv(Embedding(9)[label of every CRE]) (reference model_combined_modulator.py:168, layers.py:421-439), i.e. at most 9
// distinct key / value rows per head, each repeated count_c times in a gene's key sequence.  Softmax over the repeated keys is
// softmax over the distinct ones with log(count) added to the logit:
//     out = sum_c n_c exp(q . k_c) v_c / sum_c n_c exp(q . k_c)  =  sum_c softmax_c(q . k_c + log n_c) v_c
// -- the same function of (q, table, counts), N / 9 times less arithmetic and no [keys, 2D] gather of the table.  fp32 scores
// (bf16 / fp16 operands, products exact in fp32, summed in ascending d), base-2 logits (q carries scale * log2 e), weights
// normalised in fp32, output rounded once to the operand type: at least as precise as the tiled kernels, which round P to 16
// bits in front of the PV product; oracle.attention_counted restates it.  Vector arithmetic only (0.9 GFLOP per launch).
// Block = one sequence (gene) x a slab of its queries; thread = one (query, head) pair; the table lives in LDS as
// [c][d / 8][head][8] so that the 16-byte reads of the heads of a query fall on consecutive bank groups.
template <int DT, int DH>
__global__ __launch_bounds__(256) void attn_counted_keys_kernel(const unsigned short* __restrict__ q, int64_t q_stride,
                                                                const unsigned short* __restrict__ kv, int64_t kv_stride,
                                                                const float* __restrict__ log2_count, const int32_t* __restrict__ cu_q,
                                                                int C, int H, unsigned short* __restrict__ out, int64_t o_stride,
                                                                int q_per_block) {
    constexpr int NJ = DH / 8;
    extern __shared__ __attribute__((aligned(16))) char smem_ck[];
    u32x4_t* const sK = reinterpret_cast<u32x4_t*>(smem_ck);                    // [C][NJ][H] chunks of 8 values
    u32x4_t* const sV = sK + C * NJ * H;
    float* const sL = reinterpret_cast<float*>(sV + C * NJ * H);                // [C] log2 count of this sequence
    const int seq = blockIdx.y;
    const int tok0 = cu_q[seq], len = cu_q[seq + 1] - tok0;
    const int q0 = blockIdx.x * q_per_block;
    if (q0 >= len) return;
    const int D = H * DH;
    for (int i = threadIdx.x; i < 2 * C * NJ * H; i += 256) {
        const bool is_v = i >= C * NJ * H;
        const int ii = is_v ? i - C * NJ * H : i;
        const int h = ii % H, j = (ii / H) % NJ, c = ii / (H * NJ);
        (is_v ? sV : sK)[ii] = *reinterpret_cast<const u32x4_t*>(kv + (int64_t)c * kv_stride + (is_v ? D : 0) + h * DH + j * 8);
    }
    if (threadIdx.x < C) sL[threadIdx.x] = log2_count[(int64_t)seq * C + threadIdx.x];
    __syncthreads();
    const int q_end = (q0 + q_per_block < len) ? q0 + q_per_block : len;
    const int pairs = (q_end - q0) * H;
    auto unpack8 = [](u32x4_t v, float (&f)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = Op16<DT>::to_f32((unsigned short)(v[e] & 0xffffu));
            f[2 * e + 1] = Op16<DT>::to_f32((unsigned short)(v[e] >> 16));
        }
    };
    for (int p = threadIdx.x; p < pairs; p += 256) {
        const int t = q0 + p / H, h = p % H;
        const unsigned short* qp = q + (int64_t)(tok0 + t) * q_stride + h * DH;
        float qf[DH];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(qp + 8 * j), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[8 * j + e] = f[e];
        }
        float sc[16];
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float f[8];
                unpack8(sK[(c * NJ + j) * H + h], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) a = __builtin_fmaf(qf[8 * j + e], f[e], a);
            }
            a += sL[c];                                   // -inf for a label this sequence does not hold
            sc[c] = a;
            m = __builtin_fmaxf(m, a);
        }
        float o[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = 0.f;
        float l = 0.f;
        for (int c = 0; c < C; ++c) {
            const float pc = __builtin_amdgcn_exp2f(sc[c] - m);
            l += pc;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float f[8];
                unpack8(sV[(c * NJ + j) * H + h], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[8 * j + e] = __builtin_fmaf(pc, f[e], o[8 * j + e]);
            }
        }
        const float inv = 1.0f / l;
        unsigned short* op = out + (int64_t)(tok0 + t) * o_stride + h * DH;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u32x4_t pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = Op16<DT>::pack2(o[8 * j + 2 * e] * inv, o[8 * j + 2 * e + 1] * inv);
            *reinterpret_cast<u32x4_t*>(op + 8 * j) = pk;
        }
    }
}

template <int DT, int DH>
static int launch_counted_keys(const void* q, int64_t q_stride, const void* kv, int64_t kv_stride, const float* log2_count,
                               const int32_t* cu_q, int n_seq, int max_q, int C, int H, void* out, int64_t o_stride, hipStream_t st) {
    const int lds = 2 * C * (DH / 8) * H * 16 + C * 4;
    auto kern = attn_counted_keys_kernel<DT, DH>;
    static bool attr_set[VF_MAX_DEVICES] = {};
    const int dev = vf_current_device();
    if (dev < 0 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            vf_set_error("vf_attn_counted_keys: cannot reserve LDS");
            return VF_ERR_LAUNCH;
        }
        if (dev >= 0) attr_set[dev] = true;
    }
    // a block re-reads the whole table: give it enough queries to amortise that (>= 64 per block), but enough blocks to fill
    // the chip (>= 4 per CU when the batch allows)
    int qpb = 256;
    while (qpb > 64 && (long)n_seq * ((max_q + qpb - 1) / qpb) < 1024) qpb >>= 1;
    vf_note_kernel(1, "attn_counted_keys_kernel");
    hipLaunchKernelGGL(kern, dim3((max_q + qpb - 1) / qpb, n_seq), dim3(256), lds, st, (const unsigned short*)q, q_stride,
                       (const unsigned short*)kv, kv_stride, log2_count, cu_q, C, H, (unsigned short*)out, o_stride, qpb);
    VF_CHECK_LAUNCH("vf_attn_counted_keys");
    return VF_OK;
}

extern "C" int vf_attn_counted_keys(const void* q, int64_t q_stride, const void* kv_table, int64_t kv_stride,
                                    const float* log2_count, const int32_t* cu_seqlens_q, int n_seq, int max_seqlen_q, int C,
                                    int H, int dh, void* out, int64_t o_stride, int operand_dtype, void* stream) {
    VF_REQUIRE(q && kv_table && log2_count && cu_seqlens_q && out, "vf_attn_counted_keys: null pointer");
    VF_REQUIRE(operand_dtype == VF_BF16 || operand_dtype == VF_F16, "vf_attn_counted_keys: operand_dtype must be VF_BF16 or VF_F16");
    VF_REQUIRE(C >= 1 && C <= 16 && H >= 1 && H <= 256 && (dh == 32 || dh == 48 || dh == 64),
               "vf_attn_counted_keys: C=%d (1..16) H=%d dh=%d (32 / 48 / 64) not supported", C, H, dh);
    VF_REQUIRE(2 * C * (dh / 8) * H * 16 + C * 4 <= 160 * 1024, "vf_attn_counted_keys: the key / value table does not fit the LDS");
    VF_REQUIRE(q_stride % 8 == 0 && kv_stride % 8 == 0 && o_stride % 8 == 0 && ((uintptr_t)q % 16 == 0) &&
                   ((uintptr_t)kv_table % 16 == 0) && ((uintptr_t)out % 16 == 0),
               "vf_attn_counted_keys: rows must keep 16-byte alignment");
    if (n_seq <= 0 || max_seqlen_q <= 0) return VF_OK;
    VF_REQUIRE(n_seq <= 65535, "vf_attn_counted_keys: n_seq=%d exceeds the grid limit", n_seq);
    hipStream_t st = (hipStream_t)stream;
#define VF_CK(DT_, DH_) launch_counted_keys<DT_, DH_>(q, q_stride, kv_table, kv_stride, log2_count, cu_seqlens_q, n_seq, max_seqlen_q, C, H, out, o_stride, st)
    if (operand_dtype == VF_BF16) return dh == 32 ? VF_CK(VF_BF16, 32) : dh == 48 ? VF_CK(VF_BF16, 48) : VF_CK(VF_BF16, 64);
    return dh == 32 ? VF_CK(VF_F16, 32) : dh == 48 ? VF_CK(VF_F16, 48) : VF_CK(VF_F16, 64);
#undef VF_CK
}

// Softmax over a few distinct keys with their counts, on logits a GEMM produced (vf_gemm_ln consumer with fp32 output): the
// low-rank form of the CRE layers' context cross attention.  With C <= 16 distinct key / value rows per head the logits are
// LN(x) . (Wq_h^T k_c) -- a GEMM against an [H * Cp, D] matrix built once per weights -- and the output projection of
// sum_c w_c v_c is w . (Wo_h v_c) -- a GEMM with K = H * Cp; between them only this remains:
//   w[t, h, c] = 16-bit( exp2(s[t, h, c] + log2 n[seq(t), c] - m) / sum_c' exp2(...) ),  0 for the padding slots c >= C.
// scores fp32 [tokens, >= H * Cp] (head-major, Cp slots per head), out 16-bit [tokens, >= H * Cp].  One thread per (token, head).
template <int DT>
__global__ __launch_bounds__(256) void softmax_counted_kernel(const float* __restrict__ sc, int64_t lds, const float* __restrict__ log2_count,
                                                              const int32_t* __restrict__ cu_q, int H, int Cp, int C,
                                                              unsigned short* __restrict__ out, int64_t ldo) {
    const int seq = blockIdx.y;
    const int tok0 = cu_q[seq], len = cu_q[seq + 1] - tok0;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= len * H) return;
    const int t = p / H, h = p % H;
    const float* sp = sc + (int64_t)(tok0 + t) * lds + h * Cp;
    const float* lc = log2_count + (int64_t)seq * C;
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        v[c] = c < C ? sp[c] + lc[c] : -INFINITY;
        m = __builtin_fmaxf(m, v[c]);
    }
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        v[c] = __builtin_amdgcn_exp2f(v[c] - m);          // exp2(-inf) = 0: absent labels and padding slots
        l += v[c];
    }
    const float inv = 1.0f / l;
    unsigned short* op = out + (int64_t)(tok0 + t) * ldo + h * Cp;
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        if (c + 1 < Cp) *reinterpret_cast<unsigned int*>(op + c) = Op16<DT>::pack2(v[c] * inv, v[c + 1] * inv);
        else if (c < Cp) op[c] = (unsigned short)(Op16<DT>::pack2(v[c] * inv, 0.f) & 0xffffu);
    }
}

extern "C" int vf_softmax_counted(const float* scores, int64_t lds, const float* log2_count, const int32_t* cu_seqlens_q, int n_seq,
                                  int max_seqlen_q, int H, int Cp, int C, void* out, int64_t ldo, int out_dtype, void* stream) {
    VF_REQUIRE(scores && log2_count && cu_seqlens_q && out, "vf_softmax_counted: null pointer");
    VF_REQUIRE(out_dtype == VF_BF16 || out_dtype == VF_F16, "vf_softmax_counted: out_dtype must be VF_BF16 or VF_F16");
    VF_REQUIRE(C >= 1 && C <= Cp && Cp <= 16 && Cp % 2 == 0 && H >= 1, "vf_softmax_counted: C=%d Cp=%d (C <= Cp <= 16, Cp even) H=%d", C, Cp, H);
    VF_REQUIRE(lds >= (int64_t)H * Cp && ldo >= (int64_t)H * Cp && ldo % 2 == 0 && ((uintptr_t)out % 4 == 0),
               "vf_softmax_counted: rows hold H * Cp slots; the output keeps 4-byte alignment");
    if (n_seq <= 0 || max_seqlen_q <= 0) return VF_OK;
    VF_REQUIRE(n_seq <= 65535, "vf_softmax_counted: n_seq=%d exceeds the grid limit", n_seq);
    const dim3 grid((unsigned)(((long)max_seqlen_q * H + 255) / 256), n_seq);
    if (out_dtype == VF_BF16)
        hipLaunchKernelGGL(softmax_counted_kernel<VF_BF16>, grid, dim3(256), 0, (hipStream_t)stream, scores, lds, log2_count,
                           cu_seqlens_q, H, Cp, C, (unsigned short*)out, ldo);
    else
        hipLaunchKernelGGL(softmax_counted_kernel<VF_F16>, grid, dim3(256), 0, (hipStream_t)stream, scores, lds, log2_count,
                           cu_seqlens_q, H, Cp, C, (unsigned short*)out, ldo);
    VF_CHECK_LAUNCH("vf_softmax_counted");
    return VF_OK;
}

extern "C" int vf_attn_rows_supported(int dh, int alibi, int n_seq, int H, int max_seqlen_q, int max_seqlen_k, int flags) {
    return rows_supported(dh, alibi != 0, n_seq, H, max_seqlen_q, max_seqlen_k, (flags & VF_ATTN_Q_LOG2) != 0) ? 1 : 0;
}

extern "C" int vf_attn_varlen_fwd_rows(const void* q, const void* k, const void* v, void* out, int64_t q_stride,
                                       int64_t k_stride, int64_t v_stride, int64_t o_stride, const int32_t* cu_seqlens_q,
                                       const int32_t* cu_seqlens_k, int n_seq, int max_seqlen_q, int max_seqlen_k, int H,
                                       int dh, const float* alibi_slopes, float scale, int operand_dtype, int flags,
                                       const int64_t* q_rows, const int64_t* kv_rows, void* stream) {
    VF_REQUIRE(operand_dtype == VF_BF16 || operand_dtype == VF_F16, "vf_attn_varlen_fwd_rows: operand_dtype must be VF_BF16 or VF_F16");
    if (operand_dtype == VF_BF16)
        return attn_dispatch<VF_BF16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                                      max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, flags, stream, q_rows, kv_rows);
    return attn_dispatch<VF_F16>(q, k, v, out, q_stride, k_stride, v_stride, o_stride, cu_seqlens_q, cu_seqlens_k, n_seq,
                                 max_seqlen_q, max_seqlen_k, H, dh, alibi_slopes, scale, flags, stream, q_rows, kv_rows);
}
