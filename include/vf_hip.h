/* vf_hip.h -- C ABI of libvf_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * VariantFormer inference hot path.
 *
 * The reference (czi-ai/variantformer) has no FFI of its own: its only native code on this
 * path is the third-party flash-attn wheel plus ATen/cuBLAS behind nn.Linear / nn.LayerNorm.
 * Each entry point below names the reference call site it replaces (paths relative to the
 * reference checkout; [3p] = third-party flash-attn 2.8.3 API used at that site).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (incl. outputs and workspace);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing syncs;
 *   - bf16 tensors are raw uint16 storage (torch.bfloat16 compatible), row-major;
 *   - return value 0 = ok, otherwise a VF_ERR_* code; vf_last_error() gives the message for
 *     the calling thread; no global mutable state besides that message; re-entrant per stream;
 *   - "ld*" / "*_stride" arguments are row strides in ELEMENTS.
 */
#ifndef VF_HIP_H
#define VF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_ABI_VERSION 12

enum vf_status {
    VF_OK = 0,
    VF_ERR_INVALID_ARG = 1,   /* shape / alignment / enum not supported */
    VF_ERR_LAUNCH = 2         /* hipGetLastError() after the launch was not hipSuccess */
};

enum vf_dtype { VF_F32 = 0, VF_BF16 = 1, VF_F16 = 2 };   /* VF_F16 = IEEE half (torch.float16) */

/* Epilogues of vf_gemm_bf16 (acc = fp32 accumulator of A @ W^T, + bias[n] always) */
enum vf_epilogue {
    VF_EPI_BF16 = 0,        /* out bf16 [M,N]                      (Wqkv / Wq / Wkv projections)      */
    VF_EPI_F32 = 1,         /* out fp32 [M,N]                      (cre_map, gene_map, head layer 0)   */
    VF_EPI_RES_F32 = 2,     /* out fp32 [M,N] = acc + residual     (out_proj + res, geglu_2 + res)     */
    VF_EPI_GEGLU_BF16 = 3,  /* out bf16 [M,N/2] = a * gelu(gate); W rows interleaved in blocks of 16:
                               W'[32b+t] = W[16b+t], W'[32b+16+t] = W[N/2+16b+t], bias likewise
                               (vf_pack_geglu_rows does the permutation)                               */
    VF_EPI_GELU_F32 = 4,    /* out fp32 [M,N] = gelu(acc)          (head layer 4)                      */
    VF_EPI_GELU_BF16 = 5    /* out bf16 [M,N] = gelu(acc)                                              */
};

int vf_version(void);
const char* vf_last_error(void);
/* Diagnostic (ABI 12): the kernel the calling thread's most recent GEMM (which = 0) or attention (which = 1) entry dispatched
 * to -- a static string such as "gemm8x_kernel", "gemm_mfma_kernel<128x128>", "attn_short2_kernel<dh64,1 pass,rows>"; "" before
 * the first launch.  No reference counterpart (the reference reaches its kernels through ATen / flash-attn [3p] dispatchers);
 * exists so that a sweep over tokenizer geometries (scripts/s2r_dims_sweep.py: the real checkpoint's seq2reg width / heads /
 * depth are unknown offline, processors/model_manager.py:44-51) can record which kernel served each shape. */
const char* vf_last_kernel(int which);

/* out = epilogue(A[M,K] @ W[N,K]^T + bias[N]).   A, W bf16; bias fp32 (may be NULL = 0).
 * Replaces every nn.Linear on the path: flash_attn MHA Wqkv/Wq/Wkv/out_proj [3p]
 * (seq2reg/modules.py:140-142, seq2gene/modules/layers.py:344-351), linear_geglu_1/2
 * (layers.py:78-80,159-162; seq2reg/modules.py:145-147,184-187), cre_map/gene_map
 * (seq2gene/model_combined_modulator.py:502-507,610-612), TissueExpressionHeads Linear layers
 * (layers.py:1078-1087).  Requires K % 8 == 0, N % 8 == 0 (N % 32 == 0 for GEGLU); fast MFMA
 * path when K % 64 == 0. */
int vf_gemm_bf16(const void* A, int64_t lda, const void* W, const float* bias,
                 const float* residual, int64_t ldr, void* out, int64_t ldo,
                 int M, int N, int K, int epilogue, void* stream);

/* Same as vf_gemm_bf16 with an explicit tile configuration (tuning and tests): variant 0 = automatic
 * choice (what vf_gemm_bf16 does); 1 = 128x128, 5 = 64x64, 20 = two-group 256x256 tile, 22 = its persistent form (one
 * block per CU walks the tiles, K >= 128; see vf_gemm.hip).  All configurations give bit-identical results. */
int vf_gemm_bf16_ex(const void* A, int64_t lda, const void* W, const float* bias,
                    const float* residual, int64_t ldr, void* out, int64_t ldo,
                    int M, int N, int K, int epilogue, int variant, void* stream);

/* fp16-operand twins (IEEE half in / out where the bf16 entry has bf16, fp32 accumulation, same tiles, same MFMA rate):
 * the reference's `16-mixed` / fp16 flash-attn path (seq2gene/modules/layers.py:102-125 casts the MHA modules and
 * their inputs to fp16 when precision is fp32; utils/functions.py:12-32 maps "16-mixed" to fp16 autocast;
 * BASELINE.json configs[4] "fp16 with fp32 accumulate"). */
int vf_gemm_f16(const void* A, int64_t lda, const void* W, const float* bias,
                const float* residual, int64_t ldr, void* out, int64_t ldo,
                int M, int N, int K, int epilogue, void* stream);
int vf_gemm_f16_ex(const void* A, int64_t lda, const void* W, const float* bias,
                   const float* residual, int64_t ldr, void* out, int64_t ldo,
                   int M, int N, int K, int epilogue, int variant, void* stream);

/* LayerNorm without a pass of its own: LN(x) . W^T = rstd * (x . (gamma (.) W)^T - mean * rowsum(gamma (.) W))
 * + (W . beta + b).  The GEMM that PRODUCES the fp32 stream x (epilogue VF_EPI_F32 / VF_EPI_RES_F32) also writes
 * out16 = the bf16 copy of x [M, ld16] and part_stats [ceil(N/32), M, 2] = (sum, second moment about the part's own mean) of every
 * 32-column part of the row (give out16 + part_stats, leave row_stats / colsum NULL); vf_ln_finalize turns the parts into row_stats
 * [M, 2] = (mean, rstd); the GEMM that CONSUMES LN(x) (epilogue VF_EPI_BF16 / VF_EPI_GEGLU_BF16) takes A = out16,
 * W = bf16(gamma (.) W), bias = W . beta + b, colsum[n] = sum_k W'[n][k] and row_stats (give those two, leave out16 /
 * part_stats NULL).  A producer whose fp32 result nobody reads (only out16 and the statistics are consumed: the layers'
 * intermediate streams) may pass out = NULL: nothing is stored there.
 * vf_row_stats_cast produces (out16, row_stats) for a stream no GEMM produced.  Replaces the
 * nn.LayerNorm -> nn.Linear pairs of the layers (seq2gene/modules/layers.py:105-162, seq2reg/modules.py:155-187).
 * MFMA path only: K % 64 == 0; a producer needs N % 32 == 0. */
int vf_gemm_ln_bf16(const void* A, int64_t lda, const void* W, const float* bias, const float* residual, int64_t ldr,
                    void* out, int64_t ldo, int M, int N, int K, int epilogue, const float* row_stats,
                    const float* colsum, void* out16, int64_t ld16, float* part_stats, void* stream);
/* General form (ABI 3): operand_dtype VF_BF16 or VF_F16 (reference precision "16-mixed": the LayerNorm fold for fp16
 * operands).  An fp16 stream is stored SCALED, out16 = fp16(x * x16_scale) with a power-of-two x16_scale, so that the raw
 * residual stream cannot leave the fp16 range (LayerNorm is scale-invariant; vf_ln_finalize2 writes the pair the consumer
 * needs for the scaled operand: (mean * c, rstd / c)); bf16 streams use x16_scale = 1.  residual_dtype VF_F32: fp32
 * residual rows as above; residual_dtype == operand_dtype: the residual is read from the 16-BIT COPY of a stream
 * (value = float(res16) * res16_scale) -- the stream after a layer's self-attention block is otherwise only read through
 * LayerNorm -> Linear, so its fp32 rows need not exist (the reference's own autocast keeps that stream in 16 bits,
 * seq2gene/modules/layers.py:128-140).  vf_gemm_ln_bf16 = this with VF_BF16, VF_F32 residual, scales 1. */
int vf_gemm_ln(const void* A, int64_t lda, const void* W, const float* bias, const void* residual, int64_t ldr,
               int residual_dtype, void* out, int64_t ldo, int M, int N, int K, int epilogue, int operand_dtype,
               const float* row_stats, const float* colsum, void* out16, int64_t ld16, float* part_stats,
               float x16_scale, float res16_scale, void* stream);
/* ABI 5 -- the layer trunk as a scaled fp16 copy.  A layer's output x_out = linear_geglu_2(h) + x_in
 * (seq2gene/modules/layers.py:161-165, seq2reg/modules.py:186-190) travels to the next layer of its stack as
 * t16 = fp16(x * t16_scale) [M, ldt16] -- 11 significant bits whatever the operand type (bf16 has 8), so the per-layer
 * rounding of the trunk stays below the operand roundings -- instead of fp32 rows: this producer (epilogue VF_EPI_RES_F32
 * with the LayerNorm-producer outputs out16 / part_stats as above) reads its residual as float(residual_f16) * res_scale
 * and writes t16_out (NULL: not written -- the last layer of a stack, which passes `out` for its fp32 rows instead;
 * out may be NULL otherwise).  6 instead of 10 bytes per element through the down-projection epilogue. */
int vf_gemm_ln_t16(const void* A, int64_t lda, const void* W, const float* bias, const void* residual_f16, int64_t ldr,
                   float res_scale, void* out, int64_t ldo, int M, int N, int K, int operand_dtype, void* out16,
                   int64_t ld16, float* part_stats, float x16_scale, void* t16_out, int64_t ldt16, float t16_scale,
                   void* stream);
int vf_ln_finalize(const float* part_stats, int64_t rows, int n_parts, int D, float eps, float* row_stats, void* stream);
/* vf_ln_finalize consumes the ABI >= 5 part statistics only: per 32-column part (sum, second moment ABOUT THE PART'S MEAN),
 * D == 32 * n_parts (checked).
 * ..2 forms: x16_scale as above; alert (optional, device int, OR-ed): bit 0 when some row has |mean| > ratio_limit standard
 * deviations -- the regime where rounding the UNCENTRED row to 16 bits costs the folded form accuracy; bit 1 (ABI 6) when
 * some row may hold an element of magnitude >= abs_limit (|mean| + sqrt(D * var) >= abs_limit; 0 = no check) -- its scaled
 * fp16 copies (fp16 operand copy, fp16 trunk copy of vf_gemm_ln_t16) could overflow.  Callers read the flag back with their
 * outputs and recompute the batch with the separate LayerNorm (variantformer_amd.ops.ln_fold_alert; the reference's plain
 * nn.LayerNorm has no such regime, seq2gene/modules/layers.py:75-77,99-163). */
int vf_ln_finalize2(const float* part_stats, int64_t rows, int n_parts, int D, float eps, float x16_scale, float ratio_limit,
                    float abs_limit, int* alert, float* row_stats, void* stream);
int vf_row_stats_cast2(const float* x, int64_t rows, int D, float eps, void* out16, int out_dtype, float x16_scale,
                       float ratio_limit, float abs_limit, int* alert, float* row_stats, void* stream);
int vf_row_stats_cast(const float* x, int64_t rows, int D, float eps, void* out16, int out_dtype, float* row_stats,
                      void* stream);

/* Permute rows of a [2F, K] 16-bit (bf16 or fp16) weight (and its fp32 bias, may be NULL) into the VF_EPI_GEGLU_BF16
 * layout (one-time weight repack at checkpoint load). */
int vf_pack_geglu_rows(const void* W, const float* bias, void* W_out, float* bias_out,
                       int two_f, int K, void* stream);

/* Variable-length multi-head attention forward, non-causal:
 *   out[t, h, :] = softmax_j( scale * q[t,h,:].k[j,h,:] - slope[h] * |i + (sk - sq) - j| ) v[j,h,:]
 * over the keys j of t's sequence; i, j are positions inside the sequence.
 * Replaces flash_attn_varlen_{qkvpacked,kvpacked}_func inside flash_attn.modules.mha.MHA [3p]
 * (call sites seq2reg/modules.py:167; seq2gene/modules/layers.py:437-439,465,482).
 * q/k/v point at the first head of token 0 (so a packed [tokens,3,H,dh] buffer is passed as
 * q=base, k=base+H*dh, v=base+2*H*dh with row stride 3*H*dh).  bf16 in/out, fp32 softmax and
 * accumulation.  dh in {32, 48, 64}.  cu_seqlens_*: int32 [n_seq+1] device arrays.
 * alibi_slopes: fp32 [H] device array or NULL.  Sequences with 0 queries are skipped; the rows of
 * queries whose key sequence is empty are written as zeros (flash-attn's convention). */
int vf_attn_varlen_fwd(const void* q, const void* k, const void* v, void* out,
                       int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                       const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                       int n_seq, int max_seqlen_q, int max_seqlen_k,
                       int H, int dh, const float* alibi_slopes, float scale, void* stream);

/* Same kernel with the ALiBi query positions aligned to the START of the key sequence (query i of a sequence sits at
 * position i, bias -slope*|i - j|) instead of flash-attn's end alignment.  Used for the last gene layer, where only
 * the registry token (position 0 of each sequence) is still needed as a query (pool_outputs keeps row 0 only,
 * seq2gene/model_combined_modulator.py:391-392) while every token still serves as key / value. */
int vf_attn_varlen_fwd_qstart(const void* q, const void* k, const void* v, void* out,
                              int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                              const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                              int n_seq, int max_seqlen_q, int max_seqlen_k,
                              int H, int dh, const float* alibi_slopes, float scale, void* stream);

/* fp16 twins of the two attention entries (Q, K, V, P and the output in IEEE half; fp32 scores / softmax). */
int vf_attn_varlen_fwd_f16(const void* q, const void* k, const void* v, void* out,
                           int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                           int n_seq, int max_seqlen_q, int max_seqlen_k,
                           int H, int dh, const float* alibi_slopes, float scale, void* stream);
int vf_attn_varlen_fwd_qstart_f16(const void* q, const void* k, const void* v, void* out,
                                  int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                  const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                  int n_seq, int max_seqlen_q, int max_seqlen_k,
                                  int H, int dh, const float* alibi_slopes, float scale, void* stream);

/* General form (ABI 4): operand_dtype VF_BF16 / VF_F16 and a flag word.
 *   VF_ATTN_Q_AT_START  the _qstart alignment of the ALiBi query positions;
 *   VF_ATTN_Q_LOG2      Q was projected with weights (and bias) pre-multiplied by scale * log2(e) -- the caller folds the
 *                       softmax scale of flash-attn's MHA (softmax_scale = 1 / sqrt(dh), flash_attn.modules.mha [3p], as
 *                       the reference constructs it, seq2gene/modules/layers.py:344-351) into the Wq rows at weight-load
 *                       time, so q . k already is the base-2 logit and `scale` is ignored.  The long-stream dh = 48
 *                       kernel then runs its softmax without a running maximum (p = exp2(q . k), unnormalised sums,
 *                       one division at the end; softmax is invariant under the offset) and recomputes, inside the same
 *                       launch, any block whose denominator left the range [2^-100, 2^100] with the running-maximum
 *                       form: the result never depends on the magnitude of the logits, only the speed of such a block. */
enum vf_attn_flags { VF_ATTN_Q_AT_START = 1, VF_ATTN_Q_LOG2 = 2 };
int vf_attn_varlen_fwd_v2(const void* q, const void* k, const void* v, void* out,
                          int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                          const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                          int n_seq, int max_seqlen_q, int max_seqlen_k,
                          int H, int dh, const float* alibi_slopes, float scale,
                          int operand_dtype, int flags, void* stream);

/* Row-map form (ABI 8): q / k / v are TABLES of distinct rows and the gather happens in the kernel's loads -- token t's query
 * is row q_rows[t] of q, its key / value rows are row kv_rows[t] of k / v (int64 maps over the packed tokens; either may be
 * null = row t itself); the output row of token t is row t of out.  Serves the first layers' projection-by-lookup: the
 * encoder's first LayerNorm1 -> Wqkv depends on (token id, position) alone (seq2reg/model.py:215-220,
 * seq2reg/modules.py:152-160), the first gene layer's on the distinct chunk / registry rows
 * (seq2gene/model_combined_modulator.py:622-649 repeats them per tissue), so the [tokens, 3 D] projection a row gather
 * would materialise is never written or re-read.  Same arithmetic per query as vf_attn_varlen_fwd_v2 on the gathered rows
 * (bit-identical).  Only the geometries vf_attn_rows_supported() reports have a row-map kernel (the one-block-per-sequence
 * kernel: dh 64 without bias / dh 48 with ALiBi, sequences <= 256 tokens, VF_ATTN_Q_LOG2 set); any other geometry is
 * rejected with VF_ERR_INVALID_ARG -- gather the rows (vf_gather_rows_bf16) and call the plain entry there. */
int vf_attn_rows_supported(int dh, int alibi, int n_seq, int H, int max_seqlen_q, int max_seqlen_k, int flags);
int vf_attn_varlen_fwd_rows(const void* q, const void* k, const void* v, void* out,
                            int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                            const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                            int n_seq, int max_seqlen_q, int max_seqlen_k,
                            int H, int dh, const float* alibi_slopes, float scale,
                            int operand_dtype, int flags, const int64_t* q_rows, const int64_t* kv_rows, void* stream);

/* Cross attention against keys that are copies of a FEW DISTINCT ROWS (ABI 9): the CRE layers' context cross attention
 * (seq2gene/model_combined_modulator.py:168 `second_level_context_embedding(ref_labels)`, seq2gene/modules/layers.py:421-439:
 * K / V = Wkv of an Embedding(9) row per CRE), where softmax over the repeated keys equals softmax over the distinct ones with
 * log(count) added to the logit.  q [tokens, >= H*dh] 16-bit, pre-scaled by softmax_scale * log2 e (VF_ATTN_Q_LOG2 form);
 * kv_table [C, >= 2*H*dh] 16-bit = (K | V) of the C <= 16 distinct rows, heads packed (head, dh); log2_count fp32 [n_seq, C] =
 * log2 of how often row c occurs among sequence s's keys (-inf: not at all; every sequence holds >= 1 key);
 * out [tokens, >= H*dh] 16-bit.  fp32 scores, fp32-normalised weights, one rounding of the output. */
int vf_attn_counted_keys(const void* q, int64_t q_stride, const void* kv_table, int64_t kv_stride, const float* log2_count,
                         const int32_t* cu_seqlens_q, int n_seq, int max_seqlen_q, int C, int H, int dh,
                         void* out, int64_t o_stride, int operand_dtype, void* stream);

/* The same attention in LOW-RANK form (ABI 10): with C distinct key / value rows per head the logits are
 * LN(x) . (Wq_h^T k_c) -- vf_gemm_ln as a CONSUMER with fp32 output (epilogue VF_EPI_F32, new in ABI 10) against an
 * [H * Cp, D] matrix built once per weights -- and out_proj(sum_c w_c v_c) is w . (Wo_h v_c), a GEMM with K = H * Cp (Cp >= C,
 * even, H * Cp % 64 == 0 for the GEMM; padding slots are zero).  Between the two GEMMs:
 *   out[t, h * Cp + c] = 16-bit( softmax_c( scores[t, h * Cp + c] + log2_count[seq(t), c] ) ),  0 for c >= C.
 * scores fp32 [tokens, lds >= H * Cp] base-2 logits; out 16-bit [tokens, ldo >= H * Cp].  Replaces, for the CRE layers, Wq + the
 * flash-attn cross forward + the K = D half of out_proj (seq2gene/modules/layers.py:421-439, 156-158). */
int vf_softmax_counted(const float* scores, int64_t lds, const float* log2_count, const int32_t* cu_seqlens_q, int n_seq,
                       int max_seqlen_q, int H, int Cp, int C, void* out, int64_t ldo, int out_dtype, void* stream);

/* y = LayerNorm(x) * gamma + beta over the last dim (eps inside sqrt, biased variance), optional
 * exact-erf GELU, output fp32, bf16 or fp16 (out_dtype = vf_dtype).  x fp32 [rows, D], D % 4 == 0, D <= 8192.
 * Replaces nn.LayerNorm (seq2reg/modules.py:143-144, layers.py:75-77, head layers.py:1080-1081). */
int vf_layernorm(const float* x, const float* gamma, const float* beta, void* out,
                 int64_t rows, int D, float eps, int out_dtype, int gelu, void* stream);

/* Token embedding + optional positional table on the PACKED valid tokens of W windows:
 *   row r = cu[w] + (rank of position p among the valid positions of window w)
 *   out[r, :] = table[ids[w, p], :] + (pos_table ? pos_table[p, :] : 0)
 * ids int64 [W, L]; pad u8/bool [W, L] (non-zero = pad, reference convention
 * datasets/vcfdataset.py:209-213); cu int32 [W+1] = exclusive prefix sum of valid counts.
 * Replaces nn.Embedding + PE add + unpad_input [3p] (seq2reg/model.py:215-220,
 * seq2reg/modules.py:159-161). */
int vf_embed_pack(const int64_t* ids, const uint8_t* pad, const int32_t* cu,
                  const float* table, const float* pos_table, float* out,
                  int W, int L, int d, int vocab, void* stream);
/* The same rows as a LayerNorm-fold stream (ABI 5): out16 = 16-bit(x * x16_scale) [n_tokens, d] in out_dtype, row_stats
 * [n_tokens, 2] = (mean * x16_scale, rstd / x16_scale) exactly as vf_row_stats_cast2 would compute them from vf_embed_pack's
 * rows, t16 (optional) = fp16(x * t16_scale) the trunk copy of vf_gemm_ln_t16, out (optional) the fp32 rows.  Fuses
 * token_embedding + position_encoding (seq2reg/model.py:203-221) with the first layer's LayerNorm statistics: the fp32
 * rows of the encoder input never reach HBM.  d <= 2048. */
int vf_embed_stream(const int64_t* ids, const uint8_t* pad, const int32_t* cu, const float* table, const float* pos_table,
                    float* out, void* out16, int out_dtype, float x16_scale, void* t16, float t16_scale, float* row_stats,
                    float eps, float ratio_limit, float abs_limit, int* alert, int W, int L, int d, int vocab, void* stream);

/* keys[cu[w] + k] = id * key_L + position of the k-th valid token of window w (key_L = L, or 1: keys = ids), ids clamped to
 * [0, vocab) as vf_embed_pack does.  The encoder input row nn.Embedding(id) + positional(position) (seq2reg/model.py:205-221) takes
 * at most vocab * L distinct values: what the first layer computes per row from it alone (norm1 -> Wqkv, seq2reg/modules.py:
 * 152-160) is computed once per distinct row and looked up by this key (vf_gather_rows_bf16). */
int vf_token_keys(const int64_t* ids, const uint8_t* pad, const int32_t* cu, int64_t* keys, int W, int L, int vocab, int key_L,
                  void* stream);

/* Per-window valid-token count and exclusive prefix sum: cu[0]=0, cu[w+1]=cu[w]+#valid(w).
 * Replaces the cu_seqlens half of unpad_input [3p].  Single-block scan; W <= 2^24. */
int vf_mask_to_cu_seqlens(const uint8_t* pad, int32_t* cu, int W, int L, void* stream);

/* Masked mean over each window's packed tokens: out[w,:] = mean(x[cu[w]:cu[w+1], :]) (NaN if the
 * window is empty, as the reference's 0/0).  x fp32 [n_tok, d]; out fp32, bf16 or fp16 [W, d].
 * Replaces seq2reg/model.py:263-267. */
int vf_segment_mean(const float* x, const int32_t* cu, void* out, int W, int d, int out_dtype, void* stream);

/* The same masked mean over a 16-BIT stream x [n_tok, d] (row stride ldx elements; dtype VF_BF16 / VF_F16), values times
 * in_scale: out_f32 fp32 [W, d] and / or out_split 16-bit [W, 2 d] = [hi | lo], hi = rn16(mean), lo = rn16(mean - hi), in the
 * type of x (either may be NULL).  d a multiple of 8, at most 2048.  Used by the encoder's last layer: the mean pool
 * (seq2reg/model.py:263-267) commutes with the last Linear of the layer (seq2reg/modules.py:184-188), so
 * mean(src + W2 h + b) = mean(src) + W2 mean(h) + b is evaluated on W pooled rows instead of n_tok token rows. */
int vf_segment_mean16(const void* x, int64_t ldx, int dtype, const int32_t* cu, float in_scale, float* out_f32,
                      void* out_split, int W, int d, void* stream);

/* Row gather from two fp32 sources: out[i,:] = idx[i] >= 0 ? a[idx[i],:] : b[-idx[i]-1,:].
 * out fp32, bf16 or fp16 [n, d].  Replaces MultiRegistry/prepare_input concat (layers.py:508-521,
 * model_combined_modulator.py:357-366), nn.Embedding lookups (:166-168), pool_outputs row 0
 * (:391-392) and pad_input/unpad_input row moves [3p]. */
int vf_gather_rows_f32(const float* a, const float* b, const int64_t* idx, void* out,
                       int64_t n, int d, int out_dtype, void* stream);

/* 16-bit (bf16 or fp16) row gather: out[i,:] = src[idx[i],:], ld in elements. */
int vf_gather_rows_bf16(const void* src, int64_t ld_src, const int64_t* idx, void* out, int64_t ld_out,
                        int64_t n, int d, void* stream);

/* out[i] = softplus(dot(x[i,:], w) + b), softplus threshold 20 (nn.Softplus defaults), or the plain
 * affine value when softplus == 0.  x fp32 [n, d].  Replaces the last Linear(D,1) + Softplus of
 * TissueExpressionHeads (layers.py:1085-1086). */
int vf_rowdot_softplus(const float* x, const float* w, const float* b, float* out,
                       int64_t n, int d, int softplus, void* stream);

/* Max pool over the rows [cu[w], cu[w+1]) of x: out fp32 [W, d] (pool_outputs "max", model_combined_modulator.py:380-389).
 * Row sum with optional row indirection: out[i] = a[idx_a ? idx_a[i] : i] + b[idx_b ? idx_b[i] : i], fp32 [n, d]
 * (gene-stream residual `use_res`, model_combined_modulator.py:253-254,284-285; AddContext, layers.py:558-573). */
int vf_segment_max(const float* x, const int32_t* cu_seqlens, float* out, int W, int d, void* stream);
int vf_add_rows_f32(const float* a, const int64_t* idx_a, const float* b, const int64_t* idx_b, float* out, int64_t n,
                    int d, void* stream);

/* "linear" sequence pooling of seq2reg (seq2reg/model.py:183-184,268-272: nn.Linear(token_length, 1) over the token axis
 * of the zero-masked window): out[w,:] = sum over the valid positions p of window w of lin_w[p] * x[row(w,p),:] + lin_b[0],
 * on the packed stream (row(w,p) = cu[w] + rank of p among the valid positions).  pad u8 [W, L]; out fp32 / bf16 / fp16. */
int vf_segment_linear(const float* x, const int32_t* cu_seqlens, const uint8_t* pad, const float* lin_w,
                      const float* lin_b, void* out, int W, int L, int d, int out_dtype, void* stream);

/* out[i,:] = src[idx[i],:] * (scale ? scale[i] : 1) + (shift ? shift[i] : 0), fp32 [n, d]: the per-token context rows of a
 * use_context tokenizer (seq2reg/model.py:222-245: context_embedding(label) repeated over the tokens, or expanded per
 * position by expand_context = nn.Linear(1, token_length)). */
int vf_affine_rows_f32(const float* src, const int64_t* idx, const float* scale, const float* shift, float* out,
                       int64_t n, int d, void* stream);

/* fp32 -> bf16 / fp16 (round to nearest even), n elements. */
int vf_cast_f32_bf16(const float* x, void* out, int64_t n, void* stream);
int vf_cast_f32_f16(const float* x, void* out, int64_t n, void* stream);

/* ---- host-side (CPU) byte-pair encoder: SURVEY.md section 8f row 1 ------------------------------------------
 * Replaces the HuggingFace `tokenizers` BPE model (Rust, third party) as used by utils/seq.py:BPEEncoder.encode
 * (:52-62) with vocabs/bpe_vocabulary_500.json.  char_ids: int32[256], vocab id of each (upper-cased) byte or -1
 * for bytes that are not symbols (they split the sequence, e.g. 'N'); merges: int32[n_merges][3] = (left id,
 * right id, merged id) in rank order.  vf_bpe_encode upper-cases, encodes every maximal run of valid bytes as one
 * word and concatenates; returns the number of tokens (call with capacity 0 to size the buffers), writes at most
 * `capacity` ids and the raw-sequence start offset of each token (either output may be NULL); -1 on bad args. */
void* vf_bpe_create(const int32_t* char_ids, int n_ids, const int32_t* merges, int n_merges);
void vf_bpe_destroy(void* bpe);
int64_t vf_bpe_encode(const void* bpe, const char* seq, int64_t len, int32_t* ids_out, int64_t* starts_out,
                      int64_t capacity);
/* ABI 12: the first max_tokens tokens of vf_bpe_encode(seq, len), EXACTLY, without encoding the rest of a long word: the sample
 * builder keeps max_chunks x max_length tokens of a gene body (datasets/vcfdataset.py:338-394; utils/seq.py:52-62 encodes all of
 * it).  A cut text changes the tokens of at most n_merges x (longest token) characters in front of the cut (a difference moves
 * left by one symbol per merge rank), so a prefix with that margin is encoded and the tokens in front of the margin are kept. */
int64_t vf_bpe_encode_prefix(const void* bpe, const char* seq, int64_t len, int64_t max_tokens, int32_t* ids_out,
                             int64_t* starts_out, int64_t capacity);

/* ---- host-side (CPU) VCF reader + per-region IUPAC consensus: SURVEY.md section 8f row 1 ----------------------
 * Replaces the `samtools faidx | bcftools consensus -H I -e <filter> sample.vcf.gz` subprocess pair the reference
 * starts per CRE window and per gene body (utils/data_process.py:17-101 apply_bcftools_consensus, :367-467
 * apply_bcftools_consensus_to_gene).  PARITY UNPINNED: bcftools is third party, absent from the reference tree and
 * from this image; semantics are restated in variantformer_amd/csrc/vf_vcf.cpp.
 * vf_vcf_open reads a plain / gzip / bgzip VCF once (genotypes of `sample`, NULL or "" = first sample column) and
 * returns a handle (NULL on I/O error or unknown sample).  vf_vcf_consensus writes the consensus of the 0-based
 * interval [start0, start0 + ref_len) of `chrom`, whose reference bases are `ref`, into `out` and returns its
 * length, or a negative VF_CONS_* code.  snp_only != 0 is the reference's "SNP" filter (records with a snp ALT only);
 * indel_policy for records that are not single-base substitutions: 2 = bcftools' `-H I` rule (IUPAC codes for
 * equal-length alleles, the first non-REF genotype allele otherwise: a het indel applies its ALT), 1 = the first
 * genotype allele, 0 = refuse the region (VF_CONS_INDEL). */
enum { VF_CONS_BAD_ARG = -1, VF_CONS_REF_MISMATCH = -2, VF_CONS_INDEL = -3, VF_CONS_BAD_GT = -4 };
void* vf_vcf_open(const char* path, const char* sample);
void vf_vcf_close(void* vcf);
int64_t vf_vcf_num_records(const void* vcf, const char* chrom);
int64_t vf_vcf_consensus(const void* vcf, const char* chrom, int64_t start0, const char* ref, int64_t ref_len,
                         int snp_only, int indel_policy, char* out, int64_t out_cap, int64_t* n_applied);

/* ---- host-side (CPU) batched sample builder: all cCRE windows of a gene in one call -----------------------------
 * Replaces the per-window loop of the reference's sample builder (datasets/vcfdataset.py:219-283; one subprocess pair
 * and one tokenizer call per window).  span_ref holds the reference bases of [span_start0, span_start0 + span_len) of
 * `chrom`; window i = [starts0[i], ends0[i]) inside it: IUPAC consensus (vf_vcf_consensus; vcf NULL = reference only)
 * -> reverse complement when revcomp != 0 -> vf_bpe_encode -> the first L token ids into ids_out[i*L ..] (pad_id
 * beyond) and mask_out (1 = pad, the reference's convention).  status[i] (may be NULL): 0 consensus applied, 1 fell
 * back to the reference bases (REF mismatch), VF_CONS_INDEL window refused under indel_policy 0.  Returns n or -1. */
int64_t vf_build_windows(const void* vcf, const void* bpe, const char* chrom, int64_t span_start0, const char* span_ref,
                         int64_t span_len, int64_t n, const int64_t* starts0, const int64_t* ends0, int snp_only,
                         int indel_policy, int revcomp, int L, int64_t pad_id, int64_t* ids_out, uint8_t* mask_out,
                         int32_t* status);

/* Host staging of a batch's token ids (no GPU): src int64 [rows] rows of L ids, src_row_stride elements apart -> dst int32
 * [rows, L], clamped to [-1, INT32_MAX] (vf_embed_* clamp ids to [0, vocab): the same token either way; narrowing never
 * wraps).  Returns a bit mask -- 1: some id < 0, 2: some id >= 2^30 -- or -1 for bad arguments.  The batch dict's id tensors
 * (datasets/vcfdataset.py:18-63 `cre_sequences`, `gene_embeddings`: int64 [n, 1, L]) go through it once per batch. */
int vf_narrow_ids(const int64_t* src, int64_t src_row_stride, int32_t* dst, int64_t rows, int64_t L);

#ifdef __cplusplus
}
#endif
#endif /* VF_HIP_H */
