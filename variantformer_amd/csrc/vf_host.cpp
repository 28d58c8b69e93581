// vf_host: batched host-side (CPU) sample building -- all cCRE windows of a gene in ONE call.
//
// Replaces the per-window Python loop of the sample builder (reference datasets/vcfdataset.py:219-283: one
// `samtools | bcftools consensus` subprocess pair + one tokenizer call per cCRE window, ~1000 windows per gene) with:
// slice the window out of one reference span -> IUPAC consensus (vf_vcf_consensus) -> optional reverse complement
// (minus-strand genes, utils/seq.py reverse_complement) -> BPE (vf_bpe_encode) -> truncate / pad to L token ids and
// the pad mask, written straight into the [n, L] arrays the collate function stacks.  Token ids are bit-exact with the
// per-window path (tests/test_consensus_cpu.py).
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/vf_hip.h"

namespace {
// complement of an IUPAC code, case preserved (same table as variantformer_amd/utils/seq.py reverse_complement)
char comp(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'S'; case 'W': return 'W';
        case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B';
        case 'D': return 'H'; case 'H': return 'D'; case 'N': return 'N';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        case 'r': return 'y'; case 'y': return 'r'; case 's': return 's'; case 'w': return 'w';
        case 'k': return 'm'; case 'm': return 'k'; case 'b': return 'v'; case 'v': return 'b';
        case 'd': return 'h'; case 'h': return 'd'; case 'n': return 'n';
        default: return c;
    }
}
}  // namespace

// span_ref = reference bases of [span_start0, span_start0 + span_len) on `chrom`; window i = [starts0[i], ends0[i])
// must lie inside the span.  status[i]: 0 = consensus applied, 1 = fell back to the reference sequence (REF mismatch,
// as the reference does, utils/data_process.py:73-88), negative VF_CONS_* = window refused (only VF_CONS_INDEL under the
// strict policy; ids are left as padding).  vcf == NULL: reference sequence only.  Returns the number of windows
// written, or -1 on bad arguments.
extern "C" int64_t vf_build_windows(const void* vcf, const void* bpe, const char* chrom, int64_t span_start0,
                                    const char* span_ref, int64_t span_len, int64_t n, const int64_t* starts0,
                                    const int64_t* ends0, int snp_only, int indel_policy, int revcomp, int L,
                                    int64_t pad_id, int64_t* ids_out, uint8_t* mask_out, int32_t* status) {
    if (!bpe || !chrom || !span_ref || n < 0 || L <= 0 || !starts0 || !ends0 || !ids_out || !mask_out) return -1;
    std::string cons;
    std::vector<int32_t> ids;
    for (int64_t i = 0; i < n; ++i) {
        int64_t* row = ids_out + i * L;
        uint8_t* mrow = mask_out + i * L;
        for (int k = 0; k < L; ++k) { row[k] = pad_id; mrow[k] = 1; }
        const int64_t a = starts0[i] - span_start0, b = ends0[i] - span_start0;
        if (a < 0 || b > span_len || b < a) return -1;
        const char* ref = span_ref + a;
        const int64_t len = b - a;
        int st = 0;
        const char* seq = ref;
        int64_t seq_len = len;
        if (vcf) {
            int64_t cap = len + 4096;
            int64_t got;
            while (true) {
                cons.resize((size_t)cap);
                got = vf_vcf_consensus(vcf, chrom, starts0[i], ref, len, snp_only, indel_policy, &cons[0], cap, nullptr);
                if (got == VF_CONS_BAD_ARG && cap < 16 * (len + 4096)) { cap *= 4; continue; }    // insertions overflowed
                break;
            }
            if (got >= 0) { seq = cons.data(); seq_len = got; }
            else if (got == VF_CONS_INDEL) { if (status) status[i] = (int32_t)got; continue; }
            else st = 1;                                                                        // reference fallback
        }
        if (revcomp) {
            std::string rc((size_t)seq_len, 'N');
            for (int64_t k = 0; k < seq_len; ++k) rc[(size_t)k] = comp(seq[seq_len - 1 - k]);
            cons.swap(rc);
            seq = cons.data();
        }
        ids.resize((size_t)seq_len + 1);
        const int64_t nt = vf_bpe_encode(bpe, seq, seq_len, ids.data(), nullptr, (int64_t)ids.size());
        if (nt < 0) return -1;
        const int64_t keep = nt < L ? nt : L;
        for (int64_t k = 0; k < keep; ++k) { row[k] = ids[(size_t)k]; mrow[k] = 0; }
        if (status) status[i] = st;
    }
    return n;
}

// Host staging of a batch's token ids (prepare_batch): src int64 [rows, src_row_stride elements apart, L used] -> dst int32
// [rows, L], clamped to [-1, INT32_MAX] (the embedding kernels clamp to [0, vocab): the same token either way, and narrowing
// never wraps), in ONE pass that also says what the ids looked like.  Returns a bit mask: 1 = some id < 0, 2 = some id >= 2^30
// (the window de-duplication packs (id | pad << 30) into one word and is skipped then).  Replaces two numpy reductions and a
// converting copy per gene (19 + 8 ms per 32-gene batch on an 8-core host).
extern "C" int vf_narrow_ids(const int64_t* src, int64_t src_row_stride, int32_t* dst, int64_t rows, int64_t L) {
    if (!src || !dst || rows < 0 || L < 0) return -1;
    int neg = 0, big = 0;
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t* s = src + r * src_row_stride;
        int32_t* d = dst + r * L;
        for (int64_t i = 0; i < L; ++i) {
            int64_t v = s[i];
            neg |= (v < 0);
            big |= (v >= (1LL << 30));
            v = v < -1 ? -1 : (v > 2147483647LL ? 2147483647LL : v);
            d[i] = (int32_t)v;
        }
    }
    return neg | (big << 1);
}
