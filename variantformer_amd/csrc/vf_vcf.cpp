// vf_vcf: host-side (CPU, no HIP) VCF reader and per-region IUPAC consensus.
//
// Replaces the per-window `samtools faidx | bcftools consensus -H I -e '<filter>' sample.vcf.gz` subprocess pair
// the reference launches for every CRE window and every gene body (utils/data_process.py:17-101, 367-467) with one
// pass over the sample's VCF (kept in memory, sorted per chromosome) and an in-process apply per region.
//
// PARITY UNPINNED: bcftools (htslib) is a third-party tool that is neither vendored in the reference nor present in
// this image, so nothing here could be checked against it.  The semantics below restate bcftools-consensus' documented
// behaviour for `-H I` ("IUPAC code for all genotypes") on single-base substitutions, which is what the reference's
// "SNP" mode selects (`-e 'ALT~"<.*>" || TYPE!="snp"'`):
//   * a record takes part when no ALT allele is symbolic (`<...>`) and -- SNP mode -- EVERY ALT allele is a snp.
//     bcftools' filter compares TYPE as a bit set: `TYPE="snp"` holds when the union of the alleles' types (htslib
//     bcf_get_variant_types: each ALT classified against REF after trimming the common prefix / suffix) is exactly
//     {snp}, `TYPE~"snp"` when it contains snp (bcftools(1), EXPRESSIONS: "the equal sign to require that all alleles
//     are of the given type", filter.c filters_cmp_bit_and: `a == b` for TOK_EQ / TOK_NE, `a & b` for the regex
//     forms).  `-e 'TYPE!="snp"'` therefore drops a multi-allelic site that mixes a snp with an indel, an MNP or '*'
//     (round 2 kept such sites: the "any ALT is a snp" reading belongs to `!~`, not to `!=`);
//   * the genotype is the first sample's (or the named sample's) FORMAT/GT; a missing first allele skips the record,
//     a missing second allele counts as the first; haploid calls use their only allele;
//   * both alleles equal: that allele's base (nothing to do for 0/0); different: the IUPAC code of the two bases
//     (any base outside ACGT gives N); the substituted character is lower-case where the reference base is;
//   * REF must equal the reference base (case-insensitive), else the whole region fails -- the reference then falls
//     back to the unmodified reference sequence for that region (data_process.py:73-88), and so does the caller here;
//   * a record that starts at or before the end of an already applied one is skipped ("overlaps with another variant").
// All-variants mode (`-e 'ALT~"<.*>"'`, the vcf2exp path, utils/data_process.py:41-59) lets insertions, deletions and
// MNPs through.  `indel_policy` selects what happens to a record that is not a single-base substitution:
//   2 (default of the Python layer, "bcftools"): the `-H I` rule of bcftools consensus (consensus.c apply_variant, the
//     PICK_IUPAC / iupac_GTs branch of bcftools 1.10 ... 1.21, restated from the published source as recalled -- the
//     reference's Dockerfile builds htslib / bcftools 1.21):
//       * the alleles named by the genotype are collected (missing ones skipped); `fallback` = the first of them that
//         is not REF (REF if there is none);
//       * alleles that hold a character outside the IUPAC alphabet ('*', '<...>') do not take part;
//       * if all participating alleles have the SAME length the consensus allele is, position by position, the IUPAC
//         code of the union of their bases (het SNP 0/1 -> code of {ref, alt}; 1/2 -> code of {alt1, alt2}; equal-length
//         MNPs position-wise; hom 1/1 -> the alt itself);
//       * otherwise (an insertion / deletion against REF or against the other allele) the `fallback` allele replaces
//         REF as a whole: a het indel 0/1 APPLIES the ALT allele;
//       * a same-length replacement keeps the case of the reference bases it overwrites;
//   1 ("first_allele", round 1's opt-in guess): the first genotype allele of a non-SNP record, REF = no change;
//   0 ("error"): refuses the region (VF_CONS_INDEL) so that nothing is guessed.
// A record whose REF runs past the region end, or that starts inside the span of an applied record, is skipped, as
// bcftools does for overlaps ("The site ... overlaps with another variant, skipping").  Still PARITY UNPINNED: no
// bcftools binary exists offline to generate vectors from.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "../../include/vf_hip.h"

namespace {

struct Rec {
    int64_t pos;        // 1-based
    uint32_t ref_off;   // into pool: REF\0ALT1\0ALT2...\0
    uint16_t ref_len;
    uint8_t n_alt;
    int8_t a0, a1;      // genotype allele indices; -1 missing
    uint8_t symbolic;   // some ALT is <...> or a breakend
    uint8_t types;      // union of the ALT alleles' types (T_* bits below), htslib bcf_set_variant_type per allele
};
enum : uint8_t { T_SNP = 1, T_MNP = 2, T_INDEL = 4, T_OTHER = 8, T_OVERLAP = 16 };   // ref-identical alleles add no bit

// Type of one ALT allele against REF (htslib vcf.c bcf_set_variant_type, restated): '*' = overlap, symbolic / breakend =
// other, else trim the common prefix and the common suffix; nothing left = ref, one base against one base = snp, equal
// lengths = mnp, else indel.
uint8_t allele_type(const char* ref, size_t ref_len, const char* alt, size_t alt_len) {
    auto U = [](char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; };
    if (alt_len == 1 && alt[0] == '*') return T_OVERLAP;
    if (alt[0] == '<' || memchr(alt, '[', alt_len) || memchr(alt, ']', alt_len)) return T_OTHER;
    size_t lo = 0;
    while (lo < ref_len && lo < alt_len && U(ref[lo]) == U(alt[lo])) ++lo;
    if (lo == ref_len && lo == alt_len) return 0;
    size_t re = ref_len, ae = alt_len;
    while (re > lo && ae > lo && U(ref[re - 1]) == U(alt[ae - 1])) { --re; --ae; }
    if (re - lo == ae - lo) return (re - lo == 1) ? T_SNP : T_MNP;
    return T_INDEL;
}

struct Chrom {
    std::vector<Rec> recs;
    bool sorted = true;
};
struct Vcf {
    std::map<std::string, Chrom> chroms;
    std::vector<char> pool;
    int64_t n_records = 0;
    std::string error;
};

inline char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

char iupac_of(char a, char b) {
    static const char T[4][4] = {{'A', 'M', 'R', 'W'}, {'M', 'C', 'S', 'Y'}, {'R', 'S', 'G', 'K'}, {'W', 'Y', 'K', 'T'}};
    auto idx = [](char c) { c = up(c); return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; };
    const int i = idx(a), j = idx(b);
    return (i < 0 || j < 0) ? 'N' : T[i][j];
}

// IUPAC code <-> bit mask (A=1, C=2, G=4, T=8); -1 for a character outside the alphabet
int iupac_mask(char c) {
    switch (up(c)) {
        case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
        case 'M': return 3; case 'R': return 5; case 'W': return 9; case 'S': return 6; case 'Y': return 10; case 'K': return 12;
        case 'V': return 7; case 'H': return 11; case 'D': return 13; case 'B': return 14; case 'N': return 15;
        default: return -1;
    }
}
char mask_iupac(int m) {
    static const char T[16] = {'N', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};
    return T[m & 15];
}

// split a tab separated line in place; returns the number of fields
int split_tabs(char* s, char** f, int max_f) {
    int n = 0;
    f[n++] = s;
    for (char* p = s; *p && n < max_f; ++p)
        if (*p == '\t') { *p = 0; f[n++] = p + 1; }
    return n;
}

int parse_allele(const char*& p) {          // one GT allele: "." -> -1, digits -> index
    if (*p == '.') { ++p; return -1; }
    if (*p < '0' || *p > '9') return -1;
    int v = 0;
    while (*p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
    return v > 120 ? -1 : v;
}

bool parse_line(Vcf& V, char* line, int sample_col) {
    char* f[16];
    // fields up to the first 10 columns; later sample columns are located by walking tabs
    char* cols[10];
    int n = 0;
    char* p = line;
    cols[n++] = p;
    while (*p && n < 10) {
        if (*p == '\t') { *p = 0; cols[n++] = p + 1; }
        ++p;
    }
    (void)f;
    if (n < 5) return true;                 // malformed line: ignore
    Rec r{};
    r.pos = strtoll(cols[1], nullptr, 10);
    const char* ref = cols[3];
    char* alt = cols[4];
    const size_t ref_len = strlen(ref);
    if (ref_len > 65535) return true;
    r.ref_len = (uint16_t)ref_len;
    r.ref_off = (uint32_t)V.pool.size();
    V.pool.insert(V.pool.end(), ref, ref + ref_len + 1);
    r.types = 0;
    r.n_alt = 0;
    if (!(alt[0] == '.' && alt[1] == 0)) {
        char* a = alt;
        while (true) {
            char* comma = strchr(a, ',');
            if (comma) *comma = 0;
            const size_t al = strlen(a);
            if (a[0] == '<' || strchr(a, '[') || strchr(a, ']')) r.symbolic = 1;     // the filter is ALT~"<.*>": '*' alone passes it
            r.types |= allele_type(ref, ref_len, a, al);
            V.pool.insert(V.pool.end(), a, a + al + 1);
            if (r.n_alt < 255) ++r.n_alt;
            if (!comma) break;
            a = comma + 1;
        }
    }                                       // no ALT: TYPE is "ref" (types stays 0)
    // genotype
    r.a0 = r.a1 = 0;
    if (n >= 10 && sample_col >= 9) {
        // FORMAT is cols[8]; find GT's index
        int gt_idx = -1, k = 0;
        for (char* q = cols[8]; ; ++k) {
            char* colon = strchr(q, ':');
            const size_t len = colon ? (size_t)(colon - q) : strlen(q);
            if (len == 2 && q[0] == 'G' && q[1] == 'T') { gt_idx = k; break; }
            if (!colon) break;
            q = colon + 1;
        }
        // walk to the sample column (cols[9] is the first sample, still holding the rest of the line)
        char* s = cols[9];
        for (int c = 9; c < sample_col && s; ++c) {
            s = strchr(s, '\t');
            if (s) ++s;
        }
        if (!s || gt_idx < 0) {
            r.a0 = r.a1 = -1;
        } else {
            char* end = strchr(s, '\t');
            if (end) *end = 0;
            const char* q = s;
            for (int i = 0; i < gt_idx && q; ++i) {
                q = strchr(q, ':');
                if (q) ++q;
            }
            if (!q) {
                r.a0 = r.a1 = -1;
            } else {
                r.a0 = (int8_t)parse_allele(q);
                if (*q == '/' || *q == '|') {
                    ++q;
                    r.a1 = (int8_t)parse_allele(q);
                    if (r.a1 < 0) r.a1 = r.a0;                  // "1/." : the called allele
                    else if (r.a0 < 0) r.a0 = r.a1;             // "./1" : likewise (bcftools skips missing alleles)
                } else {
                    r.a1 = r.a0;            // haploid
                }
            }
        }
    } else if (n < 10) {
        r.a0 = r.a1 = -1;                   // sites-only VCF: no genotype to apply
    }
    Chrom& C = V.chroms[cols[0]];
    if (!C.recs.empty() && C.recs.back().pos > r.pos) C.sorted = false;
    C.recs.push_back(r);
    ++V.n_records;
    return true;
}

}  // namespace

namespace {
// the indel / MNP / multi-allelic rules are restated, not pinned against bcftools: say so once per process
// when the default policy actually applies such a record (VF_VCF_QUIET=1 silences it).
void warn_non_snp_once() {
    static bool warned = false;
    if (warned) return;
    warned = true;
    const char* q = getenv("VF_VCF_QUIET");
    if (q && q[0] == '1') return;
    fprintf(stderr, "vf_vcf: applying a non-SNP record (indel / MNP / multi-allelic) with the restated `bcftools consensus "
                    "-H I` rules; parity with bcftools 1.21 is UNPINNED for such records (INTEGRATION.md, section on "
                    "consensus). indel_policy=\"error\" refuses them instead.\n");
}
}  // namespace

extern "C" void* vf_vcf_open(const char* path, const char* sample) {
    gzFile fp = gzopen(path, "rb");         // reads plain text, gzip and bgzip alike
    if (!fp) return nullptr;
    gzbuffer(fp, 1 << 20);
    Vcf* V = new Vcf();
    std::string line;
    std::vector<char> buf(1 << 16);
    int sample_col = 9;
    bool have_header = false;
    while (true) {
        line.clear();
        bool eof = false;
        while (true) {                       // one full line, whatever its length
            if (!gzgets(fp, buf.data(), (int)buf.size())) { eof = true; break; }
            line.append(buf.data());
            if (!line.empty() && line.back() == '\n') break;
        }
        if (line.empty() && eof) break;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty()) { if (eof) break; continue; }
        if (line[0] == '#') {
            if (line.compare(0, 6, "#CHROM") == 0) {
                have_header = true;
                if (sample && sample[0]) {
                    std::vector<char> h(line.begin(), line.end());
                    h.push_back(0);
                    int col = 0, found = -1;
                    for (char* tok = strtok(h.data(), "\t"); tok; tok = strtok(nullptr, "\t"), ++col)
                        if (col >= 9 && strcmp(tok, sample) == 0) found = col;
                    if (found < 0) { V->error = "sample not in the VCF header"; gzclose(fp); delete V; return nullptr; }
                    sample_col = found;
                }
            }
            if (eof) break;
            continue;
        }
        std::vector<char> mut(line.begin(), line.end());
        mut.push_back(0);
        parse_line(*V, mut.data(), sample_col);
        if (eof) break;
    }
    gzclose(fp);
    (void)have_header;
    for (auto& kv : V->chroms)
        if (!kv.second.sorted)
            std::stable_sort(kv.second.recs.begin(), kv.second.recs.end(),
                             [](const Rec& a, const Rec& b) { return a.pos < b.pos; });
    return V;
}

extern "C" void vf_vcf_close(void* h) { delete static_cast<Vcf*>(h); }

extern "C" int64_t vf_vcf_num_records(const void* h, const char* chrom) {
    const Vcf* V = static_cast<const Vcf*>(h);
    if (!V) return -1;
    if (!chrom || !chrom[0]) return V->n_records;
    auto it = V->chroms.find(chrom);
    return it == V->chroms.end() ? 0 : (int64_t)it->second.recs.size();
}

// Consensus of [start0, start0 + ref_len) on `chrom`.  `ref` is the reference sequence of that interval; `out` must
// hold out_cap bytes.  Returns the consensus length (== ref_len when only substitutions were applied) or a negative
// VF_CONS_* code; *n_applied counts the records applied.
extern "C" int64_t vf_vcf_consensus(const void* h, const char* chrom, int64_t start0, const char* ref, int64_t ref_len,
                                    int snp_only, int indel_policy, char* out, int64_t out_cap, int64_t* n_applied) {
    const Vcf* V = static_cast<const Vcf*>(h);
    if (n_applied) *n_applied = 0;
    if (!V || !chrom || !ref || !out || ref_len < 0 || start0 < 0) return VF_CONS_BAD_ARG;
    auto it = V->chroms.find(chrom);
    if (it == V->chroms.end() || ref_len == 0) {
        if (out_cap < ref_len) return VF_CONS_BAD_ARG;
        memcpy(out, ref, (size_t)ref_len);
        return ref_len;
    }
    const std::vector<Rec>& R = it->second.recs;
    const int64_t lo1 = start0 + 1, hi1 = start0 + ref_len;          // 1-based inclusive interval
    auto first = std::lower_bound(R.begin(), R.end(), lo1, [](const Rec& r, int64_t p) { return r.pos < p; });
    int64_t o = 0;            // bytes written
    int64_t cur = 0;          // next reference offset (0-based within ref) not yet copied
    int64_t frozen = 0;       // 1-based end of the last applied record
    int64_t applied = 0;
    const char* pool = V->pool.data();
    for (auto r = first; r != R.end() && r->pos <= hi1; ++r) {
        if (r->symbolic) continue;                                   // -e 'ALT~"<.*>"'
        if (snp_only && r->types != T_SNP) continue;                 // ... || TYPE!="snp": not every ALT is a snp
        if (r->a0 < 0) continue;                                     // missing genotype
        if (r->a0 > r->n_alt || r->a1 > r->n_alt) return VF_CONS_BAD_GT;
        if (r->a0 == 0 && r->a1 == 0) continue;                      // hom-ref
        if (r->pos <= frozen) continue;                              // overlaps an applied record
        const int64_t off = r->pos - lo1;
        const char* rref = pool + r->ref_off;
        if (off + r->ref_len > ref_len) continue;                    // runs past the region end
        for (int k = 0; k < r->ref_len; ++k)
            if (up(ref[off + k]) != up(rref[k])) return VF_CONS_REF_MISMATCH;
        auto allele = [&](int a) {                                   // a-th allele string (0 = REF)
            const char* p = rref;
            for (int k = 0; k < a; ++k) p += strlen(p) + 1;
            return p;
        };
        const char* al0 = allele(r->a0);
        const char* al1 = allele(r->a1);
        const bool single = r->ref_len == 1 && al0[1] == 0 && al1[1] == 0 && al0[0] && al1[0];
        std::string repl;
        if (single && iupac_mask(al0[0]) >= 0 && iupac_mask(al1[0]) >= 0) {
            repl.assign(1, mask_iupac(iupac_mask(al0[0]) | iupac_mask(al1[0])));
        } else if (indel_policy == 2 || (snp_only && r->types == T_SNP)) {   // bcftools consensus -H I; a snp written with
            // flanking bases (REF=AT ALT=AC) passed the TYPE filter and is an equal-length replacement in every policy
            if (indel_policy == 2 && r->types != T_SNP) warn_non_snp_once();
            const int gt[2] = {r->a0, r->a1};
            const int fallback = gt[0] > 0 ? gt[0] : gt[1];          // first non-REF genotype allele (REF if none)
            auto iupac_ok = [](const char* a) {
                if (!a[0]) return false;
                for (const char* p = a; *p; ++p)
                    if (iupac_mask(*p) < 0) return false;
                return true;
            };
            const char* use[2];
            int n_use = 0;
            for (int k = 0; k < 2; ++k) {
                const char* a = allele(gt[k]);
                if (iupac_ok(a)) use[n_use++] = a;
            }
            if (n_use == 0) continue;
            const size_t l0 = strlen(use[0]);
            const bool same_len = n_use == 1 || strlen(use[1]) == l0;
            if (same_len) {
                repl.resize(l0);
                for (size_t j = 0; j < l0; ++j) {
                    int m = iupac_mask(use[0][j]);
                    if (n_use == 2) m |= iupac_mask(use[1][j]);
                    repl[j] = mask_iupac(m);
                }
            } else {
                const char* fa = allele(fallback);
                if (!iupac_ok(fa)) continue;
                repl = fa;
            }
        } else {
            if (indel_policy == 0) return VF_CONS_INDEL;
            if (r->a0 == 0) continue;                                // first allele is REF: nothing applied
            repl = al0;
        }
        if ((int64_t)repl.size() == r->ref_len) {                    // same length: keep the case of the overwritten bases
            bool same = true;
            for (int k = 0; k < r->ref_len; ++k) {
                const char rb = ref[off + k];
                if (rb >= 'a' && rb <= 'z' && repl[k] >= 'A' && repl[k] <= 'Z') repl[k] = (char)(repl[k] + 32);
                same = same && up(repl[k]) == up(rb);
            }
            if (same) continue;                                      // the consensus allele IS the reference: no-op
        }
        // copy the untouched reference up to the record, then the replacement
        const int64_t need = (off - cur) + (int64_t)repl.size();
        if (o + need + (ref_len - off - r->ref_len) > out_cap) return VF_CONS_BAD_ARG;
        memcpy(out + o, ref + cur, (size_t)(off - cur));
        o += off - cur;
        memcpy(out + o, repl.data(), repl.size());
        o += (int64_t)repl.size();
        cur = off + r->ref_len;
        frozen = r->pos + r->ref_len - 1;
        ++applied;
    }
    if (o + (ref_len - cur) > out_cap) return VF_CONS_BAD_ARG;
    memcpy(out + o, ref + cur, (size_t)(ref_len - cur));
    o += ref_len - cur;
    if (n_applied) *n_applied = applied;
    return o;
}
