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
//   * a record takes part when no ALT allele is symbolic (`<...>`) and -- SNP mode -- REF and every ALT are one base;
//   * the genotype is the first sample's (or the named sample's) FORMAT/GT; a missing first allele skips the record,
//     a missing second allele counts as the first; haploid calls use their only allele;
//   * both alleles equal: that allele's base (nothing to do for 0/0); different: the IUPAC code of the two bases
//     (any base outside ACGT gives N); the substituted character is lower-case where the reference base is;
//   * REF must equal the reference base (case-insensitive), else the whole region fails -- the reference then falls
//     back to the unmodified reference sequence for that region (data_process.py:73-88), and so does the caller here;
//   * a record that starts at or before the end of an already applied one is skipped ("overlaps with another variant").
// Insertions / deletions (all-variants mode, `-e 'ALT~"<.*>"'`, the vcf2exp path): `indel_policy` 0 refuses the region
// (VF_CONS_INDEL) so that nothing is guessed; 1 applies the FIRST genotype allele of a non-SNP record (bcftools 1.x
// source as recalled, unverifiable here) -- callers must opt in.
#include <stdint.h>
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
    uint8_t symbolic;   // some ALT is <...> (or '*')
    uint8_t all_snp;    // REF and every ALT are a single base
};
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
    r.all_snp = ref_len == 1;
    r.n_alt = 0;
    if (!(alt[0] == '.' && alt[1] == 0)) {
        char* a = alt;
        while (true) {
            char* comma = strchr(a, ',');
            if (comma) *comma = 0;
            const size_t al = strlen(a);
            if (a[0] == '<' || a[0] == '*' || strchr(a, '[') || strchr(a, ']')) r.symbolic = 1;
            if (al != 1) r.all_snp = 0;
            V.pool.insert(V.pool.end(), a, a + al + 1);
            if (r.n_alt < 255) ++r.n_alt;
            if (!comma) break;
            a = comma + 1;
        }
    } else {
        r.all_snp = 0;                      // no ALT: TYPE is "ref"
    }
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
                if (*q == '/' || *q == '|') { ++q; r.a1 = (int8_t)parse_allele(q); if (r.a1 < 0) r.a1 = r.a0; }
                else r.a1 = r.a0;           // haploid
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
        if (snp_only && !r->all_snp) continue;                       // ... || TYPE!="snp"
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
        if (single) {
            char c = r->a0 == r->a1 ? al0[0] : iupac_of(al0[0], al1[0]);
            const char rb = ref[off];
            if (rb >= 'a' && rb <= 'z' && c >= 'A' && c <= 'Z') c = (char)(c + 32);
            repl.assign(1, c);
        } else {
            if (indel_policy == 0) return VF_CONS_INDEL;
            if (r->a0 == 0) continue;                                // first allele is REF: nothing applied
            repl = al0;
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
