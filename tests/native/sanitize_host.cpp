#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "vf_hip.h"

static void write_file(const char* p, const std::string& s) { FILE* f = fopen(p, "w"); fwrite(s.data(), 1, s.size(), f); fclose(f); }

int main() {
    // ---- VCF parser on hostile input
    std::string hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n";
    std::vector<std::string> bodies = {
        "",                                                           // header only
        "chr1\t5\t.\tA\tC\t.\t.\t.\tGT\t0/1\t1/1\n",
        "chr1\n",                                                     // too few columns
        "chr1\t\t\t\t\n",
        "chr1\t5\t.\tA\t.\t.\t.\t.\tGT\t0/1\n",                     // no ALT
        "chr1\t5\t.\tA\tC,G,T\t.\t.\t.\tDP:GT\t9:3/7\t1:.\n",       // allele index beyond ALT list
        "chr1\t5\t.\tA\tC\t.\t.\t.\tDP\t9\t1\n",                     // no GT in FORMAT
        "chr1\t5\t.\tA\tC\t.\t.\t.\tGT\n",                           // FORMAT but no sample
        "chr1\t99999999999\t.\tA\tC\t.\t.\t.\tGT\t1/1\n",            // position far outside
        "chr1\t-5\t.\tA\tC\t.\t.\t.\tGT\t1/1\n",
        "chr1\t3\t.\tACGTACGTACGTACGT\tA\t.\t.\t.\tGT\t1/1\t0/0\n",   // deletion longer than the region
        "chr1\t3\t.\tA\t<DEL>,C\t.\t.\t.\tGT\t2/2\n",
        "chr1\t3\t.\tA\tC\t.\t.\t.\tGT\t1/1\nchr1\t2\t.\tC\tT\t.\t.\t.\tGT\t0/1\n",   // unsorted
        std::string("chr1\t4\t.\tT\t") + std::string(100000, 'A') + "\t.\t.\t.\tGT\t1/1\n",   // huge insertion
        "chr1\t4\t.\tT\tC\t.\t.\t.\tGT\t1/1"                         // no trailing newline
    };
    const char* ref = "ACGTACGTAC";
    for (size_t i = 0; i < bodies.size(); ++i) {
        write_file("t.vcf", hdr + bodies[i]);
        for (const char* sample : {(const char*)nullptr, "S1", "S2", "NOPE"}) {
            void* h = vf_vcf_open("t.vcf", sample);
            if (!h) continue;
            vf_vcf_num_records(h, nullptr);
            vf_vcf_num_records(h, "chr1");
            for (int snp = 0; snp < 2; ++snp)
                for (int pol = 0; pol < 2; ++pol)
                    for (int64_t start : {0LL, 2LL, 7LL}) {
                        char out[64];
                        int64_t n_app = 0;
                        const int64_t len = 10 - start;
                        int64_t n = vf_vcf_consensus(h, "chr1", start, ref + start, len, snp, pol, out, sizeof(out), &n_app);
                        if (n > (int64_t)sizeof(out)) { printf("overflow\n"); return 1; }
                        std::vector<char> big(200000);
                        vf_vcf_consensus(h, "chr1", start, ref + start, len, snp, pol, big.data(), (int64_t)big.size(), &n_app);
                        vf_vcf_consensus(h, "chrZ", start, ref + start, len, snp, pol, out, sizeof(out), nullptr);
                        vf_vcf_consensus(h, "chr1", start, ref + start, 0, snp, pol, out, 0, nullptr);
                    }
            vf_vcf_close(h);
        }
    }
    if (vf_vcf_open("/nonexistent/file.vcf", nullptr)) return 2;
    // ---- BPE on hostile input
    int32_t char_ids[256];
    for (int i = 0; i < 256; ++i) char_ids[i] = -1;
    const char* alpha = "ACGTRYSWKMBDHV";
    for (int i = 0; alpha[i]; ++i) char_ids[(unsigned char)alpha[i]] = 4 + i;
    std::vector<int32_t> merges = {4, 5, 18, 18, 6, 19, 6, 7, 20, 19, 7, 21, 4, 4, 22};     // AC, ACG, GT, ACGT, AA
    void* b = vf_bpe_create(char_ids, 23, merges.data(), (int)merges.size() / 3);
    if (!b) return 3;
    std::vector<std::string> seqs = {"", "A", "ACGTACGTNNNNACG", std::string(100000, 'A'), "nnnn", "\x01\xff\x80", "ACGT" + std::string(50, 'N') + "ACGT"};
    for (auto& s : seqs) {
        int64_t n = vf_bpe_encode(b, s.data(), (int64_t)s.size(), nullptr, nullptr, 0);
        std::vector<int32_t> ids((size_t)(n > 0 ? n : 1));
        std::vector<int64_t> st((size_t)(n > 0 ? n : 1));
        int64_t m = vf_bpe_encode(b, s.data(), (int64_t)s.size(), ids.data(), st.data(), n);
        if (m != n) { printf("bpe count mismatch %lld %lld\n", (long long)n, (long long)m); return 4; }
        vf_bpe_encode(b, s.data(), (int64_t)s.size(), ids.data(), st.data(), n / 2);     // truncated capacity
    }
    vf_bpe_destroy(b);
    printf("sanitizer harness ok\n");
    return 0;
}
