// vf_bpe: host-side (CPU, no HIP) byte-pair encoder for IUPAC DNA with the reference's semantics.
//
// Replaces the third-party HuggingFace `tokenizers` BPE model (Rust) as the reference drives it through
// utils/seq.py:BPEEncoder (normalize :32-41, encode :52-62, encode_with_position :68-174) with
// vocabs/bpe_vocabulary_500.json: no normalizer, no pre-tokenizer, no dropout -> every maximal run of valid
// characters is ONE word whose characters are merged greedily by merge rank (lowest rank first, leftmost
// first within a rank), new pairs created by a merge entering the same priority queue -- the algorithm of
// tokenizers' `Word::merge_all`, restated here.  Token ids must be bit-exact (tests/golden/bpe_ids.json).
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../../include/vf_hip.h"

namespace {
struct Bpe {
    int32_t char_id[256];          // upper-cased byte -> vocab id, -1 = not a valid symbol (splits the sequence)
    int n_ids;
    std::vector<int32_t> rank;     // [n_ids * n_ids], -1 = no merge
    std::vector<int32_t> merged;   // [n_ids * n_ids] id of the merged token
    std::vector<int16_t> rank16;   // the same ranks, 2 bytes each (500 KB for the shipped vocabulary: stays in the L2 of a
                                   // host core; the merged id follows from the rank: new_id_of_rank)
    std::vector<int32_t> new_id_of_rank;   // [n_rank] token produced by the merge of that rank
    int n_rank = 0;
    int max_token_len = 1;         // characters of the longest token a merge can produce (vf_bpe_encode_prefix's margin)
    bool monotone = true;          // every rule that involves a merged token has a higher rank than the token's own rule
};
struct Sym { int32_t c, prev, next, len; };

// Reusable per-thread work space (a gene body is one 300 kb word: no allocation per call)
struct Scratch {
    std::vector<Sym> sym;
    std::vector<std::vector<int32_t>> bucket;      // candidate positions per merge rank
    std::vector<uint8_t> used;                     // rank has candidates (short words touch few of the ~480 ranks)
    std::vector<int32_t> used_list;
};

// merge one word (a run of valid characters); appends ids and raw start offsets of the resulting tokens.
//
// tokenizers' Word::merge_all pops candidate pairs from a binary heap ordered by (rank, position) and validates
// each against the current symbols.  A pair created by a merge of rank r contains the token that merge produced,
// and every rule about that token was learned later, i.e. has a rank > r: candidates therefore only ever enter
// HIGHER ranks than the one being processed, and the heap order equals "rank buckets in ascending order, positions
// ascending inside a bucket".  Buckets of plain ints replace the heap (the heap was 75 % of the host time per gene);
// the position order inside a bucket is reproduced without sorting (see the run walk below).
void encode_word(const Bpe& B, Scratch& S, const char* s, int64_t n, int64_t raw0, std::vector<int32_t>& ids,
                 std::vector<int64_t>& starts) {
    std::vector<Sym>& sym = S.sym;
    sym.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        unsigned char ch = (unsigned char)s[i];
        if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
        sym[(size_t)i] = {B.char_id[ch], (int32_t)(i - 1), (int32_t)(i + 1 < n ? i + 1 : -1), 1};
    }
    const int N = B.n_ids;
    const int n_rank = B.n_rank;
    const int16_t* R = B.rank16.data();
    if ((int)S.bucket.size() < n_rank) { S.bucket.resize((size_t)n_rank); S.used.assign((size_t)n_rank, 0); }
    auto push = [&](int32_t r, int32_t pos) {
        if (!S.used[(size_t)r]) { S.used[(size_t)r] = 1; S.used_list.push_back(r); }
        S.bucket[(size_t)r].push_back(pos);
    };
    for (int64_t i = 0; i + 1 < n; ++i) {
        const int32_t r = R[(size_t)sym[(size_t)i].c * N + sym[(size_t)i + 1].c];
        if (r >= 0) push(r, (int32_t)i);
    }
    // ranks in ascending order (candidates only ever enter higher ranks); unused ranks cost one byte test each
    for (int r = 0; r < n_rank; ++r) {
        if (!S.used[(size_t)r]) continue;
        // Candidates of one rank are applied WITHOUT sorting them by position: two rank-r pairs can only overlap when
        // the rule is (a, a) and the text holds a run a a a ..., where the heap order means "leftmost pair first".  So
        // from any candidate the walk first moves left to the start of such a run and then merges greedily rightwards;
        // disjoint merges of one rank commute, and entries that went stale on the way fail the validation below.
        std::vector<int32_t>& bk = S.bucket[(size_t)r];
        const int32_t new_id = B.new_id_of_rank[(size_t)r];
        for (size_t e = 0; e < bk.size(); ++e) {
            int32_t pos = bk[e];
            {
                const Sym& c0 = sym[(size_t)pos];
                if (c0.len == 0 || c0.next == -1) continue;
                if (R[(size_t)c0.c * N + sym[(size_t)c0.next].c] != r) continue;      // expired candidate
            }
            while (sym[(size_t)pos].prev >= 0 &&
                   R[(size_t)sym[(size_t)sym[(size_t)pos].prev].c * N + sym[(size_t)pos].c] == r)
                pos = sym[(size_t)pos].prev;
            while (pos >= 0) {
                Sym& cur = sym[(size_t)pos];
                if (cur.next == -1) break;
                const int32_t next_pos = cur.next;
                const Sym right = sym[(size_t)next_pos];
                if (R[(size_t)cur.c * N + right.c] != r) break;
                cur.c = new_id;
                cur.len += right.len;
                cur.next = right.next;
                sym[(size_t)next_pos].len = 0;
                if (right.next > -1) sym[(size_t)right.next].prev = pos;
                if (cur.prev >= 0) {
                    const int32_t rr = R[(size_t)sym[(size_t)cur.prev].c * N + cur.c];
                    if (rr >= 0) push(rr, cur.prev);
                }
                if (cur.next >= 0) {
                    const int32_t rr = R[(size_t)cur.c * N + sym[(size_t)cur.next].c];
                    if (rr >= 0) push(rr, pos);
                }
                pos = cur.next;                                                        // next pair of the run, if any
            }
        }
    }
    for (int32_t u : S.used_list) { S.bucket[(size_t)u].clear(); S.used[(size_t)u] = 0; }
    S.used_list.clear();
    int64_t off = raw0;
    for (int32_t i = 0; i >= 0 && i < n; i = sym[(size_t)i].next) {
        ids.push_back(sym[(size_t)i].c);
        starts.push_back(off);
        off += sym[(size_t)i].len;
        if (sym[(size_t)i].next < 0) break;
    }
}
}  // namespace

extern "C" void* vf_bpe_create(const int32_t* char_ids, int n_ids, const int32_t* merges, int n_merges) {
    if (!char_ids || n_ids <= 0 || n_ids > 65536 || (n_merges > 0 && !merges)) return nullptr;
    Bpe* B = new Bpe();
    for (int i = 0; i < 256; ++i) B->char_id[i] = (char_ids[i] >= 0 && char_ids[i] < n_ids) ? char_ids[i] : -1;
    B->n_ids = n_ids;
    B->rank.assign((size_t)n_ids * n_ids, -1);
    B->merged.assign((size_t)n_ids * n_ids, -1);
    for (int r = 0; r < n_merges; ++r) {
        const int32_t a = merges[3 * r], b = merges[3 * r + 1], c = merges[3 * r + 2];
        if (a < 0 || b < 0 || c < 0 || a >= n_ids || b >= n_ids || c >= n_ids) { delete B; return nullptr; }
        const size_t k = (size_t)a * n_ids + b;
        if (B->rank[k] < 0) { B->rank[k] = r; B->merged[k] = c; }
    }
    B->n_rank = n_merges;
    if (n_merges > 32767) { delete B; return nullptr; }
    B->rank16.resize(B->rank.size());
    for (size_t k = 0; k < B->rank.size(); ++k) B->rank16[k] = (int16_t)B->rank[k];
    B->new_id_of_rank.assign((size_t)(n_merges > 0 ? n_merges : 1), -1);
    // rank at which every token is created (characters: -1); the bucket algorithm needs rank(rule) > creation rank of
    // both of its operands, which holds for any vocabulary learned by BPE training.  A token id may be the product of
    // SEVERAL active rules (HF training can emit one id from two rules): a merge at the later of them creates pairs
    // (x, c) too, so the rule for such a pair must rank above the LAST rule that creates c -- born[] is the maximum
    // creation rank, computed before any rule is checked (round-2 advice: with the first rank, a pair created at the
    // later rule could fall into a bucket already processed and be dropped silently).
    std::vector<int32_t> born((size_t)n_ids, -1);
    for (int r = 0; r < n_merges; ++r) {
        const int32_t a = merges[3 * r], b = merges[3 * r + 1], c = merges[3 * r + 2];
        if (B->rank[(size_t)a * n_ids + b] != r) continue;   // a later duplicate of an earlier rule never fires
        B->new_id_of_rank[(size_t)r] = c;
        if (born[(size_t)c] < r) born[(size_t)c] = r;
    }
    for (int r = 0; r < n_merges; ++r) {
        const int32_t a = merges[3 * r], b = merges[3 * r + 1];
        if (B->rank[(size_t)a * n_ids + b] != r) continue;
        if (born[(size_t)a] >= r || born[(size_t)b] >= r) B->monotone = false;
    }
    if (!B->monotone) { delete B; return nullptr; }          // not a BPE-trained merge list: refuse rather than mis-tokenise
    {   // token lengths in characters: 1 for the alphabet, len(a) + len(b) for a merge product (rules in rank order: operands exist)
        std::vector<int32_t> tl((size_t)n_ids, 1);
        for (int r = 0; r < n_merges; ++r) {
            const int32_t a = merges[3 * r], b = merges[3 * r + 1], c = merges[3 * r + 2];
            if (B->rank[(size_t)a * n_ids + b] != r) continue;
            tl[(size_t)c] = tl[(size_t)a] + tl[(size_t)b];
            if (tl[(size_t)c] > B->max_token_len) B->max_token_len = tl[(size_t)c];
        }
    }
    return B;
}

extern "C" void vf_bpe_destroy(void* h) { delete static_cast<Bpe*>(h); }

extern "C" int64_t vf_bpe_encode(const void* h, const char* seq, int64_t len, int32_t* ids_out, int64_t* starts_out,
                                 int64_t capacity) {
    if (!h || (!seq && len > 0) || len < 0) return -1;
    const Bpe& B = *static_cast<const Bpe*>(h);
    static thread_local Scratch S;
    static thread_local std::vector<int32_t> ids;
    static thread_local std::vector<int64_t> starts;
    ids.clear();
    starts.clear();
    int64_t i = 0;
    while (i < len) {
        auto valid = [&](int64_t j) {
            unsigned char ch = (unsigned char)seq[j];
            if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
            return B.char_id[ch] >= 0;
        };
        while (i < len && !valid(i)) ++i;
        int64_t j = i;
        while (j < len && valid(j)) ++j;
        if (j > i) encode_word(B, S, seq + i, j - i, i, ids, starts);
        i = j;
    }
    const int64_t n = (int64_t)ids.size();
    if (ids_out) for (int64_t k = 0; k < n && k < capacity; ++k) ids_out[k] = ids[(size_t)k];
    if (starts_out) for (int64_t k = 0; k < n && k < capacity; ++k) starts_out[k] = starts[(size_t)k];
    return n;
}

// The FIRST max_tokens tokens of vf_bpe_encode(seq, len), exactly, without encoding the rest of a long word (a gene body is one
// 300 kb word of which the sample builder keeps the first max_chunks x max_length = 40 000 tokens, reference
// datasets/vcfdataset.py:338-394: more than half of the merges were thrown away).
//
// Why a prefix can be encoded on its own: merges are applied rank by rank (candidates created by a merge only enter HIGHER ranks,
// see encode_word), and within one rank a pair (i, i + 1) fires iff its two symbols match the rule and symbol i was not consumed
// by the pair to its LEFT (same-rank pairs overlap only in runs a a a ..., merged leftmost first).  So if two runs of the algorithm
// hold the same symbols left of some frontier before a rank is processed, they hold the same symbols left of (frontier - 1 symbol)
// after it: a difference on the right -- the text beyond a cut -- moves left by at most ONE symbol per rank, and a symbol never
// exceeds the longest token.  Tokens that end at least  n_rank x max_token_len  characters before the cut are therefore those of
// the full encoding (tests/test_bpe_cpu.py compares with the full encoding at every cut of random and repetitive texts).
extern "C" int64_t vf_bpe_encode_prefix(const void* h, const char* seq, int64_t len, int64_t max_tokens, int32_t* ids_out,
                                        int64_t* starts_out, int64_t capacity) {
    if (!h || (!seq && len > 0) || len < 0 || max_tokens < 0) return -1;
    const Bpe& B = *static_cast<const Bpe*>(h);
    static thread_local Scratch S;
    static thread_local std::vector<int32_t> ids, tmp_ids;
    static thread_local std::vector<int64_t> starts, tmp_starts;
    ids.clear();
    starts.clear();
    const int64_t margin = (int64_t)B.n_rank * B.max_token_len;
    int64_t i = 0;
    while (i < len && (int64_t)ids.size() < max_tokens) {
        auto valid = [&](int64_t j) {
            unsigned char ch = (unsigned char)seq[j];
            if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
            return B.char_id[ch] >= 0;
        };
        while (i < len && !valid(i)) ++i;
        int64_t j = i;
        while (j < len && valid(j)) ++j;
        if (j == i) break;
        const int64_t n = j - i, need = max_tokens - (int64_t)ids.size();
        int64_t cut = need * 4 + margin;                       // ~3.6 characters per token on DNA; grown below when short
        for (;;) {
            if (cut >= n - margin) cut = n;                    // (a cut inside the last margin saves nothing)
            tmp_ids.clear();
            tmp_starts.clear();
            encode_word(B, S, seq + i, cut, i, tmp_ids, tmp_starts);
            int64_t exact = (int64_t)tmp_ids.size();
            if (cut < n) {                                     // tokens that END within the last `margin` characters may differ
                const int64_t limit = i + cut - margin;
                while (exact > 0 && (exact == (int64_t)tmp_ids.size() ? i + cut : tmp_starts[(size_t)exact]) > limit) --exact;
            }
            if (exact >= need || cut == n) {
                const int64_t take = exact < need ? exact : need;
                ids.insert(ids.end(), tmp_ids.begin(), tmp_ids.begin() + take);
                starts.insert(starts.end(), tmp_starts.begin(), tmp_starts.begin() + take);
                break;
            }
            cut += cut / 2 + margin;
        }
        i = j;
    }
    const int64_t nt = (int64_t)ids.size();
    if (ids_out) for (int64_t k = 0; k < nt && k < capacity; ++k) ids_out[k] = ids[(size_t)k];
    if (starts_out) for (int64_t k = 0; k < nt && k < capacity; ++k) starts_out[k] = starts[(size_t)k];
    return nt;
}
