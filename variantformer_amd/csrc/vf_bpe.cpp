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
#include <queue>
#include <vector>
#include "../../include/vf_hip.h"

namespace {
struct Bpe {
    int32_t char_id[256];          // upper-cased byte -> vocab id, -1 = not a valid symbol (splits the sequence)
    int n_ids;
    std::vector<int32_t> rank;     // [n_ids * n_ids], -1 = no merge
    std::vector<int32_t> merged;   // [n_ids * n_ids] id of the merged token
};
struct Sym { int32_t c, prev, next, len; };
struct Merge { int32_t pos, rank, new_id; };
struct Worse {                      // priority_queue keeps the "largest": smallest rank, then smallest pos
    bool operator()(const Merge& a, const Merge& b) const { return a.rank != b.rank ? a.rank > b.rank : a.pos > b.pos; }
};

// merge one word (a run of valid characters); appends ids and raw start offsets of the resulting tokens
void encode_word(const Bpe& B, const char* s, int64_t n, int64_t raw0, std::vector<int32_t>& ids, std::vector<int64_t>& starts) {
    std::vector<Sym> sym((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        unsigned char ch = (unsigned char)s[i];
        if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
        sym[(size_t)i] = {B.char_id[ch], (int32_t)(i - 1), (int32_t)(i + 1 < n ? i + 1 : -1), 1};
    }
    std::priority_queue<Merge, std::vector<Merge>, Worse> q;
    const int N = B.n_ids;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const int32_t r = B.rank[(size_t)sym[(size_t)i].c * N + sym[(size_t)i + 1].c];
        if (r >= 0) q.push({(int32_t)i, r, B.merged[(size_t)sym[(size_t)i].c * N + sym[(size_t)i + 1].c]});
    }
    while (!q.empty()) {
        const Merge top = q.top();
        q.pop();
        Sym& cur = sym[(size_t)top.pos];
        if (cur.len == 0 || cur.next == -1) continue;
        const int32_t next_pos = cur.next;
        const Sym right = sym[(size_t)next_pos];
        const size_t key = (size_t)cur.c * N + right.c;
        if (B.rank[key] < 0 || B.merged[key] != top.new_id) continue;      // expired queue entry
        cur.c = top.new_id;
        cur.len += right.len;
        cur.next = right.next;
        sym[(size_t)next_pos].len = 0;
        if (right.next > -1) sym[(size_t)right.next].prev = top.pos;
        if (cur.prev >= 0) {
            const size_t k = (size_t)sym[(size_t)cur.prev].c * N + cur.c;
            if (B.rank[k] >= 0) q.push({cur.prev, B.rank[k], B.merged[k]});
        }
        if (cur.next >= 0) {
            const size_t k = (size_t)cur.c * N + sym[(size_t)cur.next].c;
            if (B.rank[k] >= 0) q.push({top.pos, B.rank[k], B.merged[k]});
        }
    }
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
    return B;
}

extern "C" void vf_bpe_destroy(void* h) { delete static_cast<Bpe*>(h); }

extern "C" int64_t vf_bpe_encode(const void* h, const char* seq, int64_t len, int32_t* ids_out, int64_t* starts_out,
                                 int64_t capacity) {
    if (!h || (!seq && len > 0) || len < 0) return -1;
    const Bpe& B = *static_cast<const Bpe*>(h);
    std::vector<int32_t> ids;
    std::vector<int64_t> starts;
    ids.reserve((size_t)(len / 2 + 8));
    starts.reserve((size_t)(len / 2 + 8));
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
        if (j > i) encode_word(B, seq + i, j - i, i, ids, starts);
        i = j;
    }
    const int64_t n = (int64_t)ids.size();
    if (ids_out) for (int64_t k = 0; k < n && k < capacity; ++k) ids_out[k] = ids[(size_t)k];
    if (starts_out) for (int64_t k = 0; k < n && k < capacity; ++k) starts_out[k] = starts[(size_t)k];
    return n;
}
