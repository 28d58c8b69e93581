"""Token/index path: the in-tree C++ BPE must be BIT-EXACT with the reference's tokenizer.
Golden ids were produced by the reference's utils/seq.BPEEncoder (tests/golden/make_golden.py)."""
import json
import os

import pytest

from tests.conftest import GOLDEN


@pytest.fixture(scope="module")
def enc():
    from variantformer_amd.utils.seq import BPEEncoder
    e = BPEEncoder()
    e.load_vocabulary()
    return e


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN, "bpe_ids.json")) as f:
        return json.load(f)


def test_known_answer(enc):
    # SURVEY §8c a-priori known answer
    ids, toks, ids_r, _ = enc.encode(["ACGTNNNNACGTRYACGT", "A"])
    assert ids == [80, 14, 80, 14, 12, 17, 80, 14] and ids_r == [4]
    assert "".join(toks) == "ACGTACGTRYACGT"


def test_ids_bit_exact_vs_reference(enc, gold):
    assert len(gold["cases"]) >= 30
    for case in gold["cases"]:
        ids, toks, _, _ = enc.encode([case["seq"], "A"])
        assert ids == case["ids"], case["seq"][:60]
        assert enc.encode_forward(case["seq"]).tolist() == case["ids"]          # the sample builders' fast path
        assert enc.decode(ids) == "".join(c for c in case["seq"].upper() if c in "ACGTRYSWKMBDHV")


def test_encode_with_position_vs_reference(enc, gold):
    seqs = [c["seq"] for c in gold["cases"]]
    n = 0
    for pc in gold["position_cases"]:
        s = seqs[pc["seq_index"]]
        if "error" in pc:
            with pytest.raises(ValueError):
                enc.encode_with_position(s, pc["position"])
            continue
        r = enc.encode_with_position(s, pc["position"])
        ref = pc["result"]
        assert r["encoded_ids"] == ref["encoded_ids"] and r["position_id"] == ref["position_id"]
        assert r["position_token"] == ref["position_token"] and r["target_subsequence"] == ref["target_subsequence"]
        assert [list(o) for o in r["offsets"]] == [list(o) for o in ref["offsets"]]
        n += 1
    assert n >= 10


def test_edge_cases(enc):
    assert enc.encode(["", "A"])[0] == [] and enc.encode(["NNNN", "A"])[0] == []
    assert enc.encode(["n", "a"])[2] == [4]
    with pytest.raises(ValueError):
        enc.encode_with_position("ACGT", 4)
    with pytest.raises(ValueError):
        enc.encode_with_position("ACNT", 2)
    long = "ACGTTGCA" * 40000                      # 320 kb: a whole gene body as ONE word
    ids, _, _, _ = enc.encode([long, "A"])
    assert enc.decode(ids) == long and len(ids) < len(long) / 2


def test_live_cross_check_against_huggingface_tokenizers(enc):
    """The reference tokenises with the third-party HuggingFace `tokenizers` BPE (utils/seq.py:14-16,43-50).  Where that
    package is importable (the dev container; not required on the GPU box) the in-tree encoder is compared with it on
    random and adversarial strings: long single words, runs of one base (overlapping same-rank pairs), IUPAC codes."""
    tk = pytest.importorskip("tokenizers")
    import numpy as np
    from variantformer_amd.utils.seq import DEFAULT_VOCAB as VOCAB_FILE
    ref = tk.Tokenizer.from_file(VOCAB_FILE)
    rng = np.random.default_rng(12)
    cases = ["A" * 1, "A" * 2, "A" * 3, "A" * 7, "A" * 64, "T" * 1001, "AC" * 500, "ACG" * 333, "AACCGGTT" * 200,
             "AAACAAACAAAAC" * 50, "R" * 40 + "Y" * 41]
    for n in (5, 33, 350, 351, 4097, 60000):
        cases.append("".join(rng.choice(list("ACGT"), n)))
        cases.append("".join(rng.choice(list("ACGTRYSWKM"), n, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02])))
        cases.append("".join(rng.choice(list("AC"), n, p=[.9, .1])))              # long runs of one base
    for s in cases:
        assert enc.encode_forward(s).tolist() == ref.encode(s).ids, (len(s), s[:40])


def test_create_refuses_merge_lists_the_bucket_order_cannot_serve():
    """vf_bpe_create accepts only merge lists in which every rule ranks above EVERY rule that creates one of its operands
    (true of any BPE-trained vocabulary).  A token id produced by two rules, with a rule about that id ranked between the
    two, would put a pair created by the later rule into a rank bucket that was already processed (round-2 advice): such
    a list must be refused, not mis-tokenised."""
    import ctypes as C
    import numpy as np
    from variantformer_amd import _lib
    lib = _lib.load()
    lib.vf_bpe_create.restype = C.c_void_p
    lib.vf_bpe_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.vf_bpe_destroy.argtypes = [C.c_void_p]
    char_ids = np.full(256, -1, np.int32)
    for k, ch in enumerate("ACGT"):
        char_ids[ord(ch)] = k                                   # ids 0..3 = A C G T; 4, 5, 6 = merged tokens
    ok = np.array([[0, 1, 4], [4, 2, 5], [2, 3, 6]], np.int32)               # AC -> 4, (4, G) -> 5, GT -> 6: monotone
    h = lib.vf_bpe_create(char_ids.ctypes.data, 7, ok.ctypes.data, len(ok))
    assert h
    lib.vf_bpe_destroy(h)
    # id 4 is created at rank 0 (AC) and again at rank 2 (GT); the rule (4, A) at rank 1 lies between the two
    bad = np.array([[0, 1, 4], [4, 0, 5], [2, 3, 4]], np.int32)
    assert not lib.vf_bpe_create(char_ids.ctypes.data, 7, bad.ctypes.data, len(bad))
    # an operand that is created only later
    bad2 = np.array([[4, 0, 5], [0, 1, 4]], np.int32)
    assert not lib.vf_bpe_create(char_ids.ctypes.data, 7, bad2.ctypes.data, len(bad2))
