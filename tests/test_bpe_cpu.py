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
