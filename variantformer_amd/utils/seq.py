"""BPEEncoder with the reference's interface (utils/seq.py:8-174) on the in-tree C++ encoder (vf_bpe_* in
libvf_hip.so) instead of the HuggingFace `tokenizers` package.  Token ids are bit-exact with the reference
(tests/test_bpe_cpu.py checks every string of tests/golden/bpe_ids.json)."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from .. import _lib
from .constants import IUPAC_CODES

DEFAULT_VOCAB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vocabs", "bpe_vocabulary_500.json")


class BPEEncoder:
    def __init__(self):
        self._h = None
        self._lib = None
        self.vocab = {}
        self.id_to_token = {}

    def __del__(self):
        if self._h is not None and self._lib is not None:
            self._lib.vf_bpe_destroy(self._h)
            self._h = None

    def load_vocabulary(self, vocab_file: str = DEFAULT_VOCAB):
        """Load a HuggingFace tokenizer JSON (model.type == "BPE", no normalizer / pre-tokenizer)."""
        with open(vocab_file) as f:
            spec = json.load(f)
        model = spec["model"]
        if model.get("type") != "BPE" or spec.get("normalizer") or spec.get("pre_tokenizer") or model.get("dropout"):
            raise ValueError("only a plain BPE model without normalizer / pre-tokenizer / dropout is supported")
        self.vocab = dict(model["vocab"])
        self.id_to_token = {i: t for t, i in self.vocab.items()}
        n_ids = max(self.vocab.values()) + 1
        char_ids = np.full(256, -1, dtype=np.int32)
        for ch in IUPAC_CODES:                       # the reference splits on everything else (e.g. 'N')
            char_ids[ord(ch)] = self.vocab[ch]
        merges = []
        for m in model["merges"]:
            a, b = (m.split(" ") if isinstance(m, str) else m)
            merges.append((self.vocab[a], self.vocab[b], self.vocab[a + b]))
        merges = np.asarray(merges, dtype=np.int32).reshape(-1, 3)
        self._lib = _lib.load()
        if self._h is not None:
            self._lib.vf_bpe_destroy(self._h)
        self._h = self._lib.vf_bpe_create(char_ids.ctypes.data, n_ids, merges.ctypes.data, len(merges))
        if not self._h:
            raise _lib.VFError("vf_bpe_create failed")
        print(f"Loaded BPE vocabulary from {vocab_file}")

    # -- core -------------------------------------------------------------------------------------
    def encode_ids(self, seq: str):
        """(ids int32 array, start offset of every token in the RAW string)."""
        if self._h is None:
            self.load_vocabulary()
        raw = seq.encode("ascii", errors="replace")
        cap = len(raw) + 1
        ids = np.empty(cap, dtype=np.int32)
        starts = np.empty(cap, dtype=np.int64)
        n = self._lib.vf_bpe_encode(self._h, raw, len(raw), ids.ctypes.data, starts.ctypes.data, cap)
        if n < 0:
            raise _lib.VFError("vf_bpe_encode failed")
        return ids[:n].copy(), starts[:n].copy()

    def encode_forward(self, seq: str, max_tokens: int | None = None) -> np.ndarray:
        """Token ids of one strand, int32.  Same ids as encode([seq, "A"])[0]: the C++ encoder upper-cases and
        splits at invalid characters itself, so the Python-side normalize() pass and the token strings are skipped
        (the sample builders need neither).  max_tokens: only the first max_tokens tokens -- exactly those of the full
        encoding (vf_bpe_encode_prefix), at the cost of encoding ~5 characters per requested token instead of all of `seq`."""
        if max_tokens is None:
            return self.encode_ids(seq)[0]
        if self._h is None:
            self.load_vocabulary()
        raw = seq.encode("ascii", errors="replace")
        cap = min(len(raw), int(max_tokens)) + 1
        ids = np.empty(cap, dtype=np.int32)
        n = self._lib.vf_bpe_encode_prefix(self._h, raw, len(raw), int(max_tokens), ids.ctypes.data, None, cap)
        if n < 0:
            raise _lib.VFError("vf_bpe_encode_prefix failed")
        return ids[:n].copy()

    # -- reference interface ------------------------------------------------------------------------
    def normalize(self, sequences):
        out = []
        for seq in sequences:
            seq = seq.upper()
            out.extend(s for s in "".join(c if c in IUPAC_CODES else " " for c in seq).split() if s)
        return out

    def encode_strand(self, text):
        ids, toks = [], []
        for seq in text:
            i, _ = self.encode_ids(seq)
            ids.extend(int(v) for v in i)
            toks.extend(self.id_to_token[int(v)] for v in i)
        return ids, toks

    def encode(self, sequences):
        text = sequences.split(",") if isinstance(sequences, str) else sequences
        f_ids, f_tok = self.encode_strand(self.normalize([text[0]]))
        r_ids, r_tok = self.encode_strand(self.normalize([text[1]]))
        return f_ids, f_tok, r_ids, r_tok

    def decode(self, encoded_sequence):
        return "".join(self.id_to_token[int(i)] for i in encoded_sequence if int(i) > 3)

    def encode_with_position(self, sequence, position):
        """Token covering raw position `position` (reference :68-174; same keys, same ValueErrors)."""
        if position < 0 or position >= len(sequence):
            raise ValueError(f"Position {position} is out of range for the sequence of length {len(sequence)}.")
        sequence = sequence.upper()
        if sequence[position] not in IUPAC_CODES:
            raise ValueError(f"Position {position} points to invalid character '{sequence[position]}' "
                             f"which is filtered out during normalization.")
        ids, starts = self.encode_ids(sequence)
        tok = int(np.searchsorted(starts, position, side="right") - 1)
        # boundaries of the valid run (sub-sequence) that contains the position
        a = position
        while a > 0 and sequence[a - 1] in IUPAC_CODES:
            a -= 1
        b = position
        while b < len(sequence) and sequence[b] in IUPAC_CODES:
            b += 1
        in_run = [k for k in range(len(ids)) if a <= starts[k] < b]
        offsets = [(int(starts[k] - a), int(starts[k] - a + len(self.id_to_token[int(ids[k])]))) for k in in_run]
        return {"encoded_ids": [int(v) for v in ids], "all_tokens": [self.id_to_token[int(v)] for v in ids],
                "offsets": offsets, "position_id": tok, "position_token": self.id_to_token[int(ids[tok])],
                "target_subsequence": sequence[a:b]}
