"""Small helpers with the reference's semantics (utils/functions.py)."""
import torch


def precision2dtype(precision_str: str) -> torch.dtype:
    """Lightning precision string -> torch dtype (reference utils/functions.py:12-32): any string containing
    'bf16' -> bfloat16, else '16' -> float16, else '32' -> float32, otherwise ValueError."""
    s = precision_str.lower().strip()
    if "bf16" in s:
        return torch.bfloat16
    if "16" in s:
        return torch.float16
    if "32" in s:
        return torch.float32
    raise ValueError(f"Unknown precision string: {precision_str}")


_COMPLEMENT = str.maketrans("ACGTRYSWKMBDHVNacgtryswkmbdhvn", "TGCAYRSWMKVHDBNtgcayrswmkvhdbn")


def reverse_complement(seq: str) -> str:
    """IUPAC-aware reverse complement (reference utils/functions.py:129-172)."""
    return seq.translate(_COMPLEMENT)[::-1]
