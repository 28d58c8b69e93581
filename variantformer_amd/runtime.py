"""Run-time configuration of the hot path, held in context variables instead of module globals.

Two records, both per CONTEXT (a thread starts with the defaults; `with` blocks nest and restore):

* `Switches` -- every exact re-ordering the product applies next to the form it replaces.  All on in
  the product; the tests run a model twice under `runtime.override(<switch>=False)` to prove that a re-ordering changes no
  bit (or only the stated rounding points).  Nothing here is an environment variable and nothing is a mutable module
  attribute: a test that flips a switch cannot leak it into another thread's forward.
* `Env` -- the two deployment switches that ARE environment variables (`VF_LN_FOLD`, `VF_TRUNK16`), read ONCE per forward:
  `forward_env()` snapshots them where a forward starts (Seq2GenePredictorCombinedModulator.forward_prepared, the
  tokenizer's embed_packed when called on its own) and every layer below reads the snapshot; a layer called outside any
  forward (op-level tests) reads the environment at that moment.  The snapshot also carries the model's own sticky
  "fold off" state (the self-healing LayerNorm fold, model_combined_modulator._heal_if_ln_fold_alert), so that one model's
  alerts never change the path of another model in the same process.
"""
from __future__ import annotations

import contextvars
import os
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class Switches:
    rows_in_attention: bool = True            # looked-up projections gathered by the attention kernel's loads (section 3.13)
    counted_context_keys: bool = True         # CRE-layer context cross attention over the distinct label rows (3.15) ...
    lowrank_context: bool = True              # ... in its low-rank form (two skinny GEMMs around a 9-way softmax)
    layer0_qkv_table: bool = True             # seq2reg's first-layer Wqkv by lookup (3.12)
    pool_before_down_projection: bool = True  # seq2reg's mean pool before its last down-projection (3.11)
    overlap_cre_stream: bool = True           # the CRE layers on a side stream beside the gene layers (their small kernels fill the
                                              # tails of the gene stream's persistent GEMMs: -0.4 ... 0.8 % step time).  Bit-identical
                                              # to the single-stream order at full depth SINCE the library is built without
                                              # packed-fp32 instructions (round 6: the run-to-run differences of two streams were a
                                              # gfx950 hazard of v_pk_*_f32 op_sel forms beside another kernel's MFMAs --
                                              # csrc/build.py NO_PACKED_FP32, profiles/r06_d_*); False = one stream


_SW: contextvars.ContextVar = contextvars.ContextVar("vf_switches", default=Switches())


def switches() -> Switches:
    return _SW.get()


class override:
    """with runtime.override(rows_in_attention=False): ...   (tests and A/B scripts)"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.token = _SW.set(replace(_SW.get(), **self.kw))
        return _SW.get()

    def __exit__(self, *exc):
        _SW.reset(self.token)
        return False


def set_for_this_context(**kw) -> Switches:
    """A/B scripts: change switches for the rest of the current context (no `with` block around a whole program)."""
    _SW.set(replace(_SW.get(), **kw))
    return _SW.get()


TRUNK16_MODES = ("f16", "0")


@dataclass(frozen=True)
class Env:
    ln_fold: bool          # VF_LN_FOLD != "0" (and the model's own sticky switch, when a model opened the snapshot)
    trunk16: str           # VF_TRUNK16: "f16" (default) or "0"


def read_env() -> Env:
    mode = os.environ.get("VF_TRUNK16", "f16")
    if mode not in TRUNK16_MODES:
        raise ValueError(f"VF_TRUNK16={mode!r}: the supported values are 'f16' (default) and '0' (fp32 trunk)")
    return Env(ln_fold=os.environ.get("VF_LN_FOLD", "1") != "0", trunk16=mode)


_ENV: contextvars.ContextVar = contextvars.ContextVar("vf_env", default=None)


def env() -> Env:
    """The snapshot of the enclosing forward, or the environment as it is now (no forward around the caller)."""
    e = _ENV.get()
    return e if e is not None else read_env()


class forward_env:
    """with runtime.forward_env(fold_off=<model's sticky switch>): one read of the environment for everything below.
    Nested use keeps the outer snapshot (a tokenizer called by the model does not re-read), except that an inner
    fold_off=True still switches the fold off for its extent."""

    def __init__(self, fold_off: bool = False):
        self.fold_off = bool(fold_off)

    def __enter__(self):
        cur = _ENV.get()
        e = cur if cur is not None else read_env()
        if self.fold_off and e.ln_fold:
            e = replace(e, ln_fold=False)
        self.token = _ENV.set(e)
        return e

    def __exit__(self, *exc):
        _ENV.reset(self.token)
        return False
