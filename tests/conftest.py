import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_fixture(name):
    """(meta, arrays, state_dict, batch) for a golden fixture; weights and inputs are
    regenerated from the seeds recorded in the fixture (variantformer_amd.utils.synthetic)."""
    from variantformer_amd.utils.synthetic import make_batch, make_tensor
    from oracle.vf_oracle import alibi_slopes

    with open(os.path.join(GOLDEN, f"{name}.json")) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    sd = {}
    for k, shape in meta["state_dict_shapes"].items():
        if k.endswith(".m"):
            sd[k] = torch.tensor(alibi_slopes(shape[0]), dtype=torch.float32)
        else:
            sd[k] = torch.from_numpy(make_tensor(k, shape, meta["seed"]))
    chk = float(sum(float(v.double().abs().sum()) for v in sd.values()))
    assert abs(chk - meta["weight_abs_sum"]) <= 1e-6 * meta["weight_abs_sum"], "weight regeneration drifted"
    batch = make_batch(meta["seed"], meta["n_cres"], meta["n_chunks"], meta["tissues"], meta["token_length"],
                       cre_len_range=tuple(meta["cre_len_range"]))
    return meta, arrays, sd, batch


def load_vep_model_fixture(name: str = "vep_model"):
    """(meta, arrays, state_dict, vep_batch) of tests/golden/<name>.*: outputs of the reference's own
    variant_prediction on a seeded ref / het / hom batch (vep_model: shipped options; vep_model_opts_a / _b: option sets
    the shipped configuration leaves off)."""
    from variantformer_amd.utils.synthetic import make_tensor, make_vep_batch
    from oracle.vf_oracle import alibi_slopes

    with open(os.path.join(GOLDEN, f"{name}.json")) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    sd = {k: (torch.tensor(alibi_slopes(shape[0]), dtype=torch.float32) if k.endswith(".m")
              else torch.from_numpy(make_tensor(k, shape, meta["seed"]))) for k, shape in meta["state_dict_shapes"].items()}
    chk = float(sum(float(v.double().abs().sum()) for v in sd.values()))
    assert abs(chk - meta["weight_abs_sum"]) <= 1e-6 * meta["weight_abs_sum"], "weight regeneration drifted"
    vb = make_vep_batch(meta["seed"], meta["n_cre"], meta["n_chunks"], meta["tissues"], meta["token_length"],
                        cre_index=meta["cre_index"], gene_index=tuple(meta["gene_index"]),
                        cre_len_range=tuple(meta["cre_len_range"]))
    return meta, arrays, sd, vb


def load_s2r_opts():
    """{name: (hparams, reference output [9,1,d], state_dict, gene sample)} of tests/golden/s2r_opts.*: the reference's
    own Seq2RegPredictor(only_embed=True) for the non-shipped tokenizer options."""
    from variantformer_amd.utils.synthetic import make_gene, make_tensor
    from oracle.vf_oracle import alibi_slopes
    with open(os.path.join(GOLDEN, "s2r_opts.json")) as f:
        meta = json.load(f)["cases"]
    arrays = dict(np.load(os.path.join(GOLDEN, "s2r_opts.npz")))
    out = {}
    for name, m in meta.items():
        sd = {k: (torch.tensor(alibi_slopes(shape[0]), dtype=torch.float32) if k.endswith(".m")
                  else torch.from_numpy(make_tensor(k, shape, m["seed"]))) for k, shape in m["state_dict_shapes"].items()}
        chk = float(sum(float(v.double().abs().sum()) for v in sd.values()))
        assert abs(chk - m["weight_abs_sum"]) <= 1e-6 * m["weight_abs_sum"], "weight regeneration drifted"
        g = make_gene(m["seed"], m["n_windows"], 2, [7], m["hparams"]["token_length"], cre_len_range=tuple(m["cre_len_range"]))
        out[name] = (m["hparams"], arrays[name], sd, g)
    return out


@pytest.fixture(params=["small_sin", "small_alibi"])
def golden(request):
    return load_fixture(request.param)
