"""Deterministic synthetic weights and batches.

Everything here is a pure function of integer seeds, built on a counter-based
splitmix64 stream written with numpy uint64 arithmetic, so the golden-fixture
generator (tests/golden/make_golden.py, which runs next to the reference), the
CPU tests, the GPU tests and bench.py all regenerate bit-identical weights and
inputs without committing megabytes of tensors.

The batch layout is the reference's ``collate_fn_batching`` dict
(/root/reference/datasets/vcfdataset.py:18-63): per gene ``cre_sequences[i]``
int64 [N,1,L], ``cre_attention_masks[i]`` bool [N,1,L] (True = pad), ...
"""
from __future__ import annotations

import numpy as np
import torch

PAD_ID = 0           # reference: vocabs/bpe_vocabulary_500.json "<pad>" = 0
VOCAB_SIZE = 500
FIRST_REAL_TOKEN = 4  # ids 0..3 are <pad>,<s>,</s>,<unk>

# The 54 non-cell-line tissue ids of vocabs/tissue_vocab.yaml used by the
# headline metric: ids 7..62 minus lcl (45) and blood (14).
TISSUES_54 = [t for t in range(7, 63) if t not in (14, 45)]
assert len(TISSUES_54) == 54

MATRIX_GAIN = 0.4
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def u64_stream(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n deterministic uint64 values for (seed, stream)."""
    base = _splitmix64(np.array([seed], dtype=np.uint64))[0]
    base = _splitmix64(np.array([int(base) ^ (stream * 0xD1342543DE82EF95 & 0xFFFFFFFFFFFFFFFF)], dtype=np.uint64))[0]
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64((idx * np.uint64(0x2545F4914F6CDD1D) + base) & _M64)


def uniform(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """float32 uniform in [-1, 1) with 24 random bits (exactly representable)."""
    u = u64_stream(n, seed, stream) >> np.uint64(40)          # 24 bits
    return (u.astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)


def randint(n: int, lo: int, hi: int, seed: int, stream: int = 0) -> np.ndarray:
    """int64 uniform in [lo, hi)."""
    u = u64_stream(n, seed, stream) >> np.uint64(11)
    return (lo + (u % np.uint64(hi - lo)).astype(np.int64)).astype(np.int64)


def _name_hash(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def make_tensor(name: str, shape, seed: int, scale: float | None = None) -> np.ndarray:
    """Deterministic float32 tensor for a state-dict entry.

    LayerNorm weights ~ 1 + 0.1 u, biases ~ 0.02 u, embeddings ~ 0.5 u,
    matrices ~ u * MATRIX_GAIN * sqrt(3 / fan_in), unless ``scale`` is given.

    MATRIX_GAIN = 0.4 gives std 0.4/sqrt(fan_in) (0.010 at 1536, 0.018 at 512: the
    N(0, 0.02)-style init of SURVEY.md §8d).  Unit gain (1.0) makes every residual branch
    as large as the stream; such a random network amplifies a single bf16 ulp by ~10x per
    layer (measured), which no trainable 49-layer model does and which would turn an
    end-to-end parity test into a test of chaos rather than of the kernels.
    """
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform(n, seed, _name_hash(name)).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
    if scale is not None:
        return (u * np.float32(scale)).astype(np.float32)
    if leaf == "m":  # ALiBi slope buffer: never random, caller overwrites
        return np.zeros(shape, np.float32)
    is_norm = parent.startswith("norm") or (parent.isdigit() and len(shape) == 1 and leaf == "weight")
    if leaf == "weight" and len(shape) == 1 and is_norm:
        return (np.float32(1.0) + np.float32(0.1) * u).astype(np.float32)
    if leaf == "bias" or len(shape) == 1:
        return (np.float32(0.02) * u).astype(np.float32)
    if "embedding" in parent or "registry" in parent or "token_embedding" in name:
        return (np.float32(0.5) * u).astype(np.float32)
    fan_in = shape[-1]
    return (u * np.float32(MATRIX_GAIN * np.sqrt(3.0 / fan_in))).astype(np.float32)


def fill_state_dict(module: torch.nn.Module, seed: int) -> None:
    """Overwrite every parameter/persistent buffer of ``module`` in place with
    deterministic values (name-keyed, so two different implementations that
    share the reference's state-dict keys get identical weights)."""
    sd = module.state_dict()
    new = {}
    for name, t in sd.items():
        if name.endswith(".m") or name == "m":          # ALiBi slopes stay as built
            new[name] = t
            continue
        if not torch.is_floating_point(t):
            new[name] = t
            continue
        new[name] = torch.from_numpy(make_tensor(name, t.shape, seed)).to(t.dtype)
    module.load_state_dict(new)


def make_gene(seed: int, n_cre: int, n_chunks: int, tissues, token_length: int = 200,
              cre_len_range=(70, 126), last_chunk_len: int | None = None):
    """One gene's sample in the reference's per-sample layout
    (datasets/vcfdataset.py:305-336): CRE windows with 70-125 valid tokens
    (150-350 bp + 2x50 flank at ~3.6 bp/token), gene chunks full
    except a ragged last one, 9-class CRE labels."""
    L = token_length
    hi = min(cre_len_range[1], L + 1)
    lens = randint(n_cre, min(cre_len_range[0], max(1, hi // 2)), hi, seed, 1)
    ids = randint(n_cre * L, FIRST_REAL_TOKEN, VOCAB_SIZE, seed, 2).reshape(n_cre, L)
    pos = np.arange(L)[None, :]
    cre_mask = pos >= lens[:, None]                     # True = pad
    ids = np.where(cre_mask, PAD_ID, ids)
    g_lens = np.full(n_chunks, L, dtype=np.int64)
    if last_chunk_len is None:
        last_chunk_len = int(randint(1, 1, L + 1, seed, 3)[0])
    g_lens[-1] = last_chunk_len
    g_ids = randint(n_chunks * L, FIRST_REAL_TOKEN, VOCAB_SIZE, seed, 4).reshape(n_chunks, L)
    g_mask = pos >= g_lens[:, None]
    g_ids = np.where(g_mask, PAD_ID, g_ids)
    labels = randint(n_cre, 0, 9, seed, 5)
    return {
        "cre_sequences": torch.from_numpy(ids).long().unsqueeze(1),
        "cre_attention_masks": torch.from_numpy(cre_mask).unsqueeze(1),
        "tissue_context": torch.tensor(list(tissues), dtype=torch.long),
        "ref_cre_labels": torch.from_numpy(labels).long(),
        "strand": torch.tensor([0], dtype=torch.long),
        "gene_embeddings": torch.from_numpy(g_ids).long().unsqueeze(1),
        "gene_attention_masks": torch.from_numpy(g_mask).unsqueeze(1),
    }


def collate(genes: list[dict]) -> dict:
    """Same dict as the reference's collate_fn_batching (vcfdataset.py:18-63)."""
    return {
        "cre_sequences": [g["cre_sequences"] for g in genes],
        "cre_attention_masks": [g["cre_attention_masks"] for g in genes],
        "tissue_context": [g["tissue_context"] for g in genes],
        "cre_labels": [torch.zeros_like(g["ref_cre_labels"]) for g in genes],
        "ref_cre_labels": [g["ref_cre_labels"] for g in genes],
        "strand_val": torch.cat([g["strand"].unsqueeze(0) for g in genes], dim=0),
        "gene_embeddings": [g["gene_embeddings"] for g in genes],
        "gene_attention_masks": [g["gene_attention_masks"] for g in genes],
    }


def make_batch(seed: int, n_cres, n_chunks, tissues_per_gene, token_length: int = 200, **kw) -> dict:
    genes = [
        make_gene(seed * 1000003 + i, int(n), int(c), t, token_length, **kw)
        for i, (n, c, t) in enumerate(zip(n_cres, n_chunks, tissues_per_gene))
    ]
    return collate(genes)


def make_vep_batch(seed: int, n_cre: int, n_chunks: int, tissues, token_length: int = 200, cre_index: int = 2,
                   gene_index=(1, 1, 2), **kw) -> dict:
    """ref / het / hom batch in the layout the reference's VEPDataset.load_data returns
    (datasets/vepdataset.py:765-799): three samples of one gene that differ in the CRE window `cre_index` and in the
    gene chunks `gene_index[g]` (a few token ids changed, as an IUPAC substitution does after re-tokenisation), plus
    `cre_token_position` / `gene_token_position` float tensors [3, 1] (create_batch, :335-336)."""
    base = make_gene(seed * 1000003, n_cre, n_chunks, tissues, token_length, **kw)
    genes = []
    for g in range(3):
        s = {k: v.clone() for k, v in base.items()}
        if g > 0:
            cre_valid = int((~s["cre_attention_masks"][cre_index, 0]).sum())
            for j, pos in enumerate(randint(2, 0, max(1, cre_valid), seed, 20 + g)):
                s["cre_sequences"][cre_index, 0, int(pos)] = FIRST_REAL_TOKEN + (7 * g + 3 * j) % 14
            gi = gene_index[g]
            g_valid = int((~s["gene_attention_masks"][gi, 0]).sum())
            for j, pos in enumerate(randint(2, 0, max(1, g_valid), seed, 30 + g)):
                s["gene_embeddings"][gi, 0, int(pos)] = FIRST_REAL_TOKEN + (5 * g + j) % 14
        genes.append(s)
    b = collate(genes)
    return {"cre_sequences": b["cre_sequences"], "cre_attention_masks": b["cre_attention_masks"],
            "tissue_context": b["tissue_context"], "ref_labels": b["ref_cre_labels"], "strand": b["strand_val"],
            "gene_embeddings": b["gene_embeddings"], "gene_attention_masks": b["gene_attention_masks"],
            "cre_token_position": torch.tensor([[float(cre_index)]] * 3),
            "gene_token_position": torch.tensor([[float(i)] for i in gene_index]),
            "variant_type": "SNP"}


def cfg3_gene_sizes(n_genes: int, seed: int = 20251205):
    """SURVEY §8d cfg 3: N ~ lognormal(median 600, sigma 0.6) clipped [40, 2048],
    C ~ U{20..200}."""
    u1 = (uniform(n_genes, seed, 11).astype(np.float64) + 1.0) / 2.0
    u2 = (uniform(n_genes, seed, 12).astype(np.float64) + 1.0) / 2.0
    z = np.sqrt(-2.0 * np.log(np.clip(u1, 1e-12, 1.0))) * np.cos(2 * np.pi * u2)
    n = np.clip(np.round(600.0 * np.exp(0.6 * z)), 40, 2048).astype(np.int64)
    c = randint(n_genes, 20, 201, seed, 13)
    return n, c


def calibrate_sequence_sensitivity(model, cre_std: float = 8.0, gene_std: float = 1.0, seed: int = 999):
    """Make a random-weight model SEQUENCE-SENSITIVE (tests only; BASELINE configs[3], the variant-effect delta).

    With plain random weights every cCRE window / gene chunk embedding is 99 % one common vector (the mean over ~100
    tokens of positional encoding + token embedding; measured: |common| 12.5, window-specific part 1.5), attention is
    uniform, and the expression hardly depends on the DNA at all: replacing EVERY window of a gene moves it by 3e-4 --
    the size of the bf16 rounding noise -- so no implementation, the reference's bf16-mixed path included, could resolve a
    5-window variant.  A trained tokenizer does not behave like that.  This stands in for the training: `cre_map` /
    `gene_map` (plain Linear layers, so still an ordinary state dict) are re-centred on the population mean of the pooled
    embeddings and rescaled so that the rows entering the modulator have element standard deviation `cre_std` /
    `gene_std`:  W' = S W,  b' = b - S W mu.  Large un-normalised CRE rows are what make the gene -> CRE cross attention
    selective (its K / V projections read the raw rows).  Returns the original (weight, bias) tensors so that a caller can
    restore them.  The statistics are taken on the model's own device from a seeded calibration batch."""
    dev = model.gene_map.weight.device
    cal = make_batch(seed, [256], [64], [TISSUES_54[:1]], 200)
    saved = {}
    with torch.no_grad():
        for name, tok, ids, mask, tgt in (("cre_map", model.cre_tokenizer, cal["cre_sequences"][0], cal["cre_attention_masks"][0], cre_std),
                                          ("gene_map", model.gene_tokenizer if model.gene_tokenizer is not None else model.cre_tokenizer,
                                           cal["gene_embeddings"][0], cal["gene_attention_masks"][0], gene_std)):
            lin = getattr(model, name, None)
            if lin is None:
                continue
            x = tok(ids, mask, None, only_embed=True)[:, 0].float().to(dev)
            W, b = lin.weight.detach().clone(), lin.bias.detach().clone()
            saved[name] = (W.clone(), b.clone())
            mu = x.mean(0)
            S = tgt / float(((x - mu) @ W.t()).std())
            lin.weight.copy_(W * S)
            lin.bias.copy_(b - S * (W @ mu))
    return saved


def restore_maps(model, saved: dict) -> None:
    with torch.no_grad():
        for name, (W, b) in saved.items():
            getattr(model, name).weight.copy_(W)
            getattr(model, name).bias.copy_(b)


def trained_like_(model: torch.nn.Module, seed: int, outlier_gain: float = 25.0) -> None:
    """In place: N(0, 0.02)-style seeded weights -> the statistics TRAINED transformers show (the real 14 GB checkpoint is
    unavailable offline): LayerNorm gains spread log-normally with a few large entries, non-zero LayerNorm biases, outlier
    CHANNELS in the residual streams (embedding columns and the rows of the layers' output projections scaled up, so that a
    handful of channels carry activations tens of times the rest -- 'massive activations'), heavy-tailed projection weights,
    registry / context embeddings with a common mean.  Random numbers come from a CPU generator whatever device the model
    is on (tests/test_trained_like_gpu.py on a CPU copy, bench.py's `trained_like` pass on the resident model)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not torch.is_floating_point(p):
                continue
            dev = p.device
            D = p.shape[0]
            if ".norm" in name or name.endswith("LayerNorm.weight") or "layer_norm" in name:
                if name.endswith("weight") and p.dim() == 1:
                    gain = torch.exp(0.4 * torch.randn(D, generator=g))
                    gain[torch.randperm(D, generator=g)[:3]] *= 6.0
                    p.copy_(gain)
                elif name.endswith("bias") and p.dim() == 1:
                    p.copy_(0.2 * torch.randn(D, generator=g))
                continue
            if p.dim() == 2 and (name.endswith("out_proj.weight") or name.endswith("linear_geglu_2.weight")):
                # rows that write the residual stream: four outlier channels (the same ones in every layer of a width)
                ch = (torch.tensor([7, 101, D // 2 + 3, D - 5]) % D).to(dev)
                p[ch] *= outlier_gain
            if p.dim() == 2 and name.endswith("token_embedding.weight"):
                ch = torch.tensor([7, 101, p.shape[1] // 2 + 3, p.shape[1] - 5]).to(dev)
                p[:, ch] *= outlier_gain
            if p.dim() == 2 and ("registry_tokens" in name or "context_embedding" in name or "ctx" in name.lower()):
                p.add_((2.0 * float(p.std()) * torch.randn(1, p.shape[1], generator=g)).to(dev))      # a common mean direction
            if p.dim() == 2 and p.numel() > 4096:                                           # heavy tails: 0.1 % of the entries x 8
                mask = (torch.rand(p.shape, generator=g) < 1e-3).to(dev)
                p[mask] *= 8.0
