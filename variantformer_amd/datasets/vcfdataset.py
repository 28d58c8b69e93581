"""Batch format of the hot path (reference datasets/vcfdataset.py:18-63), the per-gene sample builder
`VCFDataset` (:66-394) on the in-process consensus + C++ BPE, and a synthetic per-gene dataset with the same sample
tuple for tests / bench where the genome artifacts are unavailable."""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch
import yaml
from torch.utils.data import Dataset

from ..utils import constants, synthetic
from ..utils.constants import MAP_REF_CRE_TO_IDX, SPECIAL_TOKENS
from ..utils.data_process import ExtractSeqFromBed
from ..utils.functions import reverse_complement
from ..utils.seq import BPEEncoder

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collate_fn_batching(batch):
    """Tuples (X, mask, tissues, labels, ref_labels, strand, gene_chunks, gene_masks) -> dict of lists, as the
    reference's collate_fn_batching."""
    keys = ["cre_sequences", "cre_attention_masks", "tissue_context", "cre_labels", "ref_cre_labels",
            "gene_embeddings", "gene_attention_masks"]
    out = {k: [] for k in keys}
    strands = []
    for X, mask, ctx, label, ref_label, strand, emb, emb_att in batch:
        out["cre_sequences"].append(X)
        out["cre_attention_masks"].append(mask)
        out["tissue_context"].append(ctx)
        out["cre_labels"].append(label)
        out["ref_cre_labels"].append(ref_label)
        out["gene_embeddings"].append(emb)
        out["gene_attention_masks"].append(emb_att)
        strands.append(strand.unsqueeze(0))
    out["strand_val"] = torch.cat(strands, dim=0)
    return out


class SyntheticGeneDataset(Dataset):
    """query_df rows (gene_id, tissues as comma separated names) -> seeded synthetic samples
    (SURVEY.md §8d geometry).  Mirrors VCFDataset's query filtering (:123-169) and sample tuple (:305-336)."""

    def __init__(self, query_df: pd.DataFrame, tissue_vocab: dict, n_cre=300, n_chunks=200, token_length=200,
                 seed=20251205, **_):
        rows = []
        for _, row in query_df.iterrows():
            names = [t for t in row["tissues"].split(",") if t in tissue_vocab]
            if names:
                rows.append({"gene_id": row["gene_id"], "tissues": [tissue_vocab[t] for t in names], "tissue_names": names})
        if not rows:
            raise ValueError("No genes found in the query df with at least one known tissue")
        self.query_df = pd.DataFrame(rows)
        self.n_cre, self.n_chunks, self.L, self.seed = n_cre, n_chunks, token_length, seed

    def __len__(self):
        return len(self.query_df)

    def __getitem__(self, idx):
        r = self.query_df.iloc[idx]
        g = synthetic.make_gene(self.seed + idx, self.n_cre, self.n_chunks, r["tissues"], self.L)
        return (g["cre_sequences"], g["cre_attention_masks"], g["tissue_context"],
                torch.zeros_like(g["ref_cre_labels"]), g["ref_cre_labels"], g["strand"],
                g["gene_embeddings"], g["gene_attention_masks"])


class VCFDataset(Dataset):
    """(gene, tissues) queries of one sample's VCF -> the hot path's sample tuple (reference :66-394).

    Per gene: CRE list from the per-gene manifest CSV, IUPAC consensus of every CRE window (+- cre_neighbour_hood)
    and of the gene body, BPE, pad / truncate each window to max_length, gene body in <= max_chunks chunks.
    Same constructor arguments as the reference plus `indel_policy` / `sample` for the in-process consensus
    (utils/data_process.py here); `gene_cre_manifest` is anything with `get_file_path(gene_id)`."""

    def __init__(self, max_length: int, max_chunks: int, cre_neighbour_hood: int, gencode_v24, gene_cre_manifest,
                 gene_upstream_neighbour_hood: int, gene_downstream_neighbour_hood: int, query_df: pd.DataFrame,
                 fasta_path: str, vcf_path: str = None, indel_policy: str = "bcftools", sample: str = None):
        self.bpe = BPEEncoder()
        self.bpe.load_vocabulary()
        self.vocab = self.bpe.vocab
        self.pad_token_id = self.vocab.get(SPECIAL_TOKENS["pad_token"])
        self.max_chunks, self.max_length = max_chunks, max_length
        self.cre_neighbour_hood = cre_neighbour_hood
        self.gene_upstream_neighbour_hood = gene_upstream_neighbour_hood
        self.gene_downstream_neighbour_hood = gene_downstream_neighbour_hood
        self.query_df = query_df
        self.gene_cre_manifest = gene_cre_manifest
        self.fasta_path = fasta_path
        self.ref_cre_to_idx = MAP_REF_CRE_TO_IDX
        self.cre_to_idx = constants.MAP_CRE_TO_IDX
        self.gencode_v24 = gencode_v24 if isinstance(gencode_v24, pd.DataFrame) else pd.read_csv(gencode_v24)
        with open(os.path.join(_PKG, "vocabs", "tissue_vocab.yaml")) as f:
            self.tissue_vocab = yaml.safe_load(f)
        self.vcf_path = vcf_path
        self.indel_policy, self.sample = indel_policy, sample
        self._check_filter_query_df()

    def _check_filter_query_df(self):
        """Keep genes known to gencode with >= 1 known tissue; tissue names -> ids (reference :123-169)."""
        assert self.query_df is not None, "Query dataframe is not provided"
        assert "gene_id" in self.query_df.columns, "Query dataframe must contain gene_id column"
        assert "tissues" in self.query_df.columns, "Query dataframe must contain tissues column"
        before = len(self.query_df)
        known_genes = set(self.gencode_v24["gene_id"].values)
        rows = []
        for _, row in self.query_df.iterrows():
            if row["gene_id"] not in known_genes:
                print(f"Gene {row['gene_id']} not found in the training set so skipping it")
                continue
            names = []
            for t in row["tissues"].split(","):
                if t in self.tissue_vocab:
                    names.append(t)
                else:
                    print(f"Tissue {t} not found in the tissue vocab so skipping it")
            if not names:
                print(f"No tissues found for gene {row['gene_id']}")
                continue
            keep = {"gene_id": row["gene_id"], "tissues": [self.tissue_vocab[t] for t in names], "tissue_names": names}
            if "vcf_path" in self.query_df.columns:
                keep["vcf_path"] = row["vcf_path"]
            rows.append(keep)
        if not rows:
            raise ValueError("No genes found in the query df that are present in the gencode v24 and have at least "
                             "one tissue in the training set of VariantFormer")
        self.query_df = pd.DataFrame(rows)
        print(f"Filtered query df to {len(self.query_df)} genes reducing from {before}")
        return True

    def __len__(self):
        return len(self.query_df)

    def __getitem__(self, idx):
        return self._load_file(idx)

    def _get_gene_info(self, gene_id: str) -> dict:
        """First gencode row of the gene as a dict (reference :171-176 filters the frame per call: 0.8 ms of boolean-mask work
        per gene on a 60 k-row table; the rows are indexed once here)."""
        idx = self.__dict__.get("_gene_rows")
        if idx is None:
            ids = self.gencode_v24["gene_id"].to_numpy()
            idx = {}
            for i in range(len(ids) - 1, -1, -1):           # the FIRST row of a gene id wins, like .iloc[0] of the filtered frame
                idx[ids[i]] = i
            self._gene_rows = idx
        return self.gencode_v24.iloc[idx[gene_id]].to_dict()

    def _adjust_length(self, token_ids):
        n = len(token_ids)
        if n < self.max_length:
            return token_ids + [self.pad_token_id] * (self.max_length - n), [0] * n + [1] * (self.max_length - n)
        return token_ids[: self.max_length], [0] * self.max_length

    def _extractor(self, neighbour_hood, upstream=None):
        return ExtractSeqFromBed(neighbour_hood=neighbour_hood, ref_fasta=self.fasta_path,
                                 upstream_neighbour_hood=upstream, indel_policy=self.indel_policy, sample=self.sample)

    def _get_cres(self, gene_id: str, gene_info: dict, vcf_path: str):
        """[N,1,L] ids, [N,1,L] pad mask, ref-cCRE class ids, cCRE labels (all "Low-DNase") -- reference :219-283."""
        table = pd.read_csv(self.gene_cre_manifest.get_file_path(gene_id), usecols=["chromosome", "start_cre", "end_cre", "cre_name"])
        fast = self._get_cres_batched(table, gene_info, vcf_path)
        if fast is not None:
            return fast
        bed = table[["chromosome", "start_cre", "end_cre", "cre_name"]].rename(
            columns={"chromosome": "chrom", "start_cre": "start", "end_cre": "end", "cre_name": "cCRE"})
        cres = self._extractor(self.cre_neighbour_hood).process_subject(vcf_file=vcf_path, bed_regions=bed)
        minus = gene_info["strand"] != "+"
        if gene_info["strand"] == "-":
            cres = cres.iloc[::-1]
        n, L = len(cres), self.max_length
        X = np.full((n, 1, L), self.pad_token_id, dtype=np.int64)
        masks = np.ones((n, 1, L), dtype=bool)
        ref_labels = np.empty(n, dtype=np.int64)
        for k, (seq, name) in enumerate(zip(cres["sequence"], cres["cCRE"])):
            ids = self.bpe.encode_forward(reverse_complement(seq) if minus else seq)[:L]
            X[k, 0, : len(ids)] = ids
            masks[k, 0, : len(ids)] = False
            ref_labels[k] = self.ref_cre_to_idx[name]
        labels = np.full(n, self.cre_to_idx["Low-DNase"], dtype=np.int64)
        return torch.from_numpy(X), torch.from_numpy(masks), torch.from_numpy(ref_labels), torch.from_numpy(labels)

    def _get_cres_batched(self, table: pd.DataFrame, gene_info: dict, vcf_path: str):
        """Same result as the per-window path above from ONE native call (vf_build_windows: consensus -> reverse
        complement -> BPE -> pad, for all windows of the gene; ~5x less host time per gene).  Returns None when the
        per-window path has to take over (windows on several chromosomes, a chromosome the genome lacks, a span too
        long to hold in memory at once)."""
        import ctypes as C

        from .. import _lib
        from ..utils.data_process import ConsensusError, _INDEL_POLICIES, open_fasta, open_vcf
        # (the manifest's own columns as arrays: building the renamed 4-column frame first cost 2 ms of pandas per gene)
        chroms = table["chromosome"].to_numpy()
        if len(chroms) == 0 or (chroms != chroms[0]).any():
            return None
        chrom = str(chroms[0])
        bed = {"start": table["start_cre"].to_numpy(), "end": table["end_cre"].to_numpy(), "cCRE": table["cre_name"].to_numpy()}
        fa = open_fasta(self.fasta_path)
        if chrom not in fa.index:
            return None
        nh = self.cre_neighbour_hood
        order = np.argsort(bed["start"], kind="stable")          # process_subject sorts by start
        starts = np.maximum(0, bed["start"][order].astype(np.int64) - nh)
        ends = np.minimum(bed["end"][order].astype(np.int64) + nh, fa.length(chrom))
        names = bed["cCRE"][order]
        keep = ends > starts                                                 # empty windows yield no row there either
        starts, ends, names = np.ascontiguousarray(starts[keep]), np.ascontiguousarray(ends[keep]), names[keep]
        if len(starts) == 0 or int(ends.max() - starts.min()) > 64_000_000:
            return None
        span0 = int(starts.min())
        ref = fa.fetch(chrom, span0, int(ends.max())).encode("ascii")
        if self.bpe._h is None:
            self.bpe.load_vocabulary()
        policy = {1: 1, 3: 2}.get(_INDEL_POLICIES[self.indel_policy], 0)
        if _INDEL_POLICIES[self.indel_policy] == 2:                          # "skip": substitutions only
            snp_only, policy = 1, 0
        else:
            snp_only = 0
        n, L = len(starts), self.max_length
        X = np.empty((n, 1, L), dtype=np.int64)
        masks = np.empty((n, 1, L), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        lib = _lib.load()
        minus = gene_info["strand"] != "+"
        vcf_h = open_vcf(vcf_path, self.sample)._h if vcf_path else None
        rc = lib.vf_build_windows(vcf_h, self.bpe._h, chrom.encode(), span0, ref, len(ref), n, starts.ctypes.data,
                                  ends.ctypes.data, snp_only, policy, int(minus), L, int(self.pad_token_id),
                                  X.ctypes.data, masks.ctypes.data, status.ctypes.data)
        if rc != n:
            raise _lib.VFError("vf_build_windows failed")
        for i in np.nonzero(status == 1)[0]:                                 # same note as the per-window path
            print(f"{chrom}:{int(starts[i]) + 1}-{int(ends[i])}")
            print("\nError building the consensus: REF allele does not match the reference genome")
            print("Falling back to ref genome")
        if (status < 0).any():
            i = int(np.nonzero(status < 0)[0][0])
            raise ConsensusError(int(status[i]), f"{chrom}:{int(starts[i]) + 1}-{int(ends[i])}")
        ref_labels = np.array([self.ref_cre_to_idx[nm] for nm in names], dtype=np.int64)
        if gene_info["strand"] == "-":
            X, masks, ref_labels = X[::-1].copy(), masks[::-1].copy(), ref_labels[::-1].copy()
        labels = np.full(n, self.cre_to_idx["Low-DNase"], dtype=np.int64)
        # (vf_build_windows writes 0 / 1 bytes: the bool view is free; torch's .bool() is a converting copy that starts an OpenMP
        # team per call -- 50 of a gene's 120 ms on an 8-core host, scripts/sample_builder_bench.py)
        return (torch.from_numpy(X), torch.from_numpy(masks.view(np.bool_)), torch.from_numpy(ref_labels),
                torch.from_numpy(labels))

    def _get_gene(self, gene_id: str, gene_info: dict, vcf_path: str):
        seq = self._extractor(self.gene_downstream_neighbour_hood, self.gene_upstream_neighbour_hood).process_gene(
            gene_info, vcf_path)
        assert len(seq) > 1000, f"Mutated sequence is less than 1000bp for gene {gene_id}"
        # only max_chunks x max_length tokens are kept (chunkify_data): the exact prefix of the full encoding (vf_bpe_encode_prefix)
        # -- a 301 kb gene body is ~83 k tokens of which 40 k survive; round 6: 14 -> 8 ms of a gene's 39 ms on the host
        ids = self.bpe.encode_forward(seq if gene_info["strand"] == "+" else reverse_complement(seq),
                                      max_tokens=self.max_chunks * self.max_length)
        return self.chunkify_data(torch.from_numpy(ids.astype(np.int64)).unsqueeze(0))

    def _load_file(self, idx: int):
        row = self.query_df.iloc[idx]
        vcf_path = row["vcf_path"] if "vcf_path" in self.query_df.columns else self.vcf_path
        gene_info = self._get_gene_info(row["gene_id"])
        assert gene_info["chromosome"] in ["chr" + str(i) for i in range(1, 23)], \
            f"Chromosome {gene_info['chromosome']} is not a valid chromosome. Sex chromosomes are not supported"
        X, mask, ref_labels, labels = self._get_cres(row["gene_id"], gene_info, vcf_path)
        chunks, chunk_masks = self._get_gene(row["gene_id"], gene_info, vcf_path)
        return (X, mask, torch.tensor(row["tissues"], dtype=torch.long), labels, ref_labels,
                torch.tensor([0] if gene_info["strand"] == "+" else [1], dtype=torch.long), chunks, chunk_masks)

    def chunkify_data(self, x):
        """[1, n_tokens] -> ([n_chunks, 1, max_length] ids, pad mask); at most max_chunks chunks (reference :338-394)."""
        n = x.size(1)
        n_chunks = min(self.max_chunks, -(-n // self.max_length))
        flat = x[0, : n_chunks * self.max_length].long().numpy()
        ids = np.full(n_chunks * self.max_length, self.pad_token_id, dtype=np.int64)
        mask = np.ones(n_chunks * self.max_length, dtype=bool)
        ids[: flat.size] = flat
        mask[: flat.size] = False
        shape = (n_chunks, 1, self.max_length)
        return torch.from_numpy(ids).view(shape), torch.from_numpy(mask).view(shape)
