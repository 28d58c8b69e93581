"""VCFProcessor: the vcf2exp API surface (reference processors/vcfprocessor.py:23-277) over the HIP model.

Kept: constructor(model_class), create_vcf_from_variant, get_tissues, get_genes, create_data, load_model, predict,
format_output with the same return types.  `create_data` builds the in-tree `VCFDataset` (FASTA + VCF + per-gene
cCRE manifest -> IUPAC consensus -> BPE, all in process: no samtools / bcftools subprocesses); the per-gene cCRE
manifest lookup (S3 + duckdb in the reference, utils/assets.py:307-368) is duck-typed: pass anything with
`get_file_path(gene_id)` as `gene_cre_manifest`.  A `dataset_factory` may replace the dataset (e.g.
datasets.SyntheticGeneDataset where no genome is available)."""
from __future__ import annotations

import gzip
import os
from pathlib import Path

import pandas as pd
import torch
import yaml
from torch.utils.data import DataLoader

from ..datasets.vcfdataset import VCFDataset, collate_fn_batching
from ..utils.config import Config, load_yaml
from .model_manager import ModelManager
from .trainer import Trainer


class VCFProcessor:
    def __init__(self, model_class: str = "v4_pcg", config_dir: str | None = None, require_gpu: bool = True,
                 gene_cre_manifest=None, indel_policy: str = "bcftools"):
        base_dir = Path(__file__).parent.parent.resolve()
        self.config_location = Path(config_dir) if config_dir else base_dir / "configs"
        self.model_config = load_yaml(str(self.config_location / "vf_model.yaml"))[model_class]
        with open(base_dir / "vocabs" / "tissue_vocab.yaml") as f:
            self.tissue_vocab = yaml.safe_load(f)
        self.vcf_loader_config = load_yaml(str(self.config_location / "vcfloader.yaml"))
        root = self.config_location.parent
        for node, key in ((self.model_config.dataset, "gencode_v24"), (self.model_config.model, "checkpoint_path"),
                          (self.model_config.model.cre_tokenizer, "path"), (self.model_config.model.gene_tokenizer, "path")):
            if not os.path.isabs(node[key]):
                node[key] = str(root / node[key])
        if not os.path.isabs(self.vcf_loader_config.fasta_path):
            self.vcf_loader_config.fasta_path = str(root / self.vcf_loader_config.fasta_path)
        self.gene_cre_manifest = gene_cre_manifest
        self.indel_policy = indel_policy        # insertions / deletions in the consensus: see utils/data_process.py
        if require_gpu:
            assert torch.cuda.is_available(), "GPU is not available"          # reference :60
        self.accelerator = "gpu"

    def create_vcf_from_variant(self, variant_df: pd.DataFrame, output_path: str, vcf_path: str = None):
        """Write (or merge into a copy of `vcf_path`) a single-sample VCF from rows chrom, pos, ref, alt, GT after
        checking every REF against the genome (reference :62-219, which shells out to samtools / bgzip / tabix /
        bcftools sort + concat -a -D).  Output is gzip text sorted by contig order of first appearance and position;
        a record already present (same chrom, pos, ref, alt) is kept once, the existing file's line first.  No
        tabix index is written: the in-process reader does not need one."""
        from ..utils.data_process import open_fasta
        for col in ("chrom", "pos", "ref", "alt", "GT"):
            assert col in variant_df.columns, f"{col} column is required"
        if len(variant_df) == 0:
            raise ValueError("variant_df is empty")
        fasta = open_fasta(self.vcf_loader_config.fasta_path)
        for _, row in variant_df.iterrows():
            pos, ref = int(row["pos"]), row["ref"]
            try:
                found = fasta.fetch(row["chrom"], pos - 1, pos - 1 + len(ref)).upper()
            except KeyError as e:
                raise ValueError(f"Failed to extract reference at {row['chrom']}:{pos}-{pos + len(ref) - 1}: {e}")
            if found != ref.upper():
                raise ValueError(f"Reference mismatch at {row['chrom']}:{pos}: expected '{ref}' but found '{found}' "
                                 f"in reference genome")
        sample, header, body = "SAMPLE", None, []
        if vcf_path is not None:
            opener = gzip.open if open(vcf_path, "rb").read(2) == b"\x1f\x8b" else open
            with opener(vcf_path, "rt") as f:
                lines = f.read().splitlines()
            header = [ln for ln in lines if ln.startswith("#")]
            body = [ln for ln in lines if ln and not ln.startswith("#")]
            names = header[-1].split("\t")[9:]
            assert len(names) == 1, "merging needs a single-sample VCF"
            sample = names[0]
        if header is None:
            header = ["##fileformat=VCFv4.2", f"##reference={self.vcf_loader_config.fasta_path}"]
            header += [f"##contig=<ID={c}>" for c in sorted(variant_df["chrom"].unique())]
            header += ['##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
                       f"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{sample}"]
        new = [f"{r['chrom']}\t{int(r['pos'])}\t.\t{r['ref']}\t{r['alt']}\t.\tPASS\t.\tGT\t{r['GT']}"
               for _, r in variant_df.sort_values(by=["chrom", "pos"]).iterrows()]
        order, seen, records = {}, set(), []
        for ln in body + new:
            f = ln.split("\t", 5)
            key = (f[0], int(f[1]), f[3], f[4])
            if key in seen:
                continue
            seen.add(key)
            order.setdefault(f[0], len(order))
            records.append((order[f[0]], int(f[1]), len(records), ln))
        records.sort()
        final_path = output_path if output_path.endswith(".vcf.gz") else f"{output_path}.vcf.gz"
        Path(final_path).parent.mkdir(parents=True, exist_ok=True)
        with gzip.open(final_path, "wt") as f:
            f.write("\n".join(header + [r[3] for r in records]) + "\n")
        return final_path

    def get_tissues(self):
        return self.tissue_vocab.keys()

    def get_genes(self):
        return pd.read_csv(self.model_config.dataset.gencode_v24)

    def create_data(self, vcf_path, query_df: pd.DataFrame, dataset_factory=None, **kwargs):
        cfg = Config(dict(self.vcf_loader_config.dataloader))
        cfg.update(kwargs)
        if dataset_factory is None:
            if self.gene_cre_manifest is None:
                raise ValueError("VCFProcessor needs gene_cre_manifest (get_file_path(gene_id) -> per-gene cCRE CSV) to "
                                 "build samples from a VCF; the reference resolves it from S3 (utils/assets.py:307-368)")
            d = self.model_config.dataset
            dataset = VCFDataset(max_length=d.max_length, max_chunks=d.max_chunks, cre_neighbour_hood=d.cre_neighbour_hood,
                                 gencode_v24=d.gencode_v24, gene_cre_manifest=self.gene_cre_manifest,
                                 gene_upstream_neighbour_hood=d.gene_upstream_neighbour_hood,
                                 gene_downstream_neighbour_hood=d.gene_downstream_neighbour_hood, query_df=query_df,
                                 fasta_path=self.vcf_loader_config.fasta_path, vcf_path=vcf_path,
                                 indel_policy=self.indel_policy)
        else:
            dataset = dataset_factory(vcf_path=vcf_path, query_df=query_df, tissue_vocab=self.tissue_vocab,
                                      dataset_config=self.model_config.dataset)
        if cfg.get("num_workers", 0) == 0:
            cfg.pop("prefetch_factor", None)
        return dataset, DataLoader(dataset, collate_fn=collate_fn_batching, **cfg)

    def load_model(self):
        model, checkpoint_path = ModelManager(self.model_config.model).load_model()
        trainer = Trainer(accelerator=self.accelerator, devices=1, logger=False,
                          precision=self.model_config.model.precision, enable_checkpointing=False)
        return model, checkpoint_path, trainer

    def predict(self, model, checkpoint_path, trainer, dataloader, vcf_dataset):
        predictions = trainer.predict(model, dataloader, ckpt_path=checkpoint_path)
        return self.format_output(vcf_dataset.query_df, predictions)

    def predict_distributed(self, model, checkpoint_path, trainer, vcf_dataset, batch_size: int | None = None, costs=None,
                            **loader_kwargs):
        """Multi-GPU vcf2exp (the reference is single-device, vcfprocessor.py:252-258): call from every rank
        of a torch.distributed job (one process per GPU, launched with `python -m torch.distributed.run` BEFORE anything
        touches the GPU; scripts/vcf2exp_dist.py shows the launcher).  The rows of `vcf_dataset.query_df` are dealt to
        the ranks by LPT on `costs` (default: `vcf_dataset.gene_costs()` when the dataset offers it, else equal),
        every rank runs the single-GPU path over its rows, and one padded all-gather per output (RCCL over xGMI)
        returns the complete frame, in query order, on every rank."""
        from ..dist import predict_sharded
        model.trainer = trainer
        model.eval()
        if costs is None and hasattr(vcf_dataset, "gene_costs"):
            costs = vcf_dataset.gene_costs()
        bs = batch_size or int(self.vcf_loader_config.dataloader.get("batch_size", 8))

        def run(batch, i):
            with torch.no_grad():
                return model.predict_step(batch, i)
        results, busy = predict_sharded(run, vcf_dataset, collate_fn_batching, costs=costs, batch_size=bs,
                                        device=model.device, loader_kwargs=loader_kwargs)
        self.last_busy_seconds = busy
        return self.format_output(vcf_dataset.query_df, [results])

    def format_output(self, df, predictions):
        pred_exp, embd = [], []
        for p in predictions:
            pred_exp.extend(p["pred_gene_exp"])
            embd.extend(p["embeddings"])
        pred_df = pd.DataFrame({"predicted_expression": pred_exp, "embeddings": embd})
        assert len(df) == len(pred_df), "DataFrame and predictions length mismatch"
        df["predicted_expression"] = pred_df["predicted_expression"]
        df["embeddings"] = pred_df["embeddings"]
        return df
