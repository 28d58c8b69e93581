"""Multi-GPU vcf2exp launcher: one process per GPU, genes sharded by LPT, results gathered in query order on every rank.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        scripts/vcf2exp_dist.py --vcf donor.vcf.gz --genes genes.csv --tissues whole_blood,liver --out pred.parquet

torch.distributed.run starts the ranks before anything touches the GPU; each rank binds cuda:LOCAL_RANK, builds a full
weight replica (ModelManager), and calls VCFProcessor.predict_distributed.  Rank 0 writes the table.
`--gene-cre-manifest DIR` = directory of per-gene cCRE CSVs named <gene_id>.csv (what utils/assets resolves from S3 in
the reference)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class DirManifest:
    def __init__(self, root):
        self.root = root

    def get_file_path(self, gene_id):
        return os.path.join(self.root, f"{gene_id}.csv")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vcf", required=True)
    ap.add_argument("--genes", required=True, help="CSV with a gene_id column")
    ap.add_argument("--tissues", required=True, help="comma-separated tissue names of vocabs/tissue_vocab.yaml")
    ap.add_argument("--gene-cre-manifest", required=True)
    ap.add_argument("--model-class", default="v4_pcg")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import pandas as pd
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from variantformer_amd.processors.vcfprocessor import VCFProcessor
    proc = VCFProcessor(a.model_class, gene_cre_manifest=DirManifest(a.gene_cre_manifest), indel_policy="bcftools")
    genes = pd.read_csv(a.genes)
    query = pd.DataFrame({"gene_id": genes["gene_id"], "tissues": [a.tissues] * len(genes)})
    dataset, _ = proc.create_data(a.vcf, query)
    model, ckpt, trainer = proc.load_model()
    df = proc.predict_distributed(model, ckpt, trainer, dataset)
    if not dist.is_initialized() or dist.get_rank() == 0:
        df.to_parquet(a.out)
        print(f"wrote {len(df)} rows to {a.out}; this rank was busy {proc.last_busy_seconds:.1f} s")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
