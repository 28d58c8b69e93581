"""End-to-end vcf2exp throughput on a synthetic genome at headline size: FASTA + donor VCF + per-gene cCRE tables ->
VCFDataset (in-process consensus + C++ BPE) in DataLoader workers -> collate -> HIP model (full 1.2B architecture,
random weights) -> expression matrix.  Shows whether the host-side sample builder keeps the GPU fed.
usage: python scripts/vcf2exp_e2e.py [n_genes] [num_workers] [batch_size]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from torch.utils.data import DataLoader
import bench
from tests.test_consensus_cpu import write_fasta, write_vcf, other_base
from variantformer_amd.datasets.vcfdataset import VCFDataset, collate_fn_batching
from variantformer_amd.datasets.vepdataset import LocalManifest
from variantformer_amd.utils.synthetic import TISSUES_54

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
batch_size = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(0)
n = 3_000_000
genome = "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])
root = tempfile.mkdtemp()
fasta = os.path.join(root, "g.fa"); write_fasta(fasta, {"chr1": genome})
pos = np.unique(rng.integers(1, n, n // 700))
write_vcf(os.path.join(root, "d.vcf.gz"), {"chr1": [(int(p), genome[p - 1], [other_base(genome[p - 1])], "0/1" if i % 3 else "1/1")
                                                    for i, p in enumerate(pos)]})
genes, paths = [], {}
import yaml
tissue_names = [k for k, v in yaml.safe_load(open(os.path.join(os.path.dirname(bench.__file__), "variantformer_amd", "vocabs", "tissue_vocab.yaml"))).items()
                if v in TISSUES_54]
for g in range(n_genes):
    start = int(rng.integers(600_000, n - 1_000_000))
    gid = f"G{g}"
    genes.append({"gene_id": gid, "gene_name": gid, "chromosome": "chr1", "start": start, "end": start + 400_000, "strand": "+-"[g % 2]})
    s = np.sort(rng.integers(start - 500_000, start + 500_000, 1024))
    paths[gid] = os.path.join(root, f"{gid}.csv")
    pd.DataFrame({"chromosome": "chr1", "start_cre": s, "end_cre": s + 250, "cre_name": "dELS"}).to_csv(paths[gid], index=False)
query = pd.DataFrame({"gene_id": [g["gene_id"] for g in genes], "tissues": [",".join(tissue_names)] * n_genes})
ds = VCFDataset(200, 200, 50, pd.DataFrame(genes), LocalManifest(paths), 1000, 300000, query, fasta, os.path.join(root, "d.vcf.gz"))
model, hp, kw = bench.build_model(torch.device("cuda:0"))
loader = DataLoader(ds, batch_size=batch_size, num_workers=workers, collate_fn=collate_fn_batching,
                    prefetch_factor=2 if workers else None, persistent_workers=bool(workers))
from variantformer_amd.processors.trainer import Trainer
trainer = Trainer(precision="bf16-mixed")
for rep in range(3):                        # first pass warms the workers' VCF / FASTA caches and the GPU
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = trainer.predict(model, loader)
    done = sum(len(o["pred_gene_exp"]) for o in outs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pass {rep}: {done} genes x {len(tissue_names)} tissues in {dt:.2f} s -> {done / dt:.1f} genes/s end to end ({workers} loader workers)")
