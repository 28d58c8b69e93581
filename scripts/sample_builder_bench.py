"""Host-side sample builder throughput: one headline-size gene (1024 cCRE windows of ~350 bp + a 301 kb gene body)
from a FASTA + a donor VCF with ~1 SNP per 700 bp, through the in-process consensus and the C++ BPE.
The reference does the same with 1025 `samtools | bcftools` subprocess pairs (utils/data_process.py:284-365)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from tests.test_consensus_cpu import write_fasta, write_vcf, other_base
from variantformer_amd.datasets.vcfdataset import VCFDataset
from variantformer_amd.datasets.vepdataset import LocalManifest
from variantformer_amd.utils import data_process as dp

rng = np.random.default_rng(0)
n = 2_000_000
genome = "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])
with tempfile.TemporaryDirectory() as root:
    fasta = os.path.join(root, "g.fa")
    write_fasta(fasta, {"chr1": genome})
    pos = np.unique(rng.integers(1, n, n // 700))
    recs = [(int(p), genome[p - 1], [other_base(genome[p - 1])], "0/1" if i % 3 else "1/1") for i, p in enumerate(pos)]
    vcf = os.path.join(root, "d.vcf.gz")
    write_vcf(vcf, {"chr1": recs})
    starts = np.sort(rng.integers(10_000, n - 10_000, 1024))
    cre_csv = os.path.join(root, "c.csv")
    pd.DataFrame({"chromosome": "chr1", "start_cre": starts, "end_cre": starts + 250, "cre_name": "dELS"}).to_csv(cre_csv, index=False)
    genes = pd.DataFrame([{"gene_id": "G", "gene_name": "g", "chromosome": "chr1", "start": 500_000, "end": 900_000, "strand": "+"}])
    t0 = time.perf_counter(); h = dp.open_vcf(vcf); t1 = time.perf_counter()
    print(f"VCF parse: {h.num_records()} records in {t1 - t0:.3f} s")
    ds = VCFDataset(200, 200, 50, genes, LocalManifest({"G": cre_csv}), 1000, 300000,
                    pd.DataFrame({"gene_id": ["G"], "tissues": ["liver"]}), fasta, vcf)
    ds[0]
    t0 = time.perf_counter()
    for _ in range(3):
        X, m, t, l, rl, s, c, cm = ds[0]
    dt = (time.perf_counter() - t0) / 3
    print(f"one gene: {X.shape[0]} windows + {c.shape[0]} chunks in {dt * 1e3:.1f} ms -> {1 / dt:.1f} genes/s per host core")
