"""End-to-end vcf2exp throughput on a synthetic genome at headline size: FASTA + donor VCF + per-gene cCRE tables ->
VCFDataset (in-process consensus + C++ BPE) in DataLoader workers -> collate -> HIP model (full 1.2B architecture,
random weights) -> expression matrix.  Shows whether the host-side sample builder keeps the GPU fed, and how many host
cores that takes.

usage: python scripts/vcf2exp_e2e.py [--genes N] [--workers W] [--batch B] [--cores C] [--overlap] [--passes P]
  --cores C    restrict THIS process (and the loader workers it starts) to C host cores with os.sched_setaffinity, set
               before anything touches the GPU (no taskset hop): what one rank of an 8-GPU node gets when the node's cores
               are shared out.
  --overlap    genes laid out like a whole-genome scan: one strand, TSS every 100 kb, cCRE windows drawn from ONE table
               along the genome, so that neighbouring genes share ~90 % of their windows byte for byte (reference
               datasets/vcfdataset.py:219-283); reports how many windows seq2reg embedded after exact de-duplication.
               Default: independent loci (BASELINE's synthetic workload; nothing to share)."""
import argparse, os, sys, tempfile, time
ap = argparse.ArgumentParser()
ap.add_argument("--genes", type=int, default=32)
ap.add_argument("--workers", type=int, default=8)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--cores", type=int, default=0)
ap.add_argument("--overlap", action="store_true")
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--no-dedupe", action="store_true")
args = ap.parse_args()
CORES = None
if args.cores:
    avail = sorted(os.sched_getaffinity(0))
    CORES = set(avail[:args.cores])
    os.sched_setaffinity(0, CORES)
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(args.cores, 4)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
if args.cores:
    torch.set_num_threads(max(1, min(args.cores, 4)))
from torch.utils.data import DataLoader
import bench
from tests.test_consensus_cpu import write_fasta, write_vcf, other_base
from variantformer_amd.datasets.vcfdataset import VCFDataset, collate_fn_batching
from variantformer_amd.datasets.vepdataset import LocalManifest
from variantformer_amd.utils.synthetic import TISSUES_54

n_genes, workers, batch_size = args.genes, args.workers, args.batch
rng = np.random.default_rng(0)
n = 3_000_000 if not args.overlap else max(3_000_000, 1_400_000 + 100_000 * n_genes)
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
if args.overlap:
    table = np.sort(rng.integers(50_000, n - 50_000, n // 977))           # ~1 cCRE per kb along the whole genome
for g in range(n_genes):
    gid = f"G{g}"
    if args.overlap:
        start = 600_000 + 100_000 * g
        genes.append({"gene_id": gid, "gene_name": gid, "chromosome": "chr1", "start": start, "end": start + 400_000, "strand": "+"})
        s = table[(table >= start - 500_000) & (table < start + 500_000)][:1024]
    else:
        start = int(rng.integers(600_000, n - 1_000_000))
        genes.append({"gene_id": gid, "gene_name": gid, "chromosome": "chr1", "start": start, "end": start + 400_000, "strand": "+-"[g % 2]})
        s = np.sort(rng.integers(start - 500_000, start + 500_000, 1024))
    paths[gid] = os.path.join(root, f"{gid}.csv")
    pd.DataFrame({"chromosome": "chr1", "start_cre": s, "end_cre": s + 250, "cre_name": "dELS"}).to_csv(paths[gid], index=False)
query = pd.DataFrame({"gene_id": [g["gene_id"] for g in genes], "tissues": [",".join(tissue_names)] * n_genes})
ds = VCFDataset(200, 200, 50, pd.DataFrame(genes), LocalManifest(paths), 1000, 300000, query, fasta, os.path.join(root, "d.vcf.gz"))
model, hp, kw = bench.build_model(torch.device("cuda:0"))
if CORES and len(os.sched_getaffinity(0)) != len(CORES):
    # (round 6: with --cores 1 the affinity set before the GPU was touched came back as all 256 CPUs after the runtime had
    # initialised -- the one-core line of r05 / r06_d ran unconfined; re-assert it here and in every loader worker)
    os.sched_setaffinity(0, CORES)


def _pin_worker(_):
    if CORES:
        os.sched_setaffinity(0, CORES)
loader = DataLoader(ds, batch_size=batch_size, num_workers=workers, collate_fn=collate_fn_batching,
                    prefetch_factor=2 if workers else None, persistent_workers=bool(workers), worker_init_fn=_pin_worker)
from variantformer_amd.processors.trainer import Trainer
trainer = Trainer(precision="bf16-mixed")
stats = {"total": 0, "embedded": 0, "prep_s": 0.0}
orig_prepare = model.prepare_batch


def prepare(batch, dedupe_windows=None):
    t0 = time.perf_counter()
    pb = orig_prepare(batch, dedupe_windows=False if args.no_dedupe else dedupe_windows)
    stats["prep_s"] += time.perf_counter() - t0
    stats["total"] += sum(pb.windows_total)
    stats["embedded"] += sum(pb.windows_embedded)
    return pb
model.prepare_batch = prepare
finished = []                                # (time, genes) of every finished batch: the steady-state rate excludes the first
orig_finish = model.predict_finish           # batch, which one loader worker builds alone (32 genes x ~20 ms) before the GPU starts


def finish(handle, i):
    o = orig_finish(handle, i)
    finished.append((time.perf_counter(), len(o["pred_gene_exp"])))
    return o
model.predict_finish = finish
tag = f"{len(os.sched_getaffinity(0))} host cores, {workers} loader workers, batches of {batch_size}" + (", overlapping loci" if args.overlap else "") + (", de-duplication off" if args.no_dedupe else "")
for rep in range(args.passes):              # first pass warms the workers' VCF / FASTA caches and the GPU
    stats.update(total=0, embedded=0, prep_s=0.0)
    finished.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = trainer.predict(model, loader)
    done = sum(len(o["pred_gene_exp"]) for o in outs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nb = max(1, len(outs))
    steady = (sum(g for _, g in finished[1:]) / (finished[-1][0] - finished[0][0])) if len(finished) > 2 else float("nan")
    first = finished[0][0] - t0 if finished else float("nan")
    print(f"pass {rep}: {done} genes x {len(tissue_names)} tissues in {dt:.2f} s -> {done / dt:.1f} genes/s end to end, "
          f"{steady:.1f} genes/s after the first batch (finished {first:.2f} s into the pass) ({tag}); "
          f"prepare_batch {1e3 * stats['prep_s'] / nb:.1f} ms per batch on the host; seq2reg embedded {stats['embedded']} of "
          f"{stats['total']} windows", flush=True)
