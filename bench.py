#!/usr/bin/env python
"""bench.py -- headline benchmark of the VariantFormer hot path on MI355X (contract: see the task statement).

Metric (BASELINE.json): genes/sec/node, 54-tissue expression at a 1 Mb cis-window, bf16 operands.
Workload at N=1 = BASELINE.json configs[1] ("Full 1.2 B checkpoint, 1 Mb window, 54 tissues, 1 donor VCF on
1xMI355X bf16") with SURVEY.md §8d's synthetic geometry: per gene N=1024 cCRE windows (70-125 valid tokens of
200), C=200 gene chunks of 200 tokens, T=54 tissues; model = full architecture (25 modulator layers, D=1536,
H=32; seq2reg d=512 / h=8 / 6 layers -- survey default until the tokenizer checkpoint can be read), random-init
weights (no checkpoint offline).

A step = one pass of the hot path (Seq2GenePredictorCombinedModulator.forward_prepared + D2H of the fp32 outputs)
over one batch of `--genes-per-step` genes whose token ids / masks are already resident in HBM.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), genes shard across ranks (weak scaling: every
rank runs `--genes-per-step` genes per step) and one all-gather reassembles the expression matrix each step.

`--workload cfg3` is BASELINE.json configs[2]: 256 ragged genes of one donor (N ~ lognormal(600, 0.6) in
[40, 2048] cCREs, C ~ U{20..200} chunks, 54 tissues).  A step is then ONE PASS OVER ALL 256 GENES: the genes are dealt to
the ranks by LPT on the FLOP model, every rank runs its shard in batches of <= `--genes-per-step` genes, and one
all-gather reassembles the [256, 54] expression matrix -- strong scaling (total work fixed as N grows); the line also
carries each rank's busy time (load imbalance).  `--dtype fp16` runs the fp16-operand path (BASELINE configs[4]).

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0

SEQ2REG_HP = dict(vocab_size=500, embedding_dim=512, num_heads=8, num_layers=6, num_tissues=1, num_classes=2,
                  learning_rate=1e-4, loss_fn=["cross_entropy", "0"], seq_pool="mean", cre_type="binary",
                  token_length=200, use_context=False, positional_encoding="sinusoidal", use_flash=True)
SEQ2GENE_KW = dict(num_tissues=63, emb_dim=1536, gene_emb_dim=512, num_heads=32, num_layers=25, use_alibi=True,
                   mlp_dout=0.1, use_context=True, token_dim=512, gene_pooling="multi_registry", multi_head=False,
                   use_bigger_head=True, only_cross_attention=False, cross_alibi=False, add_context_to_cres=False,
                   use_res=False, train_gene_tokenizer=True, use_batching=True)


def build_model(device, layers=None, seq2reg_layers=None, seed=1234):
    from variantformer_amd.seq2gene.model_combined_modulator import Seq2GenePredictorCombinedModulator
    from variantformer_amd.seq2reg.model import Seq2RegPredictor
    hp = dict(SEQ2REG_HP)
    kw = dict(SEQ2GENE_KW)
    if layers:
        kw["num_layers"] = layers
    if seq2reg_layers:
        hp["num_layers"] = seq2reg_layers
    with torch.device(device):
        cre_tok, gene_tok = Seq2RegPredictor(**hp), Seq2RegPredictor(**hp)
        model = Seq2GenePredictorCombinedModulator(cre_tokenizer=cre_tok, gene_tokenizer=gene_tok, **kw)
    for t in (cre_tok, gene_tok):
        t.position_encoding = t.position_encoding.cpu()
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if p.dim() >= 2 and "embedding" not in name and "registry" not in name:
                p.normal_(0.0, 0.4 / float(np.sqrt(p.shape[-1])), generator=g)     # same scale as utils.synthetic
            elif p.dim() >= 2:
                p.normal_(0.0, 0.29, generator=g)
            elif leaf == "weight":
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.01, generator=g)
    model.eval()
    return model, hp, kw


def host_threads() -> int:
    """Usable host cores: min(affinity, cgroup CPU quota) -- the GPU box shows 256 logical CPUs but a 16-CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def source_sha() -> str:
    """Hash of the kernel sources and build flags the loaded library was built from (ties a PMC profile to a build)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "variantformer_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "build.py":          # build.py: the compile flags are part of a build
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes of THIS command on THIS build
    (scripts/run_profile_set.sh writes profiles/r*_pmc_hbm_traffic.json with the source hash of the build it profiled;
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes).  bench.py cannot collect PMC counters itself, so a
    profile of another build is NOT used: traffic is null then."""
    import glob
    sha = source_sha()
    for src in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
        try:
            with open(src) as f:
                d = json.load(f)
            if d.get("_source_sha") != sha:
                continue
            if kernel == "gemm_all":            # launch-weighted mean over the GEMM kernels of the step
                ks = [d[k] for k in ("gemm8_kernel", "gemm_mfma_kernel") if k in d]
                if ks:
                    n = sum(k["launches"] for k in ks)
                    return sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / n, os.path.relpath(src, REPO)
            elif kernel in d:
                return d[kernel]["hbm_bytes_per_launch"], os.path.relpath(src, REPO)
        except Exception:
            continue
    return None, None


def cpu_baseline(model, hp, kw, executed_full: float, threads: int, budget_s: float = 20.0):
    """Oracle ("port" of the reference algorithm, fp32, tissue copies repeated as the reference does) timed on the
    host cores on a bounded sample of the same workload, scaled by executed FLOPs to genes/sec.  The sample is
    grown until the next size would exceed the time budget."""
    from oracle import vf_oracle as O
    from variantformer_amd.utils.flops import batch_flops
    from variantformer_amd.utils.synthetic import TISSUES_54, make_batch
    torch.set_num_threads(threads)
    sd = {k: v.detach().cpu().float() if torch.is_floating_point(v) else v.detach().cpu() for k, v in model.state_dict().items()}
    shp = O.Seq2RegHP.from_hparams(hp)
    ghp = O.Seq2GeneHP.from_kwargs(kw)
    best = None
    for n_cre, n_chunk, n_t in [(16, 3, 1), (48, 9, 3), (128, 25, 6), (256, 50, 8)]:
        sample = make_batch(777, [n_cre], [n_chunk], [TISSUES_54[:n_t]], 200)
        f_sample = batch_flops(sample, hp["embedding_dim"], hp["num_layers"], kw["emb_dim"], kw["num_layers"],
                               executed_by_reference=True)
        if best is not None and f_sample / best[1] * best[0] > budget_s:
            break
        t0 = time.perf_counter()
        O.predict_step(sample, sd, shp, shp, ghp, rounding=None, share_cre_stream=False)
        dt = time.perf_counter() - t0
        best = (dt, f_sample, n_cre, n_chunk, n_t)
    dt, f_sample, n_cre, n_chunk, n_t = best
    genes_per_s = 1.0 / (dt * executed_full / f_sample)
    return {"value": genes_per_s, "unit": "genes/sec", "cores": threads, "kind": "port",
            "sample": f"1 gene, N={n_cre} cCREs, C={n_chunk} chunks, T={n_t} tissues, full-depth model, fp32 oracle "
                      f"with per-tissue repeats as the reference executes them: {dt:.1f} s for {f_sample / 1e12:.3f} TFLOP "
                      f"({f_sample / dt / 1e12:.2f} TFLOP/s), scaled to the {executed_full / 1e12:.1f} TFLOP the reference "
                      f"executes per full-size gene"}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher around it: THIS process has made no GPU call (importing torch does
    not initialise HIP), so it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a CHILD
    process (never an exec: a process that has touched the GPU must not be replaced, and a parent that has not must not
    need to be), relays the child's stdout -- rank 0's one JSON line -- and returns the child's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if json_lines:
        print(json_lines[-1])
    sys.stdout.flush()
    return proc.returncode


def timed_steps(step, steps: int, warmup: int, use_dist: bool, sync, dev):
    """The contract's timed region: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier + device
    synchronisation on both sides; returns (seconds = MAX over ranks, the last step's result)."""
    import torch.distributed as dist
    out = None
    for _ in range(warmup):
        out = step()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    n_alloc = (lambda: torch.cuda.memory_stats(dev)["num_device_alloc"]) if dev.type == "cuda" else (lambda: 0)
    a0 = n_alloc()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        out = step()
        marks.append(time.perf_counter())        # a step ends with the D2H of its result, so this is its completion time
    sync()
    if use_dist:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timed_steps.device_allocs = n_alloc() - a0     # hipMalloc calls of the caching allocator inside the timed region
    timed_steps.last_step_ms = [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)]   # this rank's steps
    return dt, out


def run_stub(args, rank, world, use_dist):
    """--stub (tests/test_bench_launcher_cpu.py only; the line says data = "stub" and is NOT a measurement): the
    launcher, the process group (gloo, CPU), the timed-region skeleton, the per-step gather and the JSON relay with a
    trivial step standing in for the model -- rank r's "expression" of gene g, tissue t is g * 100 + t."""
    from variantformer_amd.dist import all_gather_expression
    G, T = args.genes_per_step, args.tissues
    dev = torch.device("cpu")
    if os.environ.get("VF_BENCH_STUB_FAIL_RANK") == str(rank):      # the launcher test's failing rank
        raise RuntimeError(f"stub: rank {rank} fails on request")
    owned = [list(range(r * G, (r + 1) * G)) for r in range(world)]
    local = torch.tensor([[g * 100.0 + t for t in range(T)] for g in owned[rank]], dtype=torch.float32)

    def step():
        return all_gather_expression(local, owned, world * G) if use_dist else all_gather_expression(local, [owned[0]], G)
    dt, expr = timed_steps(step, args.steps, args.warmup, use_dist, lambda: None, dev)
    want = torch.tensor([[g * 100.0 + t for t in range(T)] for g in range(world * G)], dtype=torch.float32)
    assert torch.equal(expr, want), "gathered matrix is not in query order"
    if rank == 0:
        print(json.dumps({"metric": "genes/sec/node (54-tissue expr) at 1 Mb cis-window", "value": round(world * G * args.steps / dt, 4),
                          "unit": "genes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub",
                          "config": {"workload": "STUB STEP (launcher / process-group test, not a measurement)",
                                     "parallelism": f"gene-shard x{world}"}}))


def cfg3_pass(args, model, hp, kw, dev, rank, world, use_dist, steps: int, warmup: int) -> dict:
    """One strong-scaling measurement on BASELINE configs[2] (256 ragged genes, LPT shards, one gather per pass)."""
    import torch.distributed as dist
    from variantformer_amd.dist import all_gather_expression, gene_cost, shard_genes_lpt
    from variantformer_amd.utils.flops import batch_flops
    from variantformer_amd.utils.synthetic import TISSUES_54, cfg3_gene_sizes, collate, make_gene
    n_genes, T = 256, args.tissues
    tissues = TISSUES_54[:T]
    n, c = cfg3_gene_sizes(n_genes)
    costs = [gene_cost(int(a), int(b), T) for a, b in zip(n, c)]
    owned = shard_genes_lpt(costs, world)
    mine = owned[rank]
    bs = args.genes_per_step
    flops_all = 0.0
    with torch.no_grad():
        pbs = []
        for s0 in range(0, len(mine), bs):          # inputs resident in HBM before the timed region
            ids = mine[s0:s0 + bs]
            batch = collate([make_gene(20251205 * 1000003 + g, int(n[g]), int(c[g]), tissues, 200) for g in ids])
            flops_all += batch_flops(batch, hp["embedding_dim"], hp["num_layers"], kw["emb_dim"], kw["num_layers"])
            pbs.append(model.prepare_batch(batch))
        busy_acc = [0.0]

        def one_pass():
            t0 = time.perf_counter()
            parts = [model.forward_prepared(pb)[0].view(pb.n_genes, T) for pb in pbs]
            local = torch.cat(parts) if parts else torch.empty((0, T), device=dev)
            torch.cuda.synchronize()
            busy_acc[0] += time.perf_counter() - t0
            expr = all_gather_expression(local, owned, n_genes) if use_dist else all_gather_expression(local, [mine], n_genes)
            return expr.cpu()

        for _ in range(warmup):
            one_pass()
        busy_acc[0] = 0.0
        dt, expr = timed_steps(one_pass, steps, 0, use_dist, torch.cuda.synchronize, dev)
        busy = busy_acc[0]
        busies = [busy]
        if use_dist:
            t = torch.tensor([busy, flops_all], device=dev, dtype=torch.float64)
            allt = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            busies = [float(x[0]) for x in allt]
            flops_all = float(sum(float(x[1]) for x in allt))
        assert torch.isfinite(expr).all() and tuple(expr.shape) == (n_genes, T)
    loads = [sum(costs[i] for i in o) for o in owned]
    return {"value": round(n_genes * steps / dt, 4), "unit": "genes/sec", "scaling": "strong", "steps": steps, "warmup": warmup,
            "ms_per_pass": round(dt / steps * 1e3, 3), "genes": n_genes, "batch_genes_per_launch": bs, "tissues": T,
            "algorithmic_tflop_per_pass": round(flops_all / 1e12, 2),
            "achieved_algorithmic_tflops_whole_step": round(flops_all * steps / dt / 1e12, 1),
            "rank_busy_seconds_per_pass": [round(b / steps, 4) for b in busies],
            "busy_imbalance_max_over_mean": round(max(busies) / (sum(busies) / len(busies)), 4),
            "lpt_cost_imbalance_max_over_mean": round(max(loads) / (sum(loads) / len(loads)), 4),
            "genes_per_rank": [len(o) for o in owned]}


def run_cfg3(args, model, hp, kw, dev, rank, world, use_dist):
    """--workload cfg3: the line's `value` is the strong-scaling pass over BASELINE configs[2]."""
    import torch.distributed as dist
    r = cfg3_pass(args, model, hp, kw, dev, rank, world, use_dist, args.steps, args.warmup)
    if rank == 0:
        T = args.tissues
        print(json.dumps({
            "metric": "genes/sec/node (54-tissue expr) at 1 Mb cis-window", "value": r["value"],
            "unit": "genes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_pass"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 256 ragged genes of one donor (N ~ lognormal(600, 0.6) in [40, 2048] "
                                   "cCRE windows, C ~ U{20..200} gene chunks, %d tissues), full 1.2B-architecture network, "
                                   "random-init weights; one step = one pass over all 256 genes, LPT gene shards, one "
                                   "all-gather of the [256, %d] expression matrix" % (T, T),
                       "genes": 256, "batch_genes_per_launch": args.genes_per_step, "tissues": T,
                       "parallelism": f"gene-shard (LPT) x{world}"},
            **{k: r[k] for k in ("algorithmic_tflop_per_pass", "achieved_algorithmic_tflops_whole_step",
                                 "rank_busy_seconds_per_pass", "busy_imbalance_max_over_mean",
                                 "lpt_cost_imbalance_max_over_mean", "genes_per_rank")},
            "source_sha": source_sha()}))
    if use_dist:
        dist.destroy_process_group()


def pipelined_product_flow(model, batch, n_batches: int, dev) -> dict:
    """The product flow on the same genes, host work included: processors.trainer.Trainer.predict over a DataLoader whose
    items are headline-size collate_fn_batching dicts on the HOST -- prepare_batch (concatenation, structure arrays), the
    H2D copies, the forward and the D2H of expression AND embeddings (predict_step's dict), software-pipelined by one
    batch (reference loop: processors/vcfprocessor.py:261-265).  Not `value`: `value` starts with inputs in HBM."""
    from torch.utils.data import DataLoader, Dataset
    from variantformer_amd.processors.trainer import Trainer

    class Repeat(Dataset):              # the same host-resident batch n times (what a loader worker would hand over)
        def __len__(self):
            return n_batches

        def __getitem__(self, i):
            return batch
    loader = DataLoader(Repeat(), batch_size=None, shuffle=False, num_workers=0)
    trainer = Trainer(precision="bf16-mixed" if model.operand_dtype() == torch.bfloat16 else "16-mixed")
    keep = model.precision
    trainer.predict(model, DataLoader(Repeat(), batch_size=None, num_workers=0))        # warm (allocator, caches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = trainer.predict(model, loader)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.precision, model.trainer = keep, None
    genes = sum(len(o["pred_gene_exp"]) for o in outs)
    t1 = time.perf_counter()
    for _ in range(3):
        model.prepare_batch(batch)
    torch.cuda.synchronize()
    prep_ms = (time.perf_counter() - t1) / 3 * 1e3
    return {"vcf2exp_pipelined_genes_per_s": round(genes / dt, 3), "batches": n_batches, "genes": genes,
            "prepare_batch_ms_per_batch_incl_h2d": round(prep_ms, 2),
            "note": "Trainer.predict over host-resident collate_fn_batching dicts: prepare_batch + H2D + forward + D2H of "
                    "expression and embeddings, one batch of software pipelining; includes one batch of pipeline fill"}


def trained_like_pass(model, args, G, tissues, dev, n_batches: int = 16) -> dict:
    """What the regime guard of the LayerNorm fold does on weights with trained-transformer statistics (round-5 verdict: random
    weights never trip it).  The resident model's weights are transformed IN PLACE (utils.synthetic.trained_like_: outlier
    channels x25 in both residual streams of every layer, log-normal LayerNorm gains, heavy-tailed projections -- the transform
    of tests/test_trained_like_gpu.py, here at full depth), then `n_batches` batches (4 distinct synthetic ones, resident) go
    through the product's predict_launch / predict_finish, which recomputes an alerting batch with the separate LayerNorm and
    switches the fold off for the model when 2 of its last 16 batches alerted.  Reported: alerts among those batches, whether
    the switch tripped, and the genes/s of the whole sequence (recomputations included).  Runs LAST: it rewrites the weights."""
    from variantformer_amd.seq2gene.model_combined_modulator import HealState, heal_state
    from variantformer_amd.utils.synthetic import make_batch, trained_like_
    t0 = time.perf_counter()
    trained_like_(model, 5)
    t_transform = time.perf_counter() - t0
    object.__setattr__(model, "_vf_heal", HealState())          # the counters of this pass only
    hs = heal_state(model)
    with torch.no_grad():
        pbs = [model.prepare_batch(make_batch(4000 + i, [args.n_cre] * G, [args.n_chunks] * G, [tissues] * G, 200)) for i in range(4)]
        model.predict_finish(model.predict_launch(pbs[0]), 0)     # weight re-packing and caches of the new weights: untimed
        object.__setattr__(model, "_vf_heal", HealState())
        hs = heal_state(model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        finite = True
        for i in range(n_batches):
            out = model.predict_finish(model.predict_launch(pbs[i % 4]), i)
            finite = finite and all(np.isfinite(p).all() for p in out["pred_gene_exp"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    spread = float(np.concatenate([p.ravel() for p in out["pred_gene_exp"]]).std())
    return {"batches": n_batches, "genes_per_batch": G, "alerting_batches": int(hs.batches),
            "alert_pattern_last_16": "".join("x" if a else "." for a in hs.recent),
            "fold_switched_off_for_the_model": bool(hs.off), "value": round(n_batches * G / dt, 4), "unit": "genes/sec",
            "ms_per_batch": round(dt / n_batches * 1e3, 3), "outputs_finite": bool(finite),
            "expression_spread_last_batch": round(spread, 5), "weight_transform_s": round(t_transform, 1),
            "note": "trained-like weight transform of tests/test_trained_like_gpu.py at full depth; predict_launch / predict_finish "
                    "(D2H of expression and embeddings); an alerting batch is recomputed with the separate LayerNorm pass"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)      # 3 passes prime the caching allocator of both streams: no extra pass
    ap.add_argument("--genes-per-step", type=int, default=32,
                    help="genes per rank per step; 32 = BASELINE configs[2]'s 256-gene batch over the 8 GPUs of a node (the "
                         "reference's DataLoader default, configs/vcfloader.yaml:5, is 8: that rate is reported beside "
                         "`value` as `batch_of_8`)")
    ap.add_argument("--n-cre", type=int, default=1024)
    ap.add_argument("--n-chunks", type=int, default=200)
    ap.add_argument("--tissues", type=int, default=54)
    ap.add_argument("--layers", type=int, default=None, help="override modulator depth (debug only; invalidates the number)")
    ap.add_argument("--workload", choices=["cfg2", "cfg3"], default="cfg2",
                    help="cfg2 = the headline 1 Mb / 54-tissue gene (weak scaling); cfg3 = 256 ragged genes, LPT shards, "
                         "strong scaling")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16", help="16-bit operand type (fp32 accumulation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-cfg3", action="store_true", help="N > 1: skip the extra strong-scaling pass over BASELINE configs[2]")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the host-inclusive product-flow measurement")
    ap.add_argument("--no-extra-rates", action="store_true",
                    help="skip value_single_stream / ln_fold_off / trained_like / batch_of_8 / batch_of_1 (the passes after the timed region): "
                         "the process then runs W + K + K identical steps, which is what the rocprofv3 passes of "
                         "scripts/run_profile_set.sh profile")
    ap.add_argument("--single-stream", action="store_true",
                    help="runtime.Switches.overlap_cre_stream off for the whole run (the order the per-kernel replay uses): what "
                         "the rocprofv3 --kernel-trace --stats pass profiles, because with two streams a kernel's traced "
                         "duration includes the time it shares the CUs with the other stream's kernels")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)      # launcher test: no model, gloo on CPU
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:        # no launcher around us: start the ranks as a child process
        sys.exit(self_launch(args, sys.argv[1:]))

    if args.single_stream:
        from variantformer_amd import runtime
        runtime.set_for_this_context(overlap_cre_stream=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch.distributed as dist
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with one rank: exercises RCCL)
    if args.stub:
        if use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        run_stub(args, rank, world, use_dist)
        if use_dist:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback on the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from variantformer_amd import ops
    from variantformer_amd.dist import all_gather_expression
    from variantformer_amd.utils.flops import batch_flops
    from variantformer_amd.utils.synthetic import TISSUES_54, make_batch

    model, hp, kw = build_model(dev, args.layers)
    if args.dtype == "fp16":
        model.precision = "16-mixed"               # reference utils/functions.py:12-32: fp16 operands
    if args.workload == "cfg3":
        return run_cfg3(args, model, hp, kw, dev, rank, world, use_dist)
    G = args.genes_per_step
    tissues = TISSUES_54[: args.tissues]
    batch = make_batch(20251205 + rank, [args.n_cre] * G, [args.n_chunks] * G, [tissues] * G, 200)
    flops_step = batch_flops(batch, hp["embedding_dim"], hp["num_layers"], kw["emb_dim"], kw["num_layers"])
    executed_step = batch_flops(batch, hp["embedding_dim"], hp["num_layers"], kw["emb_dim"], kw["num_layers"],
                                executed_by_reference=True)
    owned = [list(range(r * G, (r + 1) * G)) for r in range(world)]

    with torch.no_grad():
        pb = model.prepare_batch(batch)                      # inputs resident in HBM before the timed region

        def step_local():
            pred, emb = model.forward_prepared(pb)
            return pred.view(G, len(tissues)), emb

        def step():
            expr, emb = step_local()
            if use_dist:
                expr = all_gather_expression(expr, owned, world * G)
            return expr.cpu(), emb                            # D2H of the expression matrix (sync point of a step)

        # Warm-up = the contract's W untimed steps.  They also let torch's caching allocator reach its steady state: it keeps
        # adding multi-GiB segments for the first three passes (fragmentation: 37 GiB live, 61 GiB reserved at 32 genes), and one
        # such hipMalloc costs 0 or ~220 ms at random (profiles/r03_y_bench_repeat.log: the first timed step of 2 runs in 6 with
        # W = 2).  Only when the LAST of the W passes still allocated device memory (W < 3) are extra local passes run until one
        # allocates nothing; they are counted in `warmup` (round-5 verdict: with W >= 3 `warmup` is literally W).
        n_alloc = lambda: torch.cuda.memory_stats(dev)["num_device_alloc"]      # noqa: E731
        before = n_alloc()
        for _ in range(args.warmup):
            before = n_alloc()
            step()
        still_allocating = args.warmup == 0 or n_alloc() != before
        extra_warmup = 0
        while still_allocating and extra_warmup < 6:
            before = n_alloc()
            step_local()[0].cpu()
            extra_warmup += 1
            still_allocating = n_alloc() != before

        dt, (expr, _) = timed_steps(step, args.steps, 0, use_dist, torch.cuda.synchronize, dev)
        each_step_ms = timed_steps.last_step_ms
        allocs_in_timed_region = timed_steps.device_allocs
        assert torch.isfinite(expr).all() and tuple(expr.shape) == (world * G, len(tissues))

        # After the timed region: (1) the same steps on ONE stream (runtime.Switches.overlap_cre_stream off: the product default
        # runs the CRE layers on a side stream beside the gene layers; bit-identical, so max_rel_diff must be 0.0); (2) the
        # fallback the self-healing LayerNorm fold switches to (a batch whose rows left the folded form's regime; a model whose
        # alerts became sticky): every LayerNorm as a pass on fp32 rows, exactly a VF_LN_FOLD=0 run (reference: plain
        # nn.LayerNorm, seq2gene/modules/layers.py:75-77).
        single = fold_off = None
        if not args.no_extra_rates:
            from variantformer_amd import runtime
            from variantformer_amd.seq2gene.modules.layers import ln_fold_forced_off
            with runtime.override(overlap_cre_stream=False):
                dts, (expr_single, _) = timed_steps(step, args.steps, 2, use_dist, torch.cuda.synchronize, dev)
            single = {"value": round(world * G * args.steps / dts, 4), "unit": "genes/sec",
                      "ms_per_step": round(dts / args.steps * 1e3, 3), "steps": args.steps,
                      "max_rel_diff_of_expression_vs_value_path": float(((expr_single - expr).abs() / expr.abs()).max()),
                      "note": "runtime.Switches.overlap_cre_stream=False: every kernel on one stream (the order the per-kernel "
                              "replay behind `roofline` / `kernel_families` uses)"}
            k_off = max(2, min(args.steps, 4))
            with ln_fold_forced_off():
                dto, (expr_off, _) = timed_steps(step, k_off, 2, use_dist, torch.cuda.synchronize, dev)
            assert torch.isfinite(expr_off).all()
            fold_off = {"value": round(world * G * k_off / dto, 4), "unit": "genes/sec", "ms_per_step": round(dto / k_off * 1e3, 3),
                        "steps": k_off, "ratio_to_value": round((world * G * k_off / dto) / (world * G * args.steps / dt), 4),
                        "max_rel_diff_of_expression_vs_folded": float(((expr_off - expr).abs() / expr.abs()).max())}

        roof = None
        kernels = {}
        if rank == 0 and not args.no_kernel_timing:
            # Per-kernel durations: the same K steps replayed with HIP events bracketing each launch on the launch
            # stream (kept out of the timed region above so that event overhead does not enter `value`).  The replay is
            # LOCAL (no gather): the other ranks are not replaying, and a collective here would wait for them forever.
            ops.TIMER = ops.KernelTimer()
            for _ in range(args.steps):
                step_local()[0].cpu()
            summ = ops.TIMER.summary()
            ops.TIMER = None
            g = summ["gemm"]
            tf = g["flops"] / (g["total_ms"] * 1e-3) / 1e12
            step_ms = dt / args.steps * 1e3
            traffic, traffic_src = pmc_traffic("gemm_all")
            roof = {"kernel": "gemm8x_kernel + gemm8_kernel + gemm_mfma_kernel (vf_gemm_bf16 / vf_gemm_ln_bf16 / vf_gemm_f16, all "
                              "epilogues; >= 95 % of the GEMM time is the two-group 256x256 kernel: persistent form "
                              "gemm8x for 16-bit outputs, one-shot gemm8 for fp32 outputs)", "bound": "mfma", "achieved": round(tf, 1),
                    "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
                    "launches_per_step": g["launches"] // args.steps,
                    "avg_launch_us": round(g["total_ms"] * 1e3 / g["launches"], 2),
                    "flop_per_launch": g["flops"] / g["launches"],
                    "share_of_step_time": round(g["total_ms"] / args.steps / step_ms, 3),
                    "note": ("the GEMM launches also carry the LayerNorms of the layers (producer epilogues write "
                             "the 16-bit copy of the fp32 stream + per-row partial statistics, consumer epilogues apply mean / "
                             "rstd; DESIGN.md section 6) -- their bytes are in algorithmic_bytes_per_launch, their time in "
                             "avg_launch_us; VF_LN_FOLD=0 restores the separate LayerNorm pass")}
            # Per kernel family (KernelTimer families: the module that launched the kernel).  Every family is priced
            # against ITS roofline = min(dense MFMA peak, arithmetic intensity x HBM peak), SURVEY 8d.
            for key, r in sorted(summ.items()):
                if ":" not in key:
                    continue
                kind, fam = key.split(":", 1)
                sec = r["total_ms"] * 1e-3
                ent = {"launches_per_step": r["launches"] // args.steps, "ms_per_step": round(r["total_ms"] / args.steps, 3),
                       "share_of_step_time": round(r["total_ms"] / args.steps / step_ms, 4),
                       "algorithmic_GBps": round(r["bytes"] / sec / 1e9, 1)}
                if r["flops"] > 0:
                    inten = r["flops"] / r["bytes"]
                    ceil_tf = min(MFMA_PEAK_TFLOPS, inten * HBM_PEAK_GBS / 1e3)
                    ach = r["flops"] / sec / 1e12
                    ent.update({"achieved_TFLOPs": round(ach, 1), "intensity_flop_per_byte": round(inten, 1),
                                "bound": "mfma" if ceil_tf >= MFMA_PEAK_TFLOPS else "hbm", "roofline_TFLOPs": round(ceil_tf, 1),
                                "frac_of_roofline": round(ach / ceil_tf, 4), "frac_of_mfma_peak": round(ach / MFMA_PEAK_TFLOPS, 4)})
                else:
                    ent.update({"bound": "hbm", "frac_of_roofline": round(r["bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4)})
                kernels[f"{kind}/{fam}"] = ent

        small = None
        if rank == 0 and world == 1 and not args.no_kernel_timing and not args.no_extra_rates:
            # the same step at the reference's own DataLoader batch size (8 genes, configs/vcfloader.yaml:5) and for one
            # gene alone (SURVEY 8d's cfg-2 row: B = 1), inputs resident, same K / W
            small = {}
            for g in (8, 1):
                if g == G:
                    continue
                bg = make_batch(20251205 + rank, [args.n_cre] * g, [args.n_chunks] * g, [tissues] * g, 200)
                pbg = model.prepare_batch(bg)
                dtg, _ = timed_steps(lambda: model.forward_prepared(pbg)[0].view(g, len(tissues)).cpu(), args.steps,
                                     args.warmup, False, torch.cuda.synchronize, dev)
                small[g] = {"genes_per_step": g, "value": round(g * args.steps / dtg, 4), "unit": "genes/sec",
                            "ms_per_step": round(dtg / args.steps * 1e3, 3)}
                del pbg, bg

        strong = None
        if world > 1 and not args.no_cfg3:
            # the informative multi-GPU number: total work fixed (BASELINE configs[2]), LPT shards, busy time per rank
            strong = cfg3_pass(args, model, hp, kw, dev, rank, world, use_dist, steps=2, warmup=1)
        flow = None
        if rank == 0 and world == 1 and not args.no_pipelined:
            flow = pipelined_product_flow(model, batch, max(4, min(args.steps, 10)), dev)

    if rank == 0:
        value = world * G * args.steps / dt
        out = {
            "metric": "genes/sec/node (54-tissue expr) at 1 Mb cis-window", "value": round(value, 4), "unit": "genes/sec",
            # `warmup` counts EVERY untimed pass that ran before the timed region: the W requested warm-up steps plus any extra
            # allocator-priming pass (none when W >= 3: the W passes prime the allocator themselves)
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup + extra_warmup, "warmup_requested": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: full 1.2B-architecture network (%d modulator layers, width %d, %d "
                                    "heads; seq2reg width 512, 8 heads, 6 layers, assumed), random-init weights, 1 Mb "
                                    "cis-window, %d tissues, 1 donor; one step = %d such genes per GPU (batch_of_8 / "
                                    "batch_of_1 beside it)" % (kw["num_layers"], kw["emb_dim"], kw["num_heads"],
                                                               len(tissues), G)),
                       "genes_per_step_per_gpu": G, "n_cre_windows": args.n_cre, "gene_chunks": args.n_chunks,
                       "tissues": len(tissues), "tokens_per_window": 200, "parallelism": f"gene-shard x{world}",
                       "streams": "one (--single-stream)" if args.single_stream else "CRE layers on a side stream (default)"},
            "algorithmic_tflop_per_gene": round(flops_step / G / 1e12, 3),
            "reference_executed_tflop_per_gene": round(executed_step / G / 1e12, 3),
            "achieved_algorithmic_tflops_whole_step": round(world * flops_step * args.steps / dt / 1e12, 1),
            "roofline": roof, "kernel_families": kernels, "source_sha": source_sha(),
            "ms_of_each_timed_step_rank0": each_step_ms, "device_allocations_inside_timed_region": allocs_in_timed_region,
            "extra_allocator_priming_passes": extra_warmup,
        }
        if single is not None:
            out["value_single_stream"] = single
        if fold_off is not None:
            out["ln_fold_off"] = fold_off
        for g, rec in (small or {}).items():
            out[f"batch_of_{g}"] = rec
        out["peak_hbm_allocated_gb"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)
        from variantformer_amd.seq2gene.model_combined_modulator import ln_fold_state
        out["ln_fold"] = ln_fold_state()                     # the fold was ON for every timed step unless this says otherwise
        if strong is not None:
            out["cfg3_strong_scaling"] = strong
        if flow is not None:
            out.update({"vcf2exp_pipelined_genes_per_s": flow["vcf2exp_pipelined_genes_per_s"], "vcf2exp_pipelined": flow})
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, hp, kw, executed_step / G, host_threads())
        if world == 1 and not args.no_extra_rates:
            out["trained_like"] = trained_like_pass(model, args, G, tissues, dev)      # rewrites the weights: last
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
