"""The in-process consensus (vf_vcf.cpp behind variantformer_amd.utils.data_process) against vectors produced by bcftools
itself with the reference's exact commands (reference utils/data_process.py:27,41-59).

The vectors live in tests/golden/consensus_bcftools.json, written by scripts/make_consensus_golden.sh -- which needs
samtools / bcftools 1.21 (reference Dockerfile:24-48).  The offline development image has neither binaries nor sources, so
until someone runs the recipe where the tools exist this test SKIPS and the consensus stays "parity unpinned" (DESIGN.md):
its rules are restated from bcftools' published behaviour and known-answer tested in tests/test_consensus_cpu.py.

What is always checked here (no tools needed): the recipe's input generator is deterministic and self-consistent, and the
in-process consensus accepts every record of it in both modes."""
import json
import os
import subprocess
import sys

import pytest

from variantformer_amd.utils import data_process as dp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden", "consensus_bcftools.json")
GEN = os.path.join(REPO, "scripts", "make_consensus_golden.py")


def _inputs(tmp_path):
    subprocess.run([sys.executable, GEN, "gen", str(tmp_path)], check=True)
    regions = [l.split("\t") for l in open(tmp_path / "regions.tsv").read().strip().split("\n")]
    return str(tmp_path / "genome.fa"), str(tmp_path / "donor.vcf"), [(c, int(a), int(b)) for c, a, b in regions]


def test_recipe_inputs_are_deterministic_and_accepted_in_process(tmp_path):
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    fa, vcf, regions = _inputs(tmp_path / "a")
    fb, vcfb, _ = _inputs(tmp_path / "b")
    assert open(fa).read() == open(fb).read() and open(vcf).read() == open(vcfb).read()
    reader, h = dp.FastaReader(fa), dp.VCFHandle(vcf)
    assert h.num_records() >= 40 and len(regions) >= 15
    applied = 0
    for chrom, a, b in regions:
        ref = reader.fetch(chrom, a, b)
        for snp_only in (False, True):
            seq, n = h.consensus(chrom, a, ref, snp_only)        # REF of every record matches the genome: no ConsensusError
            assert isinstance(seq, str) and n >= 0
            if snp_only:
                assert len(seq) == len(ref)                      # the SNP filter admits no length change
            applied += n
    assert applied > 50


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason=(
    "tests/golden/consensus_bcftools.json is absent: no samtools / bcftools binary or source exists in the offline image; "
    "run scripts/make_consensus_golden.sh where bcftools 1.21 is installed and commit the file -- until then the "
    "in-process consensus is parity-unpinned against the tool (known-answer tests only)"))
def test_in_process_consensus_equals_bcftools(tmp_path):
    gold = json.load(open(GOLDEN))
    fa = tmp_path / "genome.fa"
    with open(fa, "w") as f:
        for name, seq in gold["genome"].items():
            f.write(f">{name}\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
    vcf = tmp_path / "donor.vcf"
    vcf.write_text(gold["vcf"])
    reader, h = dp.FastaReader(str(fa)), dp.VCFHandle(str(vcf))
    bad = []
    for r in gold["regions"]:
        ref = reader.fetch(r["chrom"], r["start0"], r["end0"])
        for mode, snp_only in (("all", False), ("snp", True)):
            want = r[mode]
            if want["returncode"] != 0:
                # the reference falls back to the reference sequence when bcftools fails (:72-83)
                seq, n = dp.ExtractSeqFromBed(0, str(fa))._consensus(r["chrom"], r["start0"], r["end0"], str(vcf),
                                                                      "SNP" if snp_only else None)
                if seq != ref:
                    bad.append((r["chrom"], r["start0"], r["end0"], mode, "fallback", seq, ref))
                continue
            seq, n = h.consensus(r["chrom"], r["start0"], ref, snp_only)
            if seq != want["sequence"] or (want["applied"] is not None and n != want["applied"]):
                bad.append((r["chrom"], r["start0"], r["end0"], mode, n, want["applied"], seq, want["sequence"]))
    assert not bad, f"{len(bad)} region/mode pairs differ from {gold['tools']}: {bad[:3]}"
