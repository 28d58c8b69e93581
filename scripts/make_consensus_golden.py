"""Golden vectors for the in-process IUPAC consensus (variantformer_amd/csrc/vf_vcf.cpp) FROM THE TOOL THE REFERENCE CALLS.

The reference builds every cCRE window / gene body with (utils/data_process.py:27,41-59)

    samtools faidx <fasta> chr:start-end | bcftools consensus -H I -e 'ALT~"<.*>"' <vcf.gz>                   (all variants)
    samtools faidx <fasta> chr:start-end | bcftools consensus -H I -e 'ALT~"<.*>" || TYPE!="snp"' <vcf.gz>    (variant_type="SNP")

with samtools / bcftools / htslib 1.21 (reference Dockerfile:24-48).  Neither binary nor source is in the offline image this
repository is developed in, so vf_vcf.cpp restates the rules and is known-answer tested only ("parity unpinned").  This script
closes that the day the tools are at hand:

    scripts/make_consensus_golden.sh          # needs samtools, bcftools (1.21), bgzip, tabix on PATH

  gen  <dir>    write the inputs: genome.fa (2 contigs, mixed case, N runs), donor.vcf (one sample; every genotype / allele
                form the consensus rules distinguish), regions.tsv (0-based half-open, as ExtractSeqFromBed computes them)
  pack <dir> <out.json>   after the shell script ran the two commands per region: collect inputs + tool outputs + tool
                versions into tests/golden/consensus_bcftools.json

tests/test_consensus_bcftools_cpu.py consumes that file when present (and is skipped, with this reason, when absent).
Everything here is deterministic (no RNG): re-running the recipe with the same tool versions reproduces the file byte for byte."""
import json
import os
import sys


def genome():
    unit1 = "ACGTTGCAAGGCTTAACCGGATCGATTACAGGCTAGCTTAGGCATCGGATCCTAGGAATTCCGGTTAACC"
    unit2 = "TTGACCATGGCAGTCAAGTCCGATAGGCTTACGATCGGCTAAGCTTGCATGCCTGCAGGTCGACTCTAGAG"
    c1 = (unit1 * 12)[:800]
    c1 = c1[:300] + c1[300:340].lower() + c1[340:500] + "NNNNNNNN" + c1[508:]          # soft-masked run, N run
    c2 = (unit2 * 6)[:400]
    return {"chr1": c1, "chr2": c2}


def records(g):
    """(chrom, pos1, ref, [alts], gt).  REF is always read off the genome (upper-cased) so that the file is self-consistent."""
    def R(chrom, pos, n=1):
        return g[chrom][pos - 1:pos - 1 + n].upper()

    def other(b, k=1):
        return "ACGT"[("ACGT".index(b) + k) % 4]
    recs = []

    def add(chrom, pos, reflen, alts, gt):
        ref = R(chrom, pos, reflen)
        assert "N" not in ref, (chrom, pos, ref)
        recs.append((chrom, pos, ref, [a(ref) if callable(a) else a for a in alts], gt))
    snp = lambda k=1: (lambda ref: other(ref[0], k))                         # noqa: E731
    ins = lambda tail: (lambda ref: ref + tail)                              # noqa: E731
    dele = lambda keep=1: (lambda ref: ref[:keep])                           # noqa: E731
    mnp = lambda: (lambda ref: "".join(other(b, 2) for b in ref))            # noqa: E731
    c = "chr1"
    add(c, 5, 1, [snp()], "0/1")                 # het SNP
    add(c, 9, 1, [snp()], "1/1")                 # hom SNP
    add(c, 13, 1, [snp(1), snp(2)], "1/2")       # two ALT SNPs, het between them
    add(c, 17, 1, [snp()], "0|1")                # phased
    add(c, 21, 1, [snp()], "1|0")
    add(c, 25, 1, [snp()], "0/0")                # hom ref
    add(c, 29, 1, [snp()], "./.")                # missing
    add(c, 33, 1, [snp()], "./1")                # half missing
    add(c, 37, 1, [snp()], "1")                  # haploid
    add(c, 41, 1, ["<DEL>"], "0/1")              # symbolic (excluded by -e in both modes)
    add(c, 45, 1, [snp()], "1/1")
    add(c, 45, 1, [snp(2)], "0/1")               # duplicate position after an applied record
    add(c, 60, 1, [ins("GG")], "0/1")            # het insertion
    add(c, 66, 1, [ins("TTT")], "1/1")           # hom insertion
    add(c, 72, 3, [dele()], "0/1")               # het deletion
    add(c, 80, 4, [dele()], "1/1")               # hom deletion
    add(c, 82, 1, [snp()], "1/1")                # SNP inside the span of the hom deletion (overlap)
    add(c, 90, 1, [ins("A"), ins("CCC")], "1/2")       # two insertions of unequal length
    add(c, 96, 2, [dele(), "*"], "1/2")          # '*' allele beside a deletion
    add(c, 102, 1, [snp(), ins("ACGT")], "0/2")  # mixed snp + insertion ALTs, genotype picks the insertion
    add(c, 108, 1, [snp(), ins("ACGT")], "0/1")  # mixed ALTs, genotype picks the snp
    add(c, 114, 2, [mnp()], "0/1")               # het MNP (equal length)
    add(c, 120, 3, [mnp()], "1/1")               # hom MNP
    add(c, 126, 2, [mnp(), dele()], "1/2")       # MNP + deletion
    add(c, 140, 1, [ins("T")], "1/1")
    add(c, 141, 1, [snp()], "0/1")               # adjacent to an insertion anchor
    add(c, 150, 5, [dele()], "1/1")
    add(c, 152, 2, [dele()], "1/1")              # deletion nested inside a deletion
    add(c, 199, 4, [dele()], "1/1")              # crosses the end of region chr1:101-200 (1-based)
    add(c, 201, 1, [snp()], "1/1")               # first base of the next region
    add(c, 298, 6, [dele(2)], "0/1")             # reaches into the lower-case run (301..340)
    add(c, 310, 1, [snp()], "0/1")               # het SNP on a lower-case base
    add(c, 315, 1, [ins("gg")], "1/1")           # lower-case ALT text
    add(c, 320, 2, [mnp()], "0/1")
    add(c, 498, 2, [dele()], "1/1")              # next to the N run (501..508)
    add(c, 510, 1, [snp()], "0/1")
    add(c, 700, 1, [snp()], "0/1")
    add(c, 799, 2, [dele()], "1/1")              # at the contig end
    c = "chr2"
    add(c, 1, 1, [snp()], "1/1")                 # first base of a contig
    add(c, 2, 3, [dele()], "0/1")
    add(c, 50, 1, [ins("ACGTACGTAC")], "1/1")    # long insertion
    add(c, 100, 12, [dele()], "1/1")             # long deletion
    add(c, 200, 1, [snp(1), snp(2), snp(3)], "2/3")
    add(c, 400, 1, [snp()], "1/1")               # last base of a contig
    return recs


def regions(g):
    """0-based half-open (chrom, start, end): region_str of the reference is chrom:start+1-end."""
    out = [("chr1", 0, 100), ("chr1", 100, 200), ("chr1", 200, 350), ("chr1", 0, 800), ("chr1", 70, 90), ("chr1", 81, 130),
           ("chr1", 150, 156), ("chr1", 152, 156), ("chr1", 290, 345), ("chr1", 480, 520), ("chr1", 690, 800),
           ("chr2", 0, 400), ("chr2", 0, 60), ("chr2", 1, 60), ("chr2", 95, 120), ("chr2", 105, 120), ("chr2", 190, 210),
           ("chr2", 390, 400), ("chr1", 595, 650)]                     # the last one holds no record at all
    for c, a, b in out:
        assert 0 <= a < b <= len(g[c])
    return out


def write_inputs(d):
    os.makedirs(d, exist_ok=True)
    g = genome()
    with open(os.path.join(d, "genome.fa"), "w") as f:
        for name, seq in g.items():
            f.write(f">{name}\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
    lines = ["##fileformat=VCFv4.2"] + [f"##contig=<ID={c},length={len(s)}>" for c, s in g.items()] + [
        '##ALT=<ID=DEL,Description="Deletion">', '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tDONOR"]
    for chrom, pos, ref, alts, gt in records(g):
        lines.append(f"{chrom}\t{pos}\t.\t{ref}\t{','.join(alts)}\t.\tPASS\t.\tGT\t{gt}")
    with open(os.path.join(d, "donor.vcf"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(d, "regions.tsv"), "w") as f:
        for c, a, b in regions(g):
            f.write(f"{c}\t{a}\t{b}\n")


def pack(d, out):
    g = genome()
    res = {"tools": open(os.path.join(d, "versions.txt")).read().strip().splitlines(),
           "commands": {"all": "samtools faidx genome.fa {chrom}:{start+1}-{end} | bcftools consensus -H I -e 'ALT~\"<.*>\"' donor.vcf.gz",
                        "snp": "samtools faidx genome.fa {chrom}:{start+1}-{end} | bcftools consensus -H I -e 'ALT~\"<.*>\" || TYPE!=\"snp\"' donor.vcf.gz"},
           "genome": g, "vcf": open(os.path.join(d, "donor.vcf")).read(), "regions": []}
    for i, (c, a, b) in enumerate(regions(g)):
        rec = {"chrom": c, "start0": a, "end0": b}
        for mode in ("all", "snp"):
            fa = open(os.path.join(d, f"out_{i}_{mode}.fa")).read().strip().split("\n")
            err = open(os.path.join(d, f"out_{i}_{mode}.err")).read()
            rc = int(open(os.path.join(d, f"out_{i}_{mode}.rc")).read().strip())
            applied = None
            for line in err.splitlines():            # "Applied N variants" -- what the reference parses (:88-99)
                w = line.split()
                if len(w) >= 3 and w[0] == "Applied" and w[1].isdigit():
                    applied = int(w[1])
            rec[mode] = {"returncode": rc, "sequence": "".join(fa[1:]) if rc == 0 else None, "applied": applied,
                         "stderr": err}
        res["regions"].append(rec)
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"wrote {out}: {len(res['regions'])} regions x 2 modes")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "gen":
        write_inputs(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "pack":
        pack(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)
