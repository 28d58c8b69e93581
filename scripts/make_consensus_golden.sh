#!/bin/bash
# Pins variantformer_amd's in-process consensus (vf_vcf.cpp) to bcftools itself: runs the reference's exact two commands
# (reference utils/data_process.py:27,41-59) on a small genome + donor VCF that cover het / hom indels, 1/2 genotypes of unequal
# lengths, '*', MNPs, multi-allelic snp + indel records under the SNP filter, overlapping and edge-crossing records, and writes
# tests/golden/consensus_bcftools.json.  Needs samtools, bcftools, bgzip, tabix on PATH -- versions 1.21 are what the reference
# image builds (reference Dockerfile:24-48); the versions actually used are recorded in the file.
# NOT runnable in the offline development image (no such binaries): run it wherever the tools exist and commit the JSON;
# tests/test_consensus_bcftools_cpu.py then stops skipping.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"; repo="$(dirname "$here")"
for t in samtools bcftools bgzip tabix; do command -v "$t" >/dev/null || { echo "make_consensus_golden.sh: $t not on PATH" >&2; exit 2; }; done
work="$(mktemp -d)"; trap 'rm -rf "$work"' EXIT
python3 "$here/make_consensus_golden.py" gen "$work"
cd "$work"
{ samtools --version | head -2; bcftools --version | head -2; } > versions.txt
samtools faidx genome.fa
bgzip -c donor.vcf > donor.vcf.gz
tabix -p vcf donor.vcf.gz
i=0
while IFS=$'\t' read -r chrom start end; do
  region="${chrom}:$((start + 1))-${end}"                     # bcftools uses 1-based coordinates (reference :25)
  set +e
  samtools faidx genome.fa "$region" | bcftools consensus -H I -e 'ALT~"<.*>"' donor.vcf.gz > "out_${i}_all.fa" 2> "out_${i}_all.err"
  echo $? > "out_${i}_all.rc"
  samtools faidx genome.fa "$region" | bcftools consensus -H I -e 'ALT~"<.*>" || TYPE!="snp"' donor.vcf.gz > "out_${i}_snp.fa" 2> "out_${i}_snp.err"
  echo $? > "out_${i}_snp.rc"
  set -e
  i=$((i + 1))
done < regions.tsv
python3 "$here/make_consensus_golden.py" pack "$work" "$repo/tests/golden/consensus_bcftools.json"
