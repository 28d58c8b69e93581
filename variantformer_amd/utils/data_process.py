"""In-process replacement of the reference's sequence extraction (utils/data_process.py:9-482).

The reference starts `samtools faidx REGION | bcftools consensus -H I -e FILTER sample.vcf.gz` once per CRE window and
per gene body and fans the ~1000 subprocess pairs of a gene over a process pool; SURVEY.md section 8f ranks this as the
end-to-end bottleneck.  Here the sample's VCF is parsed once (C++, `vf_vcf_open`, kept per path), the reference
genome is read through its .fai index, and a region's consensus is one C call (`vf_vcf_consensus`).

Same class / method names and return shapes as the reference: `ExtractSeqFromBed(neighbour_hood, ref_fasta,
upstream_neighbour_hood).process_subject(vcf_file, bed_regions, variant_type)` -> DataFrame(chrom, start_cre,
end_cre, sequence, cCRE) sorted by start; `.process_gene(gene_info, vcf_file, variant_type)` -> str.  As in the
reference, a region whose consensus fails (REF allele differs from the genome) falls back to the reference sequence
with a printed note (reference :73-88, :441-453).

PARITY UNPINNED against bcftools itself (absent offline): see variantformer_amd/csrc/vf_vcf.cpp for the restated
semantics.  `variant_type="SNP"` (the VEP loader's mode) touches single-base substitutions only and is fully specified
there.  `variant_type=None` (the vcf2exp mode) lets insertions / deletions through in the reference; here that needs
an explicit `indel_policy` ("first_allele" or "skip"), otherwise such a region raises -- nothing is guessed silently.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import pandas as pd

from .. import _lib

_INDEL_POLICIES = {"error": 0, "first_allele": 1, "skip": 2}


class FastaReader:
    """Random access to a FASTA through its samtools .fai index (built in memory when the .fai is missing)."""

    def __init__(self, path: str):
        self.path = path
        self.index = {}
        fai = path + ".fai"
        if os.path.exists(fai):
            with open(fai) as f:
                for line in f:
                    p = line.rstrip("\n").split("\t")
                    if len(p) >= 5:
                        self.index[p[0]] = tuple(int(v) for v in p[1:5])      # length, offset, linebases, linewidth
        else:
            self._scan()
        self._fh = open(path, "rb")
        self._lock = threading.Lock()

    def _scan(self):
        name, length, offset, lb, lw, pos = None, 0, 0, 0, 0, 0
        with open(self.path, "rb") as f:
            for raw in f:
                if raw.startswith(b">"):
                    if name is not None:
                        self.index[name] = (length, offset, lb, lw)
                    name = raw[1:].split()[0].decode()
                    length, offset, lb, lw = 0, pos + len(raw), 0, 0
                else:
                    body = raw.rstrip(b"\r\n")
                    if lb == 0:
                        lb, lw = len(body), len(raw)
                    length += len(body)
                pos += len(raw)
        if name is not None:
            self.index[name] = (length, offset, lb, lw)

    def length(self, chrom: str) -> int:
        return self.index[chrom][0]

    def fetch(self, chrom: str, start0: int, end0: int) -> str:
        """Bases of the 0-based half-open interval, clipped to the chromosome (as `samtools faidx chr:start0+1-end0`)."""
        if chrom not in self.index:
            raise KeyError(f"{chrom} not in {self.path}")
        length, offset, lb, lw = self.index[chrom]
        start0, end0 = max(0, int(start0)), min(int(end0), length)
        if end0 <= start0:
            return ""
        first = offset + (start0 // lb) * lw + start0 % lb
        last = offset + ((end0 - 1) // lb) * lw + (end0 - 1) % lb + 1
        with self._lock:
            self._fh.seek(first)
            raw = self._fh.read(last - first)
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode("ascii")


class VCFHandle:
    """One parsed VCF (all records of one sample, in memory)."""

    def __init__(self, path: str, sample: str | None = None):
        self._lib = _lib.load()
        self._h = self._lib.vf_vcf_open(path.encode(), (sample or "").encode())
        if not self._h:
            raise _lib.VFError(f"cannot read VCF {path}" + (f" (sample {sample})" if sample else ""))
        self.path = path

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.vf_vcf_close(self._h)
            self._h = None

    def num_records(self, chrom: str | None = None) -> int:
        return int(self._lib.vf_vcf_num_records(self._h, (chrom or "").encode()))

    def consensus(self, chrom: str, start0: int, ref: str, snp_only: bool, indel_policy: str = "error"):
        """(consensus string, n_applied); raises ConsensusError with .code on VF_CONS_* failures."""
        raw = ref.encode("ascii")
        cap = len(raw) + 4096
        policy = _INDEL_POLICIES[indel_policy]
        while True:
            out = C.create_string_buffer(cap)
            n_applied = C.c_int64(0)
            n = self._lib.vf_vcf_consensus(self._h, chrom.encode(), int(start0), raw, len(raw), int(bool(snp_only)),
                                           1 if policy == 1 else 0, out, cap, C.byref(n_applied))
            if n == -1 and cap < 16 * (len(raw) + 4096):      # insertions overflowed the buffer
                cap *= 4
                continue
            break
        if n == -3 and policy == 2:                            # "skip": substitutions only
            return self.consensus(chrom, start0, ref, True, "error")
        if n < 0:
            raise ConsensusError(int(n), f"{chrom}:{start0 + 1}-{start0 + len(raw)}")
        return out.raw[:n].decode("ascii"), int(n_applied.value)


class ConsensusError(RuntimeError):
    MESSAGES = {-1: "bad argument", -2: "REF allele does not match the reference genome",
                -3: "insertion / deletion genotype in the region and no indel_policy given",
                -4: "genotype allele index beyond the ALT list"}

    def __init__(self, code: int, region: str):
        super().__init__(f"consensus of {region} failed: {self.MESSAGES.get(code, code)}")
        self.code = code


_FASTA_CACHE: dict = {}
_VCF_CACHE: dict = {}
_CACHE_LOCK = threading.Lock()


def open_fasta(path: str) -> FastaReader:
    with _CACHE_LOCK:
        if path not in _FASTA_CACHE:
            _FASTA_CACHE[path] = FastaReader(path)
        return _FASTA_CACHE[path]


def open_vcf(path: str, sample: str | None = None) -> VCFHandle:
    key = (path, sample, os.path.getmtime(path))
    with _CACHE_LOCK:
        if key not in _VCF_CACHE:
            _VCF_CACHE[key] = VCFHandle(path, sample)
        return _VCF_CACHE[key]


class ExtractSeqFromBed:
    def __init__(self, neighbour_hood: int, ref_fasta: str, upstream_neighbour_hood: int = None,
                 indel_policy: str = "error", sample: str | None = None):
        self.neighbour_hood = neighbour_hood
        self.ref_fasta = ref_fasta
        self.upstream_neighbour_hood = upstream_neighbour_hood
        if indel_policy not in _INDEL_POLICIES:
            raise ValueError(f"indel_policy must be one of {sorted(_INDEL_POLICIES)}")
        self.indel_policy = indel_policy
        self.sample = sample

    # -- one region ---------------------------------------------------------------------------------
    def _consensus(self, chrom, start0, end0, vcf_file, variant_type):
        """(sequence or None, mutations) of [start0, end0) -- reference :17-101 / :367-467."""
        try:
            ref = open_fasta(self.ref_fasta).fetch(chrom, start0, end0)
        except KeyError as e:
            print(f"{chrom}:{start0 + 1}-{end0}")
            print(f"\nError reading the reference genome: {e}")
            return None, 0
        if not vcf_file:
            return ref, 0
        try:
            return open_vcf(vcf_file, self.sample).consensus(chrom, start0, ref, variant_type == "SNP", self.indel_policy)
        except ConsensusError as e:
            if e.code == -3:
                raise
            print(f"{chrom}:{start0 + 1}-{end0}")
            print(f"\nError building the consensus: {e}")
            print("Falling back to ref genome")
            return ref, 0

    def apply_bcftools_consensus(self, region, vcf_file, reference_fasta=None, variant_type: str = None):
        start = max(0, int(region.start) - self.neighbour_hood)
        end = int(region.end) + self.neighbour_hood
        return self._consensus(region.chrom, start, end, vcf_file, variant_type)

    def process_region(self, args):
        region, vcf_file, reference_fasta, variant_type = args
        seq, mutations = self.apply_bcftools_consensus(region, vcf_file, reference_fasta, variant_type=variant_type)
        if seq:
            return {"chrom": region.chrom, "start_cre": max(0, region.start - self.neighbour_hood),
                    "end_cre": region.end + self.neighbour_hood, "sequence": seq, "cCRE": region.cCRE}, mutations
        return None, None

    # -- all CRE windows of a gene ---------------------------------------------------------------------
    def process_subject(self, vcf_file: str, bed_regions: pd.DataFrame, variant_type: str = None):
        rows = []
        for region in bed_regions.itertuples(index=False):           # .chrom .start .end .cCRE, as a Series row would
            d, _ = self.process_region((region, vcf_file, self.ref_fasta, variant_type))
            if d:
                rows.append(d)
        df = pd.DataFrame(rows)
        if not df["start_cre"].is_monotonic_increasing:
            df = df.sort_values(by=["chrom", "start_cre"], ascending=True).reset_index(drop=True)
        return df

    # -- gene body -------------------------------------------------------------------------------------
    def apply_bcftools_consensus_to_gene(self, chrom, strand, start, end, vcf_file, variant_type: str = None):
        if strand == "-":
            start = max(int(start), int(end) - self.neighbour_hood)
            end = int(end) + self.upstream_neighbour_hood
        else:
            start = max(0, int(start) - self.upstream_neighbour_hood)
            end = min(int(end), int(start) + self.neighbour_hood)     # NB: `start` is already shifted (reference :396-401)
        seq, _ = self._consensus(chrom, start, end, vcf_file, variant_type)
        if seq is None:
            raise ValueError(f"Error extracting {chrom}:{start + 1}-{end}")
        return seq

    def process_gene(self, gene_info, vcf_file, variant_type: str = None):
        return self.apply_bcftools_consensus_to_gene(chrom=gene_info["chromosome"], strand=gene_info["strand"],
                                                     start=gene_info["start"], end=int(gene_info["end"]),
                                                     vcf_file=vcf_file, variant_type=variant_type)
