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
there.  `variant_type=None` (the vcf2exp mode) lets insertions / deletions through in the reference.  `indel_policy` selects
their treatment: "bcftools" (default) = the `-H I` rule of bcftools consensus restated in vf_vcf.cpp (IUPAC codes for
equal-length alleles, the first non-REF genotype allele otherwise, i.e. a het indel applies its ALT), so a real donor
VCF runs as it does in the reference; "error" refuses a region that holds such a genotype (strict mode: nothing is
guessed), "skip" keeps substitutions only, "first_allele" is round 1's guess.

The genome may be plain FASTA or bgzip-compressed FASTA (`.fa.gz` with its `.fai` and `.gzi`, which is what the shipped
vcfloader.yaml / veploader.yaml point at, like `samtools faidx` reads it).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import pandas as pd

from .. import _lib

_INDEL_POLICIES = {"error": 0, "first_allele": 1, "skip": 2, "bcftools": 3}
DEFAULT_INDEL_POLICY = "bcftools"


class FastaReader:
    """Random access to a FASTA through its samtools .fai index (built in memory when the .fai is missing).
    bgzip-compressed FASTA (BGZF: `.fa.gz` + `.fai` + `.gzi`, as `samtools faidx` writes them) is read block-wise: the
    .fai offsets address the UNCOMPRESSED stream and the .gzi (or, when it is missing, one pass over the block headers)
    maps them to BGZF blocks, each inflated with zlib on demand.  Plain gzip cannot be read randomly and is refused."""

    def __init__(self, path: str):
        self.path = path
        self.index = {}
        with open(path, "rb") as f:
            magic = f.read(18)
        self.bgzf = magic[:2] == b"\x1f\x8b"
        if self.bgzf and not (len(magic) == 18 and magic[3] & 4 and magic[12:14] == b"BC"):
            raise ValueError(f"{path} is gzip but not BGZF: random access needs `bgzip` compression (or a plain FASTA); "
                             "recompress with `bgzip` and index with `samtools faidx`")
        if self.bgzf:
            self._load_block_index()
            self._block_cache = (-1, b"")
        self._fh = open(path, "rb")
        self._lock = threading.Lock()
        fai = path + ".fai"
        if os.path.exists(fai):
            with open(fai) as f:
                for line in f:
                    p = line.rstrip("\n").split("\t")
                    if len(p) >= 5:
                        self.index[p[0]] = tuple(int(v) for v in p[1:5])      # length, offset, linebases, linewidth
        else:
            self._scan()

    # -- BGZF ------------------------------------------------------------------------------------------
    def _load_block_index(self):
        """self._blk_c / self._blk_u: compressed / uncompressed start offset of every BGZF block (sorted)."""
        import struct
        gzi = self.path + ".gzi"
        if os.path.exists(gzi):
            with open(gzi, "rb") as f:
                raw = f.read()
            n = struct.unpack_from("<Q", raw, 0)[0]
            vals = struct.unpack_from("<%dQ" % (2 * n), raw, 8)
            self._blk_c, self._blk_u = [0] + list(vals[0::2]), [0] + list(vals[1::2])      # block 0 is implicit
            return
        self._blk_c, self._blk_u = [], []
        c = u = 0
        size = os.path.getsize(self.path)
        with open(self.path, "rb") as f:
            while c < size:
                f.seek(c)
                h = f.read(18)
                if len(h) < 18 or h[:2] != b"\x1f\x8b" or h[12:14] != b"BC":
                    raise ValueError(f"{self.path}: malformed BGZF block at offset {c}")
                bsize = struct.unpack_from("<H", h, 16)[0] + 1
                f.seek(c + bsize - 4)
                isize = struct.unpack("<I", f.read(4))[0]
                if isize:                       # the empty EOF block carries no data
                    self._blk_c.append(c)
                    self._blk_u.append(u)
                c += bsize
                u += isize

    def _block(self, i: int) -> bytes:
        import struct
        import zlib
        if self._block_cache[0] == i:
            return self._block_cache[1]
        self._fh.seek(self._blk_c[i])
        h = self._fh.read(18)
        bsize = struct.unpack_from("<H", h, 16)[0] + 1
        xlen = struct.unpack_from("<H", h, 10)[0]
        body = self._fh.read(bsize - 18)
        data = zlib.decompress(body[xlen - 6: -8], -15)          # raw deflate between the header and CRC32 / ISIZE
        self._block_cache = (i, data)
        return data

    def _read_uncompressed(self, first: int, last: int) -> bytes:
        import bisect
        i = bisect.bisect_right(self._blk_u, first) - 1
        out = []
        while first < last and i < len(self._blk_u):
            data = self._block(i)
            a = first - self._blk_u[i]
            piece = data[a: a + (last - first)]
            if not piece:
                break
            out.append(piece)
            first += len(piece)
            i += 1
        return b"".join(out)

    def _scan(self):
        import gzip
        name, length, offset, lb, lw, pos = None, 0, 0, 0, 0, 0
        with (gzip.open(self.path, "rb") if self.bgzf else open(self.path, "rb")) as f:
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
            if self.bgzf:
                raw = self._read_uncompressed(first, last)
            else:
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

    def consensus(self, chrom: str, start0: int, ref: str, snp_only: bool, indel_policy: str = DEFAULT_INDEL_POLICY):
        """(consensus string, n_applied); raises ConsensusError with .code on VF_CONS_* failures."""
        raw = ref.encode("ascii")
        cap = len(raw) + 4096
        policy = _INDEL_POLICIES[indel_policy]
        while True:
            out = C.create_string_buffer(cap)
            n_applied = C.c_int64(0)
            n = self._lib.vf_vcf_consensus(self._h, chrom.encode(), int(start0), raw, len(raw), int(bool(snp_only)),
                                           {1: 1, 3: 2}.get(policy, 0), out, cap, C.byref(n_applied))
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
                -3: "insertion / deletion genotype in the region under indel_policy='error'",
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
                 indel_policy: str = DEFAULT_INDEL_POLICY, sample: str | None = None):
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
        df = pd.DataFrame(rows, columns=["chrom", "start_cre", "end_cre", "sequence", "cCRE"])
        if len(df) and not df["start_cre"].is_monotonic_increasing:
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
