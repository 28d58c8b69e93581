"""Output stage of the VEP flow (VariantProcessor.compile_predictions / format_scores / eqtl_scores, variant and pair
bookkeeping) against the reference's own outputs on seeded predictions (tests/golden/variantprocessor.npz, written by
tests/golden/make_golden.py:variantprocessor_fixture)."""
import os

import numpy as np
import pandas as pd
import pytest
import yaml

from tests import vep_artifacts as va
from variantformer_amd.datasets.vepdataset import Variant
from variantformer_amd.processors.multi_datasets_loader import MultiDatasetsLoader
from variantformer_amd.processors.variantprocessor import VariantProcessor
from variantformer_amd.utils.config import Config

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(os.path.join(HERE, "golden", "variantprocessor.npz"), allow_pickle=False))


def _processor(tmp_path, genes=None):
    cfg_dir = tmp_path / "configs"
    cfg_dir.mkdir(exist_ok=True)
    genes_csv = tmp_path / "genes.csv"
    (genes if genes is not None else pd.DataFrame(
        [{"gene_id": "ENSG_PLUS.3", "gene_name": "p", "chromosome": "chr1", "start": 1000, "end": 2600, "strand": "+"},
         {"gene_id": "ENSG_MINUS.1", "gene_name": "m", "chromosome": "chr1", "start": 3200, "end": 5200, "strand": "-"},
         {"gene_id": "ENSG_FAR.1", "gene_name": "f", "chromosome": "chr1", "start": 5_000_000, "end": 5_001_000, "strand": "+"},
         {"gene_id": "ENSG_OTHER.1", "gene_name": "o", "chromosome": "chr2", "start": 100, "end": 900, "strand": "+"}])
     ).to_csv(genes_csv, index=False)
    bed = tmp_path / "cres.bed"
    bed.write_text("chr1\t700\t760\tEH1\t0\t.\t700\t760\t255,0,0\tPLS\n")
    block = {"dataset": {"max_length": 20, "max_chunks": 6, "cre_neighbour_hood": 10, "gencode_v24": str(genes_csv),
                         "gene_upstream_neighbour_hood": 100, "gene_downstream_neighbour_hood": 2500},
             "model": {"model_class": "Seq2GenePredictorCombinedModulator", "checkpoint_path": str(tmp_path / "m.pth"),
                       "precision": "bf16-mixed", "emb_dim": 6, "cre_tokenizer": {"path": str(tmp_path / "t.pth")},
                       "gene_tokenizer": {"path": str(tmp_path / "t.pth")}}}
    with open(cfg_dir / "vf_model.yaml", "w") as f:
        yaml.safe_dump({"v4_pcg": block}, f)
    with open(cfg_dir / "veploader.yaml", "w") as f:
        yaml.safe_dump({"CRE_BED": str(bed), "fasta_path": str(tmp_path / "g.fa"), "af_path": str(tmp_path / "af"),
                        "precision": "bf16-mixed", "dataloader": {"num_workers": 0, "pin_memory": False}}, f)
    return VariantProcessor(config_dir=str(cfg_dir), require_gpu=False)


def _check_frame(df, golden, prefix):
    cols = [str(c) for c in golden[f"{prefix}.__columns__"]]
    assert [str(c) for c in df.columns] == cols
    for c in cols:
        want = golden[f"{prefix}.{c}"]
        col = df[c]
        if want.dtype.kind in "US":
            assert [str(v) for v in col] == list(want), c
        elif want.ndim == 2:
            np.testing.assert_array_equal(np.stack(list(col)), want, err_msg=c)
        elif want.dtype.kind == "f":
            np.testing.assert_allclose(col.to_numpy(dtype=float), want, rtol=1e-12, atol=0, equal_nan=True, err_msg=c)
        else:
            np.testing.assert_array_equal(col.to_numpy(), want, err_msg=c)


@pytest.mark.parametrize("tag,with_sample", [("pop", False), ("sample", True)])
def test_output_stage_matches_reference(tmp_path, golden, tag, with_sample):
    vp = _processor(tmp_path)
    pairs, preds, af = va.make_vp_case(Variant, with_sample=with_sample)
    va.write_af_tables(af, str(tmp_path / "af"))
    vp.gene_variant_pairs = pairs
    vp.config.output_location = str(tmp_path)
    long_df = vp.compile_predictions(preds, vcf_path="x.vcf.gz" if with_sample else None)
    assert os.path.exists(tmp_path / "vep_VF.parquet")
    _check_frame(long_df, golden, f"{tag}.long")
    wide_df = vp.format_scores(long_df.copy())
    _check_frame(wide_df, golden, f"{tag}.wide")
    score_df = vp.eqtl_scores(wide_df.copy())
    _check_frame(score_df, golden, f"{tag}.score")
    if not with_sample:
        assert "VF-agg-log2fc-weighted" in score_df.columns and score_df["VF-agg-log2fc-weighted"].notna().any()


def test_variants_and_pairs(tmp_path):
    vp = _processor(tmp_path)
    table = pd.DataFrame({"chr": ["1", "chr1", "2"], "pos": [1075, 3350, 500], "ref": ["A", "C", "G"], "alt": ["G", "T", "A"],
                          "tissue": ["liver,thyroid", "whole blood", "lung"], "gene_id": ["ENSG_PLUS.3,ENSG_NOPE", "", ""]})
    with pytest.raises(ValueError, match="Column tissue"):
        vp.load_variants(table.drop(columns=["tissue"]))
    variants = vp.load_variants(table)
    assert [v.chrom for v in variants] == ["chr1", "chr2", "chr1"]           # sorted by the raw chrom strings, then pos
    assert variants[0].gene_id == ["ENSG_PLUS", "ENSG_NOPE"] and variants[0].tissue == [vp.tissue_vocab["liver"], vp.tissue_vocab["thyroid"]]
    with pytest.raises(RuntimeError, match="load_annotations"):
        vp.multi_data_loader.get_probable_genes(variants[0])
    vp.multi_data_loader.load_annotations()
    assert list(vp.multi_data_loader.all_cres.columns)[:4] == ["chromosome", "start", "end", "name"]
    near = vp.multi_data_loader.get_probable_genes(variants[0])
    assert [g["gene_id"] for g in near] == ["ENSG_PLUS.3", "ENSG_MINUS.1"]     # within 1 Mb; ENSG_FAR is 5 Mb away
    pairs, mapped = vp.build_pairs(variants)
    # variant 0 is restricted to ENSG_PLUS by its gene_id list; an EMPTY gene_id cell parses to [""] and therefore
    # matches no gene (reference create_variant_objects :103-105 + initialize :157-163)
    assert mapped == 1 and len(pairs) == 6
    free = vp.load_variants(table.drop(columns=["gene_id"]))
    pairs, mapped = vp.build_pairs(free)
    assert mapped == 3 and len(pairs) == (2 + 1 + 2) * 6          # chr1:1075 -> 2 genes, chr2:500 -> 1, chr1:3350 -> 2
    assert [p["population"] for p in pairs[:6]] == ["REF_HG38", "EAS", "EUR", "AFR", "SAS", "AMR"]
    assert pairs[1]["sample_name"] == "HG00404" and pairs[0]["vcf_path"] is None
    pairs, _ = vp.build_pairs(variants[:1], vcf_path="d.vcf.gz", sample_name="donor")
    assert [(p["population"], p["sample_name"], p["vcf_path"]) for p in pairs] == \
        [("SAMPLE", "donor", "d.vcf.gz"), ("REF_HG38", "hg38", None)]
    # output path rules
    vp.config.output_location = str(tmp_path)
    assert vp._get_variant_output_path().endswith("vep_VF.parquet")
    vp.config.variants_file, vp.config.chunks, vp.config.chunk_id = "/data/my_vars.tsv", 4, 2
    assert vp._get_variant_output_path().endswith("my_vars_chunk2_VF.parquet")


def test_loader_is_standalone(tmp_path):
    ml = MultiDatasetsLoader(Config({"gencode": "x", "all_cres": "y"}))
    df = ml._load_variants(pd.DataFrame({"chrom": ["chr2", "chr1"], "pos": [5, 9], "ref": ["A", "C"], "alt": ["C", "G"],
                                         "tissue": ["liver", "liver"]}))
    assert list(df["chrom"]) == ["chr1", "chr2"]
