"""Constants of the data contract (values from the reference's utils/constants.py:2-31,95-100 and
vocabs/): IUPAC alphabet, the 9 reference cCRE classes, special tokens."""

IUPAC_CODES = {
    "A": ["A"], "C": ["C"], "G": ["G"], "T": ["T"],
    "R": ["A", "G"], "Y": ["C", "T"], "S": ["G", "C"], "W": ["A", "T"], "K": ["G", "T"], "M": ["A", "C"],
    "B": ["C", "G", "T"], "D": ["A", "G", "T"], "H": ["A", "C", "T"], "V": ["A", "C", "G"],
}
IGNORE_CHRS = ["chrX", "chrY", "chrM"]

# second-level context classes, index = label id fed to second_level_context_embedding
REF_CREs = [
    "CTCF-only,CTCF-bound",
    "DNase-H3K4me3",
    "DNase-H3K4me3,CTCF-bound",
    "PLS",
    "PLS,CTCF-bound",
    "dELS",
    "dELS,CTCF-bound",
    "pELS",
    "pELS,CTCF-bound",
]
MAP_REF_CRE_TO_IDX = {cre: idx for idx, cre in enumerate(REF_CREs)}

# multi-class cCRE labels of the training data; inference feeds "Low-DNase" everywhere (reference utils/constants.py:75-89)
CREs = ["Low-DNase", "DNase-only", "CTCF-only,CTCF-bound", "DNase-H3K4me3", "DNase-H3K4me3,CTCF-bound", "PLS",
        "PLS,CTCF-bound", "dELS", "dELS,CTCF-bound", "pELS", "pELS,CTCF-bound"]
MAP_CRE_TO_IDX = {cre: idx for idx, cre in enumerate(CREs)}

SPECIAL_TOKENS = {"pad_token": "<pad>", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>"}
PAD_TOKEN_ID = 0
