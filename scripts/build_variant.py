"""Build a variant of the tuning library with extra defines (library A/Bs on one box):
    python scripts/build_variant.py NAME -DFOO=1 -DBAR=2   ->   variantformer_amd/csrc/libvf_hip_NAME.so
Load it with VF_LIB=libvf_hip_NAME.so scripts/attn_bench.py / scripts/ab_lib.py; the product never loads it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variantformer_amd.csrc import build
name, defs = sys.argv[1], sys.argv[2:]
print(build._build(os.path.join(build.HERE, f"libvf_hip_{name}.so"), ["-DVF_TUNING"] + defs, False, f"var_{name}_"))
