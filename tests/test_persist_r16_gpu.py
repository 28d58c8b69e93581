"""LayerNorm-producer GEMMs with a 16-bit residual (operand-type stream copy, VF_LN_PRODUCER_R16, or the fp16 trunk copy,
VF_LN_PRODUCER_T16) in the persistent form of the 256x256 kernel (gemm8x_kernel<RES_F32, ., LN>).  Which producers take
that form are once-per-process environment switches (VF_GEMM_PERSIST_R16, VF_GEMM_PERSIST_T16), so both settings are
exercised in child processes: the bit-exact producer / race-screen cases of test_ops_gpu.py and the
model-level parity cases must pass unchanged with the switch forced on (2 = every such producer) and off (0)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["0", "2"])
def test_16bit_residual_producers_in_either_kernel_form(mode):
    env = dict(os.environ, VF_GEMM_PERSIST_R16=mode, VF_GEMM_PERSIST_T16="1" if mode == "2" else "0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_ops_gpu.py"),
                        os.path.join(REPO, "tests", "test_model_gpu.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "gemm_ln_producer or ln_fold or production_dims or full_depth or seq2reg_embeddings"],
                       env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
