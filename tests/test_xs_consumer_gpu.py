"""The LayerNorm-consumer form of the X-stationary K = 512 kernel is reachable only through an environment switch that
the library reads once per process (VF_GEMM_XS), so it is exercised in a child process: the seq2reg golden-fixture and
production-width parity tests must pass unchanged with every K = 512 consumer GEMM routed through it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_seq2reg_parity_with_the_x_stationary_consumer():
    env = dict(os.environ, VF_GEMM_XS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_model_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "seq2reg_embeddings or production_dims or seq2reg_options"],
                       env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
