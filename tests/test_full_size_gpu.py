"""GPU, BASELINE.json configs[1] size: the kernels that are only selected at full size (the ping-pong GEMM / conv flavour
needs >= 0.88 full rounds of 256 tiles) against the small-tile kernels that the oracle parity tests cover, in situ: one
whole DDIM step (reference UNet write pass, denoising UNet read pass with CFG, DDIM) with MD_GEMM_PP=0 and with the
automatic selection must agree to fp16 accumulation-order noise: relative L2 <= 2e-3, cosine >= 0.99999."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(env_extra, path):
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "full_size_step.py"), path], env=dict(os.environ, **env_extra),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return torch.load(path)


def test_full_size_step_pingpong_vs_small_tile_kernels():
    with tempfile.TemporaryDirectory() as d:
        a = _run({"MD_GEMM_PP": "0"}, os.path.join(d, "a.pt"))
        b = _run({"MD_GEMM_PP": "2"}, os.path.join(d, "b.pt"))
    assert a.shape == b.shape == (1, 4, 16, 96, 96)
    rel = float((a - b).norm() / a.norm())
    cos = float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))
    assert rel <= 2e-3 and cos >= 0.99999, (rel, cos)
