"""GPU: the persistent software-pipelined GEMM / conv flavours (mikudance_amd/csrc/gemm_sp.h) forced on for every eligible
problem (MD_GEMM_SP=1 is read once per process, hence the subprocess): parity with fp32 PyTorch + run-to-run bit identity."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile", ["auto", "192x320", "192x256", "128x256", "256x128", "192x128"])
def test_gemm_sp_parity_and_race_screen(tile):
    """auto: the launcher's own choice between the 192 x 320, 192 x 256 and 128 x 256 tiles; the other runs pin one of them wherever
    it is eligible (MD_GEMM_SP_NT is the A/B override of that choice), so every case with N % 1280 == 0 is checked on all of them
    (256x128: the tile of the N % 128 == 0-only layers, 192x128: the tile of the under-filled 12 x 12 level; both pinned on every
    N % 128 == 0 case)."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MD_GEMM_SP="1", MD_GEMM_SP_NT={"auto": "0", "192x320": "5", "192x256": "4", "128x256": "2", "256x128": "42", "192x128": "32"}[tile])
    r = subprocess.run([sys.executable, os.path.join(here, "gemm_sp_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
