"""GPU: the persistent software-pipelined GEMM / conv flavours (mikudance_amd/csrc/gemm_sp.h) forced on for every eligible
problem (MD_GEMM_SP=1 is read once per process, hence the subprocess): parity with fp32 PyTorch + run-to-run bit identity."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("knob", ["MD_GEMM_SP"])
def test_gemm_sp_parity_and_race_screen(knob):
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, **{knob: "1"})
    r = subprocess.run([sys.executable, os.path.join(here, "gemm_sp_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
