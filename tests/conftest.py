import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def full(golden_dir):
    """FULL-WIDTH SD-1.5 geometry (320/640/1280/1280, 257 x 768 context), seeded weights (checksums pinned by g8_meta.json), built
    ONCE per session and shared by every in-process full-width test (tests/test_unets_gpu.py, tests/test_e2e_parity_gpu.py,
    tests/test_full_size_gpu.py): (ref, den, ref_sd, den_sd) with the fp32 state dicts the oracle is fed with."""
    import json
    from mikudance_amd.selftest import build_models
    meta = json.load(open(os.path.join(golden_dir, "g8_meta.json")))
    geom = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768)
    ref, den, ref_sd, den_sd = build_models(geom=geom, seed_den=meta["seed_den"], seed_ref=meta["seed_ref"])
    cs = lambda sd: float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(cs(den_sd) - meta["checksum_den"]) < 1e-6 * meta["checksum_den"]
    assert abs(cs(ref_sd) - meta["checksum_ref"]) < 1e-6 * meta["checksum_ref"]
    return ref, den, ref_sd, den_sd
