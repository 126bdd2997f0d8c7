import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle evaluates many SMALL operators (reduced-width UNets, 16 x 16 latents): on the GPU box's 128 hardware threads
    # ATen's intra-op pool makes them ~7x slower than on 16 (bench.py's thread sweep: 16.5 s vs 2.45 s per configs[0] step; the two
    # 20 / 30-step window tests took 83 + 116 s on 128 threads).  The oracle's results do not depend on the thread count.
    import torch
    torch.set_num_threads(min(torch.get_num_threads(), int(os.environ.get("MD_TEST_THREADS", "16"))))


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
