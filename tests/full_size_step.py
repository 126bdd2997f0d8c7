"""One DDIM step of the denoising loop at BASELINE.json configs[1] size (768x768, 16 frames, full-width random-init UNets)
through the product path; writes the latents to argv[1].  tests/test_full_size_gpu.py runs it under different kernel
dispatch settings (read once per process) and compares the results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline  # noqa: E402
from mikudance_amd.selftest import SCHED_KWARGS, build_models  # noqa: E402
from mikudance_amd.synth import synth_inputs  # noqa: E402

dev = torch.device("cuda:0")
ref, den, _, _ = build_models(geom=dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768), device=dev,
                            keep_state_dicts=False)
pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
lat, rl, emb = synth_inputs(16, 96, 96, ctx_len=257, ctx_dim=768, seed=100)
out = pipe.denoise(lat.half().to(dev), rl.half().to(dev), emb.half().to(dev), 1, 3.5)
assert torch.isfinite(out.float()).all()
torch.save(out.float().cpu(), sys.argv[1])
print("DONE", float(out.float().abs().mean()))
