"""AutoencoderKL timing at the benchmark size (768x768): F = 16 decodes and 3F + 2 = 50 encodes per clip, the calls the
reference pipeline makes around the denoising loop (src/pipelines/pipeline_mikudance.py:115-130, :456-549).  Full-width
sd-vae-ft-mse geometry, seeded random weights, synthetic inputs.  Reported separately from bench.py's headline
(SURVEY.md 8d).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import AutoencoderKL, _lib  # noqa: E402
from mikudance_amd.synth import synth_state_dict  # noqa: E402

dev = torch.device("cuda:0")
size, frames, batch = int(os.environ.get("MD_VAE_SIZE", "768")), 16, int(os.environ.get("MD_VAE_BATCH", "8"))
vae = AutoencoderKL()
vae.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=77), strict=True)
vae = vae.half().to(dev).eval()
lat = torch.randn(frames, 4, size // 8, size // 8, device=dev).half()
imgs = (torch.rand(3 * frames + 2, 3, size, size, device=dev) * 2 - 1).half()


def decode_all():
    return [vae.decode(lat[i:i + batch]).sample for i in range(0, frames, batch)]


def encode_all():
    return [vae.encode(imgs[i:i + batch]).latent_dist.mean for i in range(0, imgs.shape[0], batch)]


res = {}
for name, fn in (("decode", decode_all), ("encode", encode_all)):
    out = fn()
    torch.cuda.synchronize()
    assert all(torch.isfinite(o.float()).all() for o in out)
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    res[name + "_ms_per_clip"] = (time.perf_counter() - t0) * 1e3
_lib.PROFILER.start()
decode_all(); encode_all()
torch.cuda.synchronize()
_lib.PROFILER.stop()
prof = _lib.PROFILER.summary()
fam = {}
for label, d in prof.items():
    f = fam.setdefault(label.split(" ")[0], dict(ms=0.0, flops=0.0))
    f["ms"] += d["ms"]; f["flops"] += d["flops"]
res.update(size=size, decodes=frames, encodes=imgs.shape[0], batch=batch,
           families={k: dict(ms=round(v["ms"], 2), tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None)
                     for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])})
if os.environ.get("MD_VAE_DUMP"):                 # per-shape table of one decode + encode pass (profiles/r04_vae_shapes.txt)
    with open(os.environ["MD_VAE_DUMP"], "w") as fh:
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            fh.write(f"{v['ms']:10.3f} ms/clip  {v['count']:6d} launches  {(v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else 0:8.1f} TF  "
                     f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f} GB/s  {k}\n")
print(json.dumps(res))
