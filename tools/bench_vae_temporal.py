"""AutoencoderKLTemporalDecoder (--video_decoder) timing at the headline size: ONE chunk of 16 frames at 96 x 96 latents -> 768 x 768 pixels,
the call src/pipelines/pipeline_mikudance.py:132-150 makes per 16 frames.  Published geometry (97.7 M parameters), seeded random weights.
Prints one JSON line (profiles/r05_vae_temporal.json)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import AutoencoderKLTemporalDecoder, _lib  # noqa: E402
from mikudance_amd.synth import synth_state_dict  # noqa: E402

dev = torch.device("cuda:0")
latent, frames = int(os.environ.get("MD_VAE_LATENT", "96")), 16
vae = AutoencoderKLTemporalDecoder()
vae.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=31), strict=True)
vae = vae.half().to(dev).eval()
z = torch.randn(frames, 4, latent, latent, device=dev).half()
out = vae.decode(z, num_frames=frames).sample
torch.cuda.synchronize()
assert torch.isfinite(out.float()).all() and tuple(out.shape) == (frames, 3, 8 * latent, 8 * latent)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    vae.decode(z, num_frames=frames)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
_lib.PROFILER.start()
vae.decode(z, num_frames=frames)
torch.cuda.synchronize()
_lib.PROFILER.stop()
fam = {}
for label, d in _lib.PROFILER.summary().items():
    f = fam.setdefault(label.split(" ")[0], dict(ms=0.0, flops=0.0))
    f["ms"] += d["ms"]; f["flops"] += d["flops"]
print(json.dumps({"what": "AutoencoderKLTemporalDecoder.decode, one chunk", "frames": frames, "latent": [latent, latent], "pixels": [8 * latent, 8 * latent],
                  "ms_per_chunk_median": sorted(ts)[1], "ms_per_chunk_all": [round(t, 1) for t in ts], "peak_hbm_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
                  "families": {k: dict(ms=round(v["ms"], 2), tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None)
                               for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}, "device": torch.cuda.get_device_name(0)}))
