"""Build-container only (needs /root/reference): times ONE UNet-pair evaluation (reference UNet write pass + denoising UNet
read pass, CFG) at config-1 shape -- full-width SD-1.5 geometry, 32x32 latents, 4 frames, fp32 -- through
  (a) the reference's own modules (imported unmodified; diffusers leaf ops from oracle/shim_diffusers) and
  (b) the CPU restatement oracle/cpu_ref.py,
on the same weights, and checks that they agree.  Purpose: show that the restatement bench.py times as
`cpu_baseline` is not a strawman.  Test infrastructure; prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as G  # noqa: E402  (sets up sys.path for the reference + shim)
from oracle import cpu_ref as O  # noqa: E402
from mikudance_amd.synth import synth_inputs  # noqa: E402
from src.models.mutual_mix_attention import ReferenceAttentionControl  # noqa: E402

torch.set_grad_enabled(False)
FULL = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768)
ref, den, ref_sd, den_sd = G.build_unets(**FULL)
writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
reader = ReferenceAttentionControl(den, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
f, h, w = 4, 32, 32
latents, ref_latents, embeds = synth_inputs(f, h, w, ctx_len=257, ctx_dim=768, seed=100)
x = latents.repeat(2, 1, 1, 1, 1)
g = ref_latents.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w)
emb_in = embeds.repeat((f, 1, 1))


def reference_pair():
    ref(g, torch.zeros((), dtype=torch.long), encoder_hidden_states=emb_in, return_dict=False)
    reader.update(writer)
    out = den(x, torch.tensor(601), encoder_hidden_states=emb_in[:2], return_dict=False)[0]
    reader.clear(); writer.clear()
    return out


def oracle_pair():
    banks, _ = O.reference_unet_forward(ref_sd, g, emb_in)
    banks = {k: v.half().float() for k, v in banks.items()}
    return O.denoising_unet_forward(den_sd, x, torch.tensor(601), embeds, banks, cfg=True)


res = {}
for name, fn in (("reference", reference_pair), ("oracle", oracle_pair)):
    fn()
    t0 = time.perf_counter()
    out = fn()
    res[name] = time.perf_counter() - t0
    res[name + "_out"] = out
rel = float((res["reference_out"] - res["oracle_out"]).norm() / res["reference_out"].norm())
print(json.dumps({"threads": torch.get_num_threads(), "reference_s": res["reference"], "oracle_s": res["oracle"],
                  "ratio": res["oracle"] / res["reference"], "rel_l2": rel}))
