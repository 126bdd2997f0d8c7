"""One UNet-pair evaluation at the G9 shape (full-width UNets, 96x96 latents, f = 2, CFG, literal reference call pattern,
weights / inputs regenerated from the seeds in tests/golden/g9_meta.json) through the product path; writes the prediction
to argv[1].  tests/test_unets_gpu.py runs it in a subprocess under kernel-dispatch settings that are read once per process
(MD_GEMM_SP=1: the persistent one-wave-per-SIMD conv / GEMM / GEGLU kernels of gemm_sp.h on every eligible problem) and compares with the golden
produced by the reference's own modules."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mikudance_amd import ReferenceAttentionControl  # noqa: E402
from mikudance_amd.selftest import build_models  # noqa: E402
from mikudance_amd.synth import synth_inputs  # noqa: E402

meta = json.load(open(os.path.join(HERE, "golden", "g9_meta.json")))
ref, den, _, _ = build_models(geom=dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768),
                              seed_den=meta["seed_den"], seed_ref=meta["seed_ref"], keep_state_dicts=False)
f, (h, w) = meta["frames"], meta["latent"]
lat, rl, emb = synth_inputs(f, h, w, ctx_len=257, ctx_dim=768, seed=meta["seed_inputs"])
writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
reader = ReferenceAttentionControl(den, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
g = rl.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w).cuda().half()
emb_in = emb.repeat((f, 1, 1)).cuda().half()
ref(g, torch.zeros((), dtype=torch.long), encoder_hidden_states=emb_in, return_dict=False)
reader.update(writer)
pred = den(lat.repeat(2, 1, 1, 1, 1).cuda().half(), torch.tensor(meta["timestep"]), encoder_hidden_states=emb_in[:2], return_dict=False)[0]
torch.save(pred.float().cpu(), sys.argv[1])
print("DONE", float(pred.float().abs().mean()))
