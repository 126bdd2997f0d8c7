"""End-to-end parity of the denoising loop at BASELINE.json configs[1]'s own width and spatial size ("fp16 latents after
20 steps vs fp32 restatement: relative L2 <= 3e-2 and cosine >= 0.999").

    python tests/e2e_parity.py --frames 16 --steps 20 --out profiles/r04_e2e_parity.json       (GPU box)

Three evaluations of the SAME clip (full-width SD-1.5 geometry, 96 x 96 latents, CFG 3.5, seeded weights / inputs):
  hip   the product path (mikudance_amd, HIP kernels through the C ABI), fp16
  o32   oracle/cpu_ref.py in fp32, evaluated through PyTorch-ROCm on the GPU (20 steps x 16 frames are ~2 PFLOP: hours on CPU
        cores).  This is THE CHECKER: the restatement pinned to the reference's goldens on the CPU, fp32 weights, fp32 arithmetic
        (rocBLAS fp32 GEMMs, PyTorch's im2col convolution with MIOpen switched off, explicit softmax(QK^T)V in fp32)
  o16   the same restatement with fp16 weights and fp16 tensors (PyTorch rounds after every operator): what the reference's own
        fp16 run (weight_dtype fp16, scripts/inference_video.py:66-69) does to the arithmetic -- the yardstick for how much of
        hip's distance from o32 is fp16 itself
and the per-step curves of relative L2 / cosine between them.  tests/test_e2e_parity_gpu.py asserts the tolerance on the
f = 4 run; the f = 16 record is committed under profiles/.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline  # noqa: E402
from mikudance_amd.selftest import SCHED_KWARGS, build_models, cosine, rel_l2  # noqa: E402
from mikudance_amd.synth import synth_inputs  # noqa: E402

FULL = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768)


def _sliced(cls, lo, hi):
    """A scheduler that keeps entries lo:hi of its N-step timestep list (N itself, and with it prev_t = t - 1000 // N, unchanged): lets a
    test run ONE step of configs[4]'s 30-step schedule -- e.g. the second, t = 966 -> prev_t = 933, which is NOT the list's next entry
    932 -- at full size."""
    class Sliced(cls):
        def set_timesteps(self, n, *a, **k):
            r = super().set_timesteps(n, *a, **k)
            self.timesteps = self.timesteps[lo:hi]
            return self.timesteps if r is not None else None
    return Sliced


def _oracle_run(O, ref_sd, den_sd, inputs, steps, guidance, dtype, dev, **win):
    """oracle/cpu_ref.denoise_loop on `dev` in `dtype`; returns (final latents, [latents after each step]) as fp32 CPU tensors."""
    cast = lambda sd: {k: v.to(device=dev, dtype=dtype if v.is_floating_point() else v.dtype) for k, v in sd.items()}
    rs, ds = cast(ref_sd), cast(den_sd)
    lat, rl, emb = (t.to(device=dev, dtype=dtype) for t in inputs)
    curve = []
    # MIOpen off: PyTorch's own im2col + rocBLAS convolution is exact fp32 arithmetic and needs no per-shape kernel search /
    # compilation on a fresh box (MD_ORACLE_MIOPEN=1 switches it back on)
    ctx = contextlib.nullcontext() if os.environ.get("MD_ORACLE_MIOPEN") == "1" else torch.backends.cudnn.flags(enabled=False)
    with torch.no_grad(), ctx:
        out = O.denoise_loop(rs, ds, lat, rl, emb, steps, guidance_scale=guidance, reduced=True,
                             on_step=lambda t, x: curve.append(x.float().cpu()), **win)
    torch.cuda.synchronize()
    del rs, ds
    torch.cuda.empty_cache()
    return out.float().cpu(), curve


def run(frames=4, steps=20, latent=96, guidance=3.5, models=None, with_fp16_oracle=True, seed=100, geom=FULL, ctx=(257, 768),
        window=None, log=None, step_slice=None):
    """Returns the record described in the module docstring.  models = (ref, den, ref_sd, den_sd) or None (built here).
    step_slice = (lo, hi): only entries lo:hi of the `steps`-step schedule are executed (on both sides)."""
    from oracle import cpu_ref as O                                      # the checker
    dev = torch.device("cuda:0")
    say = log or (lambda *a: None)
    if models is None:
        models = build_models(geom=geom, device=dev, keep_state_dicts=True)
    ref, den, ref_sd, den_sd = models
    win = dict(window or {})
    lat, rl, emb = synth_inputs(frames, latent, latent, ctx_len=ctx[0], ctx_dim=ctx[1], seed=seed)
    inputs = tuple(t.half().float() for t in (lat, rl, emb))            # every evaluation starts from the same fp16-representable values
    sched_cls, osched = DDIMScheduler, None
    n_run = steps
    if step_slice is not None:
        sched_cls = _sliced(DDIMScheduler, *step_slice)
        osched = _sliced(O.DDIM, *step_slice)()
        n_run = len(range(steps)[step_slice[0]:step_slice[1]])
        win["scheduler"] = osched
    pipe = MikuDanceVideoPipeline(None, None, ref, den, sched_cls(**SCHED_KWARGS))
    hip_curve = []
    t0 = time.time()
    hip = pipe.denoise(*(t.half().to(dev) for t in inputs), steps, guidance, callback=lambda i, t, x: hip_curve.append(x.float().cpu()),
                       **{k: v for k, v in win.items() if k != "scheduler"}).float().cpu()
    torch.cuda.synchronize()
    say(f"hip  {time.time() - t0:7.1f} s")
    t0 = time.time()
    o32, c32 = _oracle_run(O, ref_sd, den_sd, inputs, steps, guidance, torch.float32, dev, **win)
    say(f"o32  {time.time() - t0:7.1f} s")
    rec = {"config": {"frames": frames, "ddim_steps": steps, "steps_executed": list(step_slice) if step_slice else "all", "latent": [latent, latent], "guidance": guidance, "width": dict(geom)["block_out_channels"],
                      "context_tokens": ctx[0], "windows": {k: v for k, v in win.items() if k != "scheduler"} or "single", "weights": "N(0, 1/fan_in) seeds 1234/4321", "seed_inputs": seed},
           "tolerance": {"rel_l2": 3e-2, "cosine": 0.999, "source": "SURVEY.md 8c"},
           "hip_vs_o32": {"rel_l2": rel_l2(hip, o32), "cosine": cosine(hip, o32),
                          "per_step_rel_l2": [rel_l2(a, b) for a, b in zip(hip_curve, c32)]},
           "final_latent_rms": float(o32.double().pow(2).mean().sqrt())}
    assert len(hip_curve) == len(c32) == n_run
    if with_fp16_oracle:
        t0 = time.time()
        o16, c16 = _oracle_run(O, ref_sd, den_sd, inputs, steps, guidance, torch.float16, dev, **win)
        say(f"o16  {time.time() - t0:7.1f} s")
        rec["o16_vs_o32"] = {"rel_l2": rel_l2(o16, o32), "cosine": cosine(o16, o32),
                             "per_step_rel_l2": [rel_l2(a, b) for a, b in zip(c16, c32)]}
        rec["hip_vs_o16"] = {"rel_l2": rel_l2(hip, o16), "cosine": cosine(hip, o16),
                             "per_step_rel_l2": [rel_l2(a, b) for a, b in zip(hip_curve, c16)]}
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--latent", type=int, default=96)
    ap.add_argument("--no-fp16-oracle", action="store_true")
    ap.add_argument("--config4", action="store_true", help="BASELINE configs[4] geometry: 128 x 128 latents, 48 frames in 3 wrapping windows of 30 "
                    "(context 30 / overlap 8), --steps DDIM steps of the 30 (default 2: the restatement runs ~1.5 PFLOP per step in fp32)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.config4:
        a.frames, a.latent = 48, 128
        a.steps = a.steps if a.steps != 20 else 2
    r = run(a.frames, a.steps, a.latent, with_fp16_oracle=not a.no_fp16_oracle, log=lambda *m: print(*m, flush=True),
            window=dict(context_frames=30, context_stride=1, context_overlap=8) if a.config4 else None)
    r["device"] = torch.cuda.get_device_name(0)
    print(json.dumps(r))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(r, fh, indent=1)
