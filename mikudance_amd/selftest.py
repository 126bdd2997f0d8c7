"""smoke(): one tiny invocation of the hot path on cuda:0 (reduced-width UNets, 16x16 latents, 4 frames, 2 DDIM steps),
checked against the CPU oracle.  The oracle is used here ONLY as the checker (see oracle/cpu_ref.py header)."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)
MM_KWARGS = dict(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
                 use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=True,
                 motion_module_decoder_only=False, motion_module_type="Vanilla",
                 motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                                           attention_block_types=["Temporal_Self", "Temporal_Self"],
                                           temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                                           temporal_attention_dim_div=1))
SCHED_KWARGS = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                    prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")


def _cache_file(cls, geom, kw, seed, mode):
    """Synthetic fp16 weights are a pure function of (class, geometry, seed, mode): tests and bench.py build the same 2.2 G
    parameters in up to six processes per run (25-40 s of CPU randn each), so the fp16 copy is kept under the system temp
    directory and mapped back by later builds.  MD_SYNTH_CACHE=0 switches it off; nothing but seeded random weights ever goes
    through it (real checkpoints load through from_pretrained_2d / load_state_dict)."""
    import hashlib
    import tempfile
    if os.environ.get("MD_SYNTH_CACHE", "1") == "0":
        return None
    key = hashlib.sha1(repr((cls.__name__, sorted(geom.items()), sorted((k, repr(v)) for k, v in kw.items()), seed, mode, 2)).encode()).hexdigest()[:16]
    d = os.path.join(tempfile.gettempdir(), "mdance_synth_cache")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, f"{cls.__name__}_{key}_f16.safetensors")


def build_models(geom=None, seed_den=1234, seed_ref=4321, mode="fan_in", device="cuda", dtype=torch.float16, keep_state_dicts=True):
    """Both UNets with seeded synthetic weights (no checkpoints exist offline).  Returns (ref, den, ref_sd, den_sd); the fp32
    state dicts are dropped (None) unless keep_state_dicts -- they are only needed to feed the CPU oracle.  One model at a
    time is materialised on the host (8 ranks x 2.2 G fp32 parameters would otherwise cost ~150 GB of host RAM)."""
    from . import UNet2DConditionModel, UNet3DConditionModel
    from .synth import synth_state_dict
    geom = dict(SMALL if geom is None else geom)
    out = []
    for cls, kw, seed in ((UNet2DConditionModel, {}, seed_ref), (UNet3DConditionModel, MM_KWARGS, seed_den)):
        cache = _cache_file(cls, geom, kw, seed, mode) if dtype == torch.float16 else None
        if cache is not None and not keep_state_dicts and os.path.exists(cache):
            from safetensors.torch import load_file
            with torch.device("meta"):
                model = cls(sample_size=16, **geom, **kw)
            model.load_state_dict(load_file(cache, device="cpu"), strict=True, assign=True)
            out.append((model.to(device=device), None))
            continue
        with torch.device("meta"):
            shapes = {k: tuple(v.shape) for k, v in cls(sample_size=16, **geom, **kw).state_dict().items()}
        sd = synth_state_dict(shapes, seed=seed, mode=mode)
        model = cls(sample_size=16, **geom, **kw)
        model.load_state_dict(sd, strict=True)
        model = model.to(dtype=dtype)
        if cache is not None and not os.path.exists(cache) and sum(v.numel() for v in sd.values()) > 50_000_000 \
                and os.environ.get("LOCAL_RANK", "0") == "0":              # one writer per host
            from safetensors.torch import save_file
            tmp = f"{cache}.{os.getpid()}.tmp"
            try:
                save_file({k: v.contiguous() for k, v in model.state_dict().items()}, tmp)
                os.replace(tmp, cache)
            except OSError:
                pass                                                   # a full or read-only temp directory only costs the next build its time
        model = model.to(device=device)
        out.append((model, sd if keep_state_dicts else None))
        del sd
    (ref, ref_sd), (den, den_sd) = out
    return ref, den, ref_sd, den_sd


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cosine(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


def smoke():
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    from . import DDIMScheduler, MikuDanceVideoPipeline, _lib
    from .synth import synth_inputs
    _lib.load()
    torch.cuda.set_device(0)
    ref, den, ref_sd, den_sd = build_models()
    pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
    latents, ref_latents, embeds = synth_inputs(4, 16, 16, ctx_len=5, ctx_dim=64, seed=100)
    out = pipe.denoise(latents.cuda().half(), ref_latents.cuda().half(), embeds.cuda().half(), 2, 3.5)
    torch.cuda.synchronize()
    from oracle import cpu_ref as O                                   # checker only
    with torch.no_grad():
        want = O.denoise_loop(ref_sd, den_sd, latents, ref_latents, embeds, 2, guidance_scale=3.5, reduced=True)
    r, c = rel_l2(out.float(), want), cosine(out.float(), want)
    print(json.dumps({"smoke": "denoise 2 steps, 4 frames, 16x16 latents, reduced-width UNets", "rel_l2": r, "cosine": c}))
    assert r < 3e-2 and c > 0.999, (r, c)
