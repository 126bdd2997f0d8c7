"""Generate tests/golden/* by importing the REFERENCE (read-only at /root/reference) in the build container.

    python oracle/gen_golden.py            # rewrites tests/golden/

Runs only where /root/reference exists (never on the GPU box).  The reference's own files are imported
unmodified; the un-vendored `diffusers` leaf ops come from oracle/shim_diffusers (see its docstring).
Fixtures are DATA only: seeds, shapes, inputs and expected outputs.  Weights are never stored: they are
regenerated from (key, seed) by mikudance_amd.synth.synth_state_dict and guarded by a checksum.

G1 windows      : src/pipelines/context.py uniform()                       -- true reference
G2 scene motion : tools/scene_motion_tracking.py camera_to_scene_motion()  -- true reference
G3 blocks       : src/models/resnet.py ResnetBlock3D/Downsample3D/Upsample3D, man_module.MANModule -- true reference
G4 UNets        : reference glue (unet_2d_mix / unet_3d_mix / mutual_mix_attention / motion_module) + shim leaf ops
G5 loop         : the loop of src/pipelines/pipeline_mikudance.py:573-686 driven with the reference UNets,
                  ReferenceAttentionControl and context scheduler; DDIM is the restated scheduler (third party)
G6 keys         : full-size state-dict key -> shape maps of both UNets (checkpoint-compat contract)
G13 no-CFG loop : the same loop with guidance_scale = 1 (one clip-half, every row reads the bank, window SUM not average)
"""
import json
import os
import sys

import numpy as np
import torch
import yaml
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "shim_diffusers"))
sys.path.insert(0, REF)

from mikudance_amd.synth import synth_inputs, synth_state_dict  # noqa: E402
from oracle.cpu_ref import DDIM  # noqa: E402  (restated third-party scheduler)

OUT = os.path.join(ROOT, "tests", "golden")
SMALL = dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)


def checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def fill(module, seed, mode="fan_in"):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth_state_dict(shapes, seed=seed, mode=mode)
    module.load_state_dict(sd, strict=True)
    return sd


def g1_windows():
    from src.pipelines.context import uniform
    cases = [(4, 30, 8), (16, 30, 8), (24, 32, 8), (48, 30, 8), (48, 32, 8), (232, 30, 8), (6, 4, 2), (31, 30, 8)]
    out = []
    for F_, cf, ov in cases:
        out.append({"num_frames": F_, "context_frames": cf, "overlap": ov, "steps": 20,
                    "windows": [list(map(int, w)) for w in uniform(0, 20, F_, cf, 1, ov)]})
    json.dump(out, open(os.path.join(OUT, "g1_windows.json"), "w"))


def g2_scene_motion():
    from tools.scene_motion_tracking import camera_to_scene_motion
    w2c = np.load(os.path.join(REF, "demo_samples/poses/w2c-demo1.npy"))[:16]
    c2w = np.load(os.path.join(REF, "demo_samples/poses/c2w-demo1.npy"))[:16]
    depth = np.load(os.path.join(REF, "demo_samples/chars/depm-img-kamisatoayakagenshinimpact.npy")).astype(np.float64)
    depth = depth.reshape(depth.shape[-2], depth.shape[-1])
    idx = (np.arange(24) * depth.shape[0] / 24).astype(int)
    depth24 = depth[idx][:, idx][None]
    K = [3.2, 3.2, 1.6, 1.6]
    flow = camera_to_scene_motion(list(w2c), list(c2w), K, depth24, 24, 24, False)
    eye = [np.eye(4)] * 5
    flow_eye = camera_to_scene_motion(eye, eye, K, np.zeros((1, 24, 24)), 24, 24, False)
    np.savez_compressed(os.path.join(OUT, "g2_scene_motion.npz"), w2c=w2c, c2w=c2w, depth=depth24, K=np.array(K),
                        flow=flow, flow_identity=flow_eye)


def g3_blocks():
    from src.models.man_module import MANModule
    from src.models.resnet import Downsample3D, ResnetBlock3D, Upsample3D
    t = {}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 32, 3, 8, 8, generator=g)
    temb = torch.randn(2, 128, generator=g)
    with torch.no_grad():
        rb = ResnetBlock3D(in_channels=32, out_channels=64, temb_channels=128, groups=32, eps=1e-5,
                           non_linearity="silu", use_inflated_groupnorm=True).eval()
        fill(rb, 11)
        t["resnet.x"], t["resnet.temb"], t["resnet.y"] = x, temb, rb(x, temb)
        rb2 = ResnetBlock3D(in_channels=64, out_channels=64, temb_channels=128, groups=32, eps=1e-5,
                            non_linearity="silu", use_inflated_groupnorm=True).eval()
        fill(rb2, 12)
        t["resnet_same.y"] = rb2(t["resnet.y"], temb)
        dn = Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op").eval()
        fill(dn, 13)
        t["down.y"] = dn(t["resnet.y"])
        up = Upsample3D(64, use_conv=True, out_channels=64).eval()
        fill(up, 14)
        t["up.y"] = up(t["resnet.y"])
        man = MANModule(64, 2).eval()
        fill(man, 15)
        xm = torch.randn(3, 64, 8, 8, generator=g)
        mm = torch.randn(3, 2, 16, 16, generator=g)
        t["man.x"], t["man.motion"], t["man.y"] = xm, mm, man(xm, mm)
    save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, "g3_blocks.safetensors"))


def build_unets(seed_den=1234, seed_ref=4321, **geom):
    from src.models.unet_2d_mix import UNet2DConditionModel
    from src.models.unet_3d_mix import UNet3DConditionModel
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/inference/mikudance_config.yaml")))
    den = UNet3DConditionModel(sample_size=16, **geom, **cfg["unet_additional_kwargs"]).eval()
    ref = UNet2DConditionModel(sample_size=16, **geom).eval()
    den_sd = fill(den, seed_den)
    ref_sd = fill(ref, seed_ref)
    return ref, den, ref_sd, den_sd


def g4_g5_unets():
    from src.models.mutual_mix_attention import ReferenceAttentionControl
    from src.pipelines.context import get_context_scheduler
    ref, den, ref_sd, den_sd = build_unets(**SMALL)
    meta = {"geometry": {k: list(v) if isinstance(v, tuple) else v for k, v in SMALL.items()},
            "seed_den": 1234, "seed_ref": 4321, "mode": "fan_in",
            "checksum_den": checksum(den_sd), "checksum_ref": checksum(ref_sd)}
    json.dump({"denoising_unet": {k: list(v.shape) for k, v in den.state_dict().items()},
               "reference_unet": {k: list(v.shape) for k, v in ref.state_dict().items()}},
              open(os.path.join(OUT, "g6_state_dict_keys_small.json"), "w"))
    writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1,
                                       fusion_blocks="full")
    reader = ReferenceAttentionControl(den, do_classifier_free_guidance=True, mode="read", batch_size=1,
                                       fusion_blocks="full")
    F_, h, w = 6, 16, 16
    latents, ref_latents, embeds = synth_inputs(F_, h, w, ctx_len=5, ctx_dim=64, seed=100)
    t = {}
    # ---- G4: one window of f=4, one UNet pair evaluation, literal call pattern of pipeline_mikudance.py:626-660
    with torch.no_grad():
        win = [0, 1, 2, 3]
        f = len(win)
        x = latents[:, :, win].repeat(2, 1, 1, 1, 1)
        g = ref_latents[:, win].repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w)
        emb_in = embeds.repeat((f, 1, 1))
        ref_out = ref(g, torch.zeros((), dtype=torch.long), encoder_hidden_states=emb_in, return_dict=False)[0]
        reader.update(writer)
        names = {id(m): n for n, m in den.named_modules()}
        from src.models.attention import TemporalBasicTransformerBlock
        for m in den.modules():
            if isinstance(m, TemporalBasicTransformerBlock):
                assert len(m.bank) == 1 and m.bank[0].dtype == torch.float16
                t["bank." + names[id(m)]] = m.bank[0][f:].contiguous()          # cond half, fp16 (quirk 3/5)
        pred = den(x, torch.tensor(601), encoder_hidden_states=emb_in[:2], return_dict=False)[0]
        reader.clear(); writer.clear()
        t["g4.ref_out_cond"] = ref_out[f:].contiguous()
        t["g4.pred"] = pred.contiguous()
        # no-bank / no-CFG-mask path of the reader is not reachable from the pipeline; not recorded.

        # ---- G5: 4-step loop, F=6, context 4, overlap 2, cfg 3.5 (loop body of pipeline_mikudance.py:573-686)
        sch = DDIM()
        steps, gs = 4, 3.5
        timesteps = sch.set_timesteps(steps)
        sched = get_context_scheduler("uniform")
        lat = latents.clone()
        for t_ in timesteps:
            noise_pred = torch.zeros((2,) + tuple(lat.shape[1:]))
            counter = torch.zeros((1, 1, F_, 1, 1))
            queue = list(sched(0, steps, F_, 4, 1, 2))
            for c in queue:
                lmi = torch.cat([lat[:, :, c]]).repeat(2, 1, 1, 1, 1)
                b, cc, f, hh, ww = lmi.shape
                rli = torch.cat([ref_latents[:, c]]).repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, hh, ww)
                emb_in = embeds.repeat((f, 1, 1))
                ref(rli, torch.zeros_like(t_), encoder_hidden_states=emb_in, return_dict=False)
                reader.update(writer)
                pred = den(lmi, t_, encoder_hidden_states=emb_in[:b], return_dict=False)[0]
                noise_pred[:, :, c] = noise_pred[:, :, c] + pred
                counter[:, :, c] = counter[:, :, c] + 1
                reader.clear(); writer.clear()
            u, c_ = (noise_pred / counter).chunk(2)
            v = u + gs * (c_ - u)
            lat = sch.step(v, t_, lat)
            t[f"g5.latents_after_t{int(t_)}"] = lat.clone()
        meta["g5"] = {"frames": F_, "context_frames": 4, "overlap": 2, "steps": steps, "guidance": gs,
                      "timesteps": [int(x) for x in timesteps], "windows": [list(map(int, c)) for c in queue]}
    t["in.latents"], t["in.ref_latents"], t["in.embeds"] = latents, ref_latents, embeds
    save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, "g4_g5_unets.safetensors"))
    json.dump(meta, open(os.path.join(OUT, "g4_g5_meta.json"), "w"), indent=1)


def g8_fullwidth():
    """FULL-WIDTH SD-1.5 geometry at BASELINE configs[0] shape (32x32 latents, 4 frames, CFG, 257x768 context): ONE
    UNet-pair evaluation with the reference's literal call pattern (pipeline_mikudance.py:626-660).  Weights and inputs
    are regenerated from their seeds by the tests, only the prediction is stored."""
    from src.models.mutual_mix_attention import ReferenceAttentionControl
    ref, den, ref_sd, den_sd = build_unets(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768)
    writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1,
                                       fusion_blocks="full")
    reader = ReferenceAttentionControl(den, do_classifier_free_guidance=True, mode="read", batch_size=1,
                                       fusion_blocks="full")
    f, h, w = 4, 32, 32
    latents, ref_latents, embeds = synth_inputs(f, h, w, ctx_len=257, ctx_dim=768, seed=100)
    with torch.no_grad():
        x = latents.repeat(2, 1, 1, 1, 1)
        g = ref_latents.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w)
        emb_in = embeds.repeat((f, 1, 1))
        ref(g, torch.zeros((), dtype=torch.long), encoder_hidden_states=emb_in, return_dict=False)
        reader.update(writer)
        pred = den(x, torch.tensor(601), encoder_hidden_states=emb_in[:2], return_dict=False)[0]
    save_file({"g8.pred": pred.contiguous()}, os.path.join(OUT, "g8_fullwidth_pred.safetensors"))
    json.dump({"checksum_den": checksum(den_sd), "checksum_ref": checksum(ref_sd), "timestep": 601, "frames": f, "latent": [h, w],
               "seed_inputs": 100, "seed_den": 1234, "seed_ref": 4321}, open(os.path.join(OUT, "g8_meta.json"), "w"))


def g9_fullsize():
    """FULL-WIDTH UNet pair at BASELINE configs[1] spatial size (96x96 latents = 768x768 pixels, Lq = Lk = 9216 at d = 40),
    f = 2 frames, CFG: one evaluation with the reference's literal call pattern (pipeline_mikudance.py:626-660).  This is
    the size at which the GPU build's automatic dispatch picks the big-tile conv/GEMM kernels and the folded d=40
    attention, so the HIP path is pinned to the reference's own modules at its benchmark shapes.  Only `pred` is stored."""
    from src.models.mutual_mix_attention import ReferenceAttentionControl
    ref, den, ref_sd, den_sd = build_unets(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768)
    writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1,
                                       fusion_blocks="full")
    reader = ReferenceAttentionControl(den, do_classifier_free_guidance=True, mode="read", batch_size=1,
                                       fusion_blocks="full")
    f, h, w = 2, 96, 96
    latents, ref_latents, embeds = synth_inputs(f, h, w, ctx_len=257, ctx_dim=768, seed=100)
    import time
    t0 = time.time()
    with torch.no_grad():
        x = latents.repeat(2, 1, 1, 1, 1)
        g = ref_latents.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w)
        emb_in = embeds.repeat((f, 1, 1))
        ref(g, torch.zeros((), dtype=torch.long), encoder_hidden_states=emb_in, return_dict=False)
        reader.update(writer)
        pred = den(x, torch.tensor(601), encoder_hidden_states=emb_in[:2], return_dict=False)[0]
    print("g9 pair evaluated in %.1f s" % (time.time() - t0))
    save_file({"g9.pred": pred.half().contiguous()}, os.path.join(OUT, "g9_fullsize_pred.safetensors"))
    json.dump({"checksum_den": checksum(den_sd), "checksum_ref": checksum(ref_sd), "timestep": 601, "frames": f, "latent": [h, w],
               "seed_inputs": 100, "seed_den": 1234, "seed_ref": 4321, "stored": "fp16 rounding of the fp32 reference output"},
              open(os.path.join(OUT, "g9_meta.json"), "w"))


def g10_odd_and_plain_gn():
    """Two more evaluations of the reference's own UNet pair at reduced width (same seeds/weights as G4):
    g10.pred_odd      latents 18 x 20 (not a multiple of 8): the `forward_upsample_size` / `upsample_size` path
                      (src/models/unet_3d_mix.py:447-455,564-586; src/models/unet_2d_mix.py:1016-1027,1345-1346), f = 3, CFG
    g10.pred_plain_gn use_inflated_groupnorm=False: torch.nn.GroupNorm on the 5-D tensor, i.e. statistics ACROSS the frames
                      of a clip-half in every ResnetBlock3D and in conv_norm_out (src/models/resnet.py:156-191), 16 x 16, f = 4"""
    from src.models.mutual_mix_attention import ReferenceAttentionControl
    from src.models.unet_3d_mix import UNet3DConditionModel
    t = {}

    def pair(ref, den, f, h, w, seed):
        writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
        reader = ReferenceAttentionControl(den, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
        latents, ref_latents, embeds = synth_inputs(f, h, w, ctx_len=5, ctx_dim=64, seed=seed)
        x = latents.repeat(2, 1, 1, 1, 1)
        g = ref_latents.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w)
        emb_in = embeds.repeat((f, 1, 1))
        ref(g, torch.zeros((), dtype=torch.long), encoder_hidden_states=emb_in, return_dict=False)
        reader.update(writer)
        pred = den(x, torch.tensor(601), encoder_hidden_states=emb_in[:2], return_dict=False)[0]
        reader.clear(); writer.clear()
        return pred.contiguous()

    with torch.no_grad():
        ref, den, ref_sd, den_sd = build_unets(**SMALL)
        t["g10.pred_odd"] = pair(ref, den, 3, 18, 20, 31)
        cfg = yaml.safe_load(open(os.path.join(REF, "configs/inference/mikudance_config.yaml")))["unet_additional_kwargs"]
        cfg["use_inflated_groupnorm"] = False
        den2 = UNet3DConditionModel(sample_size=16, **SMALL, **cfg).eval()
        den2.load_state_dict(den_sd, strict=True)
        t["g10.pred_plain_gn"] = pair(ref, den2, 4, 16, 16, 32)
    save_file(t, os.path.join(OUT, "g10_odd_plain_gn.safetensors"))
    json.dump({"odd": {"frames": 3, "latent": [18, 20], "seed_inputs": 31}, "plain_gn": {"frames": 4, "latent": [16, 16], "seed_inputs": 32},
               "timestep": 601, "seed_den": 1234, "seed_ref": 4321, "checksum_den": checksum(den_sd), "checksum_ref": checksum(ref_sd)},
              open(os.path.join(OUT, "g10_meta.json"), "w"))


CLIP_SMALL = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14,
                  projection_dim=64, hidden_act="quick_gelu")
CLIP_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224, patch_size=14,
                projection_dim=768, hidden_act="quick_gelu")       # the image_encoder of sd-image-variations (ViT-L/14)


def g11_clip():
    """transformers' OWN CLIPVisionModelWithProjection (installed in the build container; third party, not in /root/reference)
    with seeded synthetic weights, driven exactly like src/pipelines/pipeline_mikudance.py:406-416:
    last_hidden_state -> vision_model.post_layernorm -> visual_projection, all tokens.  Two geometries: a reduced one (fp32
    tensors stored in full) and ViT-L/14 at 224x224 (257 x 768 output stored as fp16)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    t, meta = {}, {}
    for name, geom, seed in (("small", CLIP_SMALL, 21), ("l14", CLIP_L14, 22)):
        m = CLIPVisionModelWithProjection(CLIPVisionConfig(**geom)).eval()
        sd = fill(m, seed)
        g = torch.Generator().manual_seed(seed + 100)
        px = torch.randn(1, 3, geom["image_size"], geom["image_size"], generator=g)
        with torch.no_grad():
            last = m(px).last_hidden_state
            out = m.visual_projection(m.vision_model.post_layernorm(last))
        t[f"g11.{name}.embeds"] = out.half().contiguous() if name == "l14" else out.contiguous()
        if name == "small":
            t["g11.small.last_hidden_state"] = last.contiguous()
        meta[name] = {"config": geom, "seed_weights": seed, "seed_pixels": seed + 100, "checksum": checksum(sd),
                      "keys": {k: list(v.shape) for k, v in sd.items()} if name == "small" else len(sd)}
    save_file(t, os.path.join(OUT, "g11_clip.safetensors"))
    json.dump(meta, open(os.path.join(OUT, "g11_meta.json"), "w"))


def g12_interpolation():
    """src/pipelines/utils.py linear / slerp (TRUE reference, imported) on seeded latent frames, incl. the near-parallel
    fallback, plus interpolate_latents' frame layout restated line by line from src/pipelines/pipeline_mikudance.py:317-360
    (that file cannot be imported here) around the reference's own blend functions."""
    from src.pipelines import utils as RU
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 4, 4, 6, 5, generator=g)
    t = {"g12.latents": lat}
    for name, is_slerp in (("linear", False), ("slerp", True)):
        RU.set_tensor_interpolation_method(is_slerp)
        for factor in (2, 3):
            new = torch.zeros(1, 4, (lat.shape[2] - 1) * factor + 1, 6, 5)
            rate = [i / factor for i in range(factor)][1:]
            idx = 0
            for i0, i1 in zip(range(lat.shape[2]), range(lat.shape[2])[1:]):
                v0, v1 = lat[:, :, i0], lat[:, :, i1]
                new[:, :, idx] = v0
                idx += 1
                for f in rate:
                    new[:, :, idx] = RU.get_tensor_interpolation_method()(v0, v1, f)
                    idx += 1
            new[:, :, idx] = v1
            t[f"g12.{name}.x{factor}"] = new
    v = torch.randn(4, 6, 5, generator=g)
    t["g12.slerp_parallel"] = RU.slerp(v, v * 1.0001 + 1e-5, 0.3)
    t["g12.slerp_parallel_in"] = v
    save_file({k: x.contiguous() for k, x in t.items()}, os.path.join(OUT, "g12_interpolation.safetensors"))


def g13_no_cfg_loop():
    """guidance_scale <= 1 (pipeline_mikudance.py:397): no classifier-free guidance.  The reference then builds ONE clip-half
    (:626-645: `.repeat(1, ...)`), keeps the CLIP tokens alone as context (:420-423), constructs ReferenceAttentionControl with
    do_classifier_free_guidance=False (every row reads the bank, mutual_mix_attention.py:181-201) and -- the division sits
    inside `if do_classifier_free_guidance:` (:670-674) -- feeds the scheduler the window SUM, not the window average.
    3 steps, F = 6, context 4, overlap 2 (wrapping windows, frames covered twice), small geometry, same weights as G4 / G5."""
    from src.models.mutual_mix_attention import ReferenceAttentionControl
    from src.pipelines.context import get_context_scheduler
    ref, den, ref_sd, den_sd = build_unets(**SMALL)
    writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=False, mode="write", batch_size=1, fusion_blocks="full")
    reader = ReferenceAttentionControl(den, do_classifier_free_guidance=False, mode="read", batch_size=1, fusion_blocks="full")
    F_, h, w = 6, 16, 16
    latents, ref_latents, embeds = synth_inputs(F_, h, w, ctx_len=5, ctx_dim=64, seed=100)
    image_prompt_embeds = embeds[1:]                                  # the conditional tokens alone
    t = {}
    sch = DDIM()
    steps, gs = 3, 1.0
    do_cfg = gs > 1.0
    timesteps = sch.set_timesteps(steps)
    sched = get_context_scheduler("uniform")
    lat = latents.clone()
    for t_ in timesteps:
        noise_pred = torch.zeros((lat.shape[0] * (2 if do_cfg else 1),) + tuple(lat.shape[1:]))
        counter = torch.zeros((1, 1, F_, 1, 1))
        queue = list(sched(0, steps, F_, 4, 1, 2))
        for c in queue:
            lmi = torch.cat([lat[:, :, c]]).repeat(2 if do_cfg else 1, 1, 1, 1, 1)
            b, cc, f, hh, ww = lmi.shape
            rli = torch.cat([ref_latents[:, c]]).repeat(2 if do_cfg else 1, 1, 1, 1, 1).reshape(b * f, 22, hh, ww)
            emb_in = image_prompt_embeds.repeat((f, 1, 1))
            ref(rli, torch.zeros_like(t_), encoder_hidden_states=emb_in, return_dict=False)
            reader.update(writer)
            pred = den(lmi, t_, encoder_hidden_states=emb_in[:b], return_dict=False)[0]
            noise_pred[:, :, c] = noise_pred[:, :, c] + pred
            counter[:, :, c] = counter[:, :, c] + 1
            reader.clear(); writer.clear()
        if do_cfg:
            u, c_ = (noise_pred / counter).chunk(2)
            noise_pred = u + gs * (c_ - u)
        lat = sch.step(noise_pred, t_, lat)
        t[f"g13.latents_after_t{int(t_)}"] = lat.clone()
    assert float(counter.max()) > 1.0                                 # the windows really overlap
    save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, "g13_no_cfg_loop.safetensors"))
    json.dump({"frames": F_, "context_frames": 4, "overlap": 2, "steps": steps, "guidance": gs, "timesteps": [int(x) for x in timesteps],
               "windows": [list(map(int, c)) for c in queue], "seed_inputs": 100, "seed_den": 1234, "seed_ref": 4321,
               "checksum_den": checksum(den_sd)}, open(os.path.join(OUT, "g13_meta.json"), "w"))


def g6_keys():
    ref, den, _, _ = build_unets()      # full SD-1.5 geometry (constructor defaults + cross_attention_dim 768)
    json.dump({"denoising_unet": {k: list(v.shape) for k, v in den.state_dict().items()},
               "reference_unet": {k: list(v.shape) for k, v in ref.state_dict().items()}},
              open(os.path.join(OUT, "g6_state_dict_keys.json"), "w"))


def g7_ddim():
    sch = DDIM()
    out = {"alphas_cumprod_first": [float(x) for x in sch.alphas_cumprod[:3]],
           "alphas_cumprod_last": [float(x) for x in sch.alphas_cumprod[-3:]]}
    for n in (4, 20, 30):
        out[f"timesteps_{n}"] = [int(x) for x in sch.set_timesteps(n)]
    json.dump(out, open(os.path.join(OUT, "g7_ddim_restated.json"), "w"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g45", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13"]
    if "g1" in which: g1_windows()
    if "g2" in which: g2_scene_motion()
    if "g3" in which: g3_blocks()
    if "g45" in which: g4_g5_unets()
    if "g6" in which: g6_keys()
    if "g7" in which: g7_ddim()
    if "g8" in which: g8_fullwidth()
    if "g9" in which: g9_fullsize()
    if "g10" in which: g10_odd_and_plain_gn()
    if "g11" in which: g11_clip()
    if "g12" in which: g12_interpolation()
    if "g13" in which: g13_no_cfg_loop()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
