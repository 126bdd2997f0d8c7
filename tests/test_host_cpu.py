"""CPU: host-side logic of the product package (no kernels): state-dict key layout vs the reference, window
scheduler, DDIM table, config handling."""
import json
import os

import pytest
import torch

import mikudance_amd as M
from mikudance_amd.selftest import MM_KWARGS, SCHED_KWARGS, SMALL
from oracle import cpu_ref as O


@pytest.mark.parametrize("name,geom", [("", {}), ("_small", SMALL)])
def test_state_dict_keys_match_reference(golden_dir, name, geom):
    gold = json.load(open(os.path.join(golden_dir, f"g6_state_dict_keys{name}.json")))
    with torch.device("meta"):
        den = M.UNet3DConditionModel(sample_size=16, **geom, **MM_KWARGS)
        ref = M.UNet2DConditionModel(sample_size=16, **geom)
    for m, g in ((den, gold["denoising_unet"]), (ref, gold["reference_unet"])):
        sd = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert sd == g


def test_windows_match_reference(golden_dir):
    for case in json.load(open(os.path.join(golden_dir, "g1_windows.json"))):
        got = list(M.get_context_scheduler("uniform")(0, case["steps"], case["num_frames"], case["context_frames"], 1, case["overlap"]))
        assert got == case["windows"]
    with pytest.raises(ValueError):
        M.get_context_scheduler("nope")


def test_ddim_table_matches_oracle():
    s, o = M.DDIMScheduler(**SCHED_KWARGS), O.DDIM()
    assert torch.equal(s.alphas_cumprod, o.alphas_cumprod)
    for n in (4, 20, 30):
        s.set_timesteps(n); o.set_timesteps(n)
        assert torch.equal(s.timesteps, o.timesteps)
        for t in s.timesteps:
            assert s.step_coefficients(t) == o.coeffs(t)
    with pytest.raises(NotImplementedError):
        M.DDIMScheduler()                       # epsilon / leading defaults are not the MikuDance configuration


def test_reference_control_pairs_blocks_like_the_reference():
    with torch.device("meta"):
        den = M.UNet3DConditionModel(sample_size=16, **SMALL, **MM_KWARGS)
        ref = M.UNet2DConditionModel(sample_size=16, **SMALL)
    w = M.ReferenceAttentionControl(ref, mode="write", fusion_blocks="full")
    r = M.ReferenceAttentionControl(den, mode="read", do_classifier_free_guidance=True, fusion_blocks="full")
    names_r = {id(m): n for n, m in den.named_modules()}
    names_w = {id(m): n for n, m in ref.named_modules()}
    pr = [names_r[id(b)] for b in r._blocks(den)]
    pw = [names_w[id(b)] for b in w._blocks(ref)]
    assert pr == pw and len(pr) == 16
    assert [b.dim for b in r._blocks(den)] == [256] * 6 + [128] * 5 + [64] * 5
    assert pr[0].startswith("down_blocks.2") and pr[5].startswith("mid_block")


def test_unsupported_configs_fail_loudly():
    with pytest.raises(NotImplementedError):
        M.UNet3DConditionModel(use_linear_projection=True)
    with pytest.raises(RuntimeError):
        M.UNet3DConditionModel.from_pretrained_2d("/nonexistent", "/nonexistent.pth")


def test_scene_motion_matches_reference(golden_dir):
    import numpy as np
    from mikudance_amd.scene_motion import camera_to_scene_motion
    z = np.load(os.path.join(golden_dir, "g2_scene_motion.npz"))
    flow = camera_to_scene_motion(list(z["w2c"]), list(z["c2w"]), list(z["K"]), z["depth"], 24, 24, False)
    assert flow.shape == (16, 2, 24, 24) and np.abs(flow - z["flow"]).max() <= 1e-12
    eye = [np.eye(4)] * 5
    assert np.abs(camera_to_scene_motion(eye, eye, list(z["K"]), np.zeros((1, 24, 24)), 24, 24, False)).max() == 0


def test_hot_path_refuses_cpu_tensors_and_flags_odd_latents():
    from mikudance_amd.unet_3d_mix import _UNetBase
    with torch.device("meta"):
        den = M.UNet3DConditionModel(sample_size=16, **SMALL, **MM_KWARGS)
        ref = M.UNet2DConditionModel(sample_size=16, **SMALL)
    pipe = M.MikuDanceVideoPipeline(None, None, ref, den, M.DDIMScheduler(**SCHED_KWARGS))
    with pytest.raises(RuntimeError):
        pipe.denoise(torch.zeros(1, 4, 2, 16, 16), torch.zeros(1, 2, 22, 16, 16), torch.zeros(2, 5, 64), 1, 3.5)
    assert _UNetBase._needs_upsample_size(12, 16, 4)     # 12 is not a multiple of 8: the reference's upsample_size path
    assert _UNetBase._needs_upsample_size(90, 96, 4) and not _UNetBase._needs_upsample_size(96, 128, 4)
    with pytest.raises(AssertionError):
        den.forward(torch.zeros(1, 4, 16, 16), 0, torch.zeros(1, 5, 64))       # 5-D input required (transformer_3d.py:117-119)


def test_zero_context_detection_and_pe_fold_tables():
    """Host-side logic of two result-preserving shortcuts (no kernels involved): which leading frames have an all-zero
    context (cross-attention == to_out bias there), and the per-frame row term pe @ Wq^T that replaces the query-only
    positional-encoding add of the motion module (reference src/models/motion_module.py:416-417)."""
    import torch
    from mikudance_amd import UNet3DConditionModel
    from mikudance_amd.blocks import MotionModule
    from mikudance_amd.selftest import MM_KWARGS, SMALL
    den = UNet3DConditionModel(sample_size=16, **SMALL, **MM_KWARGS)
    ctx = torch.zeros(2, 5, 64)
    ctx[1] = torch.randn(5, 64)
    assert den._cross(ctx, [0] * 3 + [1] * 3, "cpu").zero_frames == 3                 # [uncond x3 | cond x3]
    assert den._cross(ctx, [1, 0, 1, 0], "cpu").zero_frames == 0                      # interleaved: no leading run
    assert den._cross(ctx.flip(0), [0] * 3 + [1] * 3, "cpu").zero_frames == 0
    assert den._cross(torch.zeros(2, 5, 64), [0, 0, 1, 1], "cpu").zero_frames == 4
    mm = MotionModule(64, max_len=32).float()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for prm in mm.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * 0.1)
    pk = mm._pack(torch.device("cpu"))
    ab = mm.temporal_transformer.transformer_blocks[0].attention_blocks[1]
    want = ab.pos_encoder.pe[0].float() @ ab.to_q.weight.float().t()
    assert tuple(pk["peq1"].shape) == (32, 192) and float(pk["peq1"][:, 64:].abs().max()) == 0.0
    assert torch.allclose(pk["peq1"][:, :64].float(), want, atol=2e-3, rtol=2e-3)
    assert torch.equal(pk["qkv1"].float()[64:128], ab.to_k.weight.half().float())


def test_cross_context_cache_never_matches_by_address_alone():
    """a freed context tensor's address is re-used by the next clip's tokens (same shape, _version 0).
    The cache entry owns its source tensor, so that cannot produce a stale hit (runs on CPU: `_cross` is host logic)."""
    with torch.device("meta"):
        den = M.UNet3DConditionModel(sample_size=16, **SMALL, **MM_KWARGS)
    idx = [0, 0, 1, 1]
    for trial in range(50):
        x = torch.cat([torch.zeros(1, 5, 64), torch.randn(1, 5, 64)], 0)
        hit = den._cross(x, idx, "cpu")
        assert torch.equal(hit.ctx.view(2, 8, 64)[:, :5].float(), x.half().float())
        assert hit.zero_frames == 2 and den._cross(x, idx, "cpu") is hit and den._cross(x[:2], idx, "cpu") is hit
        del x, hit
    x = torch.randn(2, 5, 64)
    h1 = den._cross(x, idx, "cpu")
    x.add_(1.0)
    assert den._cross(x, idx, "cpu") is not h1
    den.clear_context_cache()
    assert not den._cross_cache


def test_window_layout_rejects_non_advancing_windows():
    with pytest.raises(ValueError):
        list(M.get_context_scheduler("uniform")(0, 20, 48, 30, 1, 30))
    with pytest.raises(ValueError):
        list(M.get_context_scheduler("uniform")(0, 20, 48, 8, 1, 12))
    assert list(M.get_context_scheduler("uniform")(0, 20, 4, 8, 1, 12)) == [[0, 1, 2, 3]]     # one window: overlap unused


def test_clip_tower_key_layout_and_preprocess_match_transformers(golden_dir):
    """The CLIP tower's state-dict keys == transformers' CLIPVisionModelWithProjection's (pinned in g11_meta.json by the
    generator, which built the third-party model), and clip_preprocess == CLIPImageProcessor defaults (constants restated)."""
    import numpy as np
    from PIL import Image
    meta = json.load(open(os.path.join(golden_dir, "g11_meta.json")))["small"]
    with torch.device("meta"):
        m = M.CLIPVisionModelWithProjection(meta["config"])
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == meta["keys"]
    img = Image.fromarray(np.random.default_rng(0).integers(0, 255, (224, 224, 3), dtype=np.uint8))
    px = M.clip_preprocess(img)
    a = np.asarray(img).astype(np.float64) / 255.0
    want = (a - np.array([0.48145466, 0.4578275, 0.40821073])) / np.array([0.26862954, 0.26130258, 0.27577711])
    assert tuple(px.shape) == (1, 3, 224, 224) and np.abs(px[0].permute(1, 2, 0).numpy() - want).max() < 1e-5
    assert tuple(M.clip_preprocess(img.resize((300, 260))).shape) == (1, 3, 224, 224)
    with pytest.raises(NotImplementedError):
        M.CLIPVisionModelWithProjection(dict(meta["config"], hidden_act="gelu"))


def test_gemm_and_conv_dispatch_table_of_the_benchmark_shapes():
    """The automatic kernel choice for every MFMA-bound launch shape of configs[1] (768 x 768, 16 frames, CFG: 32-frame batches), as
    measured best on MI355X in same-box A/B runs (profiles/r03_ab_gemm_sp_tiles.log, r03_ab_gemm_sp_tile_128x256.log,
    r03_ab_transposed_sp.log).  md_gemm_plan / md_conv3x3_plan run the launcher's own decision code without
    touching a device: 135 / 134 / 124 / 132 / 142 = gemm_sp_kernel 192x320 / 192x256 / 128x256 / 192x128 / 256x128, 144 = its 256x256 GEGLU flavour, +1000 on
    swapped operands (transposed output), +2000 when a residual enters through the matrix core (round 6: K tiles > sub-tiles of the wave tile -- FF-out, the
    K >= 1280 out-projections, a resnet's second conv; short K keeps the epilogue form: 135), 210 / 220 / 230 = W-stationary streaming kernel, 301 / 303 = multi-workgroup kernel."""
    from mikudance_amd import _lib, ops
    lib = _lib.load()
    ncu, G = 256, ops.ACT_GEGLU
    gemm = {
        # level 0 (96 x 96): HBM-bound projections and the K = 320 GEGLU stream; FF-out on the big tile
        (294912, 320, 320, 0, 0, 5): 210, (294912, 960, 320, 0, 0, 0): 210, (294912, 2560, 320, G, 0, 4): 230, (294912, 320, 1280, 0, 0, 5): 2135,
        # level 1 (48 x 48)
        (73728, 640, 640, 0, 0, 5): 220, (73728, 5120, 640, G, 0, 4): 144, (73728, 640, 2560, 0, 0, 5): 2135,
        # level 2 (24 x 24): N = 1280 takes the 192 x 256 tile (480 tiles = 1.9 rounds instead of 384 = 1.5)
        (18432, 1280, 1280, 0, 0, 5): 2134, (18432, 10240, 1280, G, 0, 4): 144, (18432, 1280, 5120, 0, 0, 5): 2134, (18432, 3840, 1280, 0, 0, 0): 135,
        (18432, 2560, 1280, 0, 0, 0): 135,
        # level 3 (12 x 12): N = 1280 on 192 x 128 tiles (24 x 10 = 240 of them fill 256 CUs; 128 x 256 gives 180): +14..16 % same-box
        # (profiles/r04_ab_tile_192x128.log); the wider N keep the 192 x 256 tile
        (4608, 1280, 1280, 0, 0, 5): 2132, (4608, 10240, 1280, G, 0, 4): 144, (4608, 1280, 5120, 0, 0, 5): 2132, (4608, 2560, 1280, 0, 0, 0): 134,
        (4608, 3840, 1280, 0, 0, 0): 134,
        # V^T projections (transposed output): swapped operands
        (294912, 320, 320, 0, 1, 0): 1134, (73728, 640, 640, 0, 1, 0): 1124, (18432, 1280, 1280, 0, 1, 0): 1134, (4608, 1280, 1280, 0, 1, 0): 1124,
        # not sp: ragged / tiny N, few tiles
        (257 * 32, 768, 320, 0, 0, 4): 303, (294912, 4, 320, 0, 0, 4): 301, (2048, 320, 1280, 0, 0, 4): 303,
    }
    for args, want in gemm.items():
        assert lib.md_gemm_plan(*args, ncu) == want, (args, lib.md_gemm_plan(*args, ncu), want)
    conv = {  # (B, H = W, Cin, Cout, stride, upsample)
        (32, 96, 320, 320, 1, 0): 135, (32, 96, 640, 320, 1, 0): 135, (32, 96, 960, 320, 1, 0): 135, (32, 48, 640, 640, 1, 0): 135,
        (32, 48, 1920, 640, 1, 0): 135, (32, 24, 1280, 1280, 1, 0): 134, (32, 24, 2560, 1280, 1, 0): 134, (32, 12, 1280, 1280, 1, 0): 132,
        (32, 12, 2560, 1280, 1, 0): 132, (32, 48, 640, 640, 1, 1): 135, (32, 24, 1280, 1280, 1, 1): 135, (32, 12, 1280, 1280, 1, 1): 134,
        (32, 96, 320, 320, 2, 0): 135, (32, 48, 640, 640, 2, 0): 135, (32, 24, 1280, 1280, 2, 0): 132, (32, 96, 320, 4, 1, 0): 301,
        (2, 8, 64, 320, 1, 0): 303,
        # AutoencoderKL at 768 x 768, 8 images per call: the 128-channel layers (N % 128 only) on the 256 x 128 tile, 256 channels on 192 x 256
        (8, 768, 64, 128, 1, 0): 142, (8, 768, 128, 128, 1, 0): 142, (8, 384, 128, 256, 1, 0): 134, (8, 384, 256, 256, 1, 0): 134,
    }
    for (B, H, cin, cout, st, up), want in conv.items():
        got = lib.md_conv3x3_plan(B, H, H, cin, cout, st, up, 4, ncu)
        assert got == want, ((B, H, cin, cout, st, up), got, want)
    # a resnet's second conv (bias + shortcut): the residual through the matrix core; K = 640 with a residual (10 K tiles < 15 sub-tiles): the epilogue form
    assert lib.md_conv3x3_plan(32, 96, 96, 320, 320, 1, 0, 5, ncu) == 2135 and lib.md_conv3x3_plan(32, 24, 24, 1280, 1280, 1, 0, 5, ncu) == 2134
    assert lib.md_gemm_plan(294912, 320, 640, 0, 0, 5, ncu) == 135 and lib.md_gemm_plan(294912, 320, 960, 0, 0, 5, ncu) == 135
    # argument errors are reported, not guessed around
    assert lib.md_gemm_plan(128, 320, 100, 0, 0, 0, ncu) < 0 and lib.md_conv3x3_plan(1, 8, 8, 60, 320, 1, 0, 0, ncu) < 0
    # configs[4]: 983 040 tokens.  K = 1280 makes A 2.5 GB, beyond the sp kernel's 2^31-byte reach: planned (and launched) in row blocks
    assert lib.md_gemm_plan(983040, 320, 1280, 0, 0, 5, ncu) == 2135 and lib.md_gemm_plan(983040, 320, 320, 0, 0, 5, ncu) == 210
    assert lib.md_conv3x3_plan(60, 128, 128, 960, 320, 1, 0, 4, ncu) == 135
    # a smaller chip changes the rounds, hence the tile: the model is per device
    assert lib.md_gemm_plan(18432, 1280, 1280, 0, 0, 5, 192) in (2134, 2135, 2124)


def test_fused_normalisation_plans_are_the_measured_table():
    """md_gemm_ln_plan / md_gemm_affine_plan (no device touched): the fused normalisations exist exactly where they won their same-box A/B
    (profiles/r05_ab_fused_norms*.log) -- K = 320, plain epilogue, N a multiple of 320, >= 32768 rows in whole 16-row tiles -- and nowhere
    else; the host (blocks.ln_linear / gn_linear) runs the literal operator pair wherever a plan says no."""
    from mikudance_amd import _lib
    lib = _lib.load()
    ACT_NONE, ACT_GEGLU = 0, 3
    yes = [(294912, 960, 320, 2), (294912, 320, 320, 0), (147456, 320, 320, 0), (983040, 960, 320, 2), (32768, 640, 320, 0), (32784, 320, 320, 0)]
    for M, N, K, epi in yes:
        assert lib.md_gemm_ln_plan(M, N, K, ACT_NONE, epi) == 1, (M, N, K, epi)
    no = [(294912, 2560, 320, ACT_GEGLU, 0),          # GEGLU: memory waves VALU bound (and the fold costs the consumer its warm input)
          (73728, 1920, 640, ACT_NONE, 2), (73728, 640, 640, ACT_NONE, 0),    # K = 640: loader waves' work redone per column group
          (18432, 1280, 1280, ACT_NONE, 0), (4608, 1280, 1280, ACT_NONE, 0),  # not streaming-kernel shapes
          (16384, 320, 320, ACT_NONE, 0), (32776, 320, 320, ACT_NONE, 0),     # too few rows / not whole tiles
          (294912, 320, 320, ACT_NONE, 1), (294912, 256, 320, ACT_NONE, 0), (294912, 2880, 320, ACT_NONE, 0)]   # residual; N not k x 320; > 8 groups
    for M, N, K, act, epi in no:
        assert lib.md_gemm_ln_plan(M, N, K, act, epi) == 0, (M, N, K, act, epi)
    assert lib.md_gemm_affine_plan(294912, 320, 320, 9216) == 1 and lib.md_gemm_affine_plan(983040, 320, 320, 16384) == 1
    assert lib.md_gemm_affine_plan(73728, 640, 320, 9216) == 1                      # two column groups
    for M, N, K, rpi in [(73728, 640, 640, 2304), (294912, 320, 320, 9216 + 8), (294912, 320, 320, 9000), (294912, 320, 320, 0), (16384, 320, 320, 1024),
                         (18432, 1280, 1280, 576)]:
        assert lib.md_gemm_affine_plan(M, N, K, rpi) == 0, (M, N, K, rpi)


def test_ln_fold_algebra_and_row_mean_insensitivity():
    """packing.ln_fold: LN(x) @ W^T + bias == rstd * (x @ Wf^T - mu * s) + c.  With the fp16-rounded Wf the identity holds to the rounding
    of gamma * W (2^-11 per weight); with s summed from the ROUNDED Wf the mean term cancels exactly, so shifting every row by 1000 sigma
    changes nothing but fp64 noise -- and the GEGLU row interleave commutes with the fold."""
    import torch
    from mikudance_amd import packing
    g = torch.Generator().manual_seed(3)
    M, N, K = 64, 96, 320
    x = torch.randn(M, K, generator=g, dtype=torch.float64)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    bias = (torch.randn(N, generator=g) * 0.2).half()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).half()
    beta = (0.3 * torch.randn(K, generator=g)).half()
    wf, sc = packing.ln_fold(w, bias, gamma, beta)
    assert wf.dtype == torch.float16 and sc.dtype == torch.float32 and sc.shape == (2, N)
    assert torch.equal(sc[0].double(), wf.double().sum(1).float().double())                      # s is the sum of the ROUNDED weights
    def folded(xx):
        mu = xx.mean(1, keepdim=True)
        rstd = torch.rsqrt(((xx - mu) ** 2).mean(1, keepdim=True) + 1e-5)
        return rstd * (xx @ wf.double().t() - mu * wf.double().sum(1)) + sc[1].double()
    ref = torch.nn.functional.layer_norm(x, (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + bias.double()
    assert (folded(x) - ref).abs().max() < 2e-3 * ref.abs().max()                                # rounding of gamma * W to fp16
    assert (folded(x + 1000.0) - folded(x)).abs().max() < 1e-7 * ref.abs().max()                 # (x - mu) . Wf: the mean never reaches the sum
    wp2, bp2 = packing.geglu_weight(torch.cat([w, w[:32]]), torch.cat([bias, bias[:32]]), "cpu")
    wf2, sc2 = packing.ln_fold(wp2, bp2, gamma, beta)
    wf3, sc3 = packing.ln_fold(torch.cat([w, w[:32]]), torch.cat([bias, bias[:32]]), gamma, beta)
    wp3, cp3 = packing.geglu_weight(wf3, sc3[1], "cpu")
    assert torch.equal(wf2, wp3) and torch.allclose(sc2[1], cp3.float(), atol=1e-3)              # interleave(fold) == fold(interleave)
