"""CPU: the HOST graph of the two UNets and of the denoising step, executed on an emulation of the operator layer
(tests/fake_ops.py: every mikudance_amd.ops function restated in PyTorch with the C ABI's argument meaning) and compared with the
oracle.  What this pins without a GPU: operator order, the skip tensors born inside their concat buffers (channel slices as
inputs, `out=` slices as outputs: no torch.cat copy), row-broadcast tables, bank hand-over with the CFG row mask, packing /
unpacking at the API boundary.  The kernels themselves are pinned by the -m gpu tests."""
import pytest
import torch

import fake_ops
from mikudance_amd import ReferenceAttentionControl
from mikudance_amd.selftest import build_models, cosine, rel_l2
from mikudance_amd.synth import synth_inputs
from oracle import cpu_ref as O


@pytest.fixture(scope="module")
def small():
    return build_models(device="cpu")


def _pair(ref, den, lat, rl, emb, f, h, w, t):
    writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
    reader = ReferenceAttentionControl(den, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
    g = rl.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w).half()
    emb_in = emb.repeat((f, 1, 1)).half()
    ref_out = ref(g, torch.zeros((), dtype=torch.long), encoder_hidden_states=emb_in, return_dict=False)[0]
    reader.update(writer)
    banks = [blk.bank[0].float() for blk in reader._blocks(den)]
    pred = den(lat.repeat(2, 1, 1, 1, 1).half(), torch.tensor(t), encoder_hidden_states=emb_in[:2], return_dict=False)[0]
    reader.clear(); writer.clear()
    return ref_out, banks, pred


@pytest.mark.parametrize("hw", [(16, 16), (18, 20)])
def test_unet_pair_on_emulated_operators_matches_the_oracle(small, monkeypatch, hw):
    """(18, 20): a latent that is not a multiple of 8 -> the upsample_size path with odd intermediate sizes."""
    ref, den, ref_sd, den_sd = small
    fake_ops.install(monkeypatch)
    f, (h, w) = 3, hw
    lat, rl, emb = (t.half().float() for t in synth_inputs(f, h, w, ctx_len=5, ctx_dim=64, seed=5))
    ref_out, banks, pred = _pair(ref, den, lat, rl, emb, f, h, w, 601)
    with torch.no_grad():
        g = rl.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w)
        ctx = emb.repeat((f, 1, 1))
        want_banks, want_ref = O.reference_unet_forward(ref_sd, g, ctx)
        want = O.denoising_unet_forward(den_sd, lat.repeat(2, 1, 1, 1, 1), torch.tensor(601), ctx[:2],
                                        {k: v.half().float() for k, v in want_banks.items()}, cfg=True)
    assert rel_l2(ref_out.float(), want_ref) < 2e-2 and cosine(ref_out.float(), want_ref) > 0.999
    assert len(banks) == len(want_banks) == 16
    assert rel_l2(pred.float(), want) < 2e-2 and cosine(pred.float(), want) > 0.999, rel_l2(pred.float(), want)
    # no concat copy anywhere, and operators really consumed channel slices / produced into them
    names = [c[0] for c in fake_ops.CALLS]
    assert "concat" not in names
    sliced_in = [c for c in fake_ops.CALLS if c[0] in ("groupnorm", "conv", "instnorm") and c[1][1][-2] > c[1][0][-1]]
    sliced_out = [c for c in fake_ops.CALLS if c[0] in ("conv", "gemm") and c[1][-1] is not None]
    assert len(sliced_in) >= 20 and len(sliced_out) >= 40, (len(sliced_in), len(sliced_out))


def test_unet_pair_with_the_fused_normalisations_matches_the_oracle(small, monkeypatch):
    """The same pair with md_gemm_ln_f16 / md_gemm_affine_f16 taken EVERYWHERE (the real plans only say yes on the 96 x 96 / 48 x 48
    levels' token counts): LayerNorm folded into to_q / q|k|v / ff.net.0 (packing.ln_fold: fp16(gamma W), s, c in the GEGLU row
    order, the motion module's positional row term on top) and GroupNorm applied inside proj_in -- host side against the oracle."""
    ref, den, ref_sd, den_sd = small
    fake_ops.install(monkeypatch)
    monkeypatch.setattr(fake_ops, "FUSED", True)
    f, h, w = 3, 16, 16
    lat, rl, emb = (t.half().float() for t in synth_inputs(f, h, w, ctx_len=5, ctx_dim=64, seed=5))
    ref_out, banks, pred = _pair(ref, den, lat, rl, emb, f, h, w, 601)
    for net in (ref, den):                                   # drop the folded weights again: other tests share these models
        for m in net.modules():
            if hasattr(m, "_pk"):
                m._pk = None
    with torch.no_grad():
        g = rl.repeat(2, 1, 1, 1, 1).reshape(2 * f, 22, h, w)
        ctx = emb.repeat((f, 1, 1))
        want_banks, want_ref = O.reference_unet_forward(ref_sd, g, ctx)
        want = O.denoising_unet_forward(den_sd, lat.repeat(2, 1, 1, 1, 1), torch.tensor(601), ctx[:2],
                                        {k: v.half().float() for k, v in want_banks.items()}, cfg=True)
    assert rel_l2(ref_out.float(), want_ref) < 2e-2 and cosine(ref_out.float(), want_ref) > 0.999
    assert rel_l2(pred.float(), want) < 2e-2 and cosine(pred.float(), want) > 0.999, rel_l2(pred.float(), want)
    names = [c[0] for c in fake_ops.CALLS]
    # every LayerNorm but norm1 and every SiLU-free GroupNorm went through a fused entry point
    n_blocks, n_motion = 16, 21
    assert names.count("gemm_affine") == names.count("groupnorm_table") == 2 * n_blocks + n_motion
    assert names.count("gemm_ln") == 2 * 2 * n_blocks + 3 * n_motion, names.count("gemm_ln")


def test_skip_plan_matches_the_reference_channel_arithmetic(small):
    ref, den, _, _ = small
    for net in (ref, den):
        plan = net._skip_plan()
        ups = [r for blk in net.up_blocks for r in blk.resnets]
        assert len(plan) == 12 and all(c1 > 0 for c1 in plan)
        # hidden channels of up resnet i = output channels of the operator in front of it (reference unet_3d_blocks.py:736,877)
        outs = [net.mid_block.resnets[1].cout] + [r.cout for r in ups[:-1]]
        assert [ups[i].cin - (ups[i].cin - plan[len(ups) - 1 - i]) for i in range(12)] == outs


@pytest.mark.parametrize("chans,frames,clips", [((64, 128), 3, 2), ((64, 64), 1, 2)])
def test_temporal_vae_decoder_graph_on_emulated_operators(monkeypatch, chans, frames, clips):
    """AutoencoderKLTemporalDecoder.decode: Conv3d (3,1,1) as ONE 3 x 1 implicit GEMM on the (clips, frames, h*w, C) view (weights
    packed [Cout][tap][Cin]), clip-wide GroupNorm, blend folded into the second temporal conv -- host graph vs the F.conv3d oracle."""
    from mikudance_amd import AutoencoderKLTemporalDecoder
    from mikudance_amd.synth import synth_state_dict
    fake_ops.install(monkeypatch)
    vae = AutoencoderKLTemporalDecoder(block_out_channels=chans)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=31)
    for k in sd:
        if k.endswith("mix_factor"):
            sd[k] = torch.tensor([0.7 if "mid" in k else -0.4])
    vae.load_state_dict(sd, strict=True)
    vae = vae.half()
    z = torch.randn(clips * frames, 4, 8, 4, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = O.vae_temporal_decode(sd, z, frames)
    got = vae.decode(z.half(), num_frames=frames).sample
    assert rel_l2(got.float(), want) < 2e-2 and cosine(got.float(), want) > 0.999, rel_l2(got.float(), want)
    assert any(c[0] == "conv" and c[1][3] == 1 for c in fake_ops.CALLS)            # the 3 x 1 form was used


def test_autoencoder_kl_graph_on_emulated_operators(monkeypatch):
    from mikudance_amd import AutoencoderKL
    from mikudance_amd.synth import synth_state_dict
    fake_ops.install(monkeypatch)
    vae = AutoencoderKL(block_out_channels=(64, 64, 128, 128))
    sd = synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=77)
    vae.load_state_dict(sd, strict=True)
    vae = vae.half().eval()
    img = torch.rand(2, 3, 64, 32, generator=torch.Generator().manual_seed(5)) * 2 - 1
    z = torch.randn(2, 4, 8, 4, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        want_m, want_x = O.vae_encode_moments(sd, img), O.vae_decode(sd, z)
    assert rel_l2(vae.encode(img.half()).latent_dist.mean.float(), want_m[:, :4]) < 2e-2
    assert rel_l2(vae.decode(z.half()).sample.float(), want_x) < 2e-2


@pytest.mark.parametrize("case", ["single_window", "wrapping_windows", "no_cfg", "eta", "literal_reference_pass"])
def test_denoise_loop_on_emulated_operators_matches_the_oracle(small, monkeypatch, case):
    """MikuDanceVideoPipeline.denoise -- the loop of reference src/pipelines/pipeline_mikudance.py:573-686 -- on the emulated
    operator layer: window lists and their duplicate-frame slots, the bank cache across steps (reference UNet once per window), the
    [u,c,u,c] context quirk, fp32 window accumulation, CFG, the DDIM step (eta = 0 and eta > 0 with the generator's draws), the
    NCFHW <-> NHWC packing at both ends.  Against oracle/cpu_ref.denoise_loop on the same fp16-representable inputs."""
    from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline
    from mikudance_amd.selftest import SCHED_KWARGS
    ref, den, ref_sd, den_sd = small
    fake_ops.install(monkeypatch)
    F_, steps, guidance, kw, okw = 3, 2, 3.5, {}, {}
    if case == "wrapping_windows":
        F_, win = 6, dict(context_frames=4, context_stride=1, context_overlap=2)
        kw, okw = dict(win), dict(win)
    elif case == "no_cfg":
        guidance = 1.0
    elif case == "eta":
        kw = dict(eta=0.5, generator=torch.Generator().manual_seed(11))
        okw = dict(eta=0.5, generator=torch.Generator().manual_seed(11), noise_dtype=torch.float16)
    lat, rl, emb = (t.half() for t in synth_inputs(F_, 16, 16, ctx_len=5, ctx_dim=64, seed=9))
    if guidance <= 1.0:
        emb = emb[1:]
    pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
    pipe.reference_reuse = case != "literal_reference_pass"
    seen = []
    before = (lat.clone(), rl.clone(), emb.clone())
    got = pipe.denoise(lat, rl, emb, steps, guidance, callback=lambda i, t, x: seen.append((i, t, x.float())), **kw)
    assert got.dtype == torch.float16 and got.shape == lat.shape
    assert all(torch.equal(a, b) for a, b in zip(before, (lat, rl, emb)))               # the pipeline never mutates its inputs
    curve = []
    with torch.no_grad():
        want = O.denoise_loop(ref_sd, den_sd, lat.float(), rl.float(), emb.float(), steps, guidance_scale=guidance,
                              reduced=case != "literal_reference_pass", on_step=lambda t, x: curve.append((t, x.float())), **okw)
    r, c = rel_l2(got.float(), want), cosine(got.float(), want)
    assert r < 2e-2 and c > 0.999, (case, r, c)
    assert [s[1] for s in seen] == [int(t) for t, _ in curve] and len(seen) == steps
    for (_, _, a), (_, b) in zip(seen, curve):
        assert rel_l2(a, b) < 2e-2
    # nothing outlives the clip: banks dropped, context caches empty
    from mikudance_amd import ReferenceAttentionControl as RAC
    assert all(not blk.bank for blk in RAC(den, mode="read", fusion_blocks="full")._blocks(den))
    assert not den._cross_cache and not ref._cross_cache


def test_fp32_typed_boundary_is_the_fp16_run(small, monkeypatch):
    """`weight_dtype: fp32` (reference scripts/inference_video.py:66-69): .float() models re-pack their kernel-layout weights, fp32
    latents / context are converted at the edge, results come back in fp32 -- and equal the fp16 run on fp16-representable values."""
    from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline
    from mikudance_amd.selftest import SCHED_KWARGS
    ref, den, _, _ = small
    fake_ops.install(monkeypatch)
    lat, rl, emb = (t.half() for t in synth_inputs(2, 16, 16, ctx_len=5, ctx_dim=64, seed=3))
    pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
    out16 = pipe.denoise(lat, rl, emb, 2, 3.5)
    try:
        ref.float(); den.float()
        assert den.dtype == ref.dtype == torch.float32
        out32 = pipe.denoise(lat.float(), rl.float(), emb.float(), 2, 3.5)
    finally:
        ref.half(); den.half()
    assert out32.dtype == torch.float32 and torch.equal(out32.half(), out16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_pipeline_call_on_emulated_operators(small, monkeypatch, dtype):
    """MikuDanceVideoPipeline.__call__ (reference src/pipelines/pipeline_mikudance.py:363-704) with the reference script's positional
    order, PIL inputs and a CPU generator, duck-typed VAE / CLIP, in both weight dtypes: guidance assembly (22 channels), latent
    preparation, the loop, decode, the (1, 3, F, H, W) float32 CPU video in [0, 1]."""
    import numpy as np
    from PIL import Image
    from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline
    from mikudance_amd.selftest import SCHED_KWARGS
    ref, den, _, _ = small
    fake_ops.install(monkeypatch)
    H = W = 128
    F_ = 2
    rng = np.random.default_rng(0)
    img = lambda: Image.fromarray(rng.integers(0, 255, (160, 144, 3), dtype=np.uint8))
    flow = rng.uniform(-0.03, 0.03, (F_, 2, H // 8, W // 8))
    pipe = MikuDanceVideoPipeline(vae=fake_ops.FakeVAE(), image_encoder=fake_ops.FakeCLIP(), reference_unet=ref, denoising_unet=den,
                                  scheduler=DDIMScheduler(**SCHED_KWARGS))
    try:
        pipe = pipe.to("cpu", dtype=dtype)
        assert den.dtype == dtype
        seen = []
        v = pipe(img(), img(), [img() for _ in range(F_)], [img() for _ in range(F_)], [img() for _ in range(F_)], flow, W, H, F_, 2, 3.5,
                 generator=torch.manual_seed(42), callback=lambda i, t, x: seen.append(x.dtype)).videos
    finally:
        ref.half(); den.half()
    assert tuple(v.shape) == (1, 3, F_, H, W) and v.dtype == torch.float32 and torch.isfinite(v).all()
    assert float(v.min()) >= 0.0 and float(v.max()) <= 1.0 and float(v.std()) > 0
    assert seen == [dtype, dtype]                                    # latents keep the requested dtype at the boundary
