"""GPU: AutoencoderKL on the HIP kernels (through the C ABI) against the CPU oracle's restatement of diffusers 0.24.0
AutoencoderKL (oracle/cpu_ref.py vae_*; PARITY UNPINNED: diffusers is not part of /root/reference) on the same seeded
weights and inputs, plus the two kernels this row added (asymmetric-pad stride-2 conv, row softmax) against PyTorch ops.
Tolerances: kernels |err| <= 1e-2*maxabs + 1e-3; encoder moments / decoded image relative L2 <= 3e-2, cosine >= 0.999."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from parity_budget import check_kernel  # noqa: E402

from mikudance_amd import AutoencoderKL, ops, packing  # noqa: E402
from mikudance_amd.selftest import cosine, rel_l2  # noqa: E402
from mikudance_amd.synth import synth_state_dict  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def close(got, ref, what):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    bound = 1e-2 * ref.abs().max().item() + 1e-3
    assert got.shape == ref.shape and math.isfinite(err) and err <= bound, f"{what}: max err {err:.4g} > {bound:.4g}"
    check_kernel(what, got, ref)


@pytest.mark.parametrize("h,w,cin,cout", [(8, 8, 64, 64), (14, 10, 128, 192), (6, 12, 64, 320)])
def test_conv3x3_stride2_pad_0101(h, w, cin, cout):
    x, wt, b = rnd(2, cin, h, w, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(cout, seed=3)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), wt.float(), b.float(), stride=2).permute(0, 2, 3, 1)
    out = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), packing.conv3x3_weight(wt, "cuda"), cout, bias=b.cuda(), stride=2, pad_lo=0)
    close(out, ref, "conv pad (0,1,0,1)")


def test_conv3x3_into_channel_slice():
    x, wt, b = rnd(2, 64, 8, 8, seed=4), rnd(8, 64, 3, 3, seed=5, scale=(9 * 64) ** -0.5), rnd(8, seed=6)
    ref = F.conv2d(x.float(), wt.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    out = torch.zeros(2, 8, 8, 64, dtype=torch.float16, device="cuda")
    ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), packing.conv3x3_weight(wt, "cuda"), 8, bias=b.cuda(), out=out[..., :8])
    close(out[..., :8], ref, "conv into slice")
    assert float(out[..., 8:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,cols,ld", [(5, 64, 64), (300, 9216, 9216), (33, 96, 128)])
def test_softmax_rows(rows, cols, ld):
    x = rnd(rows, ld, seed=7, scale=3.0)
    ref = torch.softmax(x[:, :cols].float() * 0.37, dim=-1)
    buf = x.cuda()
    ops.softmax_rows_(buf[:, :cols], scale=0.37)
    close(buf[:, :cols], ref, "softmax rows")
    if ld > cols:
        assert torch.equal(buf[:, cols:].cpu(), x[:, cols:])


def _vae(chans, seed):
    vae = AutoencoderKL(block_out_channels=chans)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=seed)
    vae.load_state_dict(sd, strict=True)
    return vae.half().cuda().eval(), sd


@pytest.mark.parametrize("chans,hw", [((64, 128, 256, 256), (64, 64)), ((64, 128, 256, 256), (96, 64)), ((128, 256, 512, 512), (64, 64))])
def test_autoencoder_kl_vs_oracle(chans, hw):
    vae, sd = _vae(chans, seed=77)
    H, W = hw
    img = (torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(5)) * 2 - 1)
    z = torch.randn(2, 4, H // 8, W // 8, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        want_m = O.vae_encode_moments(sd, img)
        want_x = O.vae_decode(sd, z)
    post = vae.encode(img.cuda().half()).latent_dist
    got_m = torch.cat([post.mean, post.logvar], 1)
    want_m = torch.cat([want_m[:, :4], want_m[:, 4:].clamp(-30, 20)], 1)       # DiagonalGaussianDistribution clamps logvar
    r, c = rel_l2(got_m.float(), want_m), cosine(got_m.float(), want_m)
    assert r < 3e-2 and c > 0.999, ("encode", r, c)
    got_x = vae.decode(z.cuda().half()).sample
    assert got_x.shape == (2, 3, H, W)
    r, c = rel_l2(got_x.float(), want_x), cosine(got_x.float(), want_x)
    assert r < 3e-2 and c > 0.999, ("decode", r, c)
    s = post.sample(torch.Generator().manual_seed(1))
    assert s.shape == post.mean.shape and torch.isfinite(s.float()).all()


@pytest.mark.parametrize("chans,frames,clips", [((64, 128), 3, 1), ((64, 128, 128, 128), 4, 2), ((64, 64), 1, 2)])
def test_temporal_decoder_vs_oracle(chans, frames, clips):
    """AutoencoderKLTemporalDecoder.decode (`--video_decoder`): frame-shifted token GEMMs for the Conv3d(3,1,1)s,
    clip-wide GroupNorm, the learned spatial/temporal blend folded into the second temporal conv -- against the F.conv3d
    restatement (oracle, unpinned), several clips per call, and the encoder path it shares with AutoencoderKL."""
    from mikudance_amd import AutoencoderKLTemporalDecoder
    vae = AutoencoderKLTemporalDecoder(block_out_channels=chans)
    shapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    assert tuple(shapes["decoder.time_conv_out.weight"]) == (3, 3, 3, 1, 1) and "post_quant_conv.weight" not in shapes
    assert tuple(shapes["decoder.mid_block.resnets.0.temporal_res_block.conv1.weight"]) == (chans[-1], chans[-1], 3, 1, 1)
    assert tuple(shapes["decoder.up_blocks.0.resnets.2.time_mixer.mix_factor"]) == (1,)
    sd = synth_state_dict(shapes, seed=31)
    for k in sd:
        if k.endswith("mix_factor"):
            sd[k] = torch.tensor([0.7 if "mid" in k else -0.4])
    vae.load_state_dict(sd, strict=True)
    vae = vae.to("cuda", dtype=torch.float16)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(clips * frames, 4, 8, 4, generator=g)
    with torch.no_grad():
        want = O.vae_temporal_decode(sd, z, frames)
    got = vae.decode(z.cuda().half(), num_frames=frames).sample
    up = 2 ** (len(chans) - 1)
    assert tuple(got.shape) == (clips * frames, 3, 8 * up, 4 * up)
    r, c = rel_l2(got.float(), want), cosine(got.float(), want)
    assert r < 3e-2 and c > 0.999, (r, c)
    if frames > 1:                                                    # the frames really talk to each other
        alone = torch.cat([O.vae_temporal_decode(sd, z[i:i + 1], 1) for i in range(z.shape[0])])
        assert rel_l2(alone, want) > 5 * r
    x = torch.randn(2, 3, 8 * up, 8 * up, generator=g)
    m = vae.encode(x.cuda().half()).latent_dist.mean
    assert rel_l2(m.float(), O.vae_encode_moments(sd, x)[:, :4]) < 3e-2
    with pytest.raises(ValueError):
        vae.decode(z[:frames * clips - 1].cuda().half(), num_frames=frames) if frames > 1 else vae.decode(z[:, :3].cuda().half())


def _oracle_on_gpu(fn, sd, *args):
    """The oracle (fp32) evaluated through PyTorch-ROCm on the GPU: the sizes below are minutes of CPU time.  MIOpen is switched
    off (PyTorch's own im2col / dilated-3d convolutions on rocBLAS: exact fp32, no per-shape kernel search on a fresh box)."""
    dev = torch.device("cuda:0")
    sdg = {k: v.to(dev) for k, v in sd.items()}
    with torch.no_grad():
        try:
            with torch.backends.cudnn.flags(enabled=False):
                out = fn(sdg, *[a.to(dev) if torch.is_tensor(a) else a for a in args])
        except RuntimeError:
            out = fn(sdg, *[a.to(dev) if torch.is_tensor(a) else a for a in args])
    return out.float().cpu()


def test_autoencoder_kl_at_768x768_vs_oracle():
    """SURVEY 8f-1 at the size the pipeline runs it (src/pipelines/pipeline_mikudance.py:115-130,456-549): sd-vae-ft-mse geometry
    (128, 256, 512, 512), 768 x 768 images / 96 x 96 latents, two images per call.  At this size the 128- and 256-channel layers run on
    the persistent kernels (256 x 128 and 192 x 256 tiles), the mid-block attention is a 9216 x 9216 single-head softmax."""
    vae, sd = _vae((128, 256, 512, 512), seed=77)
    img = (torch.rand(2, 3, 768, 768, generator=torch.Generator().manual_seed(5)) * 2 - 1).half().float()
    z = torch.randn(2, 4, 96, 96, generator=torch.Generator().manual_seed(6)).half().float()
    want_m = _oracle_on_gpu(O.vae_encode_moments, sd, img)
    want_x = _oracle_on_gpu(O.vae_decode, sd, z)
    got_m = vae.encode(img.cuda().half()).latent_dist.mean
    r, c = rel_l2(got_m.float(), want_m[:, :4]), cosine(got_m.float(), want_m[:, :4])
    assert r < 3e-2 and c > 0.999, ("encode", r, c)
    got_x = vae.decode(z.cuda().half()).sample
    assert got_x.shape == (2, 3, 768, 768)
    r, c = rel_l2(got_x.float(), want_x), cosine(got_x.float(), want_x)
    assert r < 3e-2 and c > 0.999, ("decode", r, c)


def test_temporal_decoder_16_frames_at_64x64_latents_vs_oracle():
    """SURVEY 8f-4 at size: the published geometry (128, 256, 512, 512), ONE clip of 16 frames (the reference's decode_chunk_size,
    src/pipelines/pipeline_mikudance.py:132-150) at 64 x 64 latents -> 512 x 512 pixels.  Every Conv3d (3,1,1) runs as one 3 x 1
    implicit GEMM over an image whose rows are the 16 frames (up to 262 144 pixels wide), on the persistent kernels."""
    from mikudance_amd import AutoencoderKLTemporalDecoder
    vae = AutoencoderKLTemporalDecoder()
    shapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    sd = synth_state_dict(shapes, seed=31)
    for k in sd:
        if k.endswith("mix_factor"):
            sd[k] = torch.tensor([0.7 if "mid" in k else -0.4])
    vae.load_state_dict(sd, strict=True)
    vae = vae.to("cuda", dtype=torch.float16)
    frames = 16
    z = torch.randn(frames, 4, 64, 64, generator=torch.Generator().manual_seed(5)).half().float()
    want = _oracle_on_gpu(O.vae_temporal_decode, sd, z, frames)
    got = vae.decode(z.cuda().half(), num_frames=frames).sample
    assert tuple(got.shape) == (frames, 3, 512, 512)
    r, c = rel_l2(got.float(), want), cosine(got.float(), want)
    assert r < 3e-2 and c > 0.999, (r, c)


def test_temporal_decoder_16_frames_at_the_headline_size_vs_oracle():
    """--video_decoder at the size the reference runs it for the headline configuration: chunks of 16 frames at 96 x 96 latents -> 768 x 768
    pixels (src/pipelines/pipeline_mikudance.py:132-150).  The widest 3 x 1 implicit GEMM sees an image of 16 rows x 589 824 pixels
    (9.4 M pixels, 1.2 G elements: inside both limits of the conv's tap arithmetic).  Timing record: profiles/r05_vae_temporal.json."""
    from mikudance_amd import AutoencoderKLTemporalDecoder
    vae = AutoencoderKLTemporalDecoder()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=31)
    for k in sd:
        if k.endswith("mix_factor"):
            sd[k] = torch.tensor([0.7 if "mid" in k else -0.4])
    vae.load_state_dict(sd, strict=True)
    vae = vae.to("cuda", dtype=torch.float16)
    frames = 16
    z = torch.randn(frames, 4, 96, 96, generator=torch.Generator().manual_seed(5)).half().float()
    want = _oracle_on_gpu(O.vae_temporal_decode, sd, z, frames)
    got = vae.decode(z.cuda().half(), num_frames=frames).sample
    assert tuple(got.shape) == (frames, 3, 768, 768)
    r, c = rel_l2(got.float(), want), cosine(got.float(), want)
    assert r < 3e-2 and c > 0.999, (r, c)
