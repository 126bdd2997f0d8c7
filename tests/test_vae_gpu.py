"""GPU: AutoencoderKL on the HIP kernels (through the C ABI) against the CPU oracle's restatement of diffusers 0.24.0
AutoencoderKL (oracle/cpu_ref.py vae_*; PARITY UNPINNED: diffusers is not part of /root/reference) on the same seeded
weights and inputs, plus the two kernels this row added (asymmetric-pad stride-2 conv, row softmax) against PyTorch ops.
Tolerances: kernels |err| <= 1e-2*maxabs + 1e-3; encoder moments / decoded image relative L2 <= 3e-2, cosine >= 0.999."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

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
