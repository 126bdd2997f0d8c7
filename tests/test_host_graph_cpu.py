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
