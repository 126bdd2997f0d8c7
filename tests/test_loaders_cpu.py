"""CPU: config + checkpoint loading executed on real files, the way scripts/inference_video.py:81-117 does:

    unet           = UNet2DConditionModel.from_pretrained(base, subfolder="unet")          (donor, src/models/unet_2d_condition.py)
    reference_unet = UNet2DConditionModel_MIX.from_unet(unet)                                (src/models/unet_2d_mix.py:896-919)
    denoising_unet = UNet3DConditionModel.from_pretrained_2d(base, motion_module_path, subfolder="unet",
                                                             unet_additional_kwargs=cfg.unet_additional_kwargs)   (unet_3d_mix.py:600-691)
    denoising_unet.load_state_dict(torch.load(denoising_unet_path), strict=False); reference_unet.load_state_dict(torch.load(...))

An SD-1.5-shaped `unet/config.json` + `diffusion_pytorch_model.{safetensors,bin}` and an AnimateDiff-shaped motion-module
file are written to a temporary directory (reduced width where the reference's own code allows it: from_pretrained_2d reads the
widths from config.json; `from_unet` does NOT -- `cls(unet.config)` passes the config as `sample_size` and builds the default
SD-1.5 geometry (quirk 9) -- so that one runs at full width)."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

import mikudance_amd as M
from mikudance_amd.selftest import MM_KWARGS, SMALL
from mikudance_amd.synth import synth_state_dict

SD15_CONFIG = {
    "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.6.0", "act_fn": "silu", "attention_head_dim": 8,
    "block_out_channels": list(SMALL["block_out_channels"]), "center_input_sample": False,
    "cross_attention_dim": SMALL["cross_attention_dim"],
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2,
    "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 4, "sample_size": 64,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]}


class AttrDict(dict):
    """Attribute-access mapping like the OmegaConf DictConfig the script passes as unet_additional_kwargs."""
    __getattr__ = dict.__getitem__


@pytest.fixture(scope="module")
def checkpoint_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("sd15")
    os.makedirs(d / "unet")
    json.dump(SD15_CONFIG, open(d / "unet" / "config.json", "w"))
    with torch.device("meta"):
        den = M.UNet3DConditionModel(sample_size=64, **SMALL, **MM_KWARGS)
    shapes = {k: tuple(v.shape) for k, v in den.state_dict().items()}
    sd15 = synth_state_dict({k: v for k, v in shapes.items() if "motion_modules" not in k}, seed=5, mode="n002")
    mm = synth_state_dict({k: v for k, v in shapes.items() if "motion_modules" in k}, seed=6, mode="n002")
    assert len(sd15) > 600 and len(mm) > 500
    save_file(sd15, str(d / "unet" / "diffusion_pytorch_model.safetensors"))
    torch.save(mm, d / "mm_sd_v15_v2.ckpt")
    save_file(mm, str(d / "mm.safetensors"))
    return d, sd15, mm


def test_from_pretrained_2d_merges_sd15_and_motion_weights(checkpoint_dir):
    d, sd15, mm = checkpoint_dir
    kw = AttrDict({k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in MM_KWARGS.items()})
    for mm_path in (d / "mm_sd_v15_v2.ckpt", d / "mm.safetensors"):
        den = M.UNet3DConditionModel.from_pretrained_2d(d, mm_path, subfolder="unet", unet_additional_kwargs=kw)
        got = den.state_dict()
        assert set(got) == set(sd15) | set(mm)
        for k, v in {**sd15, **mm}.items():
            assert torch.equal(got[k], v), k
        assert den.config.block_out_channels == SMALL["block_out_channels"] or list(den.config.block_out_channels) == list(SMALL["block_out_channels"])
        assert den.config.use_motion_module and den.config.motion_module_kwargs["temporal_position_encoding_max_len"] == 32
        assert den.temporal_position_encoding_max_len == 32 and den.in_channels == 4
    # mm_zero_proj_out: the motion modules' proj_out keys are dropped from the merge and stay at their zero initialisation
    den = M.UNet3DConditionModel.from_pretrained_2d(d, d / "mm_sd_v15_v2.ckpt", subfolder="unet", unet_additional_kwargs=kw,
                                                    mm_zero_proj_out=True)
    n_zero = 0
    for k, v in den.state_dict().items():
        if "motion_modules" in k and "proj_out" in k:
            assert float(v.abs().max()) == 0.0, k
            n_zero += 1
        else:
            assert torch.equal(v, {**sd15, **mm}[k]), k
    assert n_zero == 2 * 21
    # the stage-2 checkpoint then overwrites everything (scripts/inference_video.py:111-114, strict=False)
    trained = synth_state_dict({k: tuple(v.shape) for k, v in den.state_dict().items()}, seed=77, mode="n002")
    torch.save(trained, d / "denoising_unet.pth")
    missing, unexpected = den.load_state_dict(torch.load(d / "denoising_unet.pth", map_location="cpu"), strict=False)
    assert not missing and not unexpected
    for k, v in den.state_dict().items():
        assert torch.equal(v, trained[k]), k


def test_from_pretrained_2d_bin_weights_and_error_paths(checkpoint_dir, tmp_path):
    d, sd15, mm = checkpoint_dir
    os.makedirs(tmp_path / "unet")
    json.dump(SD15_CONFIG, open(tmp_path / "unet" / "config.json", "w"))
    with pytest.raises(FileNotFoundError):
        M.UNet3DConditionModel.from_pretrained_2d(tmp_path, d / "mm_sd_v15_v2.ckpt", subfolder="unet", unet_additional_kwargs=MM_KWARGS)
    torch.save(sd15, tmp_path / "unet" / "diffusion_pytorch_model.bin")
    den = M.UNet3DConditionModel.from_pretrained_2d(tmp_path, d / "mm_sd_v15_v2.ckpt", subfolder="unet", unet_additional_kwargs=MM_KWARGS)
    assert all(torch.equal(den.state_dict()[k], v) for k, v in sd15.items())
    # a motion-module path that does not exist is silently skipped by the reference (:655): SD weights only, the rest untouched
    den = M.UNet3DConditionModel.from_pretrained_2d(tmp_path, tmp_path / "nope.ckpt", subfolder="unet", unet_additional_kwargs=MM_KWARGS)
    assert all(torch.equal(den.state_dict()[k], v) for k, v in sd15.items())
    open(tmp_path / "mm.weird", "w").write("x")
    with pytest.raises(RuntimeError):
        M.UNet3DConditionModel.from_pretrained_2d(tmp_path, tmp_path / "mm.weird", subfolder="unet", unet_additional_kwargs=MM_KWARGS)
    with pytest.raises(RuntimeError):
        M.UNet3DConditionModel.from_pretrained_2d(tmp_path / "missing", d / "mm_sd_v15_v2.ckpt", subfolder="unet", unet_additional_kwargs=MM_KWARGS)
    # without unet_additional_kwargs (the reference's `**None` is a TypeError, :637; here: the constructor defaults) the model
    # has no motion modules and plain (cross-frame) GroupNorm; the motion keys of the merge are then unexpected and ignored
    den = M.UNet3DConditionModel.from_pretrained_2d(tmp_path, d / "mm_sd_v15_v2.ckpt", subfolder="unet")
    assert not den.use_inflated_groupnorm and not any("motion_modules" in k for k in den.state_dict())
    assert all(torch.equal(den.state_dict()[k], v) for k, v in sd15.items())


def test_donor_from_pretrained_reads_config_and_weights(checkpoint_dir):
    d, sd15, mm = checkpoint_dir
    donor = M.UNet2DConditionModelPlain.from_pretrained(d, subfolder="unet")
    got = donor.state_dict()
    want = {k: v for k, v in sd15.items() if not k.startswith(("conv_out.", "conv_norm_out."))}
    assert set(got) == set(want)                                 # conv_out / conv_norm_out deleted (unet_2d_condition.py:645-654)
    assert all(torch.equal(got[k], v) for k, v in want.items())
    assert donor.config.cross_attention_dim == SMALL["cross_attention_dim"]
    with pytest.raises(RuntimeError):
        M.UNet2DConditionModelPlain.from_pretrained(d, subfolder="nope")


def test_from_unet_full_width_zero_pads_conv_in_and_copies_every_block():
    """`cls(unet.config)` (quirk 9) builds the DEFAULT geometry whatever the donor is, so this runs at the real SD-1.5
    width: conv_in 4 -> 20 input channels zero-padded then overwritten in [:, :4], time embedding / down / mid / up copied,
    MAN blocks left for the stage-2 checkpoint, which then loads strict (scripts/inference_video.py:115-117)."""
    donor = M.UNet2DConditionModelPlain(cross_attention_dim=768)
    with torch.no_grad():
        for i, (k, p) in enumerate(donor.named_parameters()):
            p.fill_(((i * 37) % 101 - 50) / 64.0)                # cheap, exactly representable, different per tensor
    ref = M.UNet2DConditionModel.from_unet(donor)
    assert tuple(ref.conv_in.weight.shape) == (320, 20, 3, 3)
    assert torch.equal(ref.conv_in.weight[:, :4], donor.conv_in.weight) and float(ref.conv_in.weight[:, 4:].abs().max()) == 0.0
    assert torch.equal(ref.conv_in.bias, donor.conv_in.bias)
    dsd, rsd = donor.state_dict(), ref.state_dict()
    for k, v in dsd.items():
        if k.startswith("conv_in."):
            continue
        assert torch.equal(rsd[k], v), k
    extra = sorted(set(rsd) - set(dsd))
    assert extra and all(k.startswith("man_blocks.") for k in extra)
    assert ref.config.sample_size["cross_attention_dim"] == 768    # the donor's config really went in as `sample_size`
    trained = {k: torch.full_like(v, 0.25) for k, v in rsd.items()}
    ref.load_state_dict(trained)                                   # strict, like the script
    assert all(float((v - 0.25).abs().max()) == 0.0 for v in ref.state_dict().values())
    with pytest.raises(RuntimeError):
        ref.load_state_dict({k: v for k, v in trained.items() if not k.startswith("man_blocks.0")})
