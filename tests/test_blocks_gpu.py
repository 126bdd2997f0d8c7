"""GPU: the HIP building blocks (through the C ABI) DIRECTLY against golden vectors produced by the reference's own,
unmodified modules (tests/golden/g3_blocks.safetensors <- oracle/gen_golden.py: src/models/resnet.py ResnetBlock3D /
Downsample3D / Upsample3D, src/models/man_module.py MANModule).  fp16 io / fp32 accumulate vs the fp32 reference:
|err| <= 1e-2 * maxabs(ref) + 1e-3."""
import os

import pytest
import torch
import torch.nn.functional as F
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu

from parity_budget import check_kernel  # noqa: E402

from mikudance_amd import blocks  # noqa: E402
from mikudance_amd.synth import synth_state_dict  # noqa: E402


def close(got, ref, what):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    bound = 1e-2 * ref.abs().max().item() + 1e-3
    assert got.shape == ref.shape and err <= bound, f"{what}: max err {err:.4g} > {bound:.4g}"
    check_kernel(what, got, ref)


def nhwc(x5):                      # (b, c, f, h, w) -> (b*f, h, w, c) fp16 on the GPU
    b, c, f, h, w = x5.shape
    return x5.permute(0, 2, 3, 4, 1).reshape(b * f, h, w, c).contiguous().half().cuda()


def ncfhw(y, b):                   # (b*f, h, w, c) -> (b, c, f, h, w)
    bf, h, w, c = y.shape
    return y.float().cpu().reshape(b, bf // b, h, w, c).permute(0, 4, 1, 2, 3)


def load(mod, shapes, seed):
    mod.load_state_dict(synth_state_dict(shapes, seed=seed), strict=True)
    return mod.half().cuda().eval()


def test_g3_reference_resnet_down_up_man(golden_dir):
    t = load_file(os.path.join(golden_dir, "g3_blocks.safetensors"))
    y1, temb = t["resnet.y"], t["resnet.temb"]                       # output of the reference's first ResnetBlock3D
    b, c, f, h, w = y1.shape
    rs = {"norm1.weight": (64,), "norm1.bias": (64,), "conv1.weight": (64, 64, 3, 3), "conv1.bias": (64,),
          "time_emb_proj.weight": (64, 128), "time_emb_proj.bias": (64,), "norm2.weight": (64,), "norm2.bias": (64,),
          "conv2.weight": (64, 64, 3, 3), "conv2.bias": (64,)}
    sd = synth_state_dict(rs, seed=12)
    blk = load(blocks.ResnetBlock(64, 64, 128), rs, 12)
    rows = F.linear(F.silu(temb), sd["time_emb_proj.weight"], sd["time_emb_proj.bias"]).half().cuda()   # resnet.py:226
    with torch.no_grad():
        out = blk(nhwc(y1), rows, f * h * w)
        close(ncfhw(out, b), t["resnet_same.y"], "ResnetBlock3D")
        cs = {"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}
        close(ncfhw(load(blocks.ConvSampler(64, up=False), cs, 13)(nhwc(y1)), b), t["down.y"], "Downsample3D")
        close(ncfhw(load(blocks.ConvSampler(64, up=True), cs, 14)(nhwc(y1)), b), t["up.y"], "Upsample3D")
        ms = {"mlp_shared.0.weight": (128, 2, 3, 3), "mlp_shared.0.bias": (128,), "mlp_gamma.weight": (64, 128, 3, 3),
              "mlp_gamma.bias": (64,), "mlp_beta.weight": (64, 128, 3, 3), "mlp_beta.bias": (64,)}
        man = load(blocks.MANModule(64, 2), ms, 15)
        x, motion = t["man.x"], t["man.motion"]
        m = F.interpolate(motion, size=x.shape[2:], mode="nearest")                                       # man_module.py:25
        m64 = torch.zeros(m.shape[0], x.shape[2], x.shape[3], 64)
        m64[..., :2] = m.permute(0, 2, 3, 1)
        out = man(x.permute(0, 2, 3, 1).contiguous().half().cuda(), m64.half().cuda())
        close(out.float().cpu().permute(0, 3, 1, 2), t["man.y"], "MANModule")
