"""The GEGLU epilogues (mikudance_amd/csrc/gemm.hip gelu_fast, used by gemm_kernel / gemm_pp / gemm_ppg / wsgemm) evaluate the
exact-erf GELU of the reference (diffusers GEGLU -> F.gelu(approximate='none'), src/models/attention.py:152-157) with the
Abramowitz-Stegun 7.1.26 rational form of erf: one reciprocal, one exp2, seven FMAs.  This restates the formula with the same
constants in fp32 on the CPU and pins its accuracy against the fp64 erf GELU over the range fp16 activations can take."""
import math

import torch


def gelu_fast(x):
    x = x.float()
    z = x.abs() * 0.70710678118654752
    t = 1.0 / (1.0 + 0.3275911 * z)
    poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))
    e = 1.0 - poly * torch.exp2(-1.4426950408889634 * z * z)
    return 0.5 * x * (1.0 + torch.copysign(e, x))


def test_gelu_fast_matches_erf_gelu():
    x = torch.cat([torch.linspace(-12, 12, 200001), torch.tensor([0.0, -0.0, 65504.0, -65504.0, 1e-4, -1e-4])])
    ref = 0.5 * x.double() * (1.0 + torch.erf(x.double() / math.sqrt(2.0)))
    got = gelu_fast(x).double()
    err = (got - ref).abs()
    # |erf error| <= 1.5e-7 (A&S) -> |gelu error| <= 0.5 * |x| * 1.5e-7 + fp32 rounding
    bound = 0.5 * x.double().abs() * 1.5e-7 + 4e-7 * ref.abs() + 1e-7
    assert (err <= bound).all(), float((err - bound).max())
    # far below fp16 resolution of the product h * gelu(g) the epilogue rounds to
    assert (err / ref.abs().clamp_min(1e-3)).max().item() < 5e-4       # |err| <= 5e-7 where |gelu| < 1e-3
    assert gelu_fast(torch.tensor([0.0]))[0].item() == 0.0 and gelu_fast(torch.tensor([-65504.0]))[0].item() == 0.0


def test_quick_gelu_is_the_clip_activation():
    # CLIP's quick_gelu (transformers activations.QuickGELUActivation): x * sigmoid(1.702 x), the ACT_QUICKGELU epilogue
    x = torch.linspace(-10, 10, 2001)
    assert torch.allclose(x / (1.0 + torch.exp(-1.702 * x)), x * torch.sigmoid(1.702 * x), atol=1e-6)
