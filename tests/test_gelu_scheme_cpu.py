"""The GEGLU epilogues (mikudance_amd/csrc/gemm.hip gelu_fast2, used by gemm_kernel / gemm_sp / wsgemm) evaluate
the exact-erf GELU of the reference (diffusers GEGLU -> F.gelu(approximate='none'), src/models/attention.py:152-157) as
    gelu(x) = max(x, 0) - |x| / p(|x|)^16,   p = sum c_k |x|^k,  c_k = a_k 2^(1/16) 2^(-k/2),
a_k the coefficients of Abramowitz-Stegun 7.1.28 (erf(z) = 1 - 1 / (1 + a1 z + .. + a6 z^6)^16, |error| <= 3e-7): six FMAs, four
squarings, one reciprocal -- no exponential.  This restates the formula with the kernel's constants in fp32 on the CPU and pins
its accuracy against the fp64 erf GELU over the range fp16 activations can take."""
import math

import torch

C = [1.044273782e+00, 5.207516304e-02, 2.207699846e-02, 3.422739239e-03, 3.968613701e-05, 5.105520901e-05, 5.621299664e-06]


def gelu_fast(x):
    x = x.float()
    ax = x.abs()
    p = torch.full_like(x, C[6])
    for k in range(5, -1, -1):
        p = p * ax + C[k]
    for _ in range(4):
        p = p * p
    return x.clamp_min(0.0) - ax.clamp_max(3.0e38) * (1.0 / p)        # the product takes min(|x|, 3e38): the |x| = inf guard


def test_gelu_constants_are_abramowitz_stegun_7_1_28():
    a = [1.0, 0.0705230784, 0.0422820123, 0.0092705272, 0.0001520143, 0.0002765672, 0.0000430638]
    for k in range(7):
        assert abs(C[k] - a[k] * 2 ** (1 / 16) / 2 ** (k / 2)) <= 1e-9 * max(C[k], 1e-6) + 1e-15, k


def test_gelu_fast_matches_erf_gelu():
    x = torch.cat([torch.linspace(-12, 12, 200001), torch.tensor([0.0, -0.0, 65504.0, -65504.0, 1e-4, -1e-4, 25.0, -25.0])])
    ref = 0.5 * x.double() * (1.0 + torch.erf(x.double() / math.sqrt(2.0)))
    got = gelu_fast(x).double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    # |erf error| <= 3e-7 (A&S 7.1.28) -> |gelu error| <= 0.5 |x| 3e-7, plus the fp32 rounding of p carried through the 16th power
    # (relative 16 x 2^-23 on r <= 1/2); measured maximum 1.2e-6 |x|
    assert (err <= 1.3e-6 * x.double().abs() + 1e-9).all(), float((err / x.double().abs().clamp_min(1e-9)).max())
    # positive side: relative accuracy far below the fp16 rounding (4.9e-4) of the product h * gelu(g) the epilogue produces
    pos = x > 1e-3
    assert (err[pos] / ref[pos]).max().item() < 3e-6
    # negative tail: r is the result itself (no cancellation); the A&S error of erfc limits the relative accuracy only where
    # |gelu| is below 1e-3 of the activations' scale
    assert (err / ref.abs().clamp_min(1e-3)).max().item() < 1e-3
    assert gelu_fast(torch.tensor([0.0]))[0].item() == 0.0 and gelu_fast(torch.tensor([-65504.0]))[0].item() == 0.0
    assert gelu_fast(torch.tensor([65504.0]))[0].item() == 65504.0


def test_gelu_fast_non_finite_inputs():
    """|x| = inf: p^16 = inf, r = 0, and |x| r would be inf * 0 = NaN (the round-3 / round-4 form answered NaN to +inf).  The product now
    takes min(|x|, 3e38): gelu(+inf) = +inf, gelu(-inf) = 0 -- the limits of x Phi(x) -- and NaN still propagates through p and r.  (ATen's
    own erf GELU answers NaN to all three on this build: its CPU kernel evaluates x * 0.5 * (1 + erf(x / sqrt 2)) through a vectorised path
    whose +inf case is inf * 0.)  A non-finite pre-activation needs a non-finite INPUT of the GEMM: fp16 operands, K <= 5120 terms and an
    fp16 bias bound the fp32 accumulator by 2.2e13."""
    x = torch.tensor([float("inf"), float("-inf"), float("nan")])
    got = gelu_fast(x)
    assert got[0].item() == float("inf") and got[1].item() == 0.0 and torch.isnan(got[2])
    assert torch.isnan(torch.nn.functional.gelu(torch.tensor([float("nan")]))[0])
    big = torch.tensor([1e30, -1e30, 3e38, -3e38, 3.4e38, -3.4e38])     # finite, far beyond fp16, either side of the clamp: exact
    assert torch.equal(gelu_fast(big), big.clamp_min(0.0))


def test_quick_gelu_is_the_clip_activation():
    # CLIP's quick_gelu (transformers activations.QuickGELUActivation): x * sigmoid(1.702 x), the ACT_QUICKGELU epilogue
    x = torch.linspace(-10, 10, 2001)
    assert torch.allclose(x / (1.0 + torch.exp(-1.702 * x)), x * torch.sigmoid(1.702 * x), atol=1e-6)
