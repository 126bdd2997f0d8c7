"""CPU emulation (fp32, same association as the kernels') of the statistics scheme of md_groupnorm_nhwc_f16 /
md_instnorm_spade_f16 (mikudance_amd/csrc/norm.hip): one sweep of sum(x - k), sum((x - k)^2) around a pilot k -- the group's first
channel at pixel 0 (GroupNorm) or the channel's own pixel-0 value (InstanceNorm) -- then mean = k + E[x - k],
var = E[(x - k)^2] - E[x - k]^2.  Checked against fp64 statistics where the plain E[x^2] - mean^2 form loses the variance."""
import pytest
import torch


def _stats_shifted(x, k):
    d = x.float() - k
    n = d.shape[-1]
    a = d.cumsum(-1)[..., -1]                    # sequential fp32 accumulation: the pessimistic association
    c2 = (d * d).cumsum(-1)[..., -1]
    mu = a / n
    return k.squeeze(-1) + mu, (c2 / n - mu * mu).clamp_min(0.0)


def _stats_plain(x):
    f = x.float()
    n = f.shape[-1]
    a, c2 = f.cumsum(-1)[..., -1], (f * f).cumsum(-1)[..., -1]
    mu = a / n
    return mu, (c2 / n - mu * mu).clamp_min(0.0)


@pytest.mark.parametrize("offset,sigma", [(0.0, 1.0), (20.0, 0.5), (200.0, 0.25), (-1000.0, 2.0), (3.0, 1e-3)])
def test_groupnorm_pilot_shift(offset, sigma):
    B, HW, G, cpg = 2, 9216, 8, 10
    gen = torch.Generator().manual_seed(5)
    offs = offset * (1.0 + 0.1 * torch.rand(B, 1, G, 1, generator=gen))
    x = (offs + sigma * torch.randn(B, HW, G, cpg, generator=gen)).half()        # NHWC tensor, channels grouped
    xg = x.permute(0, 2, 1, 3).reshape(B, G, HW * cpg)                           # one row per (image, group)
    ref_mu, ref_var = xg.double().mean(-1), xg.double().var(-1, unbiased=False)
    k = x[:, 0, :, 0].float().unsqueeze(-1)                                      # pilot: first channel of the group at pixel 0
    mu, var = _stats_shifted(xg, k)
    assert ((var.double() - ref_var).abs() / ref_var).max().item() < 1e-4
    assert (mu.double() - ref_mu).abs().max().item() <= 1e-5 * max(1.0, abs(offset))
    if abs(offset) / sigma >= 400:                                               # ... where the plain one-sweep form is far off
        _, pv = _stats_plain(xg)
        assert ((pv.double() - ref_var).abs() / ref_var).max().item() > 1e-2


def test_instancenorm_pilot_shift():
    B, HW, C = 2, 2304, 16
    gen = torch.Generator().manual_seed(6)
    x = ((torch.rand(B, 1, C, generator=gen) * 2 - 1) * 100 + 0.25 * torch.randn(B, HW, C, generator=gen)).half()
    xc = x.permute(0, 2, 1)                                                      # (B, C, HW)
    k = x[:, 0, :].float().unsqueeze(-1)                                         # the channel's own value at pixel 0
    mu, var = _stats_shifted(xc, k)
    ref_mu, ref_var = xc.double().mean(-1), xc.double().var(-1, unbiased=False)
    assert ((var.double() - ref_var).abs() / ref_var).max().item() < 1e-4 and (mu.double() - ref_mu).abs().max().item() < 1e-3
