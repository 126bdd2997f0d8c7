"""Exploration (CPU, not part of the product): rounding behaviour of LayerNorm FOLDED into its consumer Linear (DESIGN.md 8b, round-3 open
item 4) against the literal order, both measured against float64.

  literal:  n = fp16(LN(x) * gamma + beta);  y = fp16(fp32(n @ W^T) + b)
  folded:   W' = fp16(gamma * W), s = fp32(sum_k W'), c = fp32(beta @ W^T + b);  y = fp16(rstd * (fp32(x @ W'^T) - mu * s) + c)

x fp16 rows with mean/sigma ratios up to 100 (the regime where `acc - mu * s` cancels), W ~ N(0, 1/K), gamma ~ 1 +- 0.2, beta ~ 0.1."""
import torch

torch.manual_seed(0)


def run(M, K, N, ratio):
    x = (torch.randn(M, K, dtype=torch.float64) + ratio * torch.randn(M, 1, dtype=torch.float64)).half()
    W = (torch.randn(N, K, dtype=torch.float64) * K ** -0.5).half()
    g = (1 + 0.2 * torch.randn(K, dtype=torch.float64)).half()
    be = (0.1 * torch.randn(K, dtype=torch.float64)).half()
    b = (0.1 * torch.randn(N, dtype=torch.float64)).half()
    xd, Wd, gd, bed, bd = (t.double() for t in (x, W, g, be, b))
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    exact = ((xd - mu) * rstd * gd + bed) @ Wd.t() + bd
    # literal (fp32 statistics, one fp16 rounding of n, fp32 accumulate, one rounding of y)
    xf = x.float()
    muf, varf = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
    rf = (varf + 1e-5).rsqrt()
    n = ((xf - muf) * rf * g.float() + be.float()).half()
    lit = (n.float() @ W.float().t() + b.float()).half()
    # folded
    Wp = (g.float() * W.float()).half()
    s = Wp.float().sum(1)
    c = be.float() @ W.float().t() + b.float()
    acc = xf @ Wp.float().t()
    fold = (rf * (acc - muf * s) + c).half()
    e = lambda y: float((y.double() - exact).norm() / exact.norm())
    return e(lit), e(fold)


print("%8s %6s %6s %6s   %10s %10s" % ("|mu|/sig", "M", "K", "N", "literal", "folded"))
for ratio in (0, 1, 10, 30, 100):
    for K, N in ((320, 320), (1280, 1280), (320, 2560)):
        l, f = run(2048, K, N, ratio)
        print("%8g %6d %6d %6d   %10.2e %10.2e" % (ratio, 2048, K, N, l, f))
