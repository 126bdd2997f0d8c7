"""CPU emulation of the online-softmax bookkeeping of the d = 40 attention kernel (mikudance_amd/csrc/attention_v2.h, FOLD path):
reference r kept fp16-representable and 4 octaves above the running row maximum, P = fp16(exp2(S')) with S' = q'.k - r coming
out of the matrix core (q' = fp16(q * scale * log2 e)), lazy rescale triggered by "some P >= 2" (bit 14 of the fp16, OR-reduced
over the 32-query block of a wave), denominator accumulated from the same fp16 P.  The emulation uses torch on the CPU with the
same roundings and checks the scheme -- not the kernel -- against an fp64 softmax on inputs the GPU tests do not sweep: drifting
maxima, huge logits, late outliers, rows that never trigger, ragged last tiles."""
import math

import pytest
import torch

KT, QB, OFF = 64, 32, 4.0          # key tile, queries per wave, octaves between reference and maximum


def emulate(q, k, v, scale):
    Lq, d = q.shape
    Lk = k.shape[0]
    qs = (q.float() * (scale * 1.4426950408889634)).half().float()          # Q pre-scaled, rounded to fp16 like the fragment
    out = torch.empty(Lq, d)
    ntrig = 0
    for q0 in range(0, Lq, QB):
        qq = qs[q0:q0 + QB]
        r = torch.zeros(qq.shape[0])                                        # reference per row (fp16 representable)
        o = torch.zeros(qq.shape[0], d)
        l = torch.zeros(qq.shape[0])
        for it, j0 in enumerate(range(0, Lk, KT)):
            kk, vv = k[j0:j0 + KT].float(), v[j0:j0 + KT].float()
            s = qq @ kk.t() - r[:, None]                                    # fp32 accumulate of fp16 products, -r exact (slot D)
            first = it == 0
            p = torch.exp2(s).half()
            trig = first or bool((p.float() >= 2.0).any() or not torch.isfinite(p.float()).all())
            if trig:
                ntrig += 0 if first else 1
                mloc = s.max(dim=1).values + OFF
                want = r + (mloc if first else mloc.clamp_min(0.0))
                r_new = want.clamp(-60000.0, 60000.0).half().float()
                dd = r_new - r
                r = r_new
                if not first:
                    o = o * torch.exp2(-dd)[:, None]
                    l = l * torch.exp2(-dd)
                s = s - dd[:, None]
                p = torch.exp2(s).half()
            pf = p.float()
            o = o + pf @ vv                                                 # fp16 P x fp16 V, fp32 accumulate
            l = l + pf.sum(dim=1)                                           # the ones row: same fp16 P
        out[q0:q0 + QB] = o / l[:, None]
    return out, ntrig


def exact(q, k, v, scale, prescaled=False):
    """fp64 softmax attention; prescaled: on the fp16-rounded q * scale * log2(e) the kernel feeds the matrix core (base-2 softmax),
    which isolates the bookkeeping from that one extra rounding of Q (2^-12 relative per element, as if Q had been produced by one
    more fp16 GEMM epilogue: it moves a logit of magnitude |s| by ~ |s| * 2^-11 / sqrt(d), visible only for |s| in the hundreds)."""
    if prescaled:
        qs = (q.float() * (scale * 1.4426950408889634)).half().double()
        a = torch.softmax((qs @ k.double().t()) * math.log(2.0), dim=1)
    else:
        a = torch.softmax((q.double() @ k.double().t()) * scale, dim=1)
    return (a @ v.double()).float()


def _rnd(*shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).half()


@pytest.mark.parametrize("case", ["random", "peaky", "drift", "late_outlier", "huge", "ragged", "never"])
def test_or_check_scheme_matches_exact_softmax(case):
    d, Lq, Lk = 40, 64, 1024
    q, k, v = _rnd(Lq, d, seed=1), _rnd(Lk, d, seed=2), _rnd(Lk, d, seed=3)
    scale = d ** -0.5
    if case == "peaky":
        q = (q.float() * 6).half()
    elif case == "drift":                       # the row maximum grows a little with every tile: many small raises
        k = (k.float() * torch.linspace(0.2, 3.0, Lk)[:, None]).half()
        q = (q.float().abs() * 2).half()
        k = k.float().abs().half()
    elif case == "late_outlier":
        k[Lk - 3] = (q[5].float() * 6).half()
        k[Lk // 2 + 7] = (q[40].float() * 5).half()
    elif case == "huge":                        # logits of several hundred: the reference has to travel far in one step
        q, k = (q.float() * 30).half(), (k.float() * 30).half()
    elif case == "ragged":
        Lk = 1000 - 7
        k, v = k[:Lk], v[:Lk]
    elif case == "never":                       # first tile holds the maximum of every row: no trigger afterwards
        k[:QB] = (q[:QB].float() * 3).half()
        k[QB:2 * QB] = (q[QB:].float() * 3).half()
    got, ntrig = emulate(q, k, v, scale)
    want = exact(q, k, v, scale, prescaled=(case == "huge"))
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    # fp16 P and the fp16 rounding of the pre-scaled Q bound the agreement (the kernel tests use the same 1e-2 * max + 1e-3 bound)
    assert err <= 1e-2 * ref + 1e-3, (case, err, ref)
    rel = float((got - want).norm() / want.norm())
    assert rel < 2e-2, (case, rel)
    if case == "never":
        assert ntrig == 0
    if case in ("drift", "late_outlier"):
        assert ntrig >= 1


def test_probabilities_stay_in_the_fp16_range():
    """Between two triggers no P can reach 2 (else it triggers), and right after a raise the largest P of a raised row is 2^-4."""
    d, Lk = 40, 512
    q, k = _rnd(QB, d, seed=11, scale=3.0), _rnd(Lk, d, seed=12)
    qs = (q.float() * (d ** -0.5 * 1.4426950408889634)).half().float()
    s = qs @ k[:KT].float().t()
    r = (s.max(dim=1).values + OFF).half().float()
    p = torch.exp2(s - r[:, None])
    assert p.max().item() <= 2.0 ** (-OFF + 0.01) * 1.01 and p.max().item() >= 2.0 ** (-OFF - 0.1)
    assert math.isclose(float(torch.tensor(2.0).half().view(torch.int16)) , 0x4000)    # "P >= 2" <=> bit 14 of the fp16 pattern
