"""GPU: every HIP kernel, called through the C ABI (mikudance_amd.ops -> libmdance_hip.so), against an fp32 PyTorch
restatement of the same op (the oracle's leaf ops) fed the SAME fp16-rounded inputs.
Tolerance: |err| <= 1e-2 * maxabs(ref) + 1e-3  (fp16 io, fp32 accumulate)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from parity_budget import check_kernel  # noqa: E402

from mikudance_amd import ops, packing  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def close(got, ref, rtol=1e-2, atol=1e-3, what=""):
    got = got.float().cpu()
    ref = ref.float()
    err = (got - ref).abs().max().item()
    bound = rtol * ref.abs().max().item() + atol
    assert math.isfinite(err) and err <= bound, f"{what}: max err {err:.4g} > {bound:.4g}"
    check_kernel(what, got, ref)


# --------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 200, 192), (7, 320, 64), (130, 4, 64), (1000, 1288, 320)])
def test_gemm_plain_edges(dev, M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    # asymmetric operands: catches row/col or fragment transposes
    ref = a.float() @ w.float().t()
    out = ops.gemm(a.to(dev), w.to(dev))
    close(out, ref, what=f"gemm {M}x{N}x{K}")


def test_gemm_identity_asymmetric(dev):
    K = 128
    a = torch.eye(K).half()
    w = (torch.arange(K * K).reshape(K, K) % 97).half() / 16
    out = ops.gemm(a.to(dev), w.to(dev))            # = W^T
    assert torch.equal(out.cpu().float(), w.float().t())


def test_gemm_epilogues(dev):
    M, N, K = 384, 320, 256
    a, w = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=K ** -0.5)
    bias, res, radd = rnd(N, seed=5), rnd(M, N, seed=6), rnd(3, N, seed=7)
    base = a.float() @ w.float().t() + bias.float()
    d = lambda t: t.to(dev)
    close(ops.gemm(d(a), d(w), bias=d(bias)), base, what="bias")
    close(ops.gemm(d(a), d(w), bias=d(bias), act=ops.ACT_SILU), F.silu(base), what="silu")
    close(ops.gemm(d(a), d(w), bias=d(bias), act=ops.ACT_RELU), F.relu(base), what="relu")
    close(ops.gemm(d(a), d(w), bias=d(bias), residual=d(res)), base + res.float(), what="residual")
    ref = base + radd.float().repeat_interleave(128, 0)
    close(ops.gemm(d(a), d(w), bias=d(bias), rowadd=d(radd), rows_per_group=128), ref, what="rowadd")
    # strided A (a column slice of a wider matrix) and strided residual
    wide = rnd(M, 2 * K, seed=8)
    close(ops.gemm(d(wide)[:, K:], d(w)), wide[:, K:].float() @ w.float().t(), what="lda")


def test_gemm_transpose_out(dev):
    for M, N, K in [(264, 128, 64), (521, 320, 128), (36, 64, 64)]:
        a, w = rnd(M, K, seed=9), rnd(N, K, seed=10, scale=K ** -0.5)
        out = ops.gemm(a.to(dev), w.to(dev), transpose_out=True)
        assert out.shape == (N, M)
        close(out, (a.float() @ w.float().t()).t(), what=f"transpose {M}")


def test_gemm_geglu(dev):
    M, K, inner = 200, 128, 256
    a = rnd(M, K, seed=11)
    w, b = rnd(2 * inner, K, seed=12, scale=K ** -0.5), rnd(2 * inner, seed=13)
    hg = a.float() @ w.float().t() + b.float()
    ref = hg[:, :inner] * F.gelu(hg[:, inner:])
    wp, bp = packing.geglu_weight(w, b, dev)
    out = ops.gemm(a.to(dev), wp, bias=bp, act=ops.ACT_GEGLU)
    assert out.shape == (M, inner)
    close(out, ref, what="geglu")


@pytest.mark.parametrize("M,inner", [(40000, 1280), (294912, 1280), (33024, 256)])
def test_gemm_geglu_weight_stationary_streaming_kernel(dev, M, inner):
    """FeedForward net.0 at C = 320 (K = 320, N = 8C packed h|g, M >= 32768): the GEGLU flavour of the streaming kernel, whose four
    memory waves share the GELU arithmetic (two of them hand their finished pieces through LDS).  Every element is checked, three
    launches each (the hand-off pipeline has a one-tile skew: a synchronisation slip shows up as stale pieces)."""
    K = 320
    a = rnd(M, K, seed=21)
    w, b = rnd(2 * inner, K, seed=22, scale=K ** -0.5), rnd(2 * inner, seed=23)
    wp, bp = packing.geglu_weight(w, b, dev)
    ad = a.to(dev)
    hg = ad.float() @ w.to(dev).float().t() + b.to(dev).float()
    ref = (hg[:, :inner] * F.gelu(hg[:, inner:])).cpu()
    del hg
    for _ in range(3):
        out = ops.gemm(ad, wp, bias=bp, act=ops.ACT_GEGLU)
        assert out.shape == (M, inner)
        close(out, ref, what="ws geglu")


@pytest.mark.parametrize("M,N,K", [(40000, 320, 320), (32771, 960, 320), (36000, 640, 640), (33001, 128, 640), (294912, 320, 320)])
def test_gemm_weight_stationary_streaming_kernel(dev, M, N, K):
    """The W-stationary streaming GEMM (gemm_ws.h) that the dispatcher picks for the HBM-bound short-K projections
    (K = 320 / 640, M >= 32768): plain, bias + row-broadcast + residual, in-place residual, ragged M, several column groups."""
    a, w, b = rnd(M, K, seed=90), rnd(N, K, seed=91, scale=K ** -0.5), rnd(N, seed=92)
    rpg = 5000
    res, ra = rnd(M, N, seed=93), rnd((M + rpg - 1) // rpg, N, seed=94)
    ad, wd = a.to(dev), w.to(dev)
    ref = a.float() @ w.float().t()
    close(ops.gemm(ad, wd), ref, what="ws plain")
    full = ref + b.float() + ra.float().repeat_interleave(rpg, 0)[:M] + res.float()
    close(ops.gemm(ad, wd, bias=b.to(dev), residual=res.to(dev), rowadd=ra.to(dev), rows_per_group=rpg), full, what="ws epilogue")
    hs = res.to(dev).clone()
    ops.gemm(ad, wd, bias=b.to(dev), residual=hs, out=hs)                      # the in-place form of blocks.TransformerBlock
    close(hs, ref + b.float() + res.float(), what="ws in-place residual")
    if N >= 640:                                                               # A and C as column slices of wider matrices
        wide_a = torch.zeros(M, K + 64, device=dev, dtype=torch.float16)
        wide_a[:, 64:] = ad
        wide_c = torch.zeros(M, N + 32, device=dev, dtype=torch.float16)
        ops.gemm(wide_a[:, 64:], wd, out=wide_c[:, 32:])
        close(wide_c[:, 32:], ref, what="ws strided")
        assert float(wide_c[:, :32].abs().max()) == 0.0


def test_gemm_residual_aliasing_contract(dev):
    """include/mdance_hip.h 'Aliasing': residual == output (same pointer, same pitch) is supported by every GEMM flavour the
    dispatcher can pick; a residual that partially overlaps the output is refused."""
    from mikudance_amd._lib import MdanceHipError
    for M, N, K in [(300, 320, 320), (4608, 1280, 1280), (40000, 320, 320)]:
        a, w, r = rnd(M, K, seed=95), rnd(N, K, seed=96, scale=K ** -0.5), rnd(M, N, seed=97)
        hs = r.to(dev).clone()
        ops.gemm(a.to(dev), w.to(dev), residual=hs, out=hs)
        close(hs, a.float() @ w.float().t() + r.float(), what=f"in-place residual M={M}")
    buf = torch.zeros(301, 320, device=dev, dtype=torch.float16)
    with pytest.raises(MdanceHipError, match="overlaps"):
        ops.gemm(rnd(300, 320).to(dev), rnd(320, 320).to(dev), residual=buf[1:], out=buf[:300])


def test_gemm_rejects_bad_k(dev):
    from mikudance_amd._lib import MdanceHipError
    with pytest.raises(MdanceHipError):
        ops.gemm(rnd(8, 40).to(dev), rnd(8, 40).to(dev))


# --------------------------------------------------------------------------------------------- conv
@pytest.mark.parametrize("cin,cout,h,w,stride,up", [(64, 64, 8, 8, 1, False), (20, 64, 9, 7, 1, False), (128, 192, 12, 12, 2, False),
                                                    (64, 128, 6, 5, 1, True), (4, 4, 16, 16, 1, False), (64, 64, 7, 7, 2, False)])
def test_conv3x3(dev, cin, cout, h, w, stride, up):
    B = 3
    x = rnd(B, cin, h, w, seed=20)
    wt = rnd(cout, cin, 3, 3, seed=21, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=22)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, wt.float(), bias.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
    cp = packing.pad_to(cin, 64)
    xn = torch.zeros(B, h, w, cp, dtype=torch.float16)
    xn[..., :cin] = x.permute(0, 2, 3, 1)
    out = ops.conv3x3(xn.to(dev), packing.conv3x3_weight(wt, dev), cout, bias=bias.to(dev), stride=stride, upsample=up)
    assert tuple(out.shape) == tuple(ref.shape)
    close(out, ref, what="conv")


def test_conv3x3_fused_temb_residual(dev):
    B, c, h, w = 4, 64, 8, 8
    x, wt, bias = rnd(B, h, w, c, seed=23), rnd(c, c, 3, 3, seed=24, scale=(9 * c) ** -0.5), rnd(c, seed=25)
    temb, res = rnd(2, c, seed=26), rnd(B, h, w, c, seed=27)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), bias.float(), padding=1).permute(0, 2, 3, 1)
    ref = ref + temb.float().repeat_interleave(2, 0)[:, None, None, :] + res.float()
    out = ops.conv3x3(x.to(dev), packing.conv3x3_weight(wt, dev), c, bias=bias.to(dev), residual=res.to(dev),
                      rowadd=temb.to(dev), rows_per_group=2 * h * w)
    close(out, ref, what="conv+temb+res")


# --------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,HW,C,silu,eps", [(2, 64, 64, True, 1e-5), (3, 100, 320, False, 1e-6), (2, 37, 960, True, 1e-5),
                                             (1, 2304, 640, True, 1e-5), (2, 16, 2560, False, 1e-5)])
def test_groupnorm(dev, B, HW, C, silu, eps):
    x = rnd(B, HW, C, seed=30) * 2 + 0.5
    g, b = (1 + 0.1 * rnd(C, seed=31).float()).half(), rnd(C, seed=32)
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, g.float(), b.float(), eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    close(ops.groupnorm(x.to(dev), g.to(dev), b.to(dev), 32, eps, silu), ref, what="groupnorm")


@pytest.mark.parametrize("M,C", [(10, 64), (301, 320), (64, 1280), (5, 256), (7, 320), (2053, 320)])
def test_layernorm_and_adds(dev, M, C):
    x = rnd(M, C, seed=33) * 3 + 1
    g, b = (1 + 0.1 * rnd(C, seed=34).float()).half(), rnd(C, seed=35)
    ref = F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5)
    close(ops.layernorm(x.to(dev), g.to(dev), b.to(dev)), ref, what="ln")
    half = M // 2
    bank = rnd(M - half, C, seed=36)
    y, y2 = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), add=bank.to(dev), add_mode=1, add_row_begin=half)
    close(y, ref, what="ln y")
    ref2 = ref.clone()
    ref2[half:] += bank.float()
    close(y2, ref2, what="ln + bank")
    frames, rpf = 3, 2
    pe = rnd(frames, C, seed=37)
    y, y2 = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), add=pe.to(dev), add_mode=2, rows_per_frame=rpf, frames=frames)
    idx = (torch.arange(M) // rpf) % frames
    close(y2, ref + pe.float()[idx], what="ln + pe")


def test_instnorm_spade(dev):
    B, HW, C = 3, 144, 128
    x, gb = rnd(B, HW, C, seed=38) * 2 + 1, rnd(B, HW, 2 * C, seed=39)
    n = F.instance_norm(x.float().permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    ref = n * (1 + gb.float()[..., :C]) + gb.float()[..., C:]
    close(ops.instnorm_spade(x.to(dev), gb.to(dev)), ref, what="man")


def test_norm_statistics_with_large_mean(dev):
    """|mean| >> sigma (activation statistics of real checkpoints: channel groups riding on an offset of tens of sigma): the
    one-sweep E[x^2] - mean^2 in fp32 loses the variance there; the kernels accumulate around a pilot value instead.  Reference
    in fp64 on the same fp16-rounded inputs."""
    B, HW, C, G = 2, 9216, 320, 32
    gen = torch.Generator().manual_seed(77)
    offs = (torch.rand(B, 1, G, 1, generator=gen) * 2 - 1) * 200.0                      # per (image, group) offset up to +-200
    x = (offs + 0.25 * torch.randn(B, HW, G, C // G, generator=gen)).reshape(B, HW, C).half()
    g, b = (1 + 0.1 * rnd(C, seed=78).float()).half(), rnd(C, seed=79)
    ref = F.group_norm(x.double().permute(0, 2, 1), G, g.double(), b.double(), 1e-5).permute(0, 2, 1).float()
    out = ops.groupnorm(x.to(dev), g.to(dev), b.to(dev), G, 1e-5, False)
    close(out, ref, what="groupnorm, |mean| = 800 sigma")
    # instance norm (MAN): per-channel offsets
    B, HW, C = 2, 2304, 128
    offs = (torch.rand(B, 1, C, generator=gen) * 2 - 1) * 100.0
    x = (offs + 0.25 * torch.randn(B, HW, C, generator=gen)).half()
    gb = rnd(B, HW, 2 * C, seed=80)
    n = F.instance_norm(x.double().permute(0, 2, 1), eps=1e-5).permute(0, 2, 1).float()
    ref = n * (1 + gb.float()[..., :C]) + gb.float()[..., C:]
    close(ops.instnorm_spade(x.to(dev), gb.to(dev)), ref, what="instance norm, |mean| = 400 sigma")


# --------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, B, H, D, Lq, Lk, kv_index=None):
    qh = q.float().view(B, Lq, H, D).transpose(1, 2)
    nkv = k.shape[0] // Lk
    kh = k.float().view(nkv, Lk, H, D).transpose(1, 2)
    vh = v.float().view(nkv, Lk, H, D).transpose(1, 2)
    if kv_index is not None:
        kh, vh = kh[kv_index], vh[kv_index]
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(B * Lq, H * D)


@pytest.mark.parametrize("D,Lq,Lk", [(8, 256, 256), (16, 64, 64), (32, 16, 16), (32, 4, 4), (40, 200, 333), (80, 130, 64),
                                     (160, 144, 144), (64, 33, 257), (40, 576, 576),
                                     # fast path (Lk % 8 == 0) with a ragged last key tile at every K-row swizzle width (D/8 = 2, 4, 8, 10, 20 slots)
                                     (16, 40, 72), (32, 50, 136), (64, 100, 200), (80, 100, 200), (160, 70, 136)])
def test_attention_self(dev, D, Lq, Lk):
    B, H = 2, 8
    q, k, v = rnd(B * Lq, H * D, seed=40), rnd(B * Lk, H * D, seed=41), rnd(B * Lk, H * D, seed=42)
    ref = _attn_ref(q, k, v, B, H, D, Lq, Lk)
    vt = v.t().contiguous()
    out = ops.attention(q.to(dev), k.to(dev), vt.to(dev), B, H, D, Lq, Lk)
    close(out, ref, what=f"attn D={D}")


@pytest.mark.parametrize("D,L,qscale", [(40, 9216, 1.0), (40, 9216, 3.0), (80, 2304, 1.0), (160, 576, 1.0)])
def test_attention_at_benchmark_sequence_lengths(dev, D, L, qscale):
    """The BASELINE configs[1] self-attention shapes themselves (96x96 / 48x48 / 24x24 latents -> Lq = Lk = 9216 / 2304 / 576):
    144 key tiles of lazy rescaling, fp16 P, ones-row denominator and (d = 40) the folded softmax reference accumulate 9x
    longer than in any smaller case.  qscale = 3 makes the rows peaky (|s| up to ~12) so the running reference moves late."""
    B, H = 1, 8
    q, k, v = rnd(B * L, H * D, seed=70, scale=qscale), rnd(B * L, H * D, seed=71), rnd(B * L, H * D, seed=72)
    ref = _attn_ref(q, k, v, B, H, D, L, L)
    out = ops.attention(q.to(dev), k.to(dev), v.t().contiguous().to(dev), B, H, D, L, L)
    # the output of a 9216-key average of N(0,1) values is small (std ~ 0.01..0.3): bound the error relative to the rms too
    got, want = out.float().cpu(), ref.float()
    close(got, want, what=f"attn D={D} L={L}")
    rel = float((got - want).norm() / want.norm())
    assert rel < 2e-2, rel


def test_attention_long_forces_rescale(dev):
    """Same at a length the long-sequence flavours take (Lq >= 1024, Lk % 64 == 0): late keys dominating a few query rows, in
    different key tiles, so the lazy rescale fires after many tiles of accumulated O^T."""
    B, H, D, L = 1, 8, 40, 2048
    q, k, v = rnd(L, H * D, seed=46), rnd(L, H * D, seed=47), rnd(L, H * D, seed=48)
    k[1500] = q[5] * 4
    k[2047] = q[1030] * 5
    k[64] = q[2000] * 3
    ref = _attn_ref(q, k, v, B, H, D, L, L)
    out = ops.attention(q.to(dev), k.to(dev), v.t().contiguous().to(dev), B, H, D, L, L)
    close(out, ref, what="attn long rescale")


def test_attention_forces_rescale(dev):
    """Online softmax: a late key dominating one query row forces the max-rescale branch (CDNA guide rule 26)."""
    B, H, D, L = 1, 8, 40, 320
    q, k, v = rnd(L, H * D, seed=43), rnd(L, H * D, seed=44), rnd(L, H * D, seed=45)
    k[300] = q[5] * 4
    ref = _attn_ref(q, k, v, B, H, D, L, L)
    out = ops.attention(q.to(dev), k.to(dev), v.t().contiguous().to(dev), B, H, D, L, L)
    close(out, ref, what="attn rescale")


def test_attention_cross_padded_kv_index(dev):
    """Lk = 257 CLIP tokens, K/V batches padded to a stride of 264, query batches mapped through kv_index."""
    B, H, D, Lq, Lk, stride = 6, 8, 40, 100, 257, 264
    q = rnd(B * Lq, H * D, seed=46)
    kk, vv = rnd(2, Lk, H * D, seed=47), rnd(2, Lk, H * D, seed=48)
    kpad, vpad = torch.zeros(2, stride, H * D).half(), torch.zeros(2, stride, H * D).half()
    kpad[:, :Lk], vpad[:, :Lk] = kk, vv
    idx = torch.tensor([0, 1, 1, 0, 1, 0], dtype=torch.int32)
    ref = _attn_ref(q, kk.reshape(-1, H * D), vv.reshape(-1, H * D), B, H, D, Lq, Lk, kv_index=idx.long())
    out = ops.attention(q.to(dev), kpad.reshape(-1, H * D).to(dev), vpad.reshape(-1, H * D).t().contiguous().to(dev), B, H, D,
                        Lq, Lk, kv_stride=stride, kv_index=idx.to(dev))
    close(out, ref, what="cross attn")


def test_attention_via_gemm_vt(dev):
    """V^T produced by the GEMM's transposed store feeds attention directly (the production data flow)."""
    B, H, D, L, C = 2, 8, 40, 96, 320
    x, wv = rnd(B * L, C, seed=49), rnd(C, C, seed=50, scale=C ** -0.5)
    q, k = rnd(B * L, C, seed=51), rnd(B * L, C, seed=52)
    v = (x.float() @ wv.float().t()).half()
    ref = _attn_ref(q, k, v, B, H, D, L, L)
    vt = ops.gemm(x.to(dev), wv.to(dev), transpose_out=True)
    out = ops.attention(q.to(dev), k.to(dev), vt, B, H, D, L, L)
    close(out, ref, rtol=2e-2, what="attn via gemm")


@pytest.mark.parametrize("F_,HW,D,NB", [(4, 16, 8, 2), (16, 9, 40, 2), (6, 5, 32, 2), (24, 4, 80, 2), (32, 3, 160, 2), (30, 7, 40, 2), (16, 5, 80, 2), (16, 6, 160, 2),
                                        (8, 7, 160, 2), (5, 11, 40, 2), (13, 3, 80, 2), (16, 7, 40, 1), (3, 5, 160, 3),
                                        # frame counts that are no multiple of 16 at every head width: the Q / K / V images are packed back to
                                        # back in LDS and the DMA lanes past the last chunk are switched off, not rounded to 1-KiB rows
                                        (7, 9, 40, 2), (29, 5, 80, 1), (13, 5, 160, 2), (31, 3, 160, 1)])
def test_temporal_attention(dev, F_, HW, D, NB):
    H = 8
    C = H * D
    q, k, v = rnd(NB * F_ * HW, C, seed=53), rnd(NB * F_ * HW, C, seed=54), rnd(NB * F_ * HW, C, seed=55)

    def fold(t):  # (b f) d c -> (b d) f c -> heads
        return t.float().view(NB, F_, HW, H, D).permute(0, 2, 3, 1, 4)      # b d h f D
    o = F.scaled_dot_product_attention(fold(q), fold(k), fold(v))          # b d h f D
    ref = o.permute(0, 3, 1, 2, 4).reshape(NB * F_ * HW, C)
    out = ops.temporal_attention(q.to(dev), k.to(dev), v.to(dev), NB, F_, HW, H, D)
    close(out, ref, what="temporal")


# --------------------------------------------------------------------------------------------- elementwise
def test_pack_unpack_concat(dev):
    b, c, f, h, w = 2, 4, 3, 8, 6
    x = torch.randn(b, c, f, h, w, generator=torch.Generator().manual_seed(60))
    xd = x.to(dev)
    st = xd.stride()
    p = ops.pack_nhwc(xd, b * f, f, (st[0], st[2], st[1], st[3], st[4]), 0, c, 64, h, w)
    ref = torch.zeros(b * f, h, w, 64)
    ref[..., :c] = x.permute(0, 2, 3, 4, 1).reshape(b * f, h, w, c)
    assert torch.equal(p.cpu().float(), ref.half().float())
    # channel sub-range + nearest sub-sampling (motion map of MAN)
    m = torch.randn(5, 22, 16, 16, generator=torch.Generator().manual_seed(61)).half().to(dev)
    st = m.stride()
    pm = ops.pack_nhwc(m, 5, 1, (st[0], 0, st[1], st[2], st[3]), 20, 2, 64, 4, 4, hin=16, win=16)
    refm = F.interpolate(m[:, 20:].float().cpu(), size=(4, 4), mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(pm.cpu().float()[..., :2], refm)
    pm = ops.pack_nhwc(m, 5, 1, (st[0], 0, st[1], st[2], st[3]), 20, 2, 64, 5, 3, hin=16, win=16)
    refm = F.interpolate(m[:, 20:].float().cpu(), size=(5, 3), mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(pm.cpu().float()[..., :2], refm)
    assert pm[..., 2:].abs().max().item() == 0
    # unpack back to NCFHW fp32
    y = torch.zeros(b, c, f, h, w, device=dev)
    st = y.stride()
    ops.unpack_nhwc(p, y, b * f, f, (st[0], st[2], st[1], st[3], st[4]), c, h, w)
    assert torch.equal(y.cpu(), x.half().float())
    a, bb = rnd(7, 5, 64, seed=62), rnd(7, 5, 128, seed=63)
    assert torch.equal(ops.concat_channels(a.to(dev), bb.to(dev)).cpu(), torch.cat([a, bb], -1))


def test_window_accumulate_and_ddim(dev):
    Ftot, f, HW = 6, 4, 20
    sch = O.DDIM()
    sch.set_timesteps(4)
    lat = rnd(Ftot, HW, 4, seed=64)
    noise = torch.zeros(2, Ftot, HW, 4, device=dev)
    cnt = torch.zeros(Ftot, device=dev)
    ref_noise, ref_cnt = torch.zeros(2, Ftot, HW, 4), torch.zeros(Ftot)
    for wi, win in enumerate([[0, 1, 2, 3], [2, 3, 4, 5], [4, 5, 0, 1]]):
        pred = rnd(2 * f, HW, 4, seed=70 + wi)
        ops.window_accumulate(pred.to(dev), noise, cnt, torch.tensor(win, dtype=torch.int32, device=dev), f, Ftot, HW)
        for h_ in range(2):
            ref_noise[h_, win] += pred.float().view(2, f, HW, 4)[h_]
        ref_cnt[win] += 1
    assert torch.allclose(noise.cpu(), ref_noise, atol=1e-6) and torch.equal(cnt.cpu(), ref_cnt)
    u, c = (ref_noise / ref_cnt[None, :, None, None]).chunk(2)
    v = (u + 3.5 * (c - u))[0]
    t = 749
    ref = sch.step(v, t, lat.float())
    a_t, a_prev = sch.coeffs(t)
    latd = lat.to(dev).clone()
    ops.cfg_ddim_step(latd, noise, cnt, Ftot, HW, 3.5, a_t, a_prev)
    close(latd, ref, rtol=2e-3, atol=2e-3, what="ddim")


# --------------------------------------------------------------------------------------------- benchmark / configs[4] sizes
# Kernel parity AT the sizes the benchmark (configs[1]) and the long-clip configuration (configs[4]: 1024x1024, windows of 30
# frames -> 60-frame UNet batches, 983 040 tokens at level 0) run, where the small cases above cannot reach: 32-bit offset
# arithmetic, grid sizes, ring wrap-arounds.  The fp32 reference is evaluated ON THE GPU by PyTorch (the CPU would need
# minutes); operands are generated on the device from a seeded generator.
def _drnd(dev, *shape, seed=0, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=dev) * scale).half()


def test_temporal_attention_refuses_an_output_that_overlaps_its_inputs(dev):
    """O aliasing Q / K / V is not supported (every pixel's frames are spread over the token matrix; the matrix-core kernel parks O in
    the LDS image of Q): MD_ERR_ARG, while column-sliced siblings of one buffer (q | k | v | o of one allocation) are fine."""
    from mikudance_amd._lib import MdanceHipError
    NB, F_, HW, H, D = 2, 16, 9, 8, 40
    C = H * D
    buf = rnd(NB * F_ * HW, 4 * C, seed=3).to(dev)
    q, k, v, o = buf[:, :C], buf[:, C:2 * C], buf[:, 2 * C:3 * C], buf[:, 3 * C:]
    want = ops.temporal_attention(q, k, v, NB, F_, HW, H, D)
    assert torch.equal(ops.temporal_attention(q, k, v, NB, F_, HW, H, D, out=o), want)
    for bad in (q, k, v):
        with pytest.raises(MdanceHipError, match="overlap"):
            ops.temporal_attention(q, k, v, NB, F_, HW, H, D, out=bad)


def _close_dev(got, ref, rtol=1e-2, atol=1e-3, what=""):
    err = (got.float() - ref).abs().max().item()
    bound = rtol * ref.abs().max().item() + atol
    assert math.isfinite(err) and err <= bound, f"{what}: max err {err:.4g} > {bound:.4g}"
    check_kernel(what, value=float((got.float() - ref).norm() / ref.norm()))     # on the device: these operands are hundreds of MB


@pytest.mark.parametrize("F_,HW,D", [(16, 9216, 40), (16, 2304, 80), (16, 576, 160), (30, 16384, 40), (30, 4096, 80), (30, 1024, 160)])
def test_temporal_attention_at_benchmark_sizes(dev, F_, HW, D):
    """Motion-module attention at the benchmark's (F = 16, 96x96 / 48x48 / 24x24) and configs[4]'s (F = 30, 128x128 / 64x64 /
    32x32) shapes: the matrix-core flavours with their runtime-divisor staging arithmetic, against fp32 SDPA on the same inputs."""
    NB, H = 2, 8
    C = H * D
    q, k, v = (_drnd(dev, NB * F_ * HW, C, seed=s) for s in (53, 54, 55))

    def fold(t):
        return t.float().view(NB, F_, HW, H, D).permute(0, 2, 3, 1, 4)        # b d h f D
    ref = F.scaled_dot_product_attention(fold(q), fold(k), fold(v)).permute(0, 3, 1, 2, 4).reshape(NB * F_ * HW, C)
    out = ops.temporal_attention(q, k, v, NB, F_, HW, H, D)
    _close_dev(out, ref, what=f"temporal F={F_} HW={HW} D={D}")
    rel = float((out.float() - ref).norm() / ref.norm())
    assert rel < 1e-2, rel


def test_streaming_gemms_at_config5_token_count(dev):
    """M = 983 040 tokens (60 frames x 128 x 128): the W-stationary streaming GEMM (K = N = 320, with bias + residual) and its GEGLU
    flavour (K = 320, N = 2560), every element against fp32."""
    M, K = 983040, 320
    a = _drnd(dev, M, K, seed=1)
    w, b, res = _drnd(dev, 320, K, seed=2, scale=K ** -0.5), _drnd(dev, 320, seed=3), _drnd(dev, M, 320, seed=4)
    ref = a.float() @ w.float().t() + b.float() + res.float()
    _close_dev(ops.gemm(a, w, bias=b, residual=res), ref, what="ws gemm M=983040")
    del ref, res
    inner = 1280
    wg, bg = _drnd(dev, 2 * inner, K, seed=5, scale=K ** -0.5), _drnd(dev, 2 * inner, seed=6)
    wp, bp = packing.geglu_weight(wg.cpu(), bg.cpu(), dev)
    out = ops.gemm(a, wp, bias=bp, act=ops.ACT_GEGLU)
    for r0 in range(0, M, 245760):                                            # fp32 reference in four row blocks (10 GB each otherwise)
        hg = a[r0:r0 + 245760].float() @ wg.float().t() + bg.float()
        _close_dev(out[r0:r0 + 245760], hg[:, :inner] * F.gelu(hg[:, inner:]), what=f"ws geglu rows {r0}")
        del hg


def test_wide_k_gemm_at_config5_token_count_runs_in_row_blocks(dev):
    """M = 983 040 tokens x K = 1280 (the level-0 FeedForward output projection of configs[4]): A is 2.5 GB, beyond the 2^31-byte reach of
    gemm_sp_kernel's buffer descriptor, so the launcher cuts it into row blocks (gemm.hip sp_row_blocks); every element against fp32,
    in place on the residual like the transformer block calls it."""
    M, N, K = 983040, 320, 1280
    a = _drnd(dev, M, K, seed=11)
    w, b, res = _drnd(dev, N, K, seed=12, scale=K ** -0.5), _drnd(dev, N, seed=13), _drnd(dev, M, N, seed=14)
    hs = res.clone()
    ops.gemm(a, w, bias=b, residual=hs, out=hs)
    for r0 in range(0, M, 245760):
        ref = a[r0:r0 + 245760].float() @ w.float().t() + b.float() + res[r0:r0 + 245760].float()
        _close_dev(hs[r0:r0 + 245760], ref, what=f"row-blocked gemm rows {r0}")
        del ref


def test_attention_at_config5_sequence_length(dev):
    """Lq = Lk = 16 384 (128 x 128 latents, d = 40): 256 key tiles."""
    B, H, D, L = 1, 8, 40, 16384
    q, k, v = (_drnd(dev, B * L, H * D, seed=s) for s in (70, 71, 72))
    ref = F.scaled_dot_product_attention(*(t.float().view(B, L, H, D).transpose(1, 2) for t in (q, k, v))).transpose(1, 2).reshape(B * L, H * D)
    out = ops.attention(q, k, v.t().contiguous(), B, H, D, L, L)
    _close_dev(out, ref, what="attn L=16384")
    assert float((out.float() - ref).norm() / ref.norm()) < 2e-2


@pytest.mark.parametrize("qscale", [6.0, 9.0])
def test_attention_with_large_logits(dev, qscale):
    """Random-init weights keep the logits small (|s| <~ 5); trained SD-1.5 attention layers reach tens.  With q scaled by 6 / 9 the
    scaled scores of a 9216-key row reach |s| ~ 35 / 55: the d = 40 path's fp16-rounded, pre-scaled Q, its softmax reference folded
    into the QK^T MFMA and P in (0, 2^-4] are exercised where rows are dominated by one or two keys.  No checkpoint is available
    offline, so this is the adversarial stand-in; tolerance as everywhere: 1e-2 of max|ref| + 1e-3."""
    B, H, D, L = 1, 8, 40, 9216
    q, k, v = _drnd(dev, B * L, H * D, seed=80, scale=qscale), _drnd(dev, B * L, H * D, seed=81), _drnd(dev, B * L, H * D, seed=82)
    qh, kh, vh = (t.float().view(B, L, H, D).transpose(1, 2) for t in (q, k, v))
    smax = float((qh[:, :, :512] @ kh.transpose(-1, -2)).abs().max()) * D ** -0.5
    assert smax > 4.5 * qscale, smax
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * L, H * D)
    out = ops.attention(q, k, v.t().contiguous(), B, H, D, L, L)
    _close_dev(out, ref, what=f"attn |s| ~ {smax:.0f}")


def test_groupnorm_at_config5_batch(dev):
    """60 frames x 16 384 pixels x 320 channels (6.3 GB in flight): GroupNorm + SiLU against fp32 F.group_norm on the GPU."""
    B, HW, C = 60, 16384, 320
    x = _drnd(dev, B, HW, C, seed=30) * 2 + 0.5
    g, b = (1 + 0.1 * _drnd(dev, C, seed=31).float()).half(), _drnd(dev, C, seed=32)
    out = ops.groupnorm(x, g, b, 32, 1e-5, True)
    for b0 in range(0, B, 20):
        ref = F.silu(F.group_norm(x[b0:b0 + 20].float().permute(0, 2, 1), 32, g.float(), b.float(), 1e-5)).permute(0, 2, 1)
        _close_dev(out[b0:b0 + 20], ref, what=f"groupnorm frames {b0}")
        del ref


@pytest.mark.parametrize("B,HW,C,inplace", [(2, 589824, 128, False), (1, 16 * 36864, 256, True), (3, 150000, 128, False)])
def test_groupnorm_with_many_slabs_per_image(dev, B, HW, C, inplace):
    """Images of hundreds to thousands of slabs -- the AutoencoderKL at 768 x 768 (576 slabs), the temporal decoder's clip-wide GroupNorm on
    the (clip, frames * h * w, C) view -- take the three-launch form (statistics, ONE reduction of the partial sums, apply) instead of
    re-reducing the partials in every workgroup of the apply sweep (round 5); a ragged last slab and the in-place form included."""
    x = _drnd(dev, B, HW, C, seed=36) * 1.5 + 2.0
    g, b = (1 + 0.1 * _drnd(dev, C, seed=37).float()).half(), _drnd(dev, C, seed=38)
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), 32, g.float(), b.float(), 1e-6)).permute(0, 2, 1)
    out = ops.groupnorm(x, g, b, 32, 1e-6, True, out=x if inplace else None)
    _close_dev(out, ref, what=f"groupnorm {B}x{HW}x{C}")
    if not inplace:
        assert torch.equal(out, ops.groupnorm(x, g, b, 32, 1e-6, True))


def test_groupnorm_in_place(dev):
    """include/mdance_hip.h 'Aliasing': md_groupnorm_nhwc_f16 may run in place (y == x)."""
    B, HW, C = 4, 2304, 640
    x = rnd(B, HW, C, seed=33) * 2 + 3.0                                       # mean far from 0: a wrong pilot shows
    g, b = (1 + 0.1 * rnd(C, seed=34).float()).half(), rnd(C, seed=35)
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), 32, g.float(), b.float(), 1e-5)).permute(0, 2, 1)
    xd = x.to(dev)
    out = ops.groupnorm(xd, g.to(dev), b.to(dev), 32, 1e-5, True, out=xd)
    assert out.data_ptr() == xd.data_ptr()
    close(out, ref, what="groupnorm in place")


def test_conv3x3_rejects_images_beyond_the_24_bit_tap_arithmetic(dev):
    """The conv A gather addresses taps with 24-bit multiplies: images with Hin * Win >= 2^24 pixels (or >= 2^32 elements per
    image) must be refused, not wrapped."""
    from mikudance_amd._lib import MdanceHipError, call
    x = torch.zeros(1, 8, 8, 64, device=dev, dtype=torch.float16)
    w = torch.zeros(64, 9 * 64, device=dev, dtype=torch.float16)
    y = torch.zeros(1, device=dev, dtype=torch.float16)
    with pytest.raises(MdanceHipError, match="2\\^24"):
        call("md_conv3x3_nhwc_f16", x.data_ptr(), w.data_ptr(), y.data_ptr(), 64, 1, 4096, 4096, 64, 64, 1, 0, 0, 0, 0, 0, 0, 0, 0,
             torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------------------------- channel slices, 3 x 1 taps (round 4)
@pytest.mark.parametrize("B,H,W,C,lead,tail,silu", [(2, 8, 8, 64, 64, 0, True), (3, 10, 7, 320, 640, 0, False), (32, 96, 96, 320, 320, 0, True),
                                                   (2, 24, 24, 640, 1280, 64, True), (2, 12, 12, 1280, 0, 1280, False)])
def test_groupnorm_on_a_channel_slice(dev, B, H, W, C, lead, tail, silu):
    """x = wide[..., lead:lead+C]: a skip connection living inside the concat buffer of the up-block resnet that consumes it
    (pixel pitch lead + C + tail); the result must equal GroupNorm of the contiguous copy bit for bit."""
    wide = rnd(B, H, W, lead + C + tail, seed=60).to(dev)
    wide = wide + 3.0 * torch.arange(lead + C + tail, device=dev).remainder(7).half()        # per-channel offsets: group means differ
    x = wide[..., lead:lead + C]
    g, b = (1 + 0.1 * rnd(C, seed=61)).to(dev), rnd(C, seed=62).to(dev)
    ref = F.group_norm(x.float().cpu().permute(0, 3, 1, 2), 32, g.float().cpu(), b.float().cpu(), 1e-5).permute(0, 2, 3, 1)
    ref = F.silu(ref) if silu else ref
    got = ops.groupnorm(x, g, b, 32, 1e-5, silu=silu)
    assert got.is_contiguous() and got.shape == x.shape
    close(got, ref, what="groupnorm slice")
    assert torch.equal(got, ops.groupnorm(x.contiguous(), g, b, 32, 1e-5, silu=silu))
    with pytest.raises(Exception):
        ops.groupnorm(wide[..., 4:4 + C] if lead + tail >= 4 else x[:, :, ::2], g, b, 32, 1e-5)   # unaligned slice / non-uniform pitch


@pytest.mark.parametrize("B,HW,C,silu,lead", [(32, 144, 1280, True, 0), (32, 576, 1280, False, 0), (8, 576, 2560, True, 0), (9, 575, 1920, True, 0),
                                              (16, 144, 1280, True, 1280), (32, 576, 640, False, 64), (8, 1000, 1280, True, 0), (8, 144, 128, True, 0)])
def test_groupnorm_one_sweep_small_images(dev, B, HW, C, silu, lead):
    """gn_small_kernel: one workgroup per (image, group) holds its HW x C/32 values in registers (the 24 x 24 / 12 x 12 levels; chosen when
    C / 32 is a multiple of 4, B * 32 >= 256 and HW <= 1024 fits 24 sweeps): contiguous and channel-slice inputs, ragged HW, in place,
    exact two-pass statistics (|mean| = 800 sigma), bitwise run-to-run identical."""
    wide = rnd(B, HW, lead + C, seed=90).to(dev)
    wide = wide + 3.0 * torch.arange(lead + C, device=dev).remainder(5).half()
    x = wide[..., lead:]
    g, b = (1 + 0.1 * rnd(C, seed=91)).to(dev), rnd(C, seed=92).to(dev)
    ref = F.group_norm(x.float().cpu().permute(0, 2, 1), 32, g.float().cpu(), b.float().cpu(), 1e-5).permute(0, 2, 1)
    ref = F.silu(ref) if silu else ref
    got = ops.groupnorm(x, g, b, 32, 1e-5, silu=silu)
    close(got, ref, what="groupnorm one sweep")
    assert torch.equal(got, ops.groupnorm(x, g, b, 32, 1e-5, silu=silu))
    if lead == 0:
        xc = x.clone()
        ops.groupnorm(xc, g, b, 32, 1e-5, silu=silu, out=xc)                       # in place
        assert torch.equal(xc, got)
    # a group whose mean dwarfs its spread: statistics must not cancel
    big = (800.0 + 1.0 * torch.randn(B, HW, C, generator=torch.Generator().manual_seed(93))).half().to(dev)
    refb = F.group_norm(big.float().cpu().permute(0, 2, 1), 32, None, None, 1e-5).permute(0, 2, 1)
    one, zero = torch.ones(C, device=dev, dtype=torch.float16), torch.zeros(C, device=dev, dtype=torch.float16)
    gotb = ops.groupnorm(big, one, zero, 32, 1e-5).float().cpu()
    assert (gotb - refb).abs().max() <= 2e-2 * refb.abs().max() + 2e-3


def test_instnorm_spade_on_a_channel_slice(dev):
    B, HW, C = 3, 150, 128
    wide = rnd(B, HW, C + 192, seed=63).to(dev)
    gb = rnd(B, HW, 2 * C, seed=64).to(dev)
    x = wide[..., 192:]
    assert torch.equal(ops.instnorm_spade(x, gb), ops.instnorm_spade(x.contiguous(), gb))


@pytest.mark.parametrize("cin,cout,h,w,stride,up", [(64, 64, 8, 8, 1, False), (128, 192, 12, 12, 2, False), (64, 4, 10, 10, 1, False),
                                                    (64, 320, 5, 6, 1, True)])
def test_conv3x3_on_a_channel_slice_small_tile_kernels(dev, cin, cout, h, w, stride, up):
    """The occupancy flavours of gemm_kernel (small problems) with an input pixel pitch > Cin; the persistent kernels are covered by
    tests/gemm_sp_check.py."""
    B = 2
    wide = rnd(B, h, w, cin + 320, seed=70)
    wt, bias = rnd(cout, cin, 3, 3, seed=71, scale=(9 * cin) ** -0.5), rnd(cout, seed=72)
    xin = wide[..., 320:].float().permute(0, 3, 1, 2)
    xin = F.interpolate(xin, scale_factor=2.0, mode="nearest") if up else xin
    ref = F.conv2d(xin, wt.float(), bias.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
    got = ops.conv3x3(wide.to(dev)[..., 320:], packing.conv3x3_weight(wt, dev), cout, bias=bias.to(dev), stride=stride, upsample=up)
    close(got, ref, what="conv slice")


@pytest.mark.parametrize("clips,frames,hw,c,cout", [(2, 3, 20, 64, 64), (1, 4, 33, 128, 192), (2, 1, 16, 64, 64)])
def test_conv3x1_temporal_taps_small_tile_kernels(dev, clips, frames, hw, c, cout):
    """kw = 1: nn.Conv3d(C, Cout, (3,1,1), padding (1,0,0)) over the frames of each clip as ONE implicit GEMM (K = 3 C)."""
    x = rnd(clips, frames, hw, c, seed=80)
    w3, bias = rnd(cout, c, 3, seed=81, scale=(3 * c) ** -0.5), rnd(cout, seed=82)
    ref = F.conv3d(x.float().permute(0, 3, 1, 2)[..., None], w3.float()[..., None, None], bias.float(), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1)
    wpk = w3.permute(0, 2, 1).reshape(cout, 3 * c).contiguous().to(dev)
    close(ops.conv3x3(x.to(dev), wpk, cout, bias=bias.to(dev), kw=1), ref, what="conv3x1")


def test_eta_ddim_kernel(dev):
    """md_cfg_ddim_step_eta against the formula of diffusers DDIMScheduler.step (restated in oracle.DDIM.step)."""
    Ft, HW = 3, 50
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(Ft, HW, 4, generator=g).half()
    ns = torch.randn(2, Ft, HW, 4, generator=g)
    cnt = torch.tensor([1.0, 2.0, 3.0])
    z = torch.randn(Ft, HW, 4, generator=g).half()
    a_t, a_prev, eta, gs = 0.31, 0.47, 0.8, 3.5
    u, c = (ns / cnt.view(1, -1, 1, 1)).unbind(0)
    v = u + gs * (c - u)
    x = lat.float()
    std = eta * ((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)) ** 0.5
    want = a_prev ** 0.5 * (a_t ** 0.5 * x - (1 - a_t) ** 0.5 * v) + (1 - a_prev - std ** 2) ** 0.5 * (a_t ** 0.5 * v + (1 - a_t) ** 0.5 * x) + std * z.float()
    l = lat.to(dev).clone()
    ops.cfg_ddim_step(l, ns.to(dev), cnt.to(dev), Ft, HW, gs, a_t, a_prev, eta=eta, variance_noise=z.to(dev))
    close(l, want, rtol=2e-3, what="ddim eta")
