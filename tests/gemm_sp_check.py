"""Parity + race screen of the persistent one-wave-per-SIMD GEMM / conv flavour (gemm_sp.h).  Run as a script with MD_GEMM_SP=1 (the dispatch
override is read once per process, so tests/test_gemm_sp_gpu.py spawns this file); every case goes through the C ABI and is
compared with an fp32 PyTorch evaluation of the same fp16-rounded operands: |err| <= 1e-2 * maxabs(ref) + 1e-3.  Each case
is also run three times and must be bit-identical run to run (an LDS ring race shows up as run-to-run differences)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import ops, packing  # noqa: E402

assert os.environ.get("MD_GEMM_SP") == "1", "run with MD_GEMM_SP=1"
dev = torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def check(what, fn, ref, rtol=1e-2, atol=1e-3):
    outs = [fn() for _ in range(3)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), f"{what}: not deterministic run to run"
    got, ref = outs[0].float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    bound = rtol * ref.abs().max().item() + atol
    assert math.isfinite(err) and err <= bound, f"{what}: max err {err:.4g} > {bound:.4g}"
    print(f"ok  {what}: err {err:.3g} (bound {bound:.3g})")


d = lambda t: t.to(dev)

# plain GEMM, N % 320 == 0: K from 4 ring tiles (the ring depth) to 90, ragged M, several column tiles; the last two give the
# persistent kernel 2-3 output tiles per workgroup (57 600 / 192 = 300 tiles, 100 000 / 192 -> 521 tiles, ragged; 105 x 4 tiles in grouped order)
for M, N, K in [(256, 320, 128), (1000, 320, 192), (77, 640, 256), (2048, 320, 320), (700, 1280, 1280), (4096, 640, 2880),
                (57600, 320, 256), (100000, 320, 192), (20000, 1280, 128),
                # N % 256 == 0 only: the 192 x 256 tile (14 DMA pieces per wave and K tile instead of 16)
                (256, 256, 128), (1000, 512, 192), (77, 768, 256), (3000, 1024, 1280), (57600, 256, 256), (30000, 1280, 320),
                # N % 128 == 0 only: the 256 x 128 tile (wave tile 128 x 64; the 128-channel layers of the AutoencoderKL)
                (300, 128, 128), (5000, 384, 256), (70000, 128, 1152)]:
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    check(f"gemm {M}x{N}x{K}", lambda: ops.gemm(d(a), d(w)), a.float() @ w.float().t())

# identity: exact, catches fragment / row / column mix-ups in the 96x160 wave tile
K = 320
a = torch.eye(K).half()
w = (torch.arange(K * K).reshape(K, K) % 97).half() / 16
out = ops.gemm(d(a), d(w))
assert torch.equal(out.cpu().float(), w.float().t()), "identity"
print("ok  identity")

# epilogues
for M, N, K in [(900, 640, 256), (900, 1280, 256), (900, 512, 256)]:
    a, w = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=K ** -0.5)
    bias, res, radd = rnd(N, seed=5), rnd(M, N, seed=6), rnd(9, N, seed=7)
    base = a.float() @ w.float().t() + bias.float()
    check("bias", lambda: ops.gemm(d(a), d(w), bias=d(bias)), base)
    check("silu", lambda: ops.gemm(d(a), d(w), bias=d(bias), act=ops.ACT_SILU), F.silu(base))
    check("relu", lambda: ops.gemm(d(a), d(w), bias=d(bias), act=ops.ACT_RELU), F.relu(base))
    check("residual", lambda: ops.gemm(d(a), d(w), bias=d(bias), residual=d(res)), base + res.float())
    check("rowadd", lambda: ops.gemm(d(a), d(w), bias=d(bias), rowadd=d(radd), rows_per_group=100),
          base + radd.float().repeat_interleave(100, 0))
    check("rowadd, no bias", lambda: ops.gemm(d(a), d(w), rowadd=d(radd), rows_per_group=100),
          a.float() @ w.float().t() + radd.float().repeat_interleave(100, 0))
    check("bias + residual + rowadd", lambda: ops.gemm(d(a), d(w), bias=d(bias), residual=d(res), rowadd=d(radd), rows_per_group=100),
          base + res.float() + radd.float().repeat_interleave(100, 0))
    res_dev = d(res).clone()
    hs = res_dev.clone()
    ops.gemm(d(a), d(w), bias=d(bias), residual=hs, out=hs)                     # in place (blocks.py cross-attention)
    check("residual in place", lambda: hs, base + res.float())
    wide = rnd(M, 2 * K, seed=8)
    check("lda", lambda: ops.gemm(d(wide)[:, K:], d(w)), wide[:, K:].float() @ w.float().t())

# residual through the matrix core (RESM, round 6): K tiles > sub-tiles of the wave tile (K >= 1088 for 192 x 320), so the residual enters the
# accumulators as identity MFMAs inside the K loop and the epilogue is the bias-only one.  Ragged M, several column tiles, 2-3 output tiles
# per workgroup (420 / 1050 tiles), residual + row-broadcast term, in place, a strided residual; then the EXACT screen: A = 0 leaves
# fp16(bias + residual), one rounding, bit for bit -- any row / column / sub-tile mix-up of the injected residual shows up as a mismatch.
from mikudance_amd import _lib  # noqa: E402
_plan = _lib.load().md_gemm_plan
for M, N, K in [(900, 1280, 1280), (20000, 1280, 1088), (1000, 1280, 2560), (50000, 1280, 1152), (4608, 1280, 5120)]:
    a, w = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=K ** -0.5)
    bias, res, radd = rnd(N, seed=63), rnd(M, N, seed=64), rnd(10, N, seed=65)
    rpg = -(-M // 10)
    base = a.float() @ w.float().t() + bias.float()
    if M >= 4608 and os.environ.get("MD_SP_RESM", "1") != "0":      # (the plan query answers for the automatic dispatch: enough tiles)
        assert _plan(M, N, K, 0, 0, 5, 256) // 1000 == 2, "expected the residual-through-the-matrix-core flavour"
    check(f"resm residual {M}x{N}x{K}", lambda: ops.gemm(d(a), d(w), bias=d(bias), residual=d(res)), base + res.float())
    check("resm residual, no bias", lambda: ops.gemm(d(a), d(w), residual=d(res)), a.float() @ w.float().t() + res.float())
    check("resm bias + residual + rowadd", lambda: ops.gemm(d(a), d(w), bias=d(bias), residual=d(res), rowadd=d(radd), rows_per_group=rpg),
          base + res.float() + radd.float().repeat_interleave(rpg, 0)[:M])
    hs = d(res).clone()
    ops.gemm(d(a), d(w), bias=d(bias), residual=hs, out=hs)
    check("resm residual in place", lambda: hs, base + res.float())
    wide_r = rnd(M, N + 64, seed=66)
    check("resm strided residual", lambda: ops.gemm(d(a), d(w), bias=d(bias), residual=d(wide_r)[:, 64:]), base + wide_r[:, 64:].float())
    out0 = ops.gemm(torch.zeros(M, K, dtype=torch.float16, device=dev), d(w), bias=d(bias), residual=d(res))
    assert torch.equal(out0.cpu(), (bias.float() + res.float()).half()), f"resm exact screen {M}x{N}x{K}"
    print(f"ok  resm exact screen {M}x{N}x{K}")

# transposed output (V^T for the attention kernels): the same kernels on swapped operands, bias along the output rows; M % 256 == 0
for M, N, K, with_bias in [(2048, 320, 320, False), (8192, 640, 640, True), (4608, 1280, 1280, False), (512, 1280, 256, True), (73728, 320, 320, True)]:
    a, w = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5)
    bias = rnd(N, seed=33) if with_bias else None
    ref = (a.float() @ w.float().t() + (bias.float() if with_bias else 0)).t()
    check(f"transposed {M}x{N}x{K}{' +bias' if with_bias else ''}",
          lambda: ops.gemm(d(a), d(w), bias=d(bias) if with_bias else None, transpose_out=True), ref)

# GEGLU (256 x 256 tiles; the last two cases give the persistent kernel 2-3 output tiles per workgroup, one of them ragged in M)
for M, K, inner in [(200, 128, 256), (1500, 320, 1280), (300, 1280, 512), (10240, 128, 1024), (20000, 320, 1024)]:
    a = rnd(M, K, seed=11)
    w, b = rnd(2 * inner, K, seed=12, scale=K ** -0.5), rnd(2 * inner, seed=13)
    hg = a.float() @ w.float().t() + b.float()
    wp, bp = packing.geglu_weight(w, b, dev)
    check(f"geglu {M}x{2 * inner}x{K}", lambda: ops.gemm(d(a), wp, bias=bp, act=ops.ACT_GEGLU), hg[:, :inner] * F.gelu(hg[:, inner:]))

# 3x3 convolutions with Cout % 320 == 0: padding, stride 2, folded 2x upsample, fused time embedding + residual
for cin, cout, h, wd, stride, up in [(64, 320, 8, 8, 1, False), (320, 320, 24, 24, 1, False), (128, 640, 13, 11, 2, False),
                                     (64, 256, 9, 7, 1, False), (128, 512, 24, 24, 2, False), (64, 1280, 24, 24, 1, False), (128, 256, 10, 10, 1, True),
                                     (64, 320, 6, 5, 1, True), (640, 320, 16, 16, 1, False), (64, 320, 96, 96, 1, False)]:
    B = 8 if h == 96 else 3      # 96 x 96 x 8 = 384 output tiles: the persistent kernel wraps
    x = rnd(B, cin, h, wd, seed=20)
    wt = rnd(cout, cin, 3, 3, seed=21, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=22)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, wt.float(), bias.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wpk = packing.conv3x3_weight(wt, dev)
    check(f"conv {cin}->{cout} {h}x{wd} s{stride} up{int(up)}", lambda: ops.conv3x3(d(xn), wpk, cout, bias=d(bias), stride=stride, upsample=up), ref)
# Cout % 128 == 0 only (256 x 128 tile), input a CHANNEL SLICE of a wider NHWC tensor (a skip connection inside its concat buffer:
# pixel pitch ldx > Cin), output into a channel slice of another one
for cin, cout, h, wd, stride, up, B in [(128, 128, 64, 64, 1, False, 9), (64, 128, 9, 7, 2, False, 3), (128, 384, 12, 12, 1, True, 3),
                                        (320, 320, 24, 24, 1, False, 3), (128, 640, 48, 48, 1, False, 8)]:
    wide = rnd(B, h, wd, cin + 192, seed=40)
    xs = d(wide)[..., 192:]
    wt = rnd(cout, cin, 3, 3, seed=41, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=42)
    xin = wide[..., 192:].float().permute(0, 3, 1, 2)
    xin = F.interpolate(xin, scale_factor=2.0, mode="nearest") if up else xin
    ref = F.conv2d(xin, wt.float(), bias.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
    wpk = packing.conv3x3_weight(wt, dev)
    check(f"conv slice-in {cin}->{cout} {h}x{wd} s{stride} up{int(up)}", lambda: ops.conv3x3(xs, wpk, cout, bias=d(bias), stride=stride, upsample=up), ref)
    big = torch.full(tuple(ref.shape[:3]) + (cout + 64,), 7.0, dtype=torch.float16, device=dev)
    ops.conv3x3(xs, wpk, cout, bias=d(bias), stride=stride, upsample=up, out=big[..., 64:])
    check(f"conv slice-out {cin}->{cout}", lambda: big[..., 64:], ref)
    assert float((big[..., :64].float() - 7.0).abs().max()) == 0.0, "wrote outside the output slice"

# 3 x 1 filters (kw = 1): nn.Conv3d(C, Cout, (3, 1, 1), padding (1, 0, 0)) on the (clips, frames, h*w, C) view
for clips, frames, hw, c, cout in [(2, 5, 96, 128, 128), (1, 16, 4096, 128, 128), (3, 4, 100, 256, 256), (2, 1, 64, 64, 320), (1, 16, 1024, 512, 512)]:
    x = rnd(clips, frames, hw, c, seed=50)
    w3 = rnd(cout, c, 3, seed=51, scale=(3 * c) ** -0.5)
    bias, res = rnd(cout, seed=52), rnd(clips * frames * hw, cout, seed=53)
    ref = F.conv3d(x.float().permute(0, 3, 1, 2)[..., None], w3.float()[..., None, None], bias.float(), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1)
    wpk = d(w3.permute(0, 2, 1).reshape(cout, 3 * c).contiguous())
    check(f"conv3x1 {clips}x{frames}x{hw} {c}->{cout}", lambda: ops.conv3x3(d(x), wpk, cout, bias=d(bias), kw=1), ref)
    check(f"conv3x1 + residual", lambda: ops.conv3x3(d(x), wpk, cout, bias=d(bias), kw=1, residual=d(res)), ref + res.float().view(ref.shape))

B, c, h, wd = 4, 320, 8, 8
x, wt, bias = rnd(B, h, wd, c, seed=23), rnd(c, c, 3, 3, seed=24, scale=(9 * c) ** -0.5), rnd(c, seed=25)
temb, res = rnd(2, c, seed=26), rnd(B, h, wd, c, seed=27)
ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), bias.float(), padding=1).permute(0, 2, 3, 1)
ref = ref + temb.float().repeat_interleave(2, 0)[:, None, None, :] + res.float()
wpk = packing.conv3x3_weight(wt, dev)
check("conv+res (resm)", lambda: ops.conv3x3(d(x), wpk, c, bias=d(bias), residual=d(res)), ref - temb.float().repeat_interleave(2, 0)[:, None, None, :])
check("conv+temb+res", lambda: ops.conv3x3(d(x), wpk, c, bias=d(bias), residual=d(res), rowadd=d(temb), rows_per_group=2 * h * wd), ref)
# a conv with residual whose 384 (or more) output tiles wrap the persistent grid: the resnet's second conv at the 96 x 96 level, 8 images
B, c, h, wd = 8, 320, 96, 96
x, wt, bias, res = rnd(B, h, wd, c, seed=71), rnd(c, c, 3, 3, seed=72, scale=(9 * c) ** -0.5), rnd(c, seed=73), rnd(B, h, wd, c, seed=74)
ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), bias.float(), padding=1).permute(0, 2, 3, 1) + res.float()
wpk = packing.conv3x3_weight(wt, dev)
check("conv 96x96 + residual (resm, wrapped grid)", lambda: ops.conv3x3(d(x), wpk, c, bias=d(bias), residual=d(res)), ref)
hs = d(res).clone()
ops.conv3x3(d(x), wpk, c, bias=d(bias), residual=hs, out=hs)
check("conv 96x96 + residual in place", lambda: hs, ref)
print("ALL OK")
