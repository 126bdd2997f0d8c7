"""GPU: the normalisations that never reach HBM (round 5) -- md_gemm_ln_f16 (LayerNorm folded into its consumer Linear, row statistics
taken from the rows as they stream through LDS) and md_groupnorm_table_f16 + md_gemm_affine_f16 (GroupNorm's apply sweep inside proj_in).
Only the flavours that won their same-box A/B exist (K = 320, plain epilogue: profiles/r05_ab_fused_norms*.log); the others are refused.

Reference semantics: LayerNorm -> Linear of diffusers BasicTransformerBlock / the motion module's TemporalTransformerBlock (reference
src/models/attention.py:131-157,339-365, src/models/motion_module.py:245-272) and GroupNorm -> proj_in of Transformer3DModel / the
temporal transformer (src/models/transformer_3d.py:60-68,121-137, src/models/motion_module.py:121-124,159-170), restated in fp32
PyTorch on the same fp16 inputs.  Tolerance |err| <= 1e-2 max|ref| + 1e-3; the GroupNorm pair must additionally be
BIT-IDENTICAL to md_groupnorm_ld_nhwc_f16 + md_gemm_f16 (same statistics, same rounding points)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from parity_budget import check_kernel  # noqa: E402

from mikudance_amd import ops, packing  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().to(dev)


def close(got, ref, rtol=1e-2, atol=1e-3, what=""):
    err = (got.float() - ref.float()).abs().max().item()
    bound = rtol * ref.float().abs().max().item() + atol
    assert math.isfinite(err) and err <= bound, f"{what}: max err {err:.4g} > {bound:.4g}"
    check_kernel(what, value=float((got.float() - ref.float()).norm() / ref.float().norm()))     # on the device


def _ln_ref(x, g, b, w, bias, eps=1e-5):
    n = F.layer_norm(x.float(), (x.shape[1],), g.float(), b.float(), eps)
    y = n @ w.float().t()
    return y if bias is None else y + bias.float()


def _operands(M, N, K, dev, bias=False):
    x = rnd(M, K, seed=1, dev=dev)
    x[::7] += 3.0                                        # rows with a mean of 3 sigma ...
    x[5::11] *= 0.05                                     # ... and rows 20x quieter than the rest: rstd spans a wide range
    g, b = (1.0 + 0.2 * rnd(K, seed=2, dev=dev).float()).half(), rnd(K, seed=3, scale=0.3, dev=dev)
    w = rnd(N, K, seed=4, scale=K ** -0.5, dev=dev)
    return x, g, b, w, (rnd(N, seed=6, scale=0.2, dev=dev) if bias else None)


# M: the benchmark's token counts (294 912 / 147 456) and ragged ones (a multiple of 16 that is not a multiple of the 256 / 85 row
# streams, and fewer tiles than a stream's ring is deep)
@pytest.mark.parametrize("M,N,rowadd", [(147456, 320, False), (294912, 960, True), (32768 + 16 * 7, 320, False), (32768 + 16 * 5, 960, True),
                                        (32768 + 16 * 300, 640, False), (36864, 320, True)])
def test_layernorm_folded_into_linear(dev, M, N, rowadd):
    K = 320
    assert ops.gemm_ln_plan(M, N, K, ops.ACT_NONE, rowadd)
    x, g, b, w, _ = _operands(M, N, K, dev)
    rows_per_group = 1024
    tab = rnd((M + rows_per_group - 1) // rows_per_group, N, seed=5, dev=dev) if rowadd else None
    wf, sc = packing.ln_fold(w, None, g, b)
    kw = dict(rowadd=tab, rows_per_group=rows_per_group if rowadd else 0)
    got = ops.gemm_ln(x, wf, sc, **kw)
    ref = _ln_ref(x, g, b, w, None)
    if rowadd:
        ref = ref + tab.float()[torch.arange(M, device=dev) // rows_per_group]
    close(got, ref, what=f"gemm_ln {M}x{N}x{K}")
    # against the literal operator pair (md_layernorm_f16 + md_gemm_f16): both within fp16 rounding of the fp32 result
    lit = ops.gemm(ops.layernorm(x, g, b), w, **kw)
    e_f, e_l = (got.float() - ref).abs().mean().item(), (lit.float() - ref).abs().mean().item()
    assert e_f <= 1.5 * e_l + 1e-5, (e_f, e_l)           # the fold skips one rounding (n -> fp16): it is not less accurate than the pair
    assert torch.equal(got, ops.gemm_ln(x, wf, sc, **kw)), "run-to-run difference: a race between the loader waves' statistics and the epilogue"


def test_layernorm_fold_with_a_bias(dev):
    """c[n] = beta . W[n] + bias[n]: the bias of the consuming Linear rides in the folded constant."""
    M, N, K = 65536, 320, 320
    x, g, b, w, bias = _operands(M, N, K, dev, bias=True)
    wf, sc = packing.ln_fold(w, bias, g, b)
    close(ops.gemm_ln(x, wf, sc), _ln_ref(x, g, b, w, bias), what="gemm_ln with bias")


def test_layernorm_fold_is_insensitive_to_the_row_mean(dev):
    """|mean| = 100 sigma on every row: x . Wf^T and mu * s are both ~100x the result and cancel; s is summed from the rounded Wf so
    the cancellation is exact up to fp32 accumulation."""
    M, N, K = 65536, 320, 320
    x = (rnd(M, K, seed=1, dev=dev).float() * 0.25 + 25.0).half()
    g, b = (1.0 + 0.2 * rnd(K, seed=2, dev=dev).float()).half(), rnd(K, seed=3, scale=0.3, dev=dev)
    w = rnd(N, K, seed=4, scale=K ** -0.5, dev=dev)
    wf, sc = packing.ln_fold(w, None, g, b)
    close(ops.gemm_ln(x, wf, sc), _ln_ref(x, g, b, w, None), what="gemm_ln, |mean| = 100 sigma")


# (B, HW, C, N, pitch): the 96 x 96 level, a channel slice of a wider tensor, image counts whose contiguous row streams straddle images
# (256 streams of 9 tiles over images of 64 tiles), and a wide N (two column groups)
@pytest.mark.parametrize("B,HW,C,N,ldx", [(32, 9216, 320, 320, 320), (16, 9216, 320, 320, 960), (36, 1024, 320, 320, 320), (8, 9216, 320, 640, 640)])
def test_groupnorm_inside_proj_in_is_bit_identical_to_the_operator_pair(dev, B, HW, C, N, ldx):
    assert ops.gemm_affine_plan(B * HW, N, C, HW)
    wide = rnd(B, HW, ldx, seed=1, dev=dev)
    wide[:, :, : ldx // 2] += 1.5                                       # non-zero group means
    x = wide[:, :, ldx - C:] if ldx > C else wide
    g, b = (1.0 + 0.2 * rnd(C, seed=2, dev=dev).float()).half(), rnd(C, seed=3, scale=0.3, dev=dev)
    w, bias = rnd(N, C, seed=4, scale=C ** -0.5, dev=dev), rnd(N, seed=5, scale=0.2, dev=dev)
    table = ops.groupnorm_table(x, g, b, 32, 1e-6)
    got = ops.gemm_affine(x, table, w, bias=bias)
    n = ops.groupnorm(x, g, b, 32, 1e-6)
    lit = ops.gemm(n.reshape(-1, C), w, bias=bias)
    # the table against an fp32 restatement of GroupNorm's statistics
    xf = x.float().reshape(B, HW, 32, C // 32)
    mean, var = xf.mean((1, 3)), xf.var((1, 3), unbiased=False)
    sc_ref = (torch.rsqrt(var + 1e-6).repeat_interleave(C // 32, 1) * g.float())
    assert torch.allclose(table[:, 0], sc_ref, rtol=2e-4, atol=1e-6)
    assert torch.allclose(table[:, 1], b.float() - mean.repeat_interleave(C // 32, 1) * sc_ref, rtol=2e-4, atol=2e-4)
    close(got, F.group_norm(x.float().permute(0, 2, 1), 32, g.float(), b.float(), 1e-6).permute(0, 2, 1).reshape(-1, C) @ w.float().t() + bias.float(),
          what="gemm_affine vs fp32")
    assert torch.equal(got, lit), f"not bit-identical: {(got.float() - lit.float()).abs().max().item():.3g} at {int((got != lit).sum())} elements"
    assert torch.equal(got, ops.gemm_affine(x, table, w, bias=bias))

def test_fused_entry_points_refuse_what_they_have_no_kernel_for(dev):
    from mikudance_amd._lib import MdanceHipError
    assert not ops.gemm_ln_plan(4608, 1280, 1280) and not ops.gemm_ln_plan(294912, 2560, 320, ops.ACT_GEGLU)
    assert not ops.gemm_ln_plan(73728, 640, 640) and not ops.gemm_ln_plan(73728, 1920, 640, ops.ACT_NONE, True)      # K = 640: measured, slower
    assert not ops.gemm_affine_plan(18432, 1280, 1280, 576) and not ops.gemm_affine_plan(294912, 320, 320, 9216 + 8)
    assert not ops.gemm_affine_plan(73728, 640, 640, 2304)
    x, w = rnd(4608, 1280, dev=dev), rnd(1280, 1280, dev=dev)
    with pytest.raises(MdanceHipError):
        ops.gemm_ln(x, w, torch.zeros(2, 1280, device=dev))
    with pytest.raises(MdanceHipError):
        ops.gemm_affine(x.view(8, 576, 1280), torch.zeros(8, 2, 1280, device=dev), w)


def test_transformer_blocks_with_and_without_the_fused_normalisations(dev):
    """One spatial transformer and one motion module of the 96 x 96 level (C = 320, 4 frames x 2 halves) with FUSE_NORMS on and off:
    the fused graph stays within fp16 rounding of the literal one."""
    from mikudance_amd import blocks
    from mikudance_amd.synth import synth_state_dict
    torch.manual_seed(0)
    C, f, hw = 320, 4, 96
    st = blocks.SpatialTransformer(C, 768, "3d")
    mm = blocks.MotionModule(C)
    for i, m in enumerate((st, mm)):
        sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=40 + i)
        m.load_state_dict({k: v for k, v in sd.items()}, strict=False)
        m.to(dev).half()
    x = rnd(2 * f, hw, hw, C, seed=7, dev=dev)
    ctx = torch.zeros((2 * 264, 768), device=dev, dtype=torch.float16)
    ctx[264:264 + 257] = rnd(257, 768, seed=8, dev=dev)
    index = torch.tensor([0] * f + [1] * f, device=dev, dtype=torch.int32)
    outs = []
    for fuse in (True, False):
        blocks.FUSE_NORMS = fuse
        try:
            cross = blocks.CrossContext(ctx, index, 257, 264, zero_frames=f)
            y = st(x, cross)
            outs.append((y, mm(y, 2, f)))
        finally:
            blocks.FUSE_NORMS = True
            st._pk = mm._pk = st.transformer_blocks[0]._pk = None
    (a1, a2), (b1, b2) = outs
    close(a1, b1, rtol=5e-3, what="spatial transformer, fused vs literal")
    close(a2, b2, rtol=5e-3, what="motion module, fused vs literal")
