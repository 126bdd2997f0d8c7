"""Thin tensor-level wrappers over the C ABI (include/mdance_hip.h).  PyTorch supplies device memory and the
stream; all arithmetic happens in the HIP kernels.  Every function requires CUDA(ROCm) fp16 tensors and raises
otherwise -- there is no eager/CPU fallback."""
import torch

from . import _lib

ACT_NONE, ACT_SILU, ACT_RELU, ACT_GEGLU, ACT_QUICKGELU = 0, 1, 2, 3, 4
F16 = torch.float16


def _st():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, name, dtype=F16):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.MdanceHipError(f"{name} must live on the GPU: mikudance_amd has no CPU path")
    if t.dtype != dtype:
        raise _lib.MdanceHipError(f"{name} must be {dtype}, got {t.dtype}")


def require_gpu(t, who):
    """Entry check of the host-side loops: there is no CPU path behind them (the operators below check again, per tensor)."""
    if not t.is_cuda:
        raise RuntimeError(f"{who}: tensors must live on the MI355X; there is no CPU path")


def _rowmajor(t, name):
    _chk(t, name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.MdanceHipError(f"{name} must be a 2-D row-major matrix (stride(1) == 1)")
    return t.stride(0)


def _pixel_pitch(x, name, align=8):
    """x: (B, H, W, C) or (B, HW, C) fp16 NHWC, contiguous or a channel slice of a wider contiguous NHWC tensor (a skip connection
    living inside the concat buffer of the up-block resnet that will consume it).  Returns the pixel pitch in elements.
    align: inputs are fetched in 16-byte pieces (pitch and base multiples of 8 elements); outputs (align = 1) may have any pitch."""
    _chk(x, name)
    C = x.shape[-1]
    ld = x.stride(-2)
    ok = x.stride(-1) == 1 and ld >= C and ld % align == 0 and x.data_ptr() % (2 * align) == 0
    n = 1
    for d in range(x.dim() - 2, -1, -1):                       # every outer dimension walks whole pixels: one uniform pitch
        ok = ok and (x.shape[d] == 1 or x.stride(d) == ld * n)
        n *= x.shape[d]
    if not ok:
        raise _lib.MdanceHipError(f"{name} must be NHWC with one uniform pixel pitch (contiguous, or a channel slice of a contiguous tensor): "
                                  f"shape {tuple(x.shape)} strides {tuple(x.stride())}")
    return ld


def gemm(a, w, bias=None, residual=None, rowadd=None, rows_per_group=0, act=ACT_NONE, transpose_out=False, out=None,
         ldc_t=None):
    """out[M, N] = epi(a[M, K] @ w[N, K]^T).  transpose_out: out is [N, ldc_t] (V^T for attention)."""
    lda = _rowmajor(a, "a")
    _chk(w, "w")
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous(), (w.shape, K)
    if out is None:
        if transpose_out:
            out = torch.empty((N, ldc_t or M), device=a.device, dtype=F16)
        else:
            out = torch.empty((M, N // 2 if act == ACT_GEGLU else N), device=a.device, dtype=F16)
    ldc = _rowmajor(out, "out")
    ldr = _rowmajor(residual, "residual") if residual is not None else 0
    ldra = _rowmajor(rowadd, "rowadd") if rowadd is not None else 0
    _chk(bias, "bias")
    _lib.call("md_gemm_f16", a.data_ptr(), lda, w.data_ptr(), out.data_ptr(), ldc, M, N, K, _p(bias), _p(residual), ldr,
              _p(rowadd), ldra, rows_per_group, act, int(transpose_out), _st(),
              meta=(f"gemm M={M} N={N} K={K}" + (" geglu" if act == ACT_GEGLU else "") + (" T" if transpose_out else ""),
                    2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)))
    return out


def conv3x3(x, w, cout, bias=None, residual=None, rowadd=None, rows_per_group=0, act=ACT_NONE, stride=1, upsample=False,
            out=None, pad_lo=1, kw=3):
    """x: (B, H, W, Cin) NHWC (contiguous or a channel slice: see _pixel_pitch); w: [Cout, 3*kw*Cin] packed (ky, kx, cin);
    returns (B, Ho, Wo, Cout).
    pad_lo=0 (stride 2 only): zero padding (0,1,0,1) instead of 1 all round (the AutoencoderKL downsampler).
    kw=1: a 3 x 1 filter (taps along H only; Conv3d (3,1,1) of the temporal VAE decoder with H = frames, W = pixels).
    `out` may be a channel slice of a wider NHWC tensor (row pitch = its last-dim stride)."""
    _chk(w, "w")
    assert x.dim() == 4
    ldx = _pixel_pitch(x, "x")
    B, H, W, Cin = x.shape
    assert w.shape == (cout, 3 * kw * Cin) and w.is_contiguous(), (w.shape, cout, Cin)
    hup, wup = (H * 2, W * 2) if upsample else (H, W)
    Ho, Wo = (hup + pad_lo - 2) // stride + 1, ((wup + pad_lo - 2) // stride + 1 if kw == 3 else W)
    if out is None:
        out = torch.empty((B, Ho, Wo, cout), device=x.device, dtype=F16)
    assert tuple(out.shape) == (B, Ho, Wo, cout), (tuple(out.shape), (B, Ho, Wo, cout))
    ldy = _pixel_pitch(out, "out", align=1)
    r2 = residual.flatten(0, -2) if residual is not None else None
    ldr = _rowmajor(r2, "residual") if r2 is not None else 0
    ldra = _rowmajor(rowadd, "rowadd") if rowadd is not None else 0
    _chk(bias, "bias")
    _lib.call("md_conv_nhwc_f16", x.data_ptr(), ldx, w.data_ptr(), out.data_ptr(), ldy, B, H, W, Cin, cout, kw, stride,
              int(upsample), int(pad_lo), _p(bias), _p(r2), ldr, _p(rowadd), ldra, rows_per_group, act, _st(),
              meta=(f"conv3x{kw} B={B} {H}x{W} Cin={Cin} Cout={cout} s={stride} up={int(upsample)}",
                    2.0 * B * Ho * Wo * cout * 3 * kw * Cin, 2.0 * (B * H * W * Cin + cout * 3 * kw * Cin + B * Ho * Wo * cout)))
    return out


def _dense16(t):
    """The fused streaming kernels fetch whole 16-byte pieces of every row: base 16-byte aligned, row pitch a multiple of 8 elements."""
    return t is None or (t.data_ptr() % 16 == 0 and t.stride(-2) % 8 == 0 and t.stride(-1) == 1)


def gemm_ln_plan(M, N, K, act=ACT_NONE, rowadd=False, a=None):
    """True when md_gemm_ln_f16 has a kernel for the problem (else: layernorm + gemm on the unfolded weights).  The C-side query answers
    for dense, 16-byte aligned operands; `a` (the token matrix about to be passed) adds the pointer / pitch check, so that a channel- or
    row-sliced operand with an odd offset takes the literal pair instead of failing with MD_ERR_ARG."""
    return _dense16(a) and bool(_lib.load().md_gemm_ln_plan(M, N, K, act, 2 if rowadd else 0))


def gemm_ln(a, wf, sc, eps=1e-5, rowadd=None, rows_per_group=0, act=ACT_NONE, out=None):
    """out[M, N] = epi(LayerNorm(a) @ w^T + bias) from the RAW rows of a, with (wf, sc) = packing.ln_fold(w, bias, gamma, beta)."""
    lda = _rowmajor(a, "a")
    _chk(wf, "wf"); _chk(sc, "sc", torch.float32); _chk(rowadd, "rowadd")
    M, K = a.shape
    N = wf.shape[0]
    assert wf.shape[1] == K and wf.is_contiguous() and sc.shape == (2, N) and sc.is_contiguous(), (wf.shape, sc.shape, K)
    if out is None:
        out = torch.empty((M, N // 2 if act == ACT_GEGLU else N), device=a.device, dtype=F16)
    ldc = _rowmajor(out, "out")
    ldra = _rowmajor(rowadd, "rowadd") if rowadd is not None else 0
    if rowadd is not None and (rows_per_group <= 0 or rowadd.shape[0] * rows_per_group < M or rowadd.shape[1] < N):
        raise _lib.MdanceHipError(f"gemm_ln: row term {tuple(rowadd.shape)} does not cover M={M} rows in groups of {rows_per_group} x N={N}")
    _lib.call("md_gemm_ln_f16", a.data_ptr(), lda, wf.data_ptr(), sc.data_ptr(), out.data_ptr(), ldc, M, N, K, float(eps), _p(rowadd), ldra,
              rows_per_group, act, _st(),
              meta=(f"gemm M={M} N={N} K={K}" + (" geglu" if act == ACT_GEGLU else "") + " ln", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)))
    return out


_gn_ws = {}


def _gn_workspace(x, B, HW, C, groups):
    need = _lib.load().md_groupnorm_workspace_bytes(B, HW, C, groups)
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((max(need, 1 << 20) + 3) // 4, device=x.device, dtype=torch.float32)
        _gn_ws[key] = ws
    return ws


def gemm_affine_plan(M, N, K, rows_per_image, x=None):
    """As gemm_ln_plan: `x` (the NHWC tensor or channel slice about to be passed) adds the alignment check the C-side query cannot make."""
    return _dense16(x) and bool(_lib.load().md_gemm_affine_plan(M, N, K, rows_per_image))


def groupnorm_table(x, gamma, beta, groups, eps):
    """Statistics sweep of GroupNorm only: fp32 (B, 2, C) = [rstd * gamma, beta - mean * rstd * gamma] for gemm_affine."""
    _chk(gamma, "gamma"); _chk(beta, "beta")
    ldx = _pixel_pitch(x, "x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    ws = _gn_workspace(x, B, HW, C, groups)
    table = torch.empty((B, 2, C), device=x.device, dtype=torch.float32)
    _lib.call("md_groupnorm_table_f16", x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), B, HW, C, groups, float(eps), table.data_ptr(),
              ws.data_ptr(), ws.numel() * 4, _st(), meta=(f"groupnorm B={B} HW={HW} C={C} stats", 0.0, 2.0 * B * HW * C))
    return table


def gemm_affine(x, table, w, bias=None, out=None):
    """out[B*HW, N] = fp16(x * scale[image] + shift[image]) @ w^T + bias; x (B, HW, C) / (B, H, W, C) NHWC (a channel slice is fine)."""
    lda = _pixel_pitch(x, "x")
    _chk(w, "w"); _chk(bias, "bias"); _chk(table, "table", torch.float32)
    B, K = x.shape[0], x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous() and table.shape == (B, 2, K) and table.is_contiguous()
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=F16)
    ldc = _rowmajor(out, "out")
    _lib.call("md_gemm_affine_f16", x.data_ptr(), lda, table.data_ptr(), M // B, w.data_ptr(), out.data_ptr(), ldc, M, N, K, _p(bias), _st(),
              meta=(f"gemm M={M} N={N} K={K} gn", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)))
    return out


def groupnorm(x, gamma, beta, groups, eps, silu=False, out=None):
    """x: (B, HW, C) or (B, H, W, C) NHWC, contiguous or a channel slice (see _pixel_pitch); the output is contiguous."""
    _chk(gamma, "gamma"); _chk(beta, "beta")
    ldx = _pixel_pitch(x, "x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    ws = _gn_workspace(x, B, HW, C, groups)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=F16)
    assert out.is_contiguous() and out.shape == x.shape
    _lib.call("md_groupnorm_ld_nhwc_f16", x.data_ptr(), ldx, out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B, HW, C, groups,
              float(eps), int(silu), ws.data_ptr(), ws.numel() * 4, _st(),
              meta=(f"groupnorm B={B} HW={HW} C={C}", 0.0, 6.0 * B * HW * C))
    return out


def layernorm(x, gamma, beta, eps=1e-5, add=None, add_mode=0, add_row_begin=0, rows_per_frame=0, frames=0):
    """x: [M, C].  Returns y (add_mode 0) or (y, y2)."""
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta"); _chk(add, "add")
    assert x.dim() == 2 and x.is_contiguous()
    M, C = x.shape
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if add_mode else None
    _lib.call("md_layernorm_f16", x.data_ptr(), y.data_ptr(), _p(y2), gamma.data_ptr(), beta.data_ptr(), _p(add), M, C,
              float(eps), add_mode, add_row_begin, rows_per_frame, frames, _st(),
              meta=(f"layernorm M={M} C={C} mode={add_mode}", 0.0, 2.0 * M * C * (2 + (2 if add_mode else 0))))
    return (y, y2) if add_mode else y


def instnorm_spade(x, gamma_beta, eps=1e-5):
    """x: (B, HW, C); gamma_beta: (B, HW, 2C)."""
    _chk(gamma_beta, "gamma_beta")
    ldx = _pixel_pitch(x, "x")
    assert gamma_beta.is_contiguous()
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty(x.shape, device=x.device, dtype=F16)
    _lib.call("md_instnorm_spade_ld_f16", x.data_ptr(), ldx, gamma_beta.data_ptr(), y.data_ptr(), B, HW, C, float(eps), _st())
    return y


def attention(q, k, vt, B, H, D, Lq, Lk, kv_stride=None, kv_index=None, scale=None, out=None):
    """q [B*Lq, >=H*D], k [nkv*kv_stride, >=H*D], vt [H*D, >=nkv*kv_stride] (V transposed) -> out [B*Lq, H*D]."""
    ldq = _rowmajor(q, "q"); ldk = _rowmajor(k, "k"); ldvt = _rowmajor(vt, "vt")
    if kv_index is not None:
        _chk(kv_index, "kv_index", torch.int32)
    if out is None:
        out = torch.empty((B * Lq, H * D), device=q.device, dtype=F16)
    ldo = _rowmajor(out, "out")
    _lib.call("md_attention_fwd_f16", q.data_ptr(), ldq, k.data_ptr(), ldk, vt.data_ptr(), ldvt, out.data_ptr(), ldo,
              _p(kv_index), B, H, D, Lq, Lk, kv_stride if kv_stride is not None else Lk,
              float(scale if scale is not None else D ** -0.5), _st(),
              meta=(f"attention B={B} H={H} D={D} Lq={Lq} Lk={Lk}", 4.0 * B * H * Lq * Lk * D, 2.0 * B * H * D * (2 * Lq + 2 * Lk)))
    return out


def softmax_rows_(x, scale=1.0):
    """In-place softmax(scale * x) over the rows of a 2-D row-major fp16 matrix."""
    ld = _rowmajor(x, "x")
    rows, cols = x.shape
    _lib.call("md_softmax_rows_f16", x.data_ptr(), ld, rows, cols, float(scale), _st(),
              meta=(f"softmax_rows {rows}x{cols}", 0.0, 4.0 * rows * cols))
    return x


def temporal_attention(q, k, v, NB, F, HW, H, D, out=None):
    """q/k/v: [(NB*F)*HW, >=H*D] token-major (frame-major rows).  Attention across the F frames of every pixel."""
    ldq = _rowmajor(q, "q"); ldk = _rowmajor(k, "k"); ldv = _rowmajor(v, "v")
    if out is None:
        out = torch.empty((NB * F * HW, H * D), device=q.device, dtype=F16)
    ldo = _rowmajor(out, "out")
    _lib.call("md_temporal_attention_fwd_f16", q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, out.data_ptr(), ldo,
              NB, F, HW, H, D, float(D ** -0.5), _st(),
              meta=(f"temporal_attention NB={NB} F={F} HW={HW} D={D}", 4.0 * NB * HW * H * F * F * D, 8.0 * NB * F * HW * H * D))
    return out


def pack_nhwc(src, n, f, strides, c_begin, c_count, cpad, ho, wo, hin=None, win=None):
    """Strided gather into a fresh (n, ho, wo, cpad) fp16 NHWC tensor.  strides = (sB, sF, sC, sY, sX) in elements.
    (hin, win) != (ho, wo) applies PyTorch's nearest-neighbour resize rule."""
    if not src.is_cuda or src.dtype not in (torch.float16, torch.float32):
        raise _lib.MdanceHipError("pack_nhwc: source must be a CUDA fp16/fp32 tensor")
    dst = torch.empty((n, ho, wo, cpad), device=src.device, dtype=F16)
    sB, sF, sC, sY, sX = strides
    _lib.call("md_pack_nhwc_f16", src.data_ptr(), int(src.dtype == torch.float32), dst.data_ptr(), n, f, sB, sF, sC, sY, sX,
              c_begin, c_count, cpad, ho, wo, hin or ho, win or wo, _st())
    return dst


def unpack_nhwc(src, dst, n, f, strides, c, ho, wo):
    """Scatter NHWC fp16 `src` (n, ho, wo, ldc>=c) into the strided fp16/fp32 tensor `dst`."""
    _chk(src, "src")
    sB, sF, sC, sY, sX = strides
    _lib.call("md_unpack_nhwc_f16", src.data_ptr(), src.shape[-1], dst.data_ptr(), int(dst.dtype == torch.float32), n, f, sB,
              sF, sC, sY, sX, c, ho, wo, _st())
    return dst


def concat_channels(a, b):
    _chk(a, "a"); _chk(b, "b")
    assert a.is_contiguous() and b.is_contiguous() and a.shape[:-1] == b.shape[:-1]
    out = torch.empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), device=a.device, dtype=F16)
    _lib.call("md_concat_channels_f16", a.data_ptr(), a.shape[-1], b.data_ptr(), b.shape[-1], out.data_ptr(),
              a.numel() // a.shape[-1], _st())
    return out


def window_accumulate(pred, noise_sum, counter, window, f, ftot, hw, halves=2):
    _chk(pred, "pred"); _chk(noise_sum, "noise_sum", torch.float32); _chk(counter, "counter", torch.float32)
    _chk(window, "window", torch.int32)
    _lib.call("md_window_accumulate", pred.data_ptr(), noise_sum.data_ptr(), counter.data_ptr(), window.data_ptr(), f, ftot,
              hw, halves, _st())


def cfg_ddim_step(latents, noise_sum, counter, ftot, hw, guidance, alpha_t, alpha_prev, halves=2, eta=0.0, variance_noise=None):
    """eta > 0: `variance_noise` is the caller's N(0, 1) draw, fp16, laid out like `latents` (ftot, hw, 4)."""
    _chk(latents, "latents"); _chk(noise_sum, "noise_sum", torch.float32); _chk(counter, "counter", torch.float32)
    if eta:
        _chk(variance_noise, "variance_noise")
        assert variance_noise is not None and variance_noise.is_contiguous() and variance_noise.numel() == latents.numel()
        _lib.call("md_cfg_ddim_step_eta", latents.data_ptr(), noise_sum.data_ptr(), counter.data_ptr(), variance_noise.data_ptr(), ftot,
                  hw, halves, float(guidance), float(alpha_t), float(alpha_prev), float(eta), _st())
        return
    _lib.call("md_cfg_ddim_step", latents.data_ptr(), noise_sum.data_ptr(), counter.data_ptr(), ftot, hw, halves,
              float(guidance), float(alpha_t), float(alpha_prev), _st())
