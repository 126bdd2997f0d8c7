"""TEST INFRASTRUCTURE: a CPU emulation of mikudance_amd.ops (the tensor-level wrappers over the C ABI) in plain PyTorch.

The product has no CPU path and never sees this file.  It exists so that the HOST graph of the UNets -- which operator is called
on which tensor, with which pitch / slice / `out=` destination / row-broadcast table / bank -- can be executed and compared with
the oracle in the build container, where there is no GPU: tests/test_host_graph_cpu.py monkeypatches these functions over the
attributes of mikudance_amd.ops.  Every function follows the argument meaning documented in include/mdance_hip.h: fp16 in,
fp32 arithmetic, ONE rounding to fp16 on the way out, strides and channel slices honoured (an operator that ignored the
pitch of its input or wrote outside its `out` slice fails the comparison).
"""
import torch
import torch.nn.functional as F

F16 = torch.float16
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GEGLU, ACT_QUICKGELU = 0, 1, 2, 3, 4
CALLS = []          # (name, detail) log, inspected by the tests


def _pitch(x, align=8):
    """The checks of mikudance_amd.ops._pixel_pitch without the device requirement."""
    C, ld = x.shape[-1], x.stride(-2)
    ok = x.dtype == F16 and x.stride(-1) == 1 and ld >= C and ld % align == 0 and x.storage_offset() % align == 0
    n = 1
    for d in range(x.dim() - 2, -1, -1):
        ok = ok and (x.shape[d] == 1 or x.stride(d) == ld * n)
        n *= x.shape[d]
    assert ok, (tuple(x.shape), tuple(x.stride()))
    return ld


def gemm(a, w, bias=None, residual=None, rowadd=None, rows_per_group=0, act=ACT_NONE, transpose_out=False, out=None, ldc_t=None):
    assert a.dim() == 2 and a.stride(1) == 1 and a.dtype == F16 and w.dtype == F16 and w.is_contiguous()
    M, K = a.shape
    N = w.shape[0]
    acc = a.float() @ w.float().t()
    if bias is not None:
        acc = acc + bias.float()
    if act == ACT_GEGLU:
        q = acc.view(M, N // 64, 2, 32)
        acc = (q[:, :, 0] * F.gelu(q[:, :, 1])).reshape(M, N // 2)
    elif act == ACT_SILU:
        acc = F.silu(acc)
    elif act == ACT_RELU:
        acc = F.relu(acc)
    elif act == ACT_QUICKGELU:
        acc = acc * torch.sigmoid(1.702 * acc)
    if rowadd is not None:
        idx = torch.arange(M) // rows_per_group
        acc = acc + rowadd.float()[idx]
    if residual is not None:
        acc = acc + residual.float()
    CALLS.append(("gemm", (M, N, K, act, transpose_out, None if out is None else tuple(out.stride()))))
    if transpose_out:
        res = torch.zeros((N, ldc_t or M), dtype=F16) if out is None else out
        res[:, :M] = acc.t().to(F16)
        return res
    if out is None:
        return acc.to(F16)
    assert out.shape == acc.shape and out.stride(1) == 1
    out.copy_(acc.to(F16))
    return out


def conv3x3(x, w, cout, bias=None, residual=None, rowadd=None, rows_per_group=0, act=ACT_NONE, stride=1, upsample=False, out=None, pad_lo=1,
            kw=3):
    assert x.dim() == 4
    _pitch(x)
    if out is not None:
        _pitch(out, align=1)
    B, H, W, Cin = x.shape
    wt = w.float().view(cout, 3, kw, Cin).permute(0, 3, 1, 2)
    xi = x.float().permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    if pad_lo == 0:
        xi = F.pad(xi, (0, 1, 0, 1))
        y = F.conv2d(xi, wt, stride=stride)
    else:
        y = F.conv2d(xi, wt, stride=stride, padding=(1, 1 if kw == 3 else 0))
    y = y.permute(0, 2, 3, 1)
    if bias is not None:
        y = y + bias.float()
    if act == ACT_SILU:
        y = F.silu(y)
    elif act == ACT_RELU:
        y = F.relu(y)
    Ho, Wo = y.shape[1:3]
    if rowadd is not None:
        idx = torch.arange(B * Ho * Wo) // rows_per_group
        y = y + rowadd.float()[idx].view(B, Ho, Wo, cout)
    if residual is not None:
        y = y + residual.float().reshape(B, Ho, Wo, cout)
    CALLS.append(("conv", (tuple(x.shape), tuple(x.stride()), cout, kw, stride, upsample, None if out is None else tuple(out.stride()))))
    if out is None:
        out = torch.empty(tuple(y.shape), dtype=F16)       # like the wrapper: allocate, then validate as an output
        _pitch(out, align=1)
    assert tuple(out.shape) == tuple(y.shape), (out.shape, y.shape)
    out.copy_(y.to(F16))
    return out


def groupnorm(x, gamma, beta, groups, eps, silu=False, out=None):
    B, C = x.shape[0], x.shape[-1]
    _pitch(x)
    xi = x.float().reshape(B, -1, C).permute(0, 2, 1)
    y = F.group_norm(xi, groups, gamma.float(), beta.float(), eps)
    if silu:
        y = F.silu(y)
    y = y.permute(0, 2, 1).reshape(x.shape).to(F16).contiguous()
    CALLS.append(("groupnorm", (tuple(x.shape), tuple(x.stride()))))
    if out is not None:
        out.copy_(y)
        return out
    return y


def layernorm(x, gamma, beta, eps=1e-5, add=None, add_mode=0, add_row_begin=0, rows_per_frame=0, frames=0):
    assert x.dim() == 2 and x.is_contiguous()
    y = F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps).to(F16)
    if not add_mode:
        return y
    y2 = y.clone()
    if add_mode == 1:
        y2[add_row_begin:] = (y[add_row_begin:].float() + add.float()).to(F16)
    else:
        idx = (torch.arange(x.shape[0]) // rows_per_frame) % frames
        y2 = (y.float() + add.float()[idx]).to(F16)
    return y, y2


# The fused-normalisation entry points (md_gemm_ln_f16, md_groupnorm_table_f16 + md_gemm_affine_f16).  The real plans only say yes on
# the long token matrices of the 96 x 96 level (K = 320, >= 32768 rows); FUSED = True makes the emulation say yes everywhere, so that the host-side
# folding (packing.ln_fold, the GEGLU row order of s / c, the row term, the table layout) is executed at CPU-test sizes too.
FUSED = False


def gemm_ln_plan(M, N, K, act=ACT_NONE, rowadd=False, a=None):
    return FUSED and not (act == ACT_GEGLU and rowadd)


def gemm_ln(a, wf, sc, eps=1e-5, rowadd=None, rows_per_group=0, act=ACT_NONE, out=None):
    assert a.dim() == 2 and a.stride(1) == 1 and a.dtype == F16 and wf.dtype == F16 and sc.dtype == torch.float32 and sc.shape == (2, wf.shape[0])
    M, K = a.shape
    N = wf.shape[0]
    x = a.float()
    mu = x.mean(1, keepdim=True)
    rstd = torch.rsqrt(((x - mu) ** 2).mean(1, keepdim=True) + eps)
    acc = rstd * (x @ wf.float().t() - mu * sc[0]) + sc[1]            # the kernel's arithmetic: raw rows, folded weights
    if act == ACT_GEGLU:
        q = acc.view(M, N // 64, 2, 32)
        acc = (q[:, :, 0] * F.gelu(q[:, :, 1])).reshape(M, N // 2)
    if rowadd is not None:
        acc = acc + rowadd.float()[torch.arange(M) // rows_per_group]
    CALLS.append(("gemm_ln", (M, N, K, act)))
    if out is None:
        return acc.to(F16)
    out.copy_(acc.to(F16))
    return out


def gemm_affine_plan(M, N, K, rows_per_image, x=None):
    return FUSED


def groupnorm_table(x, gamma, beta, groups, eps):
    B, C = x.shape[0], x.shape[-1]
    _pitch(x)
    xi = x.float().reshape(B, -1, groups, C // groups)
    mean = xi.mean((1, 3))
    rstd = torch.rsqrt(xi.var((1, 3), unbiased=False) + eps)
    sc = rstd.repeat_interleave(C // groups, 1) * gamma.float()
    sf = beta.float() - mean.repeat_interleave(C // groups, 1) * sc
    CALLS.append(("groupnorm_table", (tuple(x.shape), tuple(x.stride()))))
    return torch.stack([sc, sf], 1).contiguous()


def gemm_affine(x, table, w, bias=None, out=None):
    B, K = x.shape[0], x.shape[-1]
    _pitch(x)
    assert table.shape == (B, 2, K) and table.dtype == torch.float32
    n = (x.float().reshape(B, -1, K) * table[:, :1] + table[:, 1:]).to(F16)          # one rounding, like the in-LDS apply
    acc = n.reshape(-1, K).float() @ w.float().t()
    if bias is not None:
        acc = acc + bias.float()
    CALLS.append(("gemm_affine", (tuple(x.shape), tuple(x.stride()), w.shape[0])))
    if out is None:
        return acc.to(F16)
    out.copy_(acc.to(F16))
    return out


def instnorm_spade(x, gamma_beta, eps=1e-5):
    B, C = x.shape[0], x.shape[-1]
    _pitch(x)
    xi = x.float().reshape(B, -1, C)
    gb = gamma_beta.float().reshape(B, -1, 2 * C)
    mu = xi.mean(1, keepdim=True)
    var = xi.var(1, unbiased=False, keepdim=True)
    n = (xi - mu) * torch.rsqrt(var + eps)
    CALLS.append(("instnorm", (tuple(x.shape), tuple(x.stride()))))
    return (n * (1 + gb[..., :C]) + gb[..., C:]).reshape(x.shape).to(F16).contiguous()


def attention(q, k, vt, B, H, D, Lq, Lk, kv_stride=None, kv_index=None, scale=None, out=None):
    kv_stride = kv_stride if kv_stride is not None else Lk
    scale = scale if scale is not None else D ** -0.5
    o = torch.empty((B * Lq, H * D), dtype=F16) if out is None else out
    for b in range(B):
        kb = int(kv_index[b]) if kv_index is not None else b
        qq = q[b * Lq:(b + 1) * Lq, :H * D].float().view(Lq, H, D).transpose(0, 1)
        kk = k[kb * kv_stride:kb * kv_stride + Lk, :H * D].float().view(Lk, H, D).transpose(0, 1)
        vv = vt[:H * D, kb * kv_stride:kb * kv_stride + Lk].float().view(H, D, Lk).transpose(1, 2)
        p = torch.softmax(qq @ kk.transpose(1, 2) * scale, dim=-1)
        o[b * Lq:(b + 1) * Lq] = (p @ vv).transpose(0, 1).reshape(Lq, H * D).to(F16)
    return o


def softmax_rows_(x, scale=1.0):
    x.copy_(torch.softmax(x.float() * scale, dim=-1).to(F16))
    return x


def temporal_attention(q, k, v, NB, F_, HW, H, D, out=None):
    C = H * D
    sh = lambda t: t[:, :C].float().view(NB, F_, HW, H, D).permute(0, 2, 3, 1, 4)        # (NB, HW, H, F, D)
    p = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) * D ** -0.5, dim=-1)
    o = (p @ sh(v)).permute(0, 3, 1, 2, 4).reshape(NB * F_ * HW, C).to(F16)
    if out is not None:
        out.copy_(o)
        return out
    return o


def _nearest(n_out, n_in):
    return torch.clamp(torch.floor(torch.arange(n_out, dtype=torch.float32) * (float(n_in) / float(n_out))).long(), max=n_in - 1)


def pack_nhwc(src, n, f, strides, c_begin, c_count, cpad, ho, wo, hin=None, win=None):
    hin, win = hin or ho, win or wo
    sB, sF, sC, sY, sX = strides
    flat = src.reshape(-1) if src.is_contiguous() else None
    base = src.as_strided((src.untyped_storage().nbytes() // src.element_size() - src.storage_offset(),), (1,), src.storage_offset()) if flat is None else flat
    N = torch.arange(n)
    off = ((N // f) * sB + (N % f) * sF).view(n, 1, 1, 1) + (_nearest(ho, hin) * sY).view(1, ho, 1, 1) + (_nearest(wo, win) * sX).view(1, 1, wo, 1) \
        + ((c_begin + torch.arange(c_count)) * sC).view(1, 1, 1, c_count)
    dst = torch.zeros((n, ho, wo, cpad), dtype=F16)
    dst[..., :c_count] = base[off].to(F16)
    return dst


def unpack_nhwc(src, dst, n, f, strides, c, ho, wo):
    sB, sF, sC, sY, sX = strides
    N = torch.arange(n)
    off = ((N // f) * sB + (N % f) * sF).view(n, 1, 1, 1) + (torch.arange(ho) * sY).view(1, ho, 1, 1) + (torch.arange(wo) * sX).view(1, 1, wo, 1) \
        + (torch.arange(c) * sC).view(1, 1, 1, c)
    flat = dst.as_strided((dst.untyped_storage().nbytes() // dst.element_size() - dst.storage_offset(),), (1,), dst.storage_offset())
    flat[off.reshape(-1)] = src.reshape(n, ho, wo, -1)[..., :c].reshape(-1).to(dst.dtype)
    return dst


def concat_channels(a, b):
    CALLS.append(("concat", (tuple(a.shape), tuple(b.shape))))
    return torch.cat([a, b], dim=-1).contiguous()


def window_accumulate(pred, noise_sum, counter, window, f, ftot, hw, halves=2):
    p = pred.float().view(halves, f, hw, 4)
    for i, fr in enumerate(window.tolist()):
        if fr < 0:
            continue
        noise_sum[:, fr] += p[:, i]
        counter[fr] += 1


def cfg_ddim_step(latents, noise_sum, counter, ftot, hw, guidance, alpha_t, alpha_prev, halves=2, eta=0.0, variance_noise=None):
    if halves == 2:
        u, c = (noise_sum / counter.view(1, -1, 1, 1)).unbind(0)
        v = u + guidance * (c - u)
    else:
        v = noise_sum[0]
    x = latents.float().view(ftot, hw, 4)
    x0 = alpha_t ** 0.5 * x - (1 - alpha_t) ** 0.5 * v
    ep = alpha_t ** 0.5 * v + (1 - alpha_t) ** 0.5 * x
    std = eta * ((1 - alpha_prev) / (1 - alpha_t) * (1 - alpha_t / alpha_prev)) ** 0.5 if eta else 0.0
    out = alpha_prev ** 0.5 * x0 + max(1 - alpha_prev - std ** 2, 0.0) ** 0.5 * ep
    if eta:
        out = out + std * variance_noise.float().view(ftot, hw, 4)
    latents.copy_(out.view(latents.shape).to(F16))


# ------------------------------------------------------------------ duck-typed modules either side of the loop
class FakeVAE(torch.nn.Module):
    """Duck-typed stand-in for diffusers AutoencoderKL (the pipeline only touches encode().latent_dist.mean, decode().sample,
    .dtype, .device): 8x average pooling to 4 channels and nearest upsampling back."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))

    dtype = property(lambda self: self.p.dtype)
    device = property(lambda self: self.p.device)

    def encode(self, x):
        z = torch.nn.functional.avg_pool2d(x, 8)
        z = torch.cat([z, z.mean(1, keepdim=True)], 1)
        return type("E", (), {"latent_dist": type("D", (), {"mean": z})})

    def decode(self, z, **kw):
        return type("S", (), {"sample": torch.nn.functional.interpolate(z[:, :3], scale_factor=8.0, mode="nearest")})


class FakeCLIP(torch.nn.Module):
    def __init__(self, tokens=5, dim=64):
        super().__init__()
        self.tokens, self.dim = tokens, dim
        self.vision_model = type("V", (), {"post_layernorm": torch.nn.Identity()})()
        self.visual_projection = torch.nn.Identity()
        self.p = torch.nn.Parameter(torch.zeros(1))

    dtype = property(lambda self: self.p.dtype)

    def forward(self, pixel_values):
        g = torch.Generator().manual_seed(3)
        h = torch.randn(1, self.tokens, self.dim, generator=g).to(pixel_values.device, pixel_values.dtype)
        return type("O", (), {"last_hidden_state": h + pixel_values.mean() * 0})


def require_gpu(t, who):
    pass


_NAMES = ("gemm", "gemm_ln_plan", "gemm_ln", "gemm_affine_plan", "groupnorm_table", "gemm_affine", "conv3x3", "groupnorm", "layernorm", "instnorm_spade",
          "attention", "softmax_rows_", "temporal_attention", "pack_nhwc", "unpack_nhwc", "concat_channels", "window_accumulate", "cfg_ddim_step",
          "require_gpu")


def install_process():
    """The same replacement for a whole (spawned worker) process: no monkeypatch fixture there, and nothing to restore when it exits."""
    from mikudance_amd import ops
    for name in _NAMES:
        setattr(ops, name, globals()[name])
    del CALLS[:]


def install(monkeypatch):
    """Replace the functions of mikudance_amd.ops by the emulations above for the duration of a test."""
    from mikudance_amd import ops
    for name in ("gemm", "gemm_ln_plan", "gemm_ln", "gemm_affine_plan", "groupnorm_table", "gemm_affine", "conv3x3", "groupnorm", "layernorm", "instnorm_spade", "attention", "softmax_rows_", "temporal_attention", "pack_nhwc",
                 "unpack_nhwc", "concat_channels", "window_accumulate", "cfg_ddim_step", "require_gpu"):
        monkeypatch.setattr(ops, name, globals()[name])
    del CALLS[:]
