"""State-dict tensors (reference layout, any dtype) -> the fp16 layouts the HIP kernels consume.

  Linear  [N, K]            -> as is (K contiguous)
  Conv1x1 [N, K, 1, 1]      -> [N, K]
  Conv3x3 [N, Cin, 3, 3]    -> [N, (ky, kx, cin_padded)]   cin zero-padded to a multiple of 64
  GEGLU   proj [2*I, K]     -> rows interleaved in blocks of 32: [h0..31, g0..31, h32..63, g32..63, ...] so that a
                               128-wide GEMM tile holds matching (h, g) columns for the fused GEGLU epilogue
"""
import torch

F16 = torch.float16


def pad_to(n, m):
    return (n + m - 1) // m * m


def linear_weight(w, device):
    return w.detach().to(device=device, dtype=F16).contiguous()


def conv1x1_weight(w, device):
    return w.detach().reshape(w.shape[0], w.shape[1]).to(device=device, dtype=F16).contiguous()


def conv3x3_weight(w, device, cin_pad=None):
    n, cin = w.shape[0], w.shape[1]
    cp = cin_pad or pad_to(cin, 64)
    out = torch.zeros((n, 3, 3, cp), dtype=F16, device=device)
    out[..., :cin] = w.detach().permute(0, 2, 3, 1).to(device=device, dtype=F16)
    return out.reshape(n, 9 * cp).contiguous()


def geglu_weight(w, b, device):
    """w: [2I, K], b: [2I]; I % 32 == 0."""
    inner = w.shape[0] // 2
    assert inner % 32 == 0, inner
    h = w[:inner].reshape(inner // 32, 32, -1)
    g = w[inner:].reshape(inner // 32, 32, -1)
    wp = torch.stack([h, g], dim=1).reshape(2 * inner, -1)
    bp = torch.stack([b[:inner].reshape(-1, 32), b[inner:].reshape(-1, 32)], dim=1).reshape(-1)
    return wp.detach().to(device=device, dtype=F16).contiguous(), bp.detach().to(device=device, dtype=F16).contiguous()


def ln_fold(w, bias, gamma, beta):
    """LayerNorm(gamma, beta) folded into the Linear (w [N, K], bias [N] or None) that consumes it, for md_gemm_ln_f16:
        LN(x) @ w.T + bias = rstd * (x @ wf.T - mu * s) + c,   wf = fp16(gamma * w),  s = wf.sum(1),  c = w @ beta + bias.
    w / bias are the kernel-layout fp16 tensors (any row order: the GEGLU interleave commutes with the fold); gamma / beta are
    taken at fp16 precision like every other parameter of the fp16 run.  s is summed from the ROUNDED wf so that
    x @ wf.T - mu * s == (x - mu) @ wf.T exactly.  Returns (wf fp16 [N, K], sc fp32 [2, N] = [s, c])."""
    dev = w.device
    w32 = w.detach().to(F16).float()
    g = gamma.detach().to(device=dev, dtype=F16).float()
    b = beta.detach().to(device=dev, dtype=F16).double()
    wf = (w32 * g[None, :]).to(F16).contiguous()
    s = wf.double().sum(1)
    c = w32.double() @ b
    if bias is not None:
        c = c + bias.detach().to(device=dev, dtype=F16).double()
    return wf, torch.stack([s, c]).float().contiguous()


def vec(t, device):
    return t.detach().to(device=device, dtype=F16).contiguous()
