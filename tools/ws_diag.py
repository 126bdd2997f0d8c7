"""Race screen for the W-stationary streaming GEMM: repeated launches on the benchmark shapes, every element checked against
fp32 (count of bad elements, the tiles and columns they sit in).  Debug tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import ops
dev = "cuda"
torch.manual_seed(0)
tot = 0
for M, N, K in [(294912, 320, 320), (73728, 640, 640), (294912, 960, 320)]:
    a = torch.randn(M, K).half().to(dev); w = (torch.randn(N, K) * K ** -0.5).half().to(dev); r = torch.randn(M, N).half().to(dev)
    ref0 = a.float() @ w.float().t()
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        for name, fn, ref in (("plain", lambda: ops.gemm(a, w), ref0), ("res", lambda: ops.gemm(a, w, residual=r), ref0 + r.float())):
            o = fn(); torch.cuda.synchronize()
            err = (o.float() - ref).abs()
            bad = (err > 0.1).nonzero()
            tot += bad.shape[0]
            if bad.shape[0]:
                print(f"{M}x{N}x{K} {name} trial {trial}: nbad {bad.shape[0]} max {float(err.max()):.2f} tiles {(bad[:, 0] // 16).unique()[:8].tolist()} "
                      f"rows%16 {(bad[:, 0] % 16).unique().tolist()} cols {bad[:, 1].unique()[:12].tolist()}")
# GEGLU flavour
import torch.nn.functional as F
from mikudance_amd import packing
M, K, inner = 294912, 320, 1280
a = torch.randn(M, K).half().to(dev); w = (torch.randn(2 * inner, K) * K ** -0.5).half(); b = torch.randn(2 * inner).half()
wp, bp = packing.geglu_weight(w, b, dev)
hg = a.float() @ w.to(dev).float().t() + b.to(dev).float()
ref = hg[:, :inner] * F.gelu(hg[:, inner:])
del hg
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    o = ops.gemm(a, wp, bias=bp, act=ops.ACT_GEGLU); torch.cuda.synchronize()
    err = (o.float() - ref).abs()
    bad = (err > 0.05 + 0.01 * ref.abs()).nonzero()
    tot += bad.shape[0]
    if bad.shape[0]:
        print(f"geglu trial {trial}: nbad {bad.shape[0]} max {float(err.max()):.2f} tiles {(bad[:, 0] // 16).unique()[:8].tolist()} cols {bad[:, 1].unique()[:12].tolist()}")
print("TOTAL BAD", tot)
