"""d = 40 self-attention, token-major vs head-major Q / K (ONE data-layout experiment).
    python tools/probe_attn_layout.py token     # the product layout: Q, K [B*L][8*40], head h at columns 40h..
    python tools/probe_attn_layout.py head      # needs the -DA2_HEADMAJOR build: Q, K [B][8][L][40] (80-byte rows, every fetched line fully used)
Timing only (random operands; the head-major build is an experiment, its results are not checked here)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "token"
dev = torch.device("cuda")
B, H, D, L = 32, 8, 40, 9216
C = H * D
if mode == "head":
    q = torch.randn(B * H * L, D, device=dev).half()
    k = torch.randn(B * H * L, D, device=dev).half()
else:
    q = torch.randn(B * L, C, device=dev).half()
    k = torch.randn(B * L, C, device=dev).half()
vt = torch.randn(C, B * L, device=dev).half()
o = torch.empty((B * L, C), device=dev, dtype=torch.float16)
iters = int(os.environ.get("MD_ITERS", "10"))
for _ in range(int(os.environ.get("MD_WARM", "3"))):
    ops.attention(q, k, vt, B, H, D, L, L, out=o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.attention(q, k, vt, B, H, D, L, L, out=o)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"attention d=40 B={B} L={L} layout={mode}: {ms:.3f} ms  {4.0 * B * H * L * L * D / ms / 1e9:.1f} TFLOP/s")
