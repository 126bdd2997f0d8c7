"""Debug tool: md_attention_fwd time against key tiles (Lk) and q-blocks (Lq) at d = 160 / 80 -- separates the fixed cost of a workgroup
round from the cost per 64-key tile."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mikudance_amd import ops
dev = torch.device("cuda")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, H = 32, 8
for D in (160, 80):
    for Lq, Lk in [(576, 64), (576, 320), (576, 576), (576, 1152), (576, 2304), (512, 576), (640, 576), (128, 576), (128, 2304), (2304, 2304), (2304, 576)]:
        C = H * D
        q = (torch.randn(B * Lq, C, device=dev)).half(); k = torch.randn(B * Lk, C, device=dev).half(); vt = torch.randn(C, B * Lk, device=dev).half()
        o = torch.empty((B * Lq, C), device=dev, dtype=torch.float16)
        ms = timeit(lambda: ops.attention(q, k, vt, B, H, D, Lq, Lk, out=o))
        print(f"D={D} Lq={Lq} Lk={Lk}: {ms*1000:8.1f} us  {4.0*B*H*Lq*Lk*D/ms/1e9:7.1f} TF  tiles={(Lk+63)//64}", flush=True)
