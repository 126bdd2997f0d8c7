"""Debug tool: temporal attention at the 30-frame windows of BASELINE configs[4] (128 x 128 latents)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mikudance_amd import ops
dev = torch.device("cuda")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
F_ = 30
for HW, D in [(16384, 40), (4096, 80), (1024, 160), (256, 160)]:
    C = 8 * D
    q, k, v = (torch.randn(2 * F_ * HW, C, device=dev).half() for _ in range(3))
    o = torch.empty_like(q)
    a = timeit(lambda: ops.temporal_attention(q, k, v, 2, F_, HW, 8, D, out=o))
    print(f"temporal F=30 HW={HW} D={D}: {a*1e3:8.1f} us {8.0 * 2 * F_ * HW * C / 1e6 / a:7.1f} GB/s", flush=True)
