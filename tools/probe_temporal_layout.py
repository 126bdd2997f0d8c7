"""Debug tool: temporal attention on the frame-major token order of the UNets (rows (b, f, pixel): a pixel's 16 frames lie HW rows apart) against
the same work in pixel-major order (NB = pixels, HW = 1: a pixel's frames are adjacent rows) -- what the access pattern costs."""
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
F_ = 16
for HW, D in [(9216, 40), (2304, 80), (576, 160)]:
    C = 8 * D
    q, k, v = (torch.randn(2 * F_ * HW, C, device=dev).half() for _ in range(3))
    o = torch.empty_like(q)
    a = timeit(lambda: ops.temporal_attention(q, k, v, 2, F_, HW, 8, D, out=o))
    b = timeit(lambda: ops.temporal_attention(q, k, v, 2 * HW, F_, 1, 8, D, out=o))
    gb = 8.0 * 2 * F_ * HW * C / 1e6
    print(f"HW={HW} D={D}: frame-major {a*1e3:7.1f} us {gb/a:7.1f} GB/s   pixel-major {b*1e3:7.1f} us {gb/b:7.1f} GB/s", flush=True)
