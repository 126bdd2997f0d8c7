"""FeedForward of the 96 x 96 level in row chunks: GEGLU (K = 320 -> 1280) and the output projection (K = 1280 -> 320, + residual) of chunk i back to back, so that the
hidden tensor of a chunk (755 MB for all 294 912 rows) is read back while it may still sit in the 256-MiB memory-side cache.  Micro-benchmark (HIP events).

MEASURED AND NOT ADOPTED (round 6, profiles/r06_ab_ff_chunks.log): in this loop the chunked form wins 13-17 % (1.032 -> 0.900 ms at C = 320 with 6 chunks, 0.899 -> 0.747 at
C = 640 with 3), bit-identical -- and inside the denoising loop the same chunking LOSES 1.4 % end to end: there the whole-tensor output projection already runs at the rate
the chunked one reaches here (0.304 ms; its operand's tail is still cached and it walks back to front), the chunked one gains nothing (60.85 vs 60.79 ms per clip), and the
six-times-smaller LayerNorm / GEGLU launches pay their ramps and tails (LayerNorm 14.4 -> 25.7 ms, GEGLU 115 -> 127 ms per clip).  A micro-benchmark that repeats ONE
producer / consumer pair overstates what the memory-side cache has to give."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import ops, packing  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
for M, C in ((294912, 320), (73728, 640)):
    inner = 4 * C
    n = torch.randn(M, C, device=dev).half()
    h = torch.randn(M, C, device=dev).half()
    w1 = (torch.randn(2 * inner, C) * C ** -0.5).half()
    b1 = torch.randn(2 * inner).half()
    w1p, b1p = packing.geglu_weight(w1, b1, dev)
    w2 = (torch.randn(C, inner, device=dev) * inner ** -0.5).half()
    b2 = torch.randn(C, device=dev).half()
    hid = torch.empty((M, inner), device=dev, dtype=torch.float16)
    out = torch.empty((M, C), device=dev, dtype=torch.float16)
    ref = None
    for chunks in (1, 2, 3, 4, 6, 8, 12, 16, 24):
        if M % (chunks * 192):
            continue
        rows = M // chunks

        def run():
            for i in range(chunks):
                s = slice(i * rows, (i + 1) * rows)
                ops.gemm(n[s], w1p, bias=b1p, act=ops.ACT_GEGLU, out=hid[s])
                ops.gemm(hid[s], w2, bias=b2, residual=h[s], out=out[s])
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if ref is None:
            ref = out.clone()
        same = torch.equal(out, ref)
        print(f"M={M} C={C} chunks={chunks:3d} rows/chunk={rows:7d} hidden/chunk={rows * inner * 2 / 2**20:6.0f} MiB  {ms:7.3f} ms per FeedForward  bit-identical to 1 chunk: {same}")
