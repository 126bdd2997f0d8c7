"""Diagnostic: slot timeline of the ping-pong GEMM (needs tools/ab/lib_trace.so built with -DPP_TRACE)."""
import ctypes, os, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["MD_GEMM_PP"] = "1"
import shutil
shutil.copy(os.path.join(root, "tools/ab/lib_trace.so"), os.path.join(root, "mikudance_amd/libmdance_hip.so"))
from mikudance_amd import ops, _lib  # noqa
dev = torch.device("cuda")
lib = _lib.load()
for what in sys.argv[1:] or ["gemm", "conv"]:
    if what == "gemm":
        M, N, K = 8192, 10240, 8192
        a = (torch.randn(M, K, device=dev)).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        fn = lambda: ops.gemm(a, w)
    else:
        x = torch.randn(32, 96, 96, 320, device=dev).half(); w = (torch.randn(320, 9 * 320, device=dev) * 0.02).half()
        fn = lambda: ops.conv3x3(x, w, 320)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = np.zeros((2, 12, 6), dtype=np.uint64)
    lib.md_debug_pp_trace(ctypes.c_void_p(buf.ctypes.data))
    t0 = int(buf[0, 0, 0])
    print("==", what, " (cycles relative to group-0 stamp 0 of tile 8)")
    print("stamps: 0 L-start, 1 L-done(frags landed), 2 M-start(after barrier), 3 MFMAs issued, 4 vmcnt wait done")
    for g in range(2):
        for i in range(12):
            r = [int(v) - t0 for v in buf[g, i, :6]]
            print(f"g{g} kt={8+i}: " + " ".join(f"{v:7d}" for v in r), " | ds", r[5] - r[0], " L", r[1] - r[0], " M", r[3] - r[2], " wait", (r[4] - r[3]) if g == 0 else (r[4] - r[1]))
