"""Diagnostic: per-k-step timeline of gemm_sp_kernel's main loop (needs tools/ab/lib_trace.so = tools/build_ab.sh trace -DSP_TRACE).
Stamps (shader clock, workgroup 0, each wave, its first 48 K tiles): 0 body start, 1..3 after k-steps 1..3, 4 after lgkmcnt(0) +
vmcnt + barrier; the last k-step of a body runs from stamp 4 to the next body's stamp 0 (across an output tile: + the epilogue)."""
import ctypes, os, shutil, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["MD_GEMM_SP"] = "1"
os.environ["MD_GEMM_SP_NT"] = "5"
shutil.copy(os.path.join(root, "tools/ab/lib_%s.so" % os.environ.get("SP_TRACE_LIB", "trace")), os.path.join(root, "mikudance_amd/libmdance_hip.so"))
from mikudance_amd import ops, _lib  # noqa
dev = torch.device("cuda")
lib = _lib.load()
N = 48
for what in sys.argv[1:] or ["conv", "gemm", "gemm_res"]:
    if what == "gemm":
        M, Nn, K = 73728, 640, 2560
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(Nn, K, device=dev) * K ** -0.5).half()
        fn = lambda: ops.gemm(a, w)
    elif what == "gemm_res":
        M, Nn, K = 18432, 2560, 1280
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(Nn, K, device=dev) * K ** -0.5).half()
        r = torch.randn(M, Nn, device=dev).half(); b = torch.randn(Nn, device=dev).half()
        fn = lambda: ops.gemm(a, w, bias=b, residual=r)
    elif what in ("ffout", "n1280", "k640", "n1280_nores"):
        # round 6: the GEMM-tail shapes of VERDICT r05 item 3, with their epilogues (bias + residual): FeedForward's output projection
        # (A streamed once from HBM, 20 K tiles per output tile), the 24 x 24 level's N = K = 1280 projections, the 48 x 48 level's N = K = 640
        M, Nn, K = {"ffout": (294912, 320, 1280), "n1280": (18432, 1280, 1280), "k640": (73728, 640, 640), "n1280_nores": (18432, 1280, 1280)}[what]
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(Nn, K, device=dev) * K ** -0.5).half()
        r = torch.randn(M, Nn, device=dev).half(); b = torch.randn(Nn, device=dev).half()
        fn = (lambda: ops.gemm(a, w, bias=b)) if what.endswith("_nores") else (lambda: ops.gemm(a, w, bias=b, residual=r))
    elif what == "conv_res":      # a resnet's second conv at the 96 x 96 level: bias + shortcut (45 K tiles per output tile)
        x = torch.randn(32, 96, 96, 320, device=dev).half(); w = (torch.randn(320, 9 * 320, device=dev) * 0.02).half()
        r = torch.randn(32, 96, 96, 320, device=dev).half(); b = torch.randn(320, device=dev).half()
        fn = lambda: ops.conv3x3(x, w, 320, bias=b, residual=r)
    else:
        x = torch.randn(32, 96, 96, 320, device=dev).half(); w = (torch.randn(320, 9 * 320, device=dev) * 0.02).half()
        fn = lambda: ops.conv3x3(x, w, 320)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros((4, N, 5), dtype=np.uint64)
    rc = lib.md_debug_sp_trace(ctypes.c_void_p(buf.ctypes.data))
    assert rc == 0
    b = buf.astype(np.int64)
    print("==", what, "(cycles; per wave: step1 step2 step3 wait+barrier step0(+epilogue at a tile end) | K-tile total)")
    for kt in range(N - 1):
        row = []
        for wv in range(4):
            s = b[wv, kt]
            nxt = b[wv, kt + 1, 0]
            row.append("%4d %4d %4d %4d %5d |%5d" % (s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], nxt - s[4], nxt - s[0]))
        print("kt %2d  " % kt + "   ".join(row))
    tot = b[:, 1:N - 1, :]
    d = np.stack([tot[:, :, 1] - tot[:, :, 0], tot[:, :, 2] - tot[:, :, 1], tot[:, :, 3] - tot[:, :, 2], tot[:, :, 4] - tot[:, :, 3]], -1)
    print("median per stage (all waves):", np.median(d.reshape(-1, 4), 0), " median K tile:", np.median(b[:, 2:N, 0] - b[:, 1:N - 1, 0]))
