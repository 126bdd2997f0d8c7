import sys, torch
sys.path.insert(0, "/root/repo")
from mikudance_amd import ops
from tools.bench_kernels import timeit, rnd
dev = torch.device("cuda")
M, N = 294912, 320
for K in (64, 320, 640, 1280):
    a, w, b, res = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N), rnd(M, N)
    o = torch.empty((M, N), device=dev, dtype=torch.float16)
    t0 = timeit(lambda: ops.gemm(a, w, out=o))
    t1 = timeit(lambda: ops.gemm(a, w, bias=b, out=o))
    t2 = timeit(lambda: ops.gemm(a, w, bias=b, residual=res, out=o))
    t3 = timeit(lambda: ops.gemm(a, w, residual=res, out=o))
    print(f"K={K}: plain {t0:.3f}  +bias {t1:.3f}  +bias+res {t2:.3f}  +res {t3:.3f} ms")
x = rnd(M, N); y = torch.empty_like(x)
print("copy", timeit(lambda: y.copy_(x)), "add", timeit(lambda: torch.add(x, x, out=y)))
