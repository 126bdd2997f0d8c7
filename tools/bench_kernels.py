"""Kernel micro-benchmarks at the config-2 shapes (HIP events, 10 iterations after 3 warm-ups).  Debug/tuning tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import ops  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=int(os.environ.get("MD_ITERS", "10")), warm=int(os.environ.get("MD_WARM", "3"))):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*shape, scale=1.0):
    if os.environ.get("MD_BENCH_ZERO"):        # power / clock experiment: all-zero operands toggle (almost) no datapath bits
        return torch.zeros(*shape, device=dev, dtype=torch.float16)
    return (torch.randn(*shape, device=dev) * scale).half()


def main(which):
    out = {}
    if "gemm" in which:
        for M, N, K, geglu in [(294912, 320, 320, False), (294912, 2560, 320, True), (73728, 640, 640, False), (73728, 5120, 640, True),
                               (18432, 1280, 1280, False), (18432, 10240, 1280, True), (294912, 320, 1280, False), (294912, 640, 320, False),
                               (8192, 8192, 8192, False), (73728, 640, 2560, False), (18432, 1280, 5120, False), (8192, 10240, 8192, False), (8192, 10240, 8192, True)]:
            a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
            res = rnd(M, N) if not geglu else None
            o = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.gemm(a, w, bias=b, residual=res, act=ops.ACT_GEGLU if geglu else 0, out=o))
            out[f"gemm {M}x{N}x{K}{' geglu' if geglu else ''}"] = (ms, 2.0 * M * N * K / ms / 1e9)
    if "skinny" in which:      # the HBM-bound short-K projections (W-stationary streaming kernel candidates), with their epilogues
        for M, N, K, res in [(294912, 320, 320, False), (294912, 320, 320, True), (294912, 960, 320, False), (294912, 640, 320, False),
                             (147456, 320, 320, True), (73728, 640, 640, False), (73728, 640, 640, True), (73728, 1920, 640, False),
                             (73728, 1280, 640, False), (294912, 320, 640, True)]:
            a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
            r = rnd(M, N) if res else None
            o = torch.empty((M, N), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.gemm(a, w, bias=b, residual=r, out=o))
            byts = 2.0 * (M * K + M * N * (2 if res else 1))
            out[f"gemm {M}x{N}x{K}{' +res' if res else ''}"] = (ms, 2.0 * M * N * K / ms / 1e9, byts / ms / 1e6)
    if "fused" in which:       # normalisation + consumer GEMM, literal operator pair against the fused entry points (round 5)
        from mikudance_amd import packing
        for M, N, K, ra in [(294912, 960, 320, True), (147456, 320, 320, False), (294912, 320, 320, False)]:
            x, w, g, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(K), rnd(K)
            tab = rnd(32, N) if ra else None
            rpg = M // 32
            o = torch.empty((M, N), device=dev, dtype=torch.float16)
            wf, sc = packing.ln_fold(w, None, g, b)
            t_ln = timeit(lambda: ops.layernorm(x, g, b))
            n = ops.layernorm(x, g, b)
            t_g = timeit(lambda: ops.gemm(n, w, rowadd=tab, rows_per_group=rpg, out=o))
            t_f = timeit(lambda: ops.gemm_ln(x, wf, sc, rowadd=tab, rows_per_group=rpg, out=o))
            tag = f"ln+gemm {M}x{N}x{K}{' +rowadd' if ra else ''}"
            print(f"{tag:40s} layernorm {t_ln:6.3f} + gemm {t_g:6.3f} = {t_ln + t_g:6.3f} ms | fused {t_f:6.3f} ms")
        for B, HW, C in [(32, 9216, 320)]:
            x, w, g, b, bias = rnd(B, HW, C), rnd(C, C, scale=C ** -0.5), rnd(C), rnd(C), rnd(C)
            o = torch.empty((B * HW, C), device=dev, dtype=torch.float16)
            t_n = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-6))
            n = ops.groupnorm(x, g, b, 32, 1e-6)
            t_g = timeit(lambda: ops.gemm(n.view(-1, C), w, bias=bias, out=o))
            t_t = timeit(lambda: ops.groupnorm_table(x, g, b, 32, 1e-6))
            tb = ops.groupnorm_table(x, g, b, 32, 1e-6)
            t_a = timeit(lambda: ops.gemm_affine(x, tb, w, bias=bias, out=o))
            print(f"{'gn+gemm ' + str(B) + 'x' + str(HW) + 'x' + str(C):40s} groupnorm {t_n:6.3f} + gemm {t_g:6.3f} = {t_n + t_g:6.3f} ms | table {t_t:6.3f} + affine gemm {t_a:6.3f} = {t_t + t_a:6.3f} ms")
    if "wsk" in which:         # the W-stationary streaming kernel's flavours one by one (ablation builds: tools/r06_gpu.sh kernab)
        for M, N, K, res, ra, geglu in [(294912, 960, 320, False, False, False), (294912, 960, 320, False, True, False), (294912, 320, 320, False, False, False),
                                        (294912, 320, 320, True, False, False), (73728, 640, 640, False, False, False), (73728, 640, 640, True, False, False),
                                        (73728, 1920, 640, False, True, False), (294912, 2560, 320, False, False, True)]:
            a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
            r = rnd(M, N) if res else None
            tab = rnd(32, N) if ra else None
            o = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.gemm(a, w, bias=b, residual=r, rowadd=tab, rows_per_group=M // 32 if ra else 0, act=ops.ACT_GEGLU if geglu else 0, out=o))
            byts = 2.0 * (M * K + M * (N // 2 if geglu else N) * (2 if res else 1))
            out[f"gemm {M}x{N}x{K}{' +res' if res else ''}{' +rowadd' if ra else ''}{' geglu' if geglu else ''}"] = (ms, 2.0 * M * N * K / ms / 1e9, byts / ms / 1e6)
    if "tgemm" in which:       # transposed-output projections (V^T for the attention kernels)
        for M, N, K in [(294912, 320, 320), (73728, 640, 640), (18432, 1280, 1280), (4608, 1280, 1280), (147456, 320, 320)]:
            a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
            o = torch.empty((N, M), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.gemm(a, w, transpose_out=True, out=o))
            out[f"gemm {M}x{N}x{K} T"] = (ms, 2.0 * M * N * K / ms / 1e9)
    if "small" in which:       # the 12x12 level (M = 4608): fewer workgroups than CU slots
        for M, N, K, geglu in [(4608, 1280, 1280, False), (4608, 10240, 1280, True), (4608, 1280, 5120, False), (4608, 2560, 1280, False)]:
            a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
            res = rnd(M, N) if not geglu else None
            o = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.gemm(a, w, bias=b, residual=res, act=ops.ACT_GEGLU if geglu else 0, out=o))
            out[f"gemm {M}x{N}x{K}{' geglu' if geglu else ''}"] = (ms, 2.0 * M * N * K / ms / 1e9)
        for B, H, Cin, Cout in [(32, 12, 1280, 1280), (32, 12, 2560, 1280)]:
            x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5), rnd(Cout)
            o = torch.empty((B, H, H, Cout), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.conv3x3(x, w, Cout, bias=b, out=o))
            out[f"conv {B}x{H}x{H} {Cin}->{Cout}"] = (ms, 2.0 * B * H * H * Cout * 9 * Cin / ms / 1e9)
    if "shapes" in which:      # every MFMA-bound GEMM / conv shape of the config-2 clip that the tiled kernels compete for (dispatch table)
        for M, N, K, geglu in [(18432, 3840, 1280, False), (18432, 2560, 1280, False), (18432, 1280, 2560, False), (73728, 640, 1920, False),
                               (294912, 320, 960, False), (294912, 320, 640, False), (73728, 1280, 640, False), (73728, 1920, 640, False),
                               (4608, 1280, 1280, False), (4608, 10240, 1280, True), (4608, 1280, 5120, False), (4608, 2560, 1280, False)]:
            a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
            res = rnd(M, N) if not geglu else None
            o = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.gemm(a, w, bias=b, residual=res, act=ops.ACT_GEGLU if geglu else 0, out=o))
            out[f"gemm {M}x{N}x{K}{' geglu' if geglu else ''}"] = (ms, 2.0 * M * N * K / ms / 1e9)
        for B, H, Cin, Cout, st, up in [(32, 96, 960, 320, 1, False), (32, 48, 1920, 640, 1, False), (32, 48, 960, 640, 1, False), (32, 24, 1920, 1280, 1, False),
                                        (32, 24, 640, 1280, 1, False), (32, 48, 320, 640, 1, False), (32, 12, 1280, 1280, 1, False), (32, 12, 2560, 1280, 1, False),
                                        (32, 48, 640, 640, 1, True), (32, 24, 1280, 1280, 1, True), (32, 12, 1280, 1280, 1, True),
                                        (32, 96, 320, 320, 2, False), (32, 48, 640, 640, 2, False), (32, 24, 1280, 1280, 2, False)]:
            x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5), rnd(Cout)
            ms = timeit(lambda: ops.conv3x3(x, w, Cout, bias=b, stride=st, upsample=up))
            ho = H * 2 if up else (H + 1) // st if st == 2 else H
            out[f"conv {B}x{H}x{H} {Cin}->{Cout} s{st} up{int(up)}"] = (ms, 2.0 * B * ho * ho * Cout * 9 * Cin / ms / 1e9)
    if "conv" in which:
        for B, H, Cin, Cout in [(32, 96, 320, 320), (32, 48, 640, 640), (32, 24, 1280, 1280), (32, 24, 2560, 1280), (32, 96, 640, 320),
                                (32, 48, 1280, 640)]:
            x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5), rnd(Cout)
            o = torch.empty((B, H, H, Cout), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.conv3x3(x, w, Cout, bias=b, out=o))
            out[f"conv {B}x{H}x{H} {Cin}->{Cout}"] = (ms, 2.0 * B * H * H * Cout * 9 * Cin / ms / 1e9)
    if "attn" in which:
        for B, L, D in [(8, 9216, 40), (32, 9216, 40), (32, 2304, 80), (32, 576, 160)]:
            C = 8 * D
            q, k, vt = rnd(B * L, C), rnd(B * L, C), rnd(C, B * L)
            o = torch.empty((B * L, C), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.attention(q, k, vt, B, 8, D, L, L, out=o))
            out[f"attn B={B} L={L} D={D}"] = (ms, 4.0 * B * 8 * L * L * D / ms / 1e9)
    if "xattn" in which:       # cross-attention of the UNets: 257 CLIP tokens padded to 264 per frame
        for B, L, D in [(32, 9216, 40), (32, 2304, 80), (32, 576, 160)]:
            C, Lk, ks = 8 * D, 257, 264
            q, k, vt = rnd(B * L, C), rnd(B * ks, C), rnd(C, B * ks)
            o = torch.empty((B * L, C), device=dev, dtype=torch.float16)
            ms = timeit(lambda: ops.attention(q, k, vt, B, 8, D, L, Lk, kv_stride=ks, out=o))
            out[f"xattn B={B} Lq={L} Lk={Lk} D={D}"] = (ms, 4.0 * B * 8 * L * Lk * D / ms / 1e9)
    if "temporal" in which:
        for HW, D in [(9216, 40), (2304, 80), (576, 160), (144, 160)]:
            C, F_ = 8 * D, 16
            q, k, v = rnd(2 * F_ * HW, C), rnd(2 * F_ * HW, C), rnd(2 * F_ * HW, C)
            o = torch.empty_like(q)
            ms = timeit(lambda: ops.temporal_attention(q, k, v, 2, F_, HW, 8, D, out=o))
            out[f"temporal HW={HW} D={D}"] = (ms, 8.0 * 2 * F_ * HW * C / ms / 1e6)      # GB/s
    if "norm" in which:
        for B, HW, C in [(32, 9216, 320), (32, 9216, 640), (32, 9216, 960), (32, 2304, 640), (32, 2304, 1280), (32, 2304, 1920), (32, 576, 1280), (32, 576, 2560),
                         (32, 144, 1280), (16, 9216, 320)]:
            x, g, b = rnd(B, HW, C), rnd(C), rnd(C)
            o = torch.empty_like(x)
            ms = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, True, out=o))
            out[f"groupnorm {B}x{HW}x{C}"] = (ms, 6.0 * B * HW * C / ms / 1e6)
            if C <= 1280:
                x2 = x.view(-1, C)
                ms = timeit(lambda: ops.layernorm(x2, g, b))
                out[f"layernorm {B * HW}x{C}"] = (ms, 4.0 * B * HW * C / ms / 1e6)
    for k, v in out.items():
        ms, rate = v[0], v[1]
        extra = f"  {v[2]:8.1f} GB/s (A + C{' + R' if '+res' in k else ''})" if len(v) > 2 else ""
        print(f"{k:44s} {ms:9.3f} ms  {rate:10.1f} {'GB/s' if k.startswith(('temporal', 'groupnorm', 'layernorm')) else 'TFLOP/s'}{extra}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["gemm", "conv", "attn", "temporal", "norm"])
