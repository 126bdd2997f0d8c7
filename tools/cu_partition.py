"""CU-partition microbenchmark (round 6, VERDICT r05 item 1, stage A): can the HBM-class 20 % of the clip (norms, temporal attention)
run on a few CUs of every XCD WHILE the power-limited MFMA-class kernels run on the rest?

Two streams created with hipExtStreamCreateWithCUMask.  The partition is per XCD, not by XCD: stream A owns CU slots [0, nA) of EVERY XCD,
stream B slots [nA, 32) of every XCD (mask bit i = slot i / 8 of XCD i % 8, calibrated by tools/ubench/cu_mask_map.hip).  Every XCD -- and
with it every L2 and every fabric port -- stays in both streams, workgroup b of a grid still lands on XCD b % 8 (the tile orders rely on
it), and no XCD is left without CUs for a queue.  md_set_cu_limit tells the persistent GEMM / conv launcher how many CUs its stream owns.

  (i)   MFMA-class kernels ALONE on nA = 32 / 28 / 24 / 20 / 16 CUs per XCD: throughput against the whole chip.  If the chip is power
        limited (profiles/r03_mfma_power.log) fewer CUs clock higher and keep more than nA / 32 of the rate.
  (ii)  HBM-class kernels ALONE on nB = 32 / 16 / 8 / 4 CUs per XCD: how few CUs still move the bytes.
  (iii) co-run: an MFMA kernel looping on A (nA) and an HBM kernel looping on B (32 - nA) at the same time; each against its own alone time.

Kill rule (VERDICT): (i) below 85 % of the whole-chip rate at 24 of 32 CUs, or a co-run that slows the MFMA kernel by more than 10 %.

    python tools/cu_partition.py > profiles/r06_ab_cu_partition.log        (GPU box, ~1 minute)
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mikudance_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
hip = ctypes.CDLL("libamdhip64.so")
ITERS = int(os.environ.get("MD_ITERS", "30"))


def masked_stream(lo, hi):
    """Stream owning CU slots [lo, hi) of every XCD."""
    words = [0] * 8
    for i in range(256):
        if lo <= i // 8 < hi:
            words[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, (ctypes.c_uint32 * 8)(*words))
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed ({rc})"
    return torch.cuda.ExternalStream(st.value)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).half()


def kernels():
    """name -> (callable, unit work per call, unit, class)."""
    out = {}
    B, H, Cin, Cout = 32, 96, 320, 320
    x, w, b = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5), rnd(Cout)
    o = torch.empty((B, H, H, Cout), device=dev, dtype=torch.float16)
    out["conv 96x96 320->320"] = (lambda: ops.conv3x3(x, w, Cout, bias=b, out=o), 2.0 * B * H * H * Cout * 9 * Cin / 1e12, "TF", "mfma")
    x2, w2, b2 = rnd(32, 48, 48, 640), rnd(640, 9 * 640, scale=(9 * 640) ** -0.5), rnd(640)
    o2 = torch.empty((32, 48, 48, 640), device=dev, dtype=torch.float16)
    out["conv 48x48 640->640"] = (lambda: ops.conv3x3(x2, w2, 640, bias=b2, out=o2), 2.0 * 32 * 48 * 48 * 640 * 9 * 640 / 1e12, "TF", "mfma")
    M, N, K = 73728, 5120, 640
    a3, w3, b3 = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    o3 = torch.empty((M, N // 2), device=dev, dtype=torch.float16)
    out["geglu 73728x5120x640"] = (lambda: ops.gemm(a3, w3, bias=b3, act=ops.ACT_GEGLU, out=o3), 2.0 * M * N * K / 1e12, "TF", "mfma")
    L, D = 9216, 40
    q, k, vt = rnd(32 * L, 320), rnd(32 * L, 320), rnd(320, 32 * L)
    oa = torch.empty((32 * L, 320), device=dev, dtype=torch.float16)
    out["attention d=40 L=9216"] = (lambda: ops.attention(q, k, vt, 32, 8, D, L, L, out=oa), 4.0 * 32 * 8 * L * L * D / 1e12, "TF", "mfma")
    xl, g, be = rnd(294912, 320), rnd(320), rnd(320)
    out["layernorm 294912x320"] = (lambda: ops.layernorm(xl, g, be), 4.0 * 294912 * 320 / 1e9, "GB", "hbm")
    xg = rnd(32, 9216, 320)
    og = torch.empty_like(xg)
    out["groupnorm 32x9216x320"] = (lambda: ops.groupnorm(xg, g, be, 32, 1e-5, True, out=og), 6.0 * 32 * 9216 * 320 / 1e9, "GB", "hbm")
    qt, kt, vtt = rnd(32 * 9216, 320), rnd(32 * 9216, 320), rnd(32 * 9216, 320)
    ot = torch.empty_like(qt)
    out["temporal attention HW=9216"] = (lambda: ops.temporal_attention(qt, kt, vtt, 2, 16, 9216, 8, 40, out=ot), 8.0 * 32 * 9216 * 320 / 1e9, "GB", "hbm")
    return out


def time_on(stream, fn, iters=ITERS, warm=3):
    with torch.cuda.stream(stream):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def corun(sa, fa, na, sb, fb, nb_iters):
    """fa x na on stream sa and fb x nb_iters on stream sb, started together; per-call times of each."""
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        ea0.record(sa)
    with torch.cuda.stream(sb):
        eb0.record(sb)
    # interleave the submissions so that both queues fill from the start
    ia = ib = 0
    while ia < na or ib < nb_iters:
        if ia < na:
            with torch.cuda.stream(sa):
                fa()
            ia += 1
        for _ in range(max(1, nb_iters // max(na, 1))):
            if ib < nb_iters:
                with torch.cuda.stream(sb):
                    fb()
                ib += 1
    with torch.cuda.stream(sa):
        ea1.record(sa)
    with torch.cuda.stream(sb):
        eb1.record(sb)
    torch.cuda.synchronize()
    return ea0.elapsed_time(ea1) / na, eb0.elapsed_time(eb1) / nb_iters, ea0.elapsed_time(ea1), eb0.elapsed_time(eb1)


def main():
    lib = _lib.load()
    pr = torch.cuda.get_device_properties(0)
    print(f"# device {pr.name}, {pr.multi_processor_count} CUs; {ITERS} iterations per figure after 3 warm-ups; partition = CU slots of EVERY XCD")
    ks = kernels()
    plain = torch.cuda.current_stream()
    base = {}
    print("\n== (0) whole chip, ordinary stream")
    for name, (fn, work, unit, cls) in ks.items():
        base[name] = time_on(plain, fn)
        print(f"{name:30s} {base[name]:8.3f} ms  {work / base[name] * 1e3:8.1f} {unit}/s")
    print("\n== (i) MFMA-class kernels alone on nA CU slots of every XCD (md_set_cu_limit(8 nA) for the persistent launchers)")
    alone = {}
    for na in (32, 28, 24, 20, 16):
        st = masked_stream(0, na)
        assert lib.md_set_cu_limit(8 * na) == 0
        for name, (fn, work, unit, cls) in ks.items():
            if cls != "mfma":
                continue
            ms = time_on(st, fn)
            alone[(name, na)] = ms
            print(f"nA={na:2d} ({8 * na:3d} CUs)  {name:30s} {ms:8.3f} ms  {work / ms * 1e3:8.1f} {unit}/s  = {100 * base[name] / ms:5.1f} % of the whole chip ({100 * na / 32:.0f} % of the CUs)")
        lib.md_set_cu_limit(0)
    print("\n== (ii) HBM-class kernels alone on the LAST nB CU slots of every XCD")
    for nb in (32, 16, 8, 4):
        st = masked_stream(32 - nb, 32)
        for name, (fn, work, unit, cls) in ks.items():
            if cls != "hbm":
                continue
            ms = time_on(st, fn)
            alone[(name, -nb)] = ms
            print(f"nB={nb:2d} ({8 * nb:3d} CUs)  {name:30s} {ms:8.3f} ms  {work / ms * 1e3:8.1f} {unit}/s  = {100 * base[name] / ms:5.1f} % of the whole chip")
    print("\n== (iii) co-run: MFMA kernel on slots [0, nA) of every XCD, HBM kernel on slots [nA, 32), both looping at the same time")
    for na in (28, 24):
        sa, sb = masked_stream(0, na), masked_stream(na, 32)
        assert lib.md_set_cu_limit(8 * na) == 0
        for an in ("conv 96x96 320->320", "attention d=40 L=9216", "geglu 73728x5120x640"):
            for bn in ("layernorm 294912x320", "groupnorm 32x9216x320", "temporal attention HW=9216"):
                fa, wa, ua, _ = ks[an]
                fb, wb, ub, _ = ks[bn]
                ta_alone = alone[(an, na)]
                tb_alone = time_on(sb, fb, iters=10)
                n_a = 12
                n_b = max(4, int(n_a * ta_alone / tb_alone))            # both queues busy for about the same time
                ta, tb, wall_a, wall_b = corun(sa, fa, n_a, sb, fb, n_b)
                serial = n_a * base[an] + n_b * base[bn]                  # the same work, one kernel after the other on the whole chip
                print(f"nA={na} | {an:24s} {ta:7.3f} ms ({100 * ta_alone / ta:5.1f} % of alone-on-nA, {100 * base[an] / ta:5.1f} % of whole chip) | "
                      f"{bn:28s} x{n_b:3d} {tb:7.3f} ms ({100 * tb_alone / tb:5.1f} % of alone-on-nB, {100 * base[bn] / tb:5.1f} % of whole chip) | "
                      f"both queues done in {max(wall_a, wall_b):8.2f} ms vs {serial:8.2f} ms serially on the whole chip = {serial / max(wall_a, wall_b):5.3f}x")
        lib.md_set_cu_limit(0)
    print("\n# reading: see profiles/HISTORY.md (round 6)")


if __name__ == "__main__":
    main()
