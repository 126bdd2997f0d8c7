#!/usr/bin/env python
"""bench.py -- MikuDance denoising loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input: the full denoising loop (20 DDIM steps,
reference_unet + denoising_unet + motion modules + CFG + DDIM) of ONE 768x768x16-frame clip per GPU (BASELINE.json
configs[1]; weak scaling: every added GPU brings its own clip, configs[3]).  Inputs are resident in HBM when the timed
region starts: every rank stages its own clip (--scatter: rank 0 owns the batch and scatters the per-clip conditioning over
RCCL inside the region); the final latents are gathered on rank 0 inside the region.  value = frames of all ranks /
max-over-ranks time.

Extra objects on the JSON line:
  roofline     dominant kernel: algorithmic FLOPs per launch / average launch duration measured live with HIP events on the
               launch stream, against the dense fp16 MFMA peak (2.5 PFLOP/s).  The events bracket every C-ABI launch of ONE extra
               pass of the same clip run right after the timed region (instrumenting ~17k launches per clip inside the timed
               region would add its own launch gaps to `value`; both times are reported)
  cpu_baseline the CPU oracle (a port of the reference's PyTorch path, oracle/cpu_ref.py) timed on this host's cores on a
               bounded sample after a warm-up: configs[0] in full (4 steps) and ONE DDIM step of one frame at 768x768 with the
               full-width UNets, literal reference algorithm, each on the best intra-op thread count of a sweep
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F16 = 2.5e15
PEAK_HBM = 8.0e12
FULL = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768)


_LINE_OUT = None


def flush_c_stdio():
    """Whatever a native library has left in the C-level stdio buffers (RCCL's banner) goes out NOW -- to stderr, where main() has pointed file
    descriptor 1 -- and not at exit: a caller that captures stdout and stderr into ONE stream then still finds the JSON line last."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                        # noqa: BLE001 -- cosmetics must never fail a benchmark
        pass


def emit_line(text):
    """The bench line, on the process's REAL stdout (main() points file descriptor 1 at stderr for everything else)."""
    sys.stdout.flush()
    sys.stderr.flush()
    flush_c_stdio()
    out = _LINE_OUT or sys.stdout
    out.write(text + "\n")
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--guidance", type=float, default=3.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes behind roofline.traffic (quote profiles/pmc_traffic.json)")
    ap.add_argument("--no-vae", action="store_true", help="skip the AutoencoderKL timing behind the extra keys vae_ms_per_clip / e2e_frames_per_s")
    ap.add_argument("--no-reuse", action="store_true", help="literal reference algorithm: reference UNet at every step on 2f frames")
    ap.add_argument("--no-share", action="store_true", help="A/B: evaluate conv_in + the first resnet of the denoising UNet on both CFG halves (literal)")
    ap.add_argument("--two-queues", action="store_true", help="A/B: the CFG halves of the denoising UNet as two queues of B = f kernels instead of ONE queue of "
                    "B = 2f kernels (the default; +0.2-0.35 percent, profiles/r06_ab_two_queue_halves.log)")
    ap.add_argument("--small", action="store_true", help="reduced-width UNets (debug only; NOT the benchmark)")
    ap.add_argument("--scatter", action="store_true", help="N > 1: rank 0 owns the batch and scatters the per-clip conditioning inside "
                    "the timed region (default: every rank stages its own clip before it)")
    ap.add_argument("--window-parallel", action="store_true", help="N > 1: ONE clip, the context windows of every DDIM step shared out over the "
                    "ranks (mikudance_amd.dp.WindowParallel: one all_reduce of the per-frame accumulators per step); strong scaling, meant for "
                    "--config 4 (3 windows of 30 frames).  Default: one clip per rank (weak scaling, configs[3])")
    ap.add_argument("--dry-run-cpu", action="store_true", help="plumbing test only (tests/test_bench_contract_cpu.py): CPU + gloo, the "
                    "kernels replaced by a stand-in; the line says so and carries no roofline")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 4],
                    help="BASELINE.json configs[i]: 1 = 768x768x16f/20 steps (the headline, default); 2 = the same shapes with full "
                         "guidance (scene-motion flow from real camera tracks through camera_to_scene_motion + non-zero face/hand "
                         "channels); 4 = 1024x1024, 48 frames, 30 steps (3 wrapping windows of 30 frames)")
    args = ap.parse_args()
    if args.config == 4:
        args.size, args.frames, args.ddim_steps = 1024, 48, 30

    # stdout carries ONE JSON line and nothing else.  RCCL prints a version banner to the C-level stdout when its first communicator comes up
    # ("RCCL version : ...", five lines, flushed at process exit -- i.e. AFTER the line; seen on MI355X in profiles/r06_rccl_one_rank.json's run),
    # and any other library may do the same: keep a private handle on the real stdout for the line and point file descriptor 1 at stderr.
    global _LINE_OUT
    sys.stdout.flush()
    _LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if os.environ.get("MD_BENCH_TEST_BANNER"):               # the contract test plays RCCL: a C-level printf, left in the stdio buffer until exit
        import ctypes
        ctypes.CDLL(None).printf(b"RCCL version : stand-in banner of tests/test_bench_contract_cpu.py\n")

    from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline, _lib, dp
    from mikudance_amd.selftest import SCHED_KWARGS, build_models
    from mikudance_amd.synth import synth_inputs

    rank, world = dp.init()
    try:
        _main_rank(args, rank, world, dp, _lib, DDIMScheduler, MikuDanceVideoPipeline, SCHED_KWARGS, build_models, synth_inputs)
    finally:
        dp.shutdown()                                        # every rank, also on an exception: no rank is left inside a collective


def _main_rank(args, rank, world, dp, _lib, DDIMScheduler, MikuDanceVideoPipeline, SCHED_KWARGS, build_models, synth_inputs):
    if world > 1:
        # N ranks share the host: cap the intra-op pool (weight synthesis, packing) so that 8 ranks do not spin 8 x 128 threads -- and
        # count the cores the container may actually use (the GPU boxes of this pool: a 16-CPU cgroup quota on a 128-core host)
        quota = cpu_quota()[1]
        usable = min(os.cpu_count() or world, int(quota + 0.5)) if quota else (os.cpu_count() or world)
        torch.set_num_threads(max(1, usable // world))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dry = args.dry_run_cpu
    if dry:
        assert args.small, "--dry-run-cpu is the plumbing test (reduced width, no compute): never a measurement"
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the product has no CPU path)"
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        torch.cuda.set_device(dev)
        _lib.load()
    sync = (lambda: None) if dry else torch.cuda.synchronize

    geom = None if args.small else FULL
    ctx = (5, 64) if args.small else (257, 768)
    t0 = time.time()
    want_cpu = world == 1 and not args.no_cpu_baseline and not dry
    ref, den, ref_sd, den_sd = build_models(geom=geom, device=dev, keep_state_dicts=want_cpu)
    pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
    pipe.reference_reuse = not args.no_reuse
    pipe.share_first_layers = not args.no_share
    pipe.two_queues = bool(args.two_queues)
    h = w = args.size // 8
    setup_s = time.time() - t0
    # the plumbing test (CPU, gloo) replaces the kernels by a stand-in: launch, collectives, timing protocol and the JSON line
    denoise = (lambda lat, rl, emb, steps, g: lat * 0.5) if dry else pipe.denoise

    def make_clip(seed):
        lat, rl, emb = synth_inputs(args.frames, h, w, ctx_len=ctx[0], ctx_dim=ctx[1], seed=seed)
        if args.config == 2:
            rl = full_guidance(rl, args.frames, h, w)
        return lat.half(), rl.half(), emb.half()

    # Inputs resident in HBM before the timed region.  Default: every rank stages ITS OWN clip (clip i = seed 100 + i lives on
    # rank i: the conditioning of a clip is produced by the VAE / CLIP of the rank that denoises it), so the timed region holds
    # the denoising loop and ONE gather of the final latents.  --scatter: rank 0 owns the whole batch (BASELINE configs[3] as a
    # service front-end would see it) and the timed region adds ONE scatter of the per-clip conditioning (~8.5 MB per clip).
    wp = None
    if args.window_parallel:
        assert not args.scatter, "--window-parallel replicates ONE clip on every rank: nothing to scatter"
        wp = dp.WindowParallel()
    if args.scatter:
        staged = [tuple(t.to(dev) for t in make_clip(100 + r)) for r in range(world)] if rank == 0 else None
    else:
        staged = tuple(t.to(dev) for t in make_clip(100 + (0 if wp else rank)))     # window-parallel: the SAME clip on every rank

    def run(clips):
        lat, rl, emb = dp.scatter_clips(clips, dev) if args.scatter else clips
        if wp is not None and not dry:
            return [denoise(lat, rl, emb, args.ddim_steps, args.guidance, window_parallel=wp)]     # every rank holds the full result
        out = denoise(lat, rl, emb, args.ddim_steps, args.guidance)
        return dp.gather_latents(out)

    for _ in range(args.warmup):
        res = run(staged)
    sync()
    dp.barrier()
    sync()
    flush_c_stdio()                                                  # every rank: the communicator exists by now, its banner is in the buffer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = run(staged)
    sync()
    own_elapsed = time.perf_counter() - t0                           # this rank's own time, before it waits for the others
    dp.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(elapsed, dev)
    n_ranks_seen = int(round(dp.sum_over_ranks(1.0, dev)))          # every rank of the job reached the end of the timed region

    # Diagnostics of a multi-rank run, all OUTSIDE the timed region (the first real 8-GPU run must be readable without a second lease):
    # who computed where (device UUID / PCI bus id per rank: N distinct GPUs, not LOCAL_RANK % device_count aliasing), each rank's own
    # loop time, the collective library, and the scatter / gather of one batch timed on their own.
    ident = dict(dp.device_identity(None if dry else dev), own_elapsed_s=own_elapsed, own_ms_per_step=own_elapsed / args.steps * 1e3)
    comm_ms = {}
    if world > 1 or dp.active():                                     # dp.active(): a forced one-rank group (MD_DIST_FORCE=1) times them too
        for name, fn in (("gather_latents", lambda: dp.gather_latents(torch.zeros((1, 4, args.frames, h, w), device=dev, dtype=torch.float16))),
                         ("scatter_clips", (lambda: dp.scatter_clips(staged, dev)) if args.scatter else None),
                         ("window_all_reduce", (lambda: wp.reduce(torch.zeros((2, args.frames, h * w, 4), device=dev), torch.zeros((args.frames,), device=dev)))
                          if wp is not None else None)):
            if fn is None:
                continue
            fn()                                                     # first call: connection set-up
            sync(); dp.barrier(); sync()
            t1 = time.perf_counter()
            fn()
            sync()
            comm_ms[name] = dp.max_over_ranks((time.perf_counter() - t1) * 1e3, dev)
    per_rank = dp.gather_objects(ident)

    # Per-launch HIP-event instrumentation (roofline, kernel families, executed FLOPs) runs on ONE extra pass of the same
    # clip right after the timed region, on rank 0 only and without collectives: two event records per launch x ~17k launches
    # per clip add launch gaps that would otherwise be charged to `value` (see "instrumented_ms_per_step" next to "ms_per_step").
    inst_steps = 1
    inst_elapsed = 0.0
    if rank == 0 and not dry:
        clip0 = staged[0] if args.scatter else staged
        # the two clip-halves run as two kernel queues (pipe.two_queues): for THIS pass queue 1 waits for queue 0 -- the same launches, one at a
        # time -- so that a launch's HIP events time that launch alone (side by side they would time whatever the other queue squeezed in)
        den.serialize_queues = True
        _lib.PROFILER.start()
        sync()
        t1 = time.perf_counter()
        for _ in range(inst_steps):
            pipe.denoise(*clip0, args.ddim_steps, args.guidance)
        sync()
        inst_elapsed = time.perf_counter() - t1
        _lib.PROFILER.stop()
        den.serialize_queues = False
    dp.barrier()

    if rank != 0:
        return
    assert len(res) == (1 if wp else world) and all(torch.isfinite(r.float()).all() for r in res), "missing or non-finite latents"
    gpus = [(r.get("host"), r.get("uuid") or r.get("pci_bus_id") or r.get("device_index")) for r in per_rank]
    multi = {"per_rank": per_rank, "n_distinct_gpus": len(set(gpus)) if not dry else None, "gpu_aliasing": (len(set(gpus)) < world) if not dry else None,
             "collectives": dp.collective_library(), "comm_ms_outside_timed_region": comm_ms,
             "mode": "window-parallel (one clip, windows of a step over ranks, all_reduce per step)" if wp else "clip data-parallel (one clip per rank)"}
    if dry:
        # plumbing line of the CPU test: same launch / collective / timing protocol, no kernels -> no throughput claim
        emit_line(json.dumps({"metric": f"frames/sec ({args.size}x{args.size}, {args.frames}f, {args.ddim_steps} DDIM steps)", "value": None,
                          "unit": "frames/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f16", "data": "dry-run (CPU + gloo plumbing test: kernels replaced by a stand-in, NOT a measurement)",
                          "config": {"workload": "dry-run", "parallelism": f"dp{world}", "input_staging": "scatter" if args.scatter else "rank-local"},
                          "clips_gathered": len(res), "clip_means": [float(r.float().mean()) for r in res], "multi_gpu": multi}))
        return
    prof = _lib.PROFILER.summary()
    total_flops = sum(d["flops"] for d in prof.values()) / inst_steps
    kernel_ms = sum(d["ms"] for d in prof.values()) / inst_steps

    def family(label):
        return label.split(" ")[0] + (" D=" + label.split("D=")[1].split(" ")[0] if label.startswith("attention") else "")
    fam = {}
    for label, d in prof.items():
        f = fam.setdefault(family(label), dict(ms=0.0, flops=0.0, count=0, bytes=0.0))
        for k in ("ms", "flops", "count", "bytes"):
            f[k] += d[k]
    dom_label, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
    dom_flops_per_launch = dom["flops"] / dom["count"]
    dom_ms = dom["ms"] / dom["count"]
    if dom_flops_per_launch > 0:
        achieved = dom_flops_per_launch / (dom_ms * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel=dom_label, achieved=achieved, peak=PEAK_MFMA_F16 / 1e12, unit="TFLOP/s",
                        frac=achieved / (PEAK_MFMA_F16 / 1e12), launches=dom["count"], avg_ms=dom_ms, traffic=None)
    else:
        achieved = dom["bytes"] / dom["count"] / (dom_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel=dom_label, achieved=achieved, peak=PEAK_HBM / 1e9, unit="GB/s",
                        frac=achieved / (PEAK_HBM / 1e9), launches=dom["count"], avg_ms=dom_ms, traffic=None)
    # roofline.traffic: HBM-side bytes per launch of the dominant kernel.  PMC counters cannot be read inside a timed region, so they are
    # collected right here, after it, by two rocprofv3 --pmc passes (kernel-trace only, FETCH_SIZE and WRITE_SIZE in separate passes as
    # the MI355X guide prescribes) over a few launches of the SAME kernel and shape in a child process on this GPU; when that is not
    # possible (rocprofv3 missing, multi-rank run, --no-pmc) the builder's record in profiles/pmc_traffic.json is quoted, and the line
    # says which of the two it carries.
    live = None if (world > 1 or args.no_pmc or args.small) else measure_traffic(dom_label, dev.index or 0)
    if live is not None:
        roofline.update(traffic=live["hbm_bytes_corrected"], algorithmic_bytes=live["algorithmic_bytes"], traffic_source=live["source"])
    else:
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        rec = json.load(open(pmc)).get(dom_label) if os.path.exists(pmc) else None
        roofline["traffic"] = rec["hbm_bytes_corrected"] if rec else None   # HBM-side bytes per launch (profiles/pmc_traffic.json)
        if rec:
            roofline["algorithmic_bytes"] = rec["algorithmic_bytes"]
            roofline["traffic_source"] = {"file": "profiles/pmc_traffic.json", "collected": rec.get("collected", "round 1"),
                                          "method": rec.get("method", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                                                      "FETCH_SIZE x2 (gfx950 correction, MI355X guide HBM section)")}

    peak_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    frames_total = args.frames * args.steps * (1 if wp else world)   # window-parallel: ONE clip, whatever the rank count (strong scaling)
    value = frames_total / elapsed
    line = {
        "metric": f"frames/sec ({args.size}x{args.size}, {args.frames}f, {args.ddim_steps} DDIM steps)", "value": value, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if wp else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic", "n_ranks_seen": n_ranks_seen,
        "config": {"workload": f"configs[{args.config}]: {args.size}x{args.size}, {args.frames}-frame clip, {args.ddim_steps} DDIM steps, fp16, "
                               "reference_unet + denoising_unet + motion_module, CFG 3.5, one clip per GPU per step"
                               + (", full guidance: scene-motion flow from the demo camera tracks (tests/golden/g2) through "
                                  "camera_to_scene_motion + non-zero face/hand latents" if args.config == 2 else "")
                               + (", context 30 / overlap 8 -> 3 wrapping windows, 60-frame UNet batches" if args.config == 4 else ""),
                   "parallelism": (f"wp{world}" if wp else f"dp{world}"), "input_staging": "scatter from rank 0 inside the timed region" if args.scatter
                   else "rank-local (every rank stages its own clip in HBM before the timed region)", "reference_reuse": pipe.reference_reuse,
                   "queues": "2 (unconditional | conditional half of the denoising UNet side by side)" if pipe.two_queues and pipe.share_first_layers and args.guidance > 1 else "1",
                   "weights": "random-init SD-1.5 geometry "
                   "(N(0,1/fan_in), seeds 1234/4321)", "width": "reduced(debug)" if args.small else "full"},
        "executed_tflop_per_clip": total_flops / 1e12, "mfma_frac_whole_loop": total_flops / (elapsed / args.steps) / PEAK_MFMA_F16,
        "peak_hbm_gb": peak_gb, "kernel_ms_per_clip": kernel_ms, "instrumented_ms_per_step": inst_elapsed / inst_steps * 1e3, "setup_s": setup_s,
        # tflops for the contraction kernels; gbps = algorithmic bytes (inputs + outputs once) / time for every family: it is THE
        # figure for the HBM-class ones (norms, temporal attention whose f x f core is 4 bytes per 2 f MACs, the skinny GEMMs)
        "kernel_families": {k: dict(ms_per_clip=v["ms"] / inst_steps,
                                    tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] and k != "temporal_attention" else None,
                                    gbps=v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                                    launches=v["count"] // inst_steps) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])},
        "roofline": roofline,
        "multi_gpu": multi,
    }
    top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:16]
    line["top_launch_shapes"] = [dict(label=k, ms_per_clip=v["ms"] / inst_steps, launches=v["count"] // inst_steps,
                                      tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] else None,
                                      gbps=v["bytes"] / (v["ms"] * 1e-3) / 1e9) for k, v in top]
    if os.environ.get("MD_BENCH_DUMP"):
        with open(os.environ["MD_BENCH_DUMP"], "w") as fh:
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                fh.write(f"{v['ms'] / inst_steps:10.3f} ms/clip  {v['count'] // inst_steps:6d} launches  "
                         f"{(v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else 0:8.1f} TF  "
                         f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:8.1f} GB/s  {k}\n")
    if world == 1 and not args.small and not args.no_vae:
        # SURVEY 8d reports the VAE beside the headline, never inside it: 3F + 2 encodes and F decodes per clip (the calls of
        # src/pipelines/pipeline_mikudance.py:456-549 and :115-130), batches of 8 images as the product pipeline issues them
        del ref, den, pipe
        torch.cuda.empty_cache()
        vae = vae_ms_per_clip(dev, args.size, args.frames)
        # the loop's time is the same for configs[1] and configs[2] (same shapes); the VAE's is not: absent face / hand guidance is 2F copies
        # of one black frame, encoded once
        mine = "configs[2]" if args.config == 2 else "configs[1]"
        line["vae_ms_per_clip"] = vae[mine]["ms"]
        line["vae_by_config"] = vae
        line["e2e_frames_per_s"] = args.frames / (elapsed / args.steps + vae[mine]["ms"] * 1e-3)
        line["e2e_frames_per_s_by_config"] = {k: args.frames / (elapsed / args.steps + v["ms"] * 1e-3) for k, v in vae.items()}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(ref_sd, den_sd, args, ctx)
    emit_line(json.dumps(line))


def measure_traffic(label, device_index=0):
    """HBM-side bytes per launch of the attention shape `label` ("attention B=.. H=.. D=.. Lq=.. Lk=..") from rocprofv3 PMC counters:
    FETCH_SIZE (KiB; doubled: on gfx950 a 128-byte request is tallied at 64 B, MI355X guide, HBM section) and WRITE_SIZE (KiB), each in
    its own --kernel-trace-only pass over a child process that launches the kernel a few times through the same C ABI.  Returns None
    when the label is not an attention shape or anything fails (the caller then quotes the committed record)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    m = re.match(r"attention B=(\d+) H=(\d+) D=(\d+) Lq=(\d+) Lk=(\d+)$", label)
    if not m or shutil.which("rocprofv3") is None:
        return None
    B, H, D, Lq, Lk = map(int, m.groups())
    child = (f"import sys, torch; sys.path.insert(0, {ROOT!r}); from mikudance_amd import ops; C = {H * D}; g = torch.Generator(device='cuda').manual_seed(1); "
             f"q, k = (torch.randn({B * Lq}, C, device='cuda', generator=g).half() for _ in range(2)); "
             f"vt = torch.randn(C, {B * Lk}, device='cuda', generator=g).half(); o = torch.empty_like(q); "
             f"[ops.attention(q, k, vt, {B}, {H}, {D}, {Lq}, {Lk}, out=o) for _ in range(4)]; torch.cuda.synchronize()")
    if Lq != Lk:
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="md_pmc_")
    # the child sees ONLY the GPU this rank measured on (on an 8-GPU node the profiler would otherwise open all of them)
    vis = [v for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v.strip() != ""]
    child_env = dict(os.environ, TMPDIR=tmp, HIP_VISIBLE_DEVICES=vis[device_index] if device_index < len(vis) else str(device_index))
    try:
        for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
            d = os.path.join(tmp, name)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", *counters, "-d", d, "-o", name, "--",
                                sys.executable, "-c", child], cwd=tmp, env=child_env, capture_output=True, text=True, timeout=150)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "attn" in row["Kernel_Name"] and row["Counter_Name"] == counters[0]:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 2:
                return None
            out[name] = sum(vals[1:]) / len(vals[1:])                     # KiB per launch, first launch (cold) left out
    except Exception:                                                     # noqa: BLE001 -- a missing profiler never fails the benchmark
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    alg = 2.0 * B * H * D * (2 * Lq + 2 * Lk)
    return {"hbm_bytes_corrected": (2.0 * out["fetch"] + out["write"]) * 1024.0, "algorithmic_bytes": alg,
            "source": {"measured": "this run, after the timed region: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                                   "passes over 4 launches of the same kernel and shape in a child process (mean of the last 3)",
                       "FETCH_SIZE_KiB": out["fetch"], "WRITE_SIZE_KiB": out["write"],
                       "corrections": "FETCH_SIZE x2 (gfx950 tallies 128-byte requests at 64 B; MI355X guide, HBM section); WRITE_SIZE as read "
                                      "(calibrated 1.00 on 16-byte coalesced stores; this kernel's 8-byte row pieces are written back as partial "
                                      "lines, which the counter tallies per request: it reads 1.2-2.0x the 189 MB written); Infinity-Cache hits "
                                      "are counted on both"}}


def full_guidance(ref_latents, frames, h, w):
    """BASELINE configs[2]: the 2 scene-motion channels come from real camera tracks (the first 16 w2c / c2w matrices of the
    reference's demo clip and its depth map, stored in tests/golden/g2_scene_motion.npz by oracle/gen_golden.py) pushed through
    the product's camera_to_scene_motion at latent resolution (scripts/inference_video.py:185-189, K = [3.2, 3.2, 1.6, 1.6]);
    the face / hand latent channels (12..19) are already non-zero in synth_inputs.  Same tensor shapes as configs[1]."""
    import numpy as np
    from mikudance_amd.scene_motion import camera_to_scene_motion
    z = np.load(os.path.join(ROOT, "tests", "golden", "g2_scene_motion.npz"))
    n = z["w2c"].shape[0]
    idx = [i % n for i in range(frames)]
    d24 = z["depth"][0]
    yi = (np.arange(h) * d24.shape[0] / h).astype(int)
    xi = (np.arange(w) * d24.shape[1] / w).astype(int)
    depth = d24[yi][:, xi][None]
    flow = camera_to_scene_motion([z["w2c"][i] for i in idx], [z["c2w"][i] for i in idx], list(z["K"]), depth, w, h, False)
    out = ref_latents.clone()
    out[0, :, 20:22] = torch.from_numpy(flow).to(out.dtype)
    assert float(out[0, :, 12:20].abs().max()) > 0 and float(out[0, 1:, 20:22].abs().max()) > 0
    return out


def vae_ms_per_clip(dev, size, frames, batch=8):
    """AutoencoderKL (sd-vae-ft-mse geometry, seeded random weights) at the benchmark size through the PRODUCT's own calls: F decodes in
    batches of 8 + MikuDanceVideoPipeline._encode_many over the 3F + 2 condition images (batches of 8, an image that occurs several times is
    encoded once).  Two input sets, median of three wall-clock passes each after a warm-up pass:
      configs[1]  pose frames distinct, face / hand guidance absent = 2F black frames (scripts/inference_video.py:156-180): F + 3 encodes
      configs[2]  every image distinct: 3F + 2 encodes
    Images and latents are device-resident when a pass starts (like `value`; the host -> device copy of the images, 3.5 MB each in fp16,
    is not in these figures).  Not part of `value` (the metric is the denoising loop)."""
    from mikudance_amd import AutoencoderKL, MikuDanceVideoPipeline
    from mikudance_amd.synth import synth_state_dict
    vae = AutoencoderKL()
    vae.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed=77), strict=True)
    vae = vae.half().to(dev).eval()
    pipe = MikuDanceVideoPipeline(vae, None, None, None, None)
    pipe.vae_batch = batch
    g = torch.Generator(device=dev).manual_seed(7)
    lat = torch.randn(frames, 4, size // 8, size // 8, device=dev, generator=g).half()
    distinct = (torch.rand(3 * frames + 2, 3, size, size, device=dev, generator=g) * 2 - 1).half()
    black = distinct.clone()
    black[2 + frames:] = 0                                               # face + hand: the script's Image.new("RGB", size, (0, 0, 0))
    out = {}
    with torch.no_grad():
        for name, imgs in (("configs[1]", black), ("configs[2]", distinct)):
            views = [imgs[i:i + 1] for i in range(imgs.shape[0])]        # one tensor per image, as __call__ hands them over

            def once():
                for i in range(0, frames, batch):
                    vae.decode(lat[i:i + batch]).sample
                pipe._encode_many(views)
            once()                                                       # warm-up: allocator growth, packed weights
            times = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                once()
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) * 1e3)
            out[name] = dict(ms=sorted(times)[1], **pipe.last_encode_stats)       # median of three
    return out


def cpu_quota(root="/sys/fs/cgroup"):
    """(raw text, CPUs) of the container's CPU bandwidth limit: cgroup v2 `cpu.max` ("<quota> <period>" or "max <period>") or cgroup v1
    `cpu/cpu.cfs_quota_us` + `cpu.cfs_period_us` (-1 = none).  CPUs is None when there is no limit (or no cgroup file)."""
    try:
        raw = open(os.path.join(root, "cpu.max")).read().strip()
        q, per = raw.split()[:2]
        return raw, (None if q == "max" else int(q) / int(per))
    except (OSError, ValueError):
        pass
    try:
        raw = open(os.path.join(root, "cpu", "cpu.cfs_quota_us")).read().strip()
        per = int(open(os.path.join(root, "cpu", "cpu.cfs_period_us")).read())
        return raw, (int(raw) / per if int(raw) > 0 else None)
    except (OSError, ValueError):
        return None, None


def cpu_baseline(ref_sd, den_sd, args, ctx):
    """The CPU oracle (a port of the reference's PyTorch path: the same ATen ops, fp32) on this host's cores, on a bounded
    sample, after one untimed warm-up pass (thread pool, allocator, oneDNN primitive caches):
      (1) BASELINE configs[0] IN FULL: 256x256 (32x32 latents), 4 frames, 4 DDIM steps, CFG, literal reference algorithm;
      (2) ONE DDIM step at the benchmark resolution on ONE frame (reference_unet + denoising_unet on the [uncond | cond]
          pair) -- `value` = 1 frame / (ddim_steps x that time), i.e. the frame-step time extrapolated linearly over frames and
          steps; a full 16-frame clip is ~2 PFLOP = hours of CPU.
    Both legs run on the BEST intra-op thread count of a sweep (configs[0]: 8 / 16 / 32 / 64 / all on one of its steps ->
    `config1_cores`, `thread_sweep_s_per_step`; the 768x768 step: 16 / 32 / 64 / all -> `cores`, `thread_sweep_s_per_step_at_size`);
    the physical core count of the host is reported beside them."""
    from oracle import cpu_ref as O                                     # cpu_baseline leg only
    from mikudance_amd.synth import synth_inputs
    try:
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:
        physical = None
    full_ctx = (257, 768) if not args.small else ctx
    avail = torch.get_num_threads()
    sweep = {}
    q_raw0, q_cpus0 = cpu_quota()
    # untimed warm-up passes run on what the container may actually use: with a 16-CPU quota on a 128-thread host a warm-up on every
    # hardware thread is a 36-s throttling exercise (the timed sweeps below still include `avail`, as the evidence for the quota story)
    warm_threads = min(avail, int(q_cpus0 + 0.5)) if q_cpus0 else avail
    with torch.no_grad():
        lat1, rl1, emb1 = synth_inputs(4, 32, 32, ctx_len=full_ctx[0], ctx_dim=full_ctx[1], seed=100)
        torch.set_num_threads(warm_threads)
        O.denoise_loop(ref_sd, den_sd, lat1, rl1, emb1, 1, guidance_scale=args.guidance)          # warm-up (untimed)
        # intra-op thread sweep on ONE DDIM step of configs[0]: ATen's small convs / GEMMs at this size stop scaling long before
        # 128 threads (the survey measured 8.5 s per step on 8 cores where 128 threads take ~18 s); the best count is used below
        counts0 = (8, 16, 32, 64, avail) if not q_cpus0 else (max(1, int(q_cpus0 + 0.5) // 2), int(q_cpus0 + 0.5), 2 * int(q_cpus0 + 0.5), avail)
        for n in sorted({n for n in counts0 if 0 < n <= avail}):
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            O.denoise_loop(ref_sd, den_sd, lat1, rl1, emb1, 1, guidance_scale=args.guidance)
            sweep[n] = time.perf_counter() - t0
        threads = min(sweep, key=sweep.get)
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        O.denoise_loop(ref_sd, den_sd, lat1, rl1, emb1, 4, guidance_scale=args.guidance)
        dt1 = time.perf_counter() - t0
        # The step at the benchmark resolution: ONE frame (the per-frame cost of both UNets is linear in the frame count; the
        # f x f temporal core is ~0.1 % of the FLOPs), one untimed warm-up step, then one timed step per intra-op thread count
        # of the sweep -- `value` is the BEST of them, `cores` the count that achieved it.
        h = w = args.size // 8
        f = 1
        lat, rl, emb = synth_inputs(f, h, w, ctx_len=full_ctx[0], ctx_dim=full_ctx[1], seed=100)
        torch.set_num_threads(warm_threads)
        O.denoise_loop(ref_sd, den_sd, lat, rl, emb, 1, guidance_scale=args.guidance)             # warm-up (untimed)
        sweep2 = {}
        counts = (16, 32, 64, avail) if not q_cpus0 else (int(q_cpus0 + 0.5), 2 * int(q_cpus0 + 0.5), avail)   # a quota: the quota, twice it, everything
        for n in sorted({n for n in counts if 0 < n <= avail}, reverse=True):
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            O.denoise_loop(ref_sd, den_sd, lat, rl, emb, 1, guidance_scale=args.guidance)
            sweep2[n] = time.perf_counter() - t0
        best = min(sweep2, key=sweep2.get)
        forced = os.environ.get("MD_CPU_WORKER_THREADS")
        if forced:                                                       # experiments / tests: pin the per-worker thread count (recorded in the line)
            best = int(forced)
            torch.set_num_threads(best)
            t0 = time.perf_counter()
            O.denoise_loop(ref_sd, den_sd, lat, rl, emb, 1, guidance_scale=args.guidance)
            sweep2[best] = time.perf_counter() - t0
        dt2 = sweep2[best]
        # (2b) the extrapolation shown, not assumed (SURVEY.md 8d asked for one full step): ONE step of FOUR frames on the same thread count.
        # `value` stays the one-frame figure only if this is within 10 % of 4 x the one-frame step; otherwise the four-frame step is the sample.
        f4 = 4
        torch.set_num_threads(best)
        lat4, rl4, emb4 = synth_inputs(f4, h, w, ctx_len=full_ctx[0], ctx_dim=full_ctx[1], seed=100)
        t0 = time.perf_counter()
        O.denoise_loop(ref_sd, den_sd, lat4, rl4, emb4, 1, guidance_scale=args.guidance)
        dt2_f4 = time.perf_counter() - t0
        lin = dt2_f4 / (f4 * dt2)                                        # 1.0 = exactly linear in the frame count
        linear_ok = abs(lin - 1.0) <= 0.10
        # (3) ALL usable cores.  "Usable" = min(CPU affinity, cgroup CPU quota): on the GPU boxes of this pool the host has 128 cores / 256
        # hardware threads and the container a quota of 16 CPUs (cpu.max "1600000 100000"), which is why 16 intra-op threads are best and
        # every further thread only adds throttling.  When the quota leaves room for more than one `best`-thread worker, P = usable // best
        # independent frames run at once (P Python threads of this process: every thread owns its own OpenMP team, the GIL is released
        # inside ATen, the weights are shared) and `value` is the better of the two legs.
        affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else avail
        quota_raw, quota_cpus = cpu_quota()
        usable = min(affinity, avail, int(quota_cpus + 0.5)) if quota_cpus else min(affinity, avail)
        P = max(1, usable // best)
        frames_in = [synth_inputs(f, h, w, ctx_len=full_ctx[0], ctx_dim=full_ctx[1], seed=100 + i) for i in range(P)]

        def worker(i):
            torch.set_num_threads(best)                                  # per calling thread (OpenMP ICV)
            with torch.no_grad():
                O.denoise_loop(ref_sd, den_sd, *frames_in[i], 1, guidance_scale=args.guidance)

        from concurrent.futures import ThreadPoolExecutor
        dt3 = None
        if P > 1:
            with ThreadPoolExecutor(P) as ex:                            # untimed: every worker thread builds its OpenMP team and touches its buffers
                list(ex.map(worker, range(P)))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(P) as ex:
                list(ex.map(worker, range(P)))
            dt3 = time.perf_counter() - t0
        torch.set_num_threads(avail)
    single = f / (dt2 * args.ddim_steps) if linear_ok else f4 / (dt2_f4 * args.ddim_steps)
    allcore = P * f / (dt3 * args.ddim_steps) if dt3 else single
    eff = (dt2 / dt3) if dt3 else 1.0                                     # 1.0 = P workers finish in the time of one
    use_all = allcore > single
    top = max(sweep2)
    slower = sweep2[top] > 1.05 * sweep2[best] and top > best
    sweep_txt = ", ".join(f"{k}: {v:.1f} s" for k, v in sorted(sweep2.items()))
    if quota_cpus and quota_cpus < min(affinity, avail):
        why = (f"the container's CPU quota is {quota_cpus:g} CPUs (cgroup cpu.max '{quota_raw}') on a host with {physical} physical cores / {affinity} hardware "
               f"threads in the affinity mask: ONE call by thread count {sweep_txt} -- threads beyond the quota are throttled, so {best} threads ARE all usable cores")
    elif dt3:
        why = (f"{'more intra-op threads are slower for ONE call' if slower else 'ONE call by thread count'} ({sweep_txt}) while {P} independent workers x {best} "
               f"threads finish in {dt3:.1f} s ({100 * eff:.0f} % of linear); no CPU quota (cgroup '{quota_raw}', affinity {affinity}): ATen's per-operator "
               f"fork/join over 1-2 images does not spread further, independent frames do")
    else:
        why = f"single worker (affinity {affinity}, cgroup '{quota_raw}', {avail} intra-op threads; ONE call by thread count {sweep_txt})"
    quota = quota_raw
    return {"value": allcore if use_all else single, "unit": "frames/s", "cores": P * best if use_all else best, "physical_cores": physical,
            "usable_cores": usable, "affinity": affinity, "cgroup_cpu_max": quota, "cgroup_quota_cpus": quota_cpus, "kind": "port",
            "single_process": {"frames_per_s": single, "threads": best, "s_per_frame_step": round(dt2, 2), "threads_forced_by_env": bool(forced)},
            "frames_linearity": {"s_per_step_1_frame": round(dt2, 2), "s_per_step_4_frames": round(dt2_f4, 2), "ratio_to_linear": round(lin, 3),
                                 "within_10_percent": linear_ok, "value_from": "1-frame step" if linear_ok else "4-frame step"},
            "all_cores": {"workers": P, "threads_per_worker": best, "s_for_all_workers": round(dt3, 2) if dt3 else None,
                          "frames_per_s": allcore, "parallel_efficiency": round(eff, 3)},
            "thread_sweep_s_per_step_at_size": {str(k): round(v, 2) for k, v in sorted(sweep2.items())},
            "config1_cores": threads, "thread_sweep_s_per_step": {str(k): round(v, 2) for k, v in sweep.items()},
            "config1_full_s": dt1, "config1_frames_per_s": 4.0 / dt1,
            "sample": f"after one warm-up pass each: (1) configs[0] in full (256x256, 4 frames, 4 DDIM steps, fp32, literal algorithm) on {threads} "
                      f"intra-op threads (best of the sweep) = {dt1:.1f} s; (2) 1 DDIM step of {f} frame at {args.size}x{args.size} (reference_unet + "
                      f"denoising_unet, CFG pair, fp32, full-width random-init weights, literal algorithm) timed on "
                      f"{'/'.join(str(k) for k in sorted(sweep2))} threads, best = {best} threads{' (forced by MD_CPU_WORKER_THREADS)' if forced else ''} = {dt2:.1f} s; "
                      f"(2b) 1 DDIM step of {f4} frames on {best} threads = {dt2_f4:.1f} s = {lin:.2f} x linear; (3) the same frame-step on "
                      f"{P} independent frame(s) at once, {best} threads each = {P * best} of {usable} usable cores; value = "
                      f"{'(3)' if use_all else '(2)'}: frames / ({args.ddim_steps} steps x wall time), the {'one' if linear_ok else 'four'}-frame step extrapolated linearly over "
                      f"frames and steps (measured: 4 frames = {lin:.2f} x 4 one-frame steps).  Cores: {why}"}


if __name__ == "__main__":
    main()
