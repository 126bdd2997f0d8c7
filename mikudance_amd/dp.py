"""Data-parallel clips: one process per GPU, independent (ref_image, pose_sequence) clips sharded across ranks.

The reference has no inference-time parallelism (batch_size = 1 hard-coded, src/pipelines/pipeline_mikudance.py:403);
clips share no state, so the only communication is ONE scatter of per-clip conditioning (~8.5 MB/clip at 768x768x16f)
and ONE gather of the final latents (~1.2 MB/clip) per batch, over torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for the tests).  There is no per-step collective.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def shard(items, rank, world):
    """clip i -> rank i mod world."""
    return [x for i, x in enumerate(items) if i % world == rank]


def scatter_clips(clips, device, src=0):
    """clips: on `src` a list (one per rank) of tuples of tensors with identical shapes/dtypes across ranks; None
    elsewhere.  Returns this rank's tuple.  One dist.scatter per tensor slot."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tuple(t.to(device) for t in clips[0])
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = [None]
    if rank == src:
        assert len(clips) == world
        meta[0] = [(tuple(t.shape), t.dtype) for t in clips[0]]
    dist.broadcast_object_list(meta, src=src)
    out = []
    for slot, (shape, dtype) in enumerate(meta[0]):
        recv = torch.empty(shape, dtype=dtype, device=device)
        send = [c[slot].to(device).contiguous() for c in clips] if rank == src else None
        dist.scatter(recv, send, src=src)
        out.append(recv)
    return tuple(out)


def gather_latents(latents, dst=0):
    """Gather every rank's final latents on `dst` (list ordered by rank); other ranks get None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [latents]
    rank, world = dist.get_rank(), dist.get_world_size()
    bucket = [torch.empty_like(latents) for _ in range(world)] if rank == dst else None
    dist.gather(latents.contiguous(), bucket, dst=dst)
    return bucket


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
