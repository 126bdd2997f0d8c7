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
        backend = backend or os.environ.get("MD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
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
    # RCCL moves device tensors directly; the gloo backend (CPU tests, single-GPU dry runs) stages through host memory
    comm_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    for slot, (shape, dtype) in enumerate(meta[0]):
        recv = torch.empty(shape, dtype=dtype, device=comm_dev)
        send = [c[slot].to(comm_dev).contiguous() for c in clips] if rank == src else None
        dist.scatter(recv, send, src=src)
        out.append(recv.to(device))
    return tuple(out)


def gather_latents(latents, dst=0):
    """Gather every rank's final latents on `dst` (list ordered by rank); other ranks get None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [latents]
    rank, world = dist.get_rank(), dist.get_world_size()
    device = latents.device
    if dist.get_backend() != "nccl":
        latents = latents.cpu()
    bucket = [torch.empty_like(latents) for _ in range(world)] if rank == dst else None
    dist.gather(latents.contiguous(), bucket, dst=dst)
    return [b.to(device) for b in bucket] if bucket is not None else None


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])     # the rank's own GPU (set in init), no device guessing
        else:
            dist.barrier()


def _reduce(value, device, op):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    # fp32 on the wire (RCCL reduces it natively on every build; fp64 is the rarer code path and nothing here needs it)
    t = torch.tensor([value], dtype=torch.float32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(value, device):
    return _reduce(value, device, dist.ReduceOp.MAX)


def sum_over_ranks(value, device):
    return _reduce(value, device, dist.ReduceOp.SUM)


def shutdown():
    """Every rank leaves the job the same way (also on an exception): a rank that returns without destroy_process_group() makes
    RCCL's watchdog of the others log aborts, or hang them inside their last collective."""
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:                                   # noqa: BLE001 -- shutting down after a failure must not mask it
            pass
