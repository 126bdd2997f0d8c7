"""Data-parallel clips: one process per GPU, independent (ref_image, pose_sequence) clips sharded across ranks.

The reference has no inference-time parallelism (batch_size = 1 hard-coded, src/pipelines/pipeline_mikudance.py:403);
clips share no state, so the only communication is ONE scatter of per-clip conditioning (~8.5 MB/clip at 768x768x16f)
and ONE gather of the final latents (~1.2 MB/clip) per batch, over torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for the tests).  There is no per-step collective.

Second mode, for ONE long clip (BASELINE configs[4]: 48 frames = 3 windows of 30, 37 s per clip on one GPU): the context windows of a
DDIM step are independent UNet evaluations (src/pipelines/pipeline_mikudance.py:625-668) whose predictions are summed per frame
(:662-674).  `WindowParallel` gives window i to rank i mod N; every rank keeps the full latents, accumulates its own windows and ONE
all_reduce(sum) of (noise_sum, counter) per step -- 4.7 MB at configs[4] -- precedes the CFG + DDIM update, which every rank then
executes identically.  `MikuDanceVideoPipeline.denoise(..., window_parallel=WindowParallel())`.
"""
import os

import torch
import torch.distributed as dist


def _forced():
    """MD_DIST_FORCE=1: a ONE-rank job still builds its process group and sends every collective of this module through the backend
    (RCCL on a GPU box): the only way to execute the RCCL code path -- scatter, gather, all_reduce, the object collectives, the
    device-bound barrier -- on a machine with a single GPU (tools/r06_gpu.sh rccl1; profiles/r06_rccl_one_rank.json)."""
    return os.environ.get("MD_DIST_FORCE") == "1"


def active():
    """True when the helpers below really communicate: more than one rank, or a forced one-rank group."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not _forced():
        return 0, 1
    if world == 1:                                          # forced one-rank group started without torchrun: its own rendezvous
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
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
    if not active():
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
    if not active():
        return [latents]
    rank, world = dist.get_rank(), dist.get_world_size()
    device = latents.device
    if dist.get_backend() != "nccl":
        latents = latents.cpu()
    bucket = [torch.empty_like(latents) for _ in range(world)] if rank == dst else None
    dist.gather(latents.contiguous(), bucket, dst=dst)
    return [b.to(device) for b in bucket] if bucket is not None else None


def barrier():
    if active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])     # the rank's own GPU (set in init), no device guessing
        else:
            dist.barrier()


def _reduce(value, device, op):
    if not active():
        return value
    # fp32 on the wire (RCCL reduces it natively on every build; fp64 is the rarer code path and nothing here needs it)
    t = torch.tensor([value], dtype=torch.float32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(value, device):
    return _reduce(value, device, dist.ReduceOp.MAX)


def sum_over_ranks(value, device):
    return _reduce(value, device, dist.ReduceOp.SUM)


def gather_objects(obj, dst=0):
    """Small picklable per-rank records (timings, device identity) on `dst`, ordered by rank; None elsewhere."""
    if not active():
        return [obj]
    bucket = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, bucket, dst=dst)
    return bucket


def device_identity(device=None):
    """Who this rank really computes on: proves N DISTINCT GPUs behind N ranks (LOCAL_RANK % device_count would silently alias two ranks
    onto one GPU on a node that exposes fewer devices than ranks).  UUID and PCI bus id come from the HIP runtime's device properties."""
    import socket
    rec = {"host": socket.gethostname(), "pid": os.getpid(), "rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
           "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")}
    if torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda"):
        idx = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
        pr = torch.cuda.get_device_properties(idx)
        rec.update(device_index=idx, device_count=torch.cuda.device_count(), name=pr.name, cus=pr.multi_processor_count,
                   hbm_gib=round(pr.total_memory / 2 ** 30, 1), uuid=str(getattr(pr, "uuid", "")) or None,
                   pci_bus_id="%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
                   if hasattr(pr, "pci_bus_id") else None)
    else:
        rec.update(device_index=None, name="cpu")
    return rec


def collective_library():
    """What moves the bytes: backend name and, for "nccl" (= RCCL on ROCm), the library version torch was built against / loaded."""
    out = {"backend": dist.get_backend() if dist.is_initialized() else None, "world": dist.get_world_size() if dist.is_initialized() else 1}
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                       # noqa: BLE001 -- a CPU-only build has no RCCL to ask
        out["rccl_version"] = None
    out["hip"] = getattr(torch.version, "hip", None)
    return out


class WindowParallel:
    """Window-level parallelism of ONE clip across the ranks of a process group (module docstring).  Passed to
    MikuDanceVideoPipeline.denoise(window_parallel=...): `mine(i)` says whether this rank evaluates window i of a step, `reduce` sums the
    per-frame accumulators of all ranks.  Every rank must enter denoise() with the same latents / conditioning (and, for eta > 0,
    generators in the same state): the latents are replicated, never communicated.  Frames covered by at most two windows (the usual
    overlap) are summed commutatively, i.e. bit-identically to the one-GPU order; with three or more covering windows the fp32 sum may
    differ in the last bit."""

    def __init__(self, group=None):
        self.group = group
        on = dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1

    def mine(self, window_index):
        return window_index % self.world == self.rank

    def reduce(self, noise_sum, counter):
        if self.world == 1 and not active():
            return
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(noise_sum, group=self.group)
            dist.all_reduce(counter, group=self.group)
            return
        for t in (noise_sum, counter):                      # gloo (CPU tests, single-GPU dry runs): through host memory
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)


def shutdown():
    """Every rank leaves the job the same way (also on an exception): a rank that returns without destroy_process_group() makes
    RCCL's watchdog of the others log aborts, or hang them inside their last collective."""
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:                                   # noqa: BLE001 -- shutting down after a failure must not mask it
            pass
