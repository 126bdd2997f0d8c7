"""Sliding-window frame scheduler for long clips -- API mirror of reference src/pipelines/context.py:7-49.
Host-side integer work; windows wrap around (closed loop) and are identical at every step because the pipeline
always passes step=0 (quirk 7)."""
import math
from typing import Callable, Optional


def ordered_halving(val: int) -> float:
    """Bit-reversed fraction of a 64-bit integer: van-der-Corput style offset in [0, 1)."""
    rev = 0
    for _ in range(64):
        rev = (rev << 1) | (val & 1)
        val >>= 1
    return rev / (1 << 64)


def uniform(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
            context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True):
    if num_frames <= context_size:
        yield list(range(num_frames))
        return
    context_stride = min(context_stride, int(math.ceil(math.log2(num_frames / context_size))) + 1)
    frac = ordered_halving(step)
    for level in range(context_stride):
        context_step = 1 << level
        pad = int(round(num_frames * frac))
        start = int(frac * context_step) + pad
        stop = num_frames + pad + (0 if closed_loop else -context_overlap)
        stride = context_size * context_step - context_overlap
        for j in range(start, stop, stride):
            yield [e % num_frames for e in range(j, j + context_size * context_step, context_step)]


def get_context_scheduler(name: str) -> Callable:
    if name == "uniform":
        return uniform
    raise ValueError(f"Unknown context_overlap policy {name}")


def get_total_steps(scheduler, timesteps, num_steps=None, num_frames=..., context_size=None, context_stride=3,
                    context_overlap=4, closed_loop=True):
    return sum(len(list(scheduler(i, num_steps, num_frames, context_size, context_stride, context_overlap)))
               for i in range(len(timesteps)))
