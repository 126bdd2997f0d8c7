"""Reference-compatible names for the window scheduler (src/pipelines/context.py); the logic lives in windows.py."""
from typing import Callable, Optional

from .windows import bit_reversed_fraction as ordered_halving  # noqa: F401
from .windows import iter_windows


def uniform(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
            context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True):
    return iter_windows(step, num_frames, context_size, context_stride, context_overlap, closed_loop)


def get_context_scheduler(name: str) -> Callable:
    if name != "uniform":
        raise ValueError(f"Unknown context_overlap policy {name}")
    return uniform


def get_total_steps(scheduler, timesteps, num_steps=None, num_frames=..., context_size=None, context_stride=3,
                    context_overlap=4, closed_loop=True):
    total = 0
    for i in range(len(timesteps)):
        total += sum(1 for _ in scheduler(i, num_steps, num_frames, context_size, context_stride, context_overlap))
    return total
