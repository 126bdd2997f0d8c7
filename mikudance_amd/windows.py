"""Frame windows for clips longer than the temporal context (host-side integer work).

Behavioural mirror of the reference's sliding-window scheduler (src/pipelines/context.py:7-49, called from
src/pipelines/pipeline_mikudance.py:577-589 as `scheduler(0, steps, F, context_frames, context_stride, context_overlap)`):
a clip that fits the context is one window; otherwise windows of `size` frames are laid out at dilations 1, 2, 4, ...
(`levels` of them), each level starting at a step-dependent phase and advancing by `size * dilation - overlap`, with
frame indices wrapping around the clip (closed loop).  The pipeline always passes step = 0, so the phase is 0 and the
same windows are used at every denoising step.  Pinned by tests/golden/g1_windows.json."""
import math
from typing import Iterator, List


def bit_reversed_fraction(value: int, bits: int = 64) -> float:
    """value's `bits`-bit pattern mirrored and read as a binary fraction in [0, 1) (0 -> 0.0, 1 -> 0.5, 2 -> 0.25, ...)."""
    mirrored = 0
    for k in range(bits):
        if (value >> k) & 1:
            mirrored |= 1 << (bits - 1 - k)
    return mirrored / float(1 << bits)


class WindowLayout:
    """All windows of one denoising step."""

    def __init__(self, frames: int, size: int, max_levels: int, overlap: int, wrap: bool = True):
        self.frames, self.size, self.overlap, self.wrap = frames, size, overlap, wrap
        if frames > size and size - overlap <= 0:
            # the reference's range() would be given a step <= 0 here (src/pipelines/context.py:33-37): ValueError for 0,
            # an empty schedule (then a 0/0 average) for a negative one -- refuse both up front
            raise ValueError(f"context_overlap ({overlap}) must be smaller than context_frames ({size})")
        self.levels = 0 if frames <= size else min(max_levels, int(math.ceil(math.log2(frames / size))) + 1)

    def at_step(self, step: int) -> List[List[int]]:
        if self.frames <= self.size:
            return [list(range(self.frames))]
        phase = bit_reversed_fraction(step)
        shift = int(round(self.frames * phase))
        last = self.frames + shift - (0 if self.wrap else self.overlap)
        out = []
        for level in range(self.levels):
            dilation = 2 ** level
            first = int(phase * dilation) + shift
            advance = self.size * dilation - self.overlap
            span = self.size * dilation
            begin = first
            while begin < last:
                out.append([idx % self.frames for idx in range(begin, begin + span, dilation)])
                begin += advance
        return out


def iter_windows(step, frames, size, max_levels, overlap, wrap=True) -> Iterator[List[int]]:
    return iter(WindowLayout(frames, size, max_levels, overlap, wrap).at_step(step))
