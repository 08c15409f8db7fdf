"""spp-split across GPUs: every (pixel, frameCounter) sample is independent (seed f5:320-323, Sobol index
f5:883), so rank r of `world` renders its own frameCounter ranges of the whole image into a private fp32 SUM buffer
(scene replicated) and the buffers are combined with ONE reduce per step; the root divides by the total sample count.
Equivalent to the reference's running mean (f5:954-955) up to fp32 summation order.  No other collective exists on
this path."""
from __future__ import annotations


def frame_range(step: int, rank: int, world: int, spp: int) -> tuple[int, int]:
    """(frame_begin, frame_count) of `rank` in `step`: step s covers frames [s*world*spp, (s+1)*world*spp), split into
    `world` contiguous blocks of `spp` frames."""
    if not (0 <= rank < world) or spp < 0 or step < 0:
        raise ValueError("bad partition arguments")
    return (step * world + rank) * spp, spp


def frames_covered(steps: int, world: int, spp: int) -> int:
    return steps * world * spp


def reduce_sum(accum, dist=None, dst: int = 0):
    """One sum-reduce of the per-rank accumulation buffers to `dst` (RCCL over xGMI on GPUs, gloo in CPU tests)."""
    if dist is not None and dist.is_initialized():  # (a single rank reduces onto itself: used to rehearse the path)
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def normalize(accum_sum, total_spp: int):
    """Sum buffer -> mean radiance (what lastFrame holds after total_spp frames)."""
    return accum_sum / float(total_spp)
