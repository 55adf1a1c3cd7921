"""Schedule generators.  Only ``gen_default`` (all-true) and the FORA-style
``gen_recompute_all_every_n`` are provided; the reference's other heuristic families
(``ecad/schedulers/cache_scheduler/generators/``) are out of scope.
"""
from __future__ import annotations

from typing import Iterator

from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule


def _uniform_step(num_blocks: int, attn1: bool, attn2: bool, ff: bool) -> dict:
    return {str(b): {"attn1": attn1, "attn2": attn2, "ff": ff} for b in range(num_blocks)}


def gen_default(num_blocks: int, num_inference_steps: int) -> Iterator[PixArtCacheSchedule]:
    """All components recompute at every step (reference: pixart_schedule_generators.py:20-30)."""
    yield PixArtCacheSchedule(
        num_blocks=num_blocks,
        num_inference_steps=num_inference_steps,
        name="default",
        schedule={s: _uniform_step(num_blocks, True, True, True) for s in range(num_inference_steps)},
        attributes={},
    )


def gen_recompute_all_every_n(num_blocks: int, num_inference_steps: int, n: int) -> PixArtCacheSchedule:
    sched = {s: _uniform_step(num_blocks, *(3 * [s % n == 0])) for s in range(num_inference_steps)}
    return PixArtCacheSchedule(num_blocks, num_inference_steps, f"recompute_all_every_{n:03}", sched, attributes={"n": n})
