"""FLUX schedule generators: ``gen_default`` (all-true, with the optional height/width/guidance config block) and the
FORA-style ``gen_recompute_all_every_n``.  Reference: ``ecad/schedulers/cache_scheduler/generators/
flux_schedule_generators.py:157-300``; the other heuristic families are out of scope.
"""
from __future__ import annotations

from typing import Iterator

from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FULL_COMPONENTS, SINGLE_COMPONENTS, FluxCacheSchedule


def _uniform_step(num_blocks: int, num_single_blocks: int, value: bool) -> dict:
    step = {str(b): {c: value for c in FULL_COMPONENTS} for b in range(num_blocks)}
    step.update({f"single_{b}": {c: value for c in SINGLE_COMPONENTS} for b in range(num_single_blocks)})
    return step


def gen_default(num_blocks: int, num_single_blocks: int, num_inference_steps: int, height: int | None = None, width: int | None = None,
                guidance_scale: float | None = None) -> Iterator[FluxCacheSchedule]:
    cfg = None if height is None or width is None or guidance_scale is None else {"height": height, "width": width, "guidance_scale": guidance_scale}
    yield FluxCacheSchedule(
        num_blocks=num_blocks,
        num_inference_steps=num_inference_steps,
        name=f"default_{height}x{width}_gs_{guidance_scale}",
        num_single_blocks=num_single_blocks,
        schedule={s: _uniform_step(num_blocks, num_single_blocks, True) for s in range(num_inference_steps)},
        top_level_config=cfg,
        attributes={},
    )


def gen_recompute_all_every_n(num_blocks: int, num_single_blocks: int, num_inference_steps: int, n: int, **cfg) -> FluxCacheSchedule:
    sched = {s: _uniform_step(num_blocks, num_single_blocks, s % n == 0) for s in range(num_inference_steps)}
    return FluxCacheSchedule(num_blocks, num_inference_steps, f"recompute_all_every_{n:03}", sched, top_level_config=cfg or None,
                             attributes={"n": n}, num_single_blocks=num_single_blocks)
