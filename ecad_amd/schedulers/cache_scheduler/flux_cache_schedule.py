"""FLUX cache schedule: three cacheable components per joint ("full") block and three per single-stream block.

Mirror of ``ecad/schedulers/cache_scheduler/flux_cache_schedule.py:13-90``: same constructor (``num_single_blocks`` is
required), ``components`` order, ``to_dict`` (adds ``num_single_blocks``) and flat ``to_numpy`` layout -- per step, all
full blocks ``(full_attn, full_ff, full_ff_context)`` then all single blocks ``(single_attn, single_proj_mlp,
single_proj_out)``.  ``to_device_bits`` splits that vector into the two ``[steps][blocks][3]`` byte arrays
``ecad_flux_set_schedule`` takes.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from ecad_amd.schedulers.cache_scheduler.cache_schedule import CacheSchedule

FULL_COMPONENTS = ["full_attn", "full_ff", "full_ff_context"]
SINGLE_COMPONENTS = ["single_attn", "single_proj_mlp", "single_proj_out"]


class FluxCacheSchedule(CacheSchedule):
    def __init__(
        self,
        num_blocks: int,
        num_inference_steps: int,
        name: str,
        schedule: dict,
        top_level_config: dict[str, Any] | None = None,
        attributes: dict[str, Any] | None = None,
        metrics: dict[str, Any] | None = None,
        num_single_blocks: int | None = None,
        **kwargs: Any,
    ) -> None:
        super().__init__(num_blocks, num_inference_steps, name, schedule, top_level_config, attributes, metrics)
        if num_single_blocks is None:
            raise ValueError("num_single_blocks must be provided for FluxCacheSchedule")
        self.num_single_blocks = int(num_single_blocks)

    def to_dict(self) -> dict[str, Any]:
        data = super().to_dict()
        data["cache_schedule"]["num_single_blocks"] = self.num_single_blocks
        return data

    @property
    def components(self) -> list[str]:
        return SINGLE_COMPONENTS + FULL_COMPONENTS

    def to_device_bits(self) -> tuple[np.ndarray, np.ndarray]:
        """(full [steps, num_blocks, 3], single [steps, num_single_blocks, 3]) bool arrays."""
        full = np.zeros((self.num_inference_steps, self.num_blocks, 3), dtype=np.bool_)
        single = np.zeros((self.num_inference_steps, self.num_single_blocks, 3), dtype=np.bool_)
        for step, per_block in self.schedule.items():
            if step >= self.num_inference_steps:
                continue
            for block_num, comp in per_block.items():
                if str(block_num).startswith("single_"):
                    single[step, int(str(block_num)[7:])] = [comp[c] for c in SINGLE_COMPONENTS]
                else:
                    full[step, int(block_num)] = [comp[c] for c in FULL_COMPONENTS]
        return full, single

    def to_numpy(self, flatten: bool = True) -> np.ndarray:
        if not flatten:
            raise NotImplementedError("FluxCacheSchedule only supports flatten=True")
        full, single = self.to_device_bits()
        steps = self.num_inference_steps
        return np.concatenate([full.reshape(steps, -1), single.reshape(steps, -1)], axis=1).reshape(-1)

    @classmethod
    def from_numpy(cls, bits: np.ndarray, name: str = "from_numpy", num_blocks: int = 19, num_single_blocks: int = 38,
                   num_inference_steps: int = 20, **kw) -> "FluxCacheSchedule":
        """Inverse of ``to_numpy`` (``binary_vector_to_schedule_dict``, flux_population_io_manager.py:221-265)."""
        arr = np.asarray(bits).reshape(num_inference_steps, (num_blocks + num_single_blocks) * 3)
        schedule = {}
        for s in range(num_inference_steps):
            per_block: dict[str, dict[str, bool]] = {}
            for b in range(num_blocks):
                per_block[str(b)] = {c: bool(arr[s, b * 3 + i]) for i, c in enumerate(FULL_COMPONENTS)}
            for b in range(num_single_blocks):
                per_block[f"single_{b}"] = {c: bool(arr[s, (num_blocks + b) * 3 + i]) for i, c in enumerate(SINGLE_COMPONENTS)}
            schedule[s] = per_block
        return cls(num_blocks, num_inference_steps, name, schedule, num_single_blocks=num_single_blocks, **kw)
