"""PixArt cache schedule: components ``attn1, attn2, ff`` per block.

Mirror of ``ecad/schedulers/cache_scheduler/pixart_cache_schedule.py:9-37``; the bit
layout of ``to_numpy`` -- ``(step, block, [attn1, attn2, ff])`` -- is the layout the C
ABI's ``ecad_set_schedule`` takes.
"""
from __future__ import annotations

import numpy as np

from ecad_amd.schedulers.cache_scheduler.cache_schedule import CacheSchedule


class PixArtCacheSchedule(CacheSchedule):
    @property
    def components(self) -> list[str]:
        return ["attn1", "attn2", "ff"]

    def to_numpy(self, flatten: bool = False) -> np.ndarray:
        arr = np.zeros((self.num_inference_steps, self.num_blocks, 3), dtype=np.bool_)
        for step, per_block in self.schedule.items():
            for block_num, comp in per_block.items():
                arr[step, int(block_num)] = [comp[c] for c in self.components]
        return arr.flatten() if flatten else arr

    @classmethod
    def from_numpy(cls, bits: np.ndarray, name: str = "from_numpy", num_blocks: int = 28, num_inference_steps: int = 20, **kw) -> "PixArtCacheSchedule":
        """Inverse of ``to_numpy`` (``binary_vector_to_schedule_dict``, pixart_population_io_manager.py:213-240)."""
        arr = np.asarray(bits).reshape(num_inference_steps, num_blocks, 3)
        schedule = {
            s: {str(b): {"attn1": bool(arr[s, b, 0]), "attn2": bool(arr[s, b, 1]), "ff": bool(arr[s, b, 2])} for b in range(num_blocks)}
            for s in range(num_inference_steps)
        }
        return cls(num_blocks, num_inference_steps, name, schedule, **kw)

    def get_custom_compute_attn(self, block_num: str) -> dict:
        return self.schedule[self.curr_step][str(block_num)].get("custom_compute_attn", {})

    def get_custom_compute_ff(self, block_num: str) -> dict:
        return self.schedule[self.curr_step][str(block_num)].get("custom_compute_ff", {})

    def _named_functions(self):
        """(step index, block index, key, registry, component columns, name) of every compute function a schedule row names."""
        from ecad_amd.transformer_blocks.custom_attn_ff import ComputeAttnRegistry, ComputeFFRegistry

        for step_key, per_block in self.schedule.items():
            for block_num, comp in per_block.items():
                for key, reg, cols in (("custom_compute_attn", ComputeAttnRegistry, (0, 1)), ("custom_compute_ff", ComputeFFRegistry, (2,))):
                    name = (comp.get(key) or {}).get("name")
                    if name is not None:
                        yield int(step_key), int(block_num), key, reg, cols, name

    def custom_component_mask(self) -> np.ndarray:
        """[blocks, 3] (attn1, attn2, ff): 1 where SOME step names a compute function only Python can run for that block (``custom_compute_attn`` covers attn1 and
        attn2, ``custom_compute_ff`` covers ff) -- those components run through the registered Python callables (``ecad_set_custom_components``) at every step; at
        steps that name the default or a native variant, that callable is what gets called.  Functions with a native variant (``ComputeRegistry.native_variant``)
        do not set the mask: the engine runs their rule itself."""
        mask = np.zeros((self.num_blocks, 3), dtype=np.uint8)
        for _, block, _, reg, cols, name in self._named_functions():
            if reg.native_variant(name) is None:
                mask[block, list(cols)] = 1
        return mask

    def native_variant_kinds(self) -> np.ndarray:
        """[steps, blocks, 3] ECAD_NATIVE_* ids for ``ecad_set_native_variants``: the variant of the function each (step, block) names, for the components the
        host-callback mask does not claim.  ``compute_attn_cfg_mean_cache`` applies to attn2 only (its attn1 is the default rule)."""
        kinds = np.zeros((self.num_inference_steps, self.num_blocks, 3), dtype=np.uint8)
        host = self.custom_component_mask()
        for step, block, key, reg, cols, name in self._named_functions():
            kind = reg.native_variant(name)
            if not kind:
                continue
            for c in cols:
                if host[block, c] or (kind == 2 and c != 1):
                    continue
                kinds[step, block, c] = kind
        return kinds

    def uses_only_native_compute(self) -> bool:
        """True when every function the schedule names is one the engine runs itself: the cached defaults or a native variant (the HIP path end to end)."""
        return not self.custom_component_mask().any()
