"""Cache schedule: which (step, block, component) recomputes and which replays its cache.

Host-side mirror of the reference's ``CacheSchedule``
(``ecad/schedulers/cache_scheduler/cache_schedule.py:18-112``): same constructor
keywords, JSON schema (``"000"``-padded step keys, string block keys), step counter
(``curr_step = last_step + 1``; callbacks fire AFTER a step) and ``ValueError`` on an
unknown component.  The device engine consumes ``to_numpy()`` once per schedule
(``ecad_set_schedule``); the per-component dict lookup is no longer on the hot path.
"""
from __future__ import annotations

import json
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any

import numpy as np


class CacheSchedule(ABC):
    def __init__(
        self,
        num_blocks: int,
        num_inference_steps: int,
        name: str,
        schedule: dict,
        top_level_config: dict[str, Any] | None = None,
        attributes: dict[str, Any] | None = None,
        metrics: dict[str, Any] | None = None,
        **kwargs: Any,
    ) -> None:
        self.num_blocks = int(num_blocks)
        self.num_inference_steps = int(num_inference_steps)
        self.name = name
        self.attributes = attributes if attributes is not None else {}
        self.metrics = metrics if metrics is not None else {}
        self.top_level_config = top_level_config if top_level_config is not None else {}
        # step keys arrive as "000" strings from JSON or ints from generators
        self.schedule = {int(step): per_block for step, per_block in schedule.items()}
        self._last_step = -1

    # -- component layout -------------------------------------------------
    @property
    @abstractmethod
    def components(self) -> list[str]: ...

    @abstractmethod
    def to_numpy(self, flatten: bool = False) -> np.ndarray: ...

    # -- step counter -------------------------------------------------------
    def reset_step(self) -> None:
        self._last_step = -1

    @property
    def curr_step(self) -> int:
        return self._last_step + 1

    def per_step_callback(self, step: int, timestep: Any = None, **kwargs: Any) -> None:
        self._last_step = step

    def get_recompute(self, block_num: str, component: str) -> bool:
        if component not in self.components:
            raise ValueError(f"Invalid component {component}. Must be one of {self.components}.")
        return self.schedule[self.curr_step][str(block_num)][component]

    # -- (de)serialisation ----------------------------------------------------
    def to_dict(self) -> dict[str, Any]:
        return {
            "cache_schedule": {
                "num_blocks": self.num_blocks,
                "num_inference_steps": self.num_inference_steps,
                "name": self.name,
                "attributes": self.attributes,
                "schedule": {f"{step:03}": per_block for step, per_block in self.schedule.items()},
            },
            "config": self.top_level_config,
            "metrics": self.metrics,
        }

    def to_json(self, file_path: Path) -> None:
        with Path(file_path).open("w") as f:
            json.dump(self.to_dict(), f, indent=4)

    @classmethod
    def from_json(cls, file_path: Path) -> "CacheSchedule":
        with Path(file_path).open("r") as f:
            data = json.load(f)
        # a file without "cache_schedule" raises KeyError, which callers use to
        # fall back to the default schedule (image_generator.py:118-124)
        body = data["cache_schedule"]
        return cls(**body, top_level_config=data.get("config"), metrics=data.get("metrics"))
