"""Which (step, block, component) recomputes and which replays its cache: the host-side schedule object.

Keeps the surface of the reference's ``CacheSchedule`` (``ecad/schedulers/cache_scheduler/cache_schedule.py``): constructor keywords,
``curr_step`` / ``per_step_callback`` / ``reset_step`` (callbacks fire AFTER a step, so the current step is the last seen one plus one),
``get_recompute`` (``ValueError`` on an unknown component), and the JSON schema -- ``"000"``-padded step keys, string block keys, the
``cache_schedule`` / ``config`` / ``metrics`` top-level blocks.  The device engine consumes ``to_numpy()`` once per schedule
(``ecad_set_schedule``); the per-component dictionary lookup is no longer on the hot path.
"""
from __future__ import annotations

import json
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any

import numpy as np

_BODY_KEYS = ("num_blocks", "num_inference_steps", "name", "attributes")


def _or_empty(value):
    return {} if value is None else value


class CacheSchedule(ABC):
    def __init__(self, num_blocks: int, num_inference_steps: int, name: str, schedule: dict, top_level_config: dict[str, Any] | None = None,
                 attributes: dict[str, Any] | None = None, metrics: dict[str, Any] | None = None, **kwargs: Any) -> None:
        self.name = name
        self.num_blocks, self.num_inference_steps = int(num_blocks), int(num_inference_steps)
        self.top_level_config, self.attributes, self.metrics = _or_empty(top_level_config), _or_empty(attributes), _or_empty(metrics)
        # JSON gives "000" strings, generators give ints
        self.schedule = {int(step_key): blocks for step_key, blocks in schedule.items()}
        self._seen = -1  # index of the last step whose callback fired

    # ---- subclass contract
    @property
    @abstractmethod
    def components(self) -> list[str]: ...

    @abstractmethod
    def to_numpy(self, flatten: bool = False) -> np.ndarray: ...

    # ---- the step counter the pipeline callbacks drive
    @property
    def curr_step(self) -> int:
        return self._seen + 1

    def per_step_callback(self, step: int, timestep: Any = None, **kwargs: Any) -> None:
        self._seen = step

    def reset_step(self) -> None:
        self._seen = -1

    # kept for code that pokes the reference's private counter
    @property
    def _last_step(self) -> int:
        return self._seen

    @_last_step.setter
    def _last_step(self, value: int) -> None:
        self._seen = value

    def get_recompute(self, block_num: str, component: str) -> bool:
        if component not in self.components:
            raise ValueError(f"Invalid component {component}. Must be one of {self.components}.")
        return self.schedule[self.curr_step][str(block_num)][component]

    # ---- files
    def to_dict(self) -> dict[str, Any]:
        body = {key: getattr(self, key) for key in _BODY_KEYS}
        body["schedule"] = {format(step, "03"): blocks for step, blocks in self.schedule.items()}
        return {"cache_schedule": body, "config": self.top_level_config, "metrics": self.metrics}

    def to_json(self, file_path: Path) -> None:
        Path(file_path).write_text(json.dumps(self.to_dict(), indent=4))

    @classmethod
    def from_json(cls, file_path: Path) -> "CacheSchedule":
        data = json.loads(Path(file_path).read_text())
        # a file without "cache_schedule" raises KeyError: callers fall back to the default schedule (image_generator.py:118-124)
        return cls(**data["cache_schedule"], top_level_config=data.get("config"), metrics=data.get("metrics"))
