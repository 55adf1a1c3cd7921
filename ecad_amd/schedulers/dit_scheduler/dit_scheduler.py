"""Block-routing step counter.

The reference routes hidden states through a per-step torch.fx graph
(``ecad/schedulers/dit_scheduler/dit_scheduler.py:27-59``,
``pixart_dit_schedule.py:24-40``); every shipped schedule uses the default graph,
which is the plain ``for block in range(28)`` loop.  The native
engine hard-wires that loop, so this class keeps only what the boundary touches:
``curr_step``, ``per_step_callback``, ``reset_step`` and ``schedule.num_inference_steps``.
A JSON carrying a non-default ``"dit_schedule"`` is rejected.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any


class PixArtDiTSchedule:
    def __init__(self, num_blocks: int, num_inference_steps: int, name: str = "default", top_level_config=None, attributes=None, metrics=None, **kw):
        self.num_blocks = num_blocks
        self.num_inference_steps = num_inference_steps
        self.name = name
        self.top_level_config = top_level_config or {}
        self.attributes = attributes or {}
        self.metrics = metrics or {}

    @classmethod
    def from_json(cls, file_path: Path) -> "PixArtDiTSchedule":
        with Path(file_path).open() as f:
            data = json.load(f)
        body = data["dit_schedule"]  # KeyError => caller falls back to the default
        if not str(body.get("name", "default")).startswith("default"):
            raise ValueError(
                f"DiT schedule {body.get('name')!r}: only the default sequential block graph is supported by the native engine."
            )
        return cls(body["num_blocks"], body["num_inference_steps"], body.get("name", "default"), data.get("config"), body.get("attributes"), data.get("metrics"))


class DiTScheduler:
    def __init__(self, schedule: PixArtDiTSchedule, verbose: bool = False):
        self.schedule = schedule
        self.verbose = verbose
        self._last_step = -1

    @property
    def curr_step(self) -> int:
        return self._last_step + 1

    def reset_step(self) -> None:
        self._last_step = -1

    def per_step_callback(self, step: int, timestep: Any = None, **kwargs: Any) -> None:
        self._last_step = step


class FluxDiTSchedule(PixArtDiTSchedule):
    """Reference: ``ecad/schedulers/dit_scheduler/flux_dit_schedule.py:16-149``.  The reference's own forward already
    short-circuits the fx graph (``flux_transformer_2d_edited.py:191-214,288-293``), so only the step counter and
    ``num_single_blocks`` are kept."""

    def __init__(self, num_blocks: int, num_inference_steps: int, name: str = "default", top_level_config=None, attributes=None, metrics=None,
                 num_single_blocks: int = 38, **kw):
        super().__init__(num_blocks, num_inference_steps, name, top_level_config, attributes, metrics)
        self.num_single_blocks = num_single_blocks

    @classmethod
    def from_json(cls, file_path: Path) -> "FluxDiTSchedule":
        with Path(file_path).open() as f:
            data = json.load(f)
        body = data["dit_schedule"]  # KeyError => caller falls back to the default
        return cls(body["num_blocks"], body["num_inference_steps"], body.get("name", "default"), data.get("config"), body.get("attributes"), data.get("metrics"),
                   num_single_blocks=body["num_single_blocks"])
