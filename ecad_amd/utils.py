"""Argument bundles the reference hands to its block-routing layer (``ecad/utils.py``): kept as plain records with the same field
names, ``fields()`` and ``to_dict()``, generated from one field table per model family."""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import torch


def _record(name: str, base, extra: list[tuple[str, Any]]):
    return dataclasses.make_dataclass(name, [(n, t) for n, t in extra], bases=(base,))


@dataclasses.dataclass
class ForwardArgs:
    hidden_states: torch.Tensor

    @classmethod
    def fields(cls) -> list[str]:
        return [f.name for f in dataclasses.fields(cls)]

    def to_dict(self) -> dict[str, Any]:
        return {k: getattr(self, k) for k in type(self).fields()}


_OT = Optional[torch.Tensor]
PixArtForwardArgs = _record(
    "PixArtForwardArgs", ForwardArgs,
    [("attention_mask", _OT), ("encoder_hidden_states", _OT), ("encoder_attention_mask", _OT), ("timestep", _OT), ("cross_attention_kwargs", Optional[dict[str, Any]])],
)
# encoder_hidden_states is consumed by the joint blocks only; the single-stream blocks see the concatenated sequence
FluxForwardArgs = _record("FluxForwardArgs", ForwardArgs, [("encoder_hidden_states", _OT), ("temb", _OT), ("image_rotary_emb", Any)])
