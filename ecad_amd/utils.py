"""Argument bundles handed to the block-routing layer (reference: ``ecad/utils.py:8-46``)."""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import torch


@dataclasses.dataclass
class ForwardArgs:
    hidden_states: torch.Tensor

    @classmethod
    def fields(cls) -> list[str]:
        return [f.name for f in dataclasses.fields(cls)]

    def to_dict(self) -> dict[str, Any]:
        return {name: getattr(self, name) for name in self.fields()}


@dataclasses.dataclass
class PixArtForwardArgs(ForwardArgs):
    attention_mask: Optional[torch.Tensor]
    encoder_hidden_states: Optional[torch.Tensor]
    encoder_attention_mask: Optional[torch.Tensor]
    timestep: Optional[torch.Tensor]
    cross_attention_kwargs: Optional[dict[str, Any]]
