"""TypedDicts shared by the host-side plug-in surface.

Same names and keys as the reference's ``ecad/types.py:6-95`` so schedule JSONs,
prompt-embedding ``.pt`` files and config blocks are interchangeable.
"""
from __future__ import annotations

from typing import Any, TypedDict

import torch

try:  # py3.11+
    from typing import NotRequired
except ImportError:  # pragma: no cover
    from typing_extensions import NotRequired


class PromptEmbedding(TypedDict):
    pass


class PixArtPromptEmbedding(PromptEmbedding):
    prompt_embeds: torch.Tensor
    prompt_attention_mask: torch.Tensor
    negative_prompt_embeds: torch.Tensor
    negative_prompt_attention_mask: torch.Tensor


class FluxPromptEmbedding(PromptEmbedding):
    """Reference: ecad/types.py:20-35 (text_ids is recomputed at generation time and may be None)."""

    prompt_embeds: torch.Tensor
    pooled_prompt_embeds: torch.Tensor
    text_ids: "torch.Tensor | None"


class PipelineConfig(TypedDict):
    name: str
    kwargs: dict[str, Any]


class ImageGeneratorConfig(TypedDict):
    pipeline: NotRequired[PipelineConfig]
    transformer_weights: NotRequired[str]
    pipeline_weights: NotRequired[str]
    image_generator: NotRequired[str]
    height: NotRequired[int]
    width: NotRequired[int]


class CustomFuncDict(TypedDict):
    name: str
    kwargs: dict[str, Any]


class PixArtComponentScheduleDict(TypedDict):
    attn1: bool
    attn2: bool
    ff: bool
    custom_compute_attn: NotRequired[CustomFuncDict]
    custom_compute_ff: NotRequired[CustomFuncDict]


class FluxSingleComponentScheduleDict(TypedDict):
    single_attn: bool
    single_proj_mlp: bool
    single_proj_out: bool


class FluxFullComponentScheduleDict(TypedDict):
    full_attn: bool
    full_ff: bool
    full_ff_context: bool


FluxBlockScheduleDict = dict[str, Any]
FluxCacheScheduleDict = dict[int, FluxBlockScheduleDict]
PixArtBlockScheduleDict = dict[str, PixArtComponentScheduleDict]
PixArtCacheScheduleDict = dict[int, PixArtBlockScheduleDict]
CacheScheduleDict = dict[int, dict[str, Any]]
