"""Key sets of the dictionaries that cross the plug-in surface (schedule JSON entries, prompt-embedding ``.pt`` files, the
``config`` block of a schedule file).  Names and keys equal the reference's ``ecad/types.py`` so files are interchangeable;
declared with the functional ``TypedDict`` form, one table per family."""
from __future__ import annotations

from typing import Any, TypedDict

import torch

try:  # py3.11+
    from typing import NotRequired
except ImportError:  # pragma: no cover
    from typing_extensions import NotRequired

_T = torch.Tensor

# ---- prompt embeddings (what encode_and_save_prompts writes, what generate_images reads)
PromptEmbedding = TypedDict("PromptEmbedding", {})
PixArtPromptEmbedding = TypedDict(
    "PixArtPromptEmbedding",
    {"prompt_embeds": _T, "prompt_attention_mask": _T, "negative_prompt_embeds": _T, "negative_prompt_attention_mask": _T},
)
# text_ids is recomputed at generation time and may be None (reference :20-35)
FluxPromptEmbedding = TypedDict("FluxPromptEmbedding", {"prompt_embeds": _T, "pooled_prompt_embeds": _T, "text_ids": "torch.Tensor | None"})

# ---- the "config" block of a schedule JSON
PipelineConfig = TypedDict("PipelineConfig", {"name": str, "kwargs": dict[str, Any]})
ImageGeneratorConfig = TypedDict(
    "ImageGeneratorConfig",
    {
        "pipeline": NotRequired[PipelineConfig],
        "transformer_weights": NotRequired[str],
        "pipeline_weights": NotRequired[str],
        "image_generator": NotRequired[str],
        "height": NotRequired[int],
        "width": NotRequired[int],
        "guidance_scale": NotRequired[float],
    },
)

# ---- one block's entry inside schedule[step]
CustomFuncDict = TypedDict("CustomFuncDict", {"name": str, "kwargs": dict[str, Any]})
PixArtComponentScheduleDict = TypedDict(
    "PixArtComponentScheduleDict",
    {"attn1": bool, "attn2": bool, "ff": bool, "custom_compute_attn": NotRequired[CustomFuncDict], "custom_compute_ff": NotRequired[CustomFuncDict]},
)
FluxSingleComponentScheduleDict = TypedDict("FluxSingleComponentScheduleDict", {"single_attn": bool, "single_proj_mlp": bool, "single_proj_out": bool})
FluxFullComponentScheduleDict = TypedDict("FluxFullComponentScheduleDict", {"full_attn": bool, "full_ff": bool, "full_ff_context": bool})

# ---- step -> block name -> entry
PixArtBlockScheduleDict = dict[str, PixArtComponentScheduleDict]
PixArtCacheScheduleDict = dict[int, PixArtBlockScheduleDict]
FluxBlockScheduleDict = dict[str, Any]
FluxCacheScheduleDict = dict[int, FluxBlockScheduleDict]
CacheScheduleDict = dict[int, dict[str, Any]]
