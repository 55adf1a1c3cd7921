"""``CachedTransformerBlock``: the block a compute plug-in sees.

Mirror of the plug-in-facing part of ``ecad/transformer_blocks/cached_transformer_block.py`` (:116-165 attributes and dispatch,
:326-391 the two default functions).  The block's arithmetic lives in the HIP engine; this view exposes what the reference's
registered functions touch -- ``block_num``, ``cache_schedule``, the three ``cached_*_output`` attributes, ``reset_cache()``,
``compute_attn`` / ``compute_ff`` (name + kwargs looked up in the schedule row of the current step) -- and ``attn1`` / ``attn2`` /
``ff`` callables that run the engine's own sub-layers (``ecad_block_attn`` / ``ecad_block_ff``) on device tensors.

Blocks whose schedule only ever names the defaults never come through here: the engine resolves their gate statically.  A block
that names another function is executed as native sub-ops around these Python callables, and its cache is the Python attribute,
exactly as in the reference.
"""
from __future__ import annotations

from typing import Any

import torch


class _NativeSublayer:
    """``block.attn1`` / ``block.attn2`` / ``block.ff`` of the reference are torch modules; here a callable over the engine's kernels."""

    def __init__(self, block: "CachedTransformerBlock", kind: str):
        self.block, self.kind = block, kind

    def __call__(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor | None = None, attention_mask: torch.Tensor | None = None,
                 **cross_attention_kwargs: Any) -> torch.Tensor:
        b = self.block
        eng = b.transformer.engine
        if self.kind == "ff":
            return eng.block_ff(b.index, hidden_states)
        if self.kind == "attn1":
            if encoder_hidden_states is not None or attention_mask is not None:
                raise ValueError("attn1 of a PixArt block is plain self-attention (only_cross_attention False, attention_mask None)")
            return eng.block_attn(b.index, 1, hidden_states)
        # attn2 attends to the text of the batch that has begun: its K/V were projected once for all steps (step-invariant)
        ctx = b.transformer.batch_context()[0]
        if encoder_hidden_states is not None and encoder_hidden_states.data_ptr() != ctx.data_ptr():
            raise ValueError("native attn2 attends to the encoder_hidden_states the block was handed (the batch's projected text); another tensor cannot be substituted")
        return eng.block_attn(b.index, 2, hidden_states)


class CachedTransformerBlock:
    def __init__(self, transformer, block_num: int | str):
        self.transformer = transformer
        self.index = int(block_num)
        self.block_num = str(block_num)
        self.attn1, self.attn2, self.ff = _NativeSublayer(self, "attn1"), _NativeSublayer(self, "attn2"), _NativeSublayer(self, "ff")
        self.cached_attn1_output: torch.Tensor | None = None
        self.cached_attn2_output: torch.Tensor | None = None
        self.cached_ff_output: torch.Tensor | None = None
        self._chunk_size = None  # diffusers' chunked feed-forward is never enabled by the reference's pipelines

    @property
    def cache_schedule(self):
        return self.transformer.cache_schedule

    def reset_cache(self) -> None:  # :120-123
        self.cached_attn1_output = None
        self.cached_attn2_output = None
        self.cached_ff_output = None

    # -- dispatch (:125-165) --------------------------------------------------------------------
    def compute_attn(self, attn: str, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor | None, attention_mask: torch.Tensor | None,
                     **cross_attention_kwargs: Any) -> torch.Tensor:
        from ecad_amd.transformer_blocks.custom_attn_ff import ComputeAttnRegistry

        cfg = self.cache_schedule.get_custom_compute_attn(self.block_num)
        func = ComputeAttnRegistry.get(cfg.get("name", None), False)
        assert func is not None, f"Function {cfg.get('name')} not found in registry"
        return func(self, attn, hidden_states, encoder_hidden_states, attention_mask, **cfg.get("kwargs", {}), **cross_attention_kwargs)

    def compute_ff(self, norm_hidden_states: torch.Tensor) -> torch.Tensor:
        from ecad_amd.transformer_blocks.custom_attn_ff import ComputeFFRegistry

        cfg = self.cache_schedule.get_custom_compute_ff(self.block_num)
        func = ComputeFFRegistry.get(cfg.get("name", None), False)
        assert func is not None, f"Function {cfg.get('name')} not found in registry"
        return func(self, norm_hidden_states, **cfg.get("kwargs", {}))


# -- the two defaults (:326-391).  Engine-resolved for blocks that never name anything else; callable from a plug-in or for a block that
#    names a plug-in at some steps only.
def compute_attn_cached(self: CachedTransformerBlock, attn: str, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor | None,
                        attention_mask: torch.Tensor | None, **cross_attention_kwargs: Any) -> torch.Tensor:
    if attn not in ["attn1", "attn2"]:
        raise ValueError(f"Invalid attention type: {attn}. Must be attn1 or attn2")
    _require_block(self, "compute_attn_cached")
    recompute = self.cache_schedule.get_recompute(self.block_num, attn)
    cache_attr = f"cached_{attn}_output"
    no_cache = getattr(self, cache_attr) is None
    if not recompute and no_cache:
        print(f"WARNING: No cached {attn} found. Recomputing.")
    if recompute or no_cache:
        out = getattr(self, attn)(hidden_states, encoder_hidden_states, attention_mask, **cross_attention_kwargs)
    else:
        out = getattr(self, cache_attr)
    setattr(self, cache_attr, out)
    return out


def compute_ff_cached(self: CachedTransformerBlock, norm_hidden_states: torch.Tensor) -> torch.Tensor:
    _require_block(self, "compute_ff_cached")
    recompute = self.cache_schedule.get_recompute(self.block_num, "ff")
    no_cache = self.cached_ff_output is None
    if not recompute and no_cache:
        print("WARNING: No cached ff found. Recomputing.")
    out = self.ff(norm_hidden_states) if (recompute or no_cache) else self.cached_ff_output
    self.cached_ff_output = out
    return out


# -- native variants (ecad_set_native_variants): registered functions with a rule the engine also has as kernels.  In a schedule whose blocks name only native
#    functions nothing below is ever called -- the engine resolves the rule statically, per-step and inside the whole-loop hipGraph; the callables serve a block
#    that ALSO names a host-only plug-in (then Python drives all of that block's components) and other plug-ins that want to delegate to them.
def compute_attn_recompute(self: CachedTransformerBlock, attn: str, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor | None,
                           attention_mask: torch.Tensor | None, **cross_attention_kwargs: Any) -> torch.Tensor:
    """Ignore the schedule's flag: evaluate the sub-layer at this step and cache it ("never reuse here")."""
    if attn not in ["attn1", "attn2"]:
        raise ValueError(f"Invalid attention type: {attn}. Must be attn1 or attn2")
    _require_block(self, "compute_attn_recompute")
    out = getattr(self, attn)(hidden_states, encoder_hidden_states, attention_mask, **cross_attention_kwargs)
    setattr(self, f"cached_{attn}_output", out)
    return out


def compute_ff_recompute(self: CachedTransformerBlock, norm_hidden_states: torch.Tensor) -> torch.Tensor:
    _require_block(self, "compute_ff_recompute")
    out = self.ff(norm_hidden_states)
    self.cached_ff_output = out
    return out


def compute_attn_cfg_mean_cache(self: CachedTransformerBlock, attn: str, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor | None,
                                attention_mask: torch.Tensor | None, **cross_attention_kwargs: Any) -> torch.Tensor:
    """attn1: the cached default.  attn2: evaluated per the schedule's flag like the default, but what is CACHED is the mean of the unconditional and the text half
    of the CFG-doubled batch (in both halves) while the un-averaged output is returned -- the caching rule ``compute_attn_tgate`` applies at its gate step
    (``cached_transformer_block.py:443-452``), here at every evaluation so that later reuse steps add the CFG-averaged cross-attention."""
    if attn not in ["attn1", "attn2"]:
        raise ValueError(f"Invalid attention type: {attn}. Must be attn1 or attn2")
    _require_block(self, "compute_attn_cfg_mean_cache")
    if attn == "attn1":
        return compute_attn_cached(self, attn, hidden_states, encoder_hidden_states, attention_mask, **cross_attention_kwargs)
    recompute = self.cache_schedule.get_recompute(self.block_num, attn) or self.cached_attn2_output is None
    if not recompute:
        return self.cached_attn2_output
    out = self.attn2(hidden_states, encoder_hidden_states, attention_mask, **cross_attention_kwargs)
    if out.shape[0] % 2:
        raise ValueError("compute_attn_cfg_mean_cache needs a CFG-doubled batch (unconditional half, text half)")
    uncond, text = out.chunk(2)
    mean = ((uncond.float() + text.float()) * 0.5).to(out.dtype)
    self.cached_attn2_output = torch.cat([mean, mean])
    return out


compute_attn_recompute.native_variant = 1       # ECAD_NATIVE_RECOMPUTE
compute_ff_recompute.native_variant = 1
compute_attn_cfg_mean_cache.native_variant = 2  # ECAD_NATIVE_CFG_MEAN_CACHE (attn2; attn1 runs the default rule)


def _require_block(block, name: str) -> None:
    if not isinstance(block, CachedTransformerBlock):
        raise RuntimeError(f"{name} runs on the HIP engine (libecad_hip.so) through a CachedTransformerBlock of PixArtTransformer2DEdited; it has no CPU fallback.")


def _register_defaults() -> None:
    from ecad_amd.transformer_blocks.custom_attn_ff import ComputeAttnRegistry, ComputeFFRegistry

    ComputeAttnRegistry.register(compute_attn_cached)
    ComputeFFRegistry.register(compute_ff_cached)
    ComputeAttnRegistry.register(compute_attn_recompute)
    ComputeFFRegistry.register(compute_ff_recompute)
    ComputeAttnRegistry.register(compute_attn_cfg_mean_cache)
