"""Oracle: cache-schedule semantics + closed-form MAC model (numpy, integer/bool).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Follows:
* bit layout ``(step, block, [attn1, attn2, ff])``:
  ``ecad/schedulers/cache_scheduler/pixart_cache_schedule.py:15-27``
* lookup ``schedule[curr_step][block][component]`` and ``curr_step = last+1``:
  ``ecad/schedulers/cache_scheduler/cache_schedule.py:58-73``
* gate ``recompute or cache is None``:
  ``ecad/transformer_blocks/cached_transformer_block.py:340-347,367-373``
* reset after the last step: ``ecad/image_generators/image_generator.py:193-202``
* MAC accounting (what calflops counts in ``ecad/benchmark/compute_macs.py:255-303``):
  nn.Linear / nn.Conv only, batch 2.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

COMPONENTS = ("attn1", "attn2", "ff")


def bits_from_schedule_json(path: str | Path) -> np.ndarray:
    """(steps, blocks, 3) bool array from a schedule JSON file."""
    with open(path) as f:
        data = json.load(f)
    cs = data["cache_schedule"]
    steps, blocks = cs["num_inference_steps"], cs["num_blocks"]
    bits = np.zeros((steps, blocks, 3), dtype=np.bool_)
    for s_key, per_block in cs["schedule"].items():
        for b_key, comp in per_block.items():
            for c, name in enumerate(COMPONENTS):
                bits[int(s_key), int(b_key), c] = bool(comp[name])
    return bits


def resolve_executed(bits: np.ndarray) -> np.ndarray:
    """Which components actually execute, given ``recompute or cache-empty``.

    A cache entry is empty until its component executed once since the last
    reset; the reset happens after the final step, so inside one image batch a
    component's cache is empty exactly until its first execution.
    """
    bits = np.asarray(bits, dtype=np.bool_)
    steps = bits.shape[0]
    executed = np.zeros_like(bits)
    have_cache = np.zeros(bits.shape[1:], dtype=np.bool_)
    for s in range(steps):
        executed[s] = bits[s] | ~have_cache
        have_cache |= executed[s]
    return executed


class StepCounter:
    """The two counters the callbacks drive (cache_schedule.py:58-66)."""

    def __init__(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        self._last_step = -1
        self.resets = 0

    @property
    def curr_step(self) -> int:
        return self._last_step + 1

    def per_step_callback(self, step: int) -> None:
        self._last_step = step
        # _reset_schedules_callback runs LAST (image_generator.py:153-159)
        if step >= self.num_inference_steps - 1:
            self._last_step = -1
            self.resets += 1


# --------------------------------------------------------------------------
# Closed-form MAC model
# --------------------------------------------------------------------------
def pixart_mac_model(
    executed: np.ndarray,
    tokens: int = 256,
    dim: int = 1152,
    text_tokens: int = 120,
    caption_channels: int = 4096,
    patch_in: int = 16,  # in_channels * patch**2
    patch_out: int = 32,  # out_channels * patch**2
    size_embedder_macs: int = 0,  # 737_280 for the 1024-MS alpha weights
    batch: int = 2,
) -> np.ndarray:
    """Per-step Linear/Conv MACs exactly as calflops reports them at batch 2."""
    n, d, t = tokens, dim, text_tokens
    cost = np.array(
        [
            4 * n * d * d,  # attn1: q,k,v,out
            2 * n * d * d + 2 * t * d * d,  # attn2: q,out on image; k,v on text
            8 * n * d * d,  # ff: d->4d->d
        ],
        dtype=np.int64,
    )
    overhead = (
        t * (caption_channels * d + d * d)  # caption projection
        + n * patch_in * d  # patch-embed conv
        + (256 * d + d * d + 6 * d * d)  # timestep MLP + adaLN-single linear
        + n * d * patch_out  # proj_out
        + size_embedder_macs
    )
    ex = np.asarray(executed, dtype=np.int64)
    per_step = overhead + (ex * cost[None, None, :]).sum(axis=(1, 2))
    return batch * per_step


def attention_core_macs(executed: np.ndarray, tokens=256, dim=1152, text_tokens=120, batch=2):
    """QK^T + PV MACs, which calflops does not count."""
    ex = np.asarray(executed, dtype=np.int64)
    self_c = 2 * tokens * tokens * dim
    cross_c = 2 * tokens * text_tokens * dim
    return batch * (ex[:, :, 0].sum() * self_c + ex[:, :, 1].sum() * cross_c)


# --------------------------------------------------------------------------
# FLUX
# --------------------------------------------------------------------------
FLUX_FULL = ("full_attn", "full_ff", "full_ff_context")
FLUX_SINGLE = ("single_attn", "single_proj_mlp", "single_proj_out")


def flux_bits_from_schedule_json(path: str | Path):
    """(bits_full [steps, num_blocks, 3], bits_single [steps, num_single_blocks, 3]) from a FLUX schedule JSON.

    Component order is the one of ``FluxCacheSchedule.components`` / ``to_numpy``
    (``ecad/schedulers/cache_scheduler/flux_cache_schedule.py:51-88``); block keys are ``"0".."18"`` and
    ``"single_0".."single_37"``.
    """
    with open(path) as f:
        data = json.load(f)
    cs = data["cache_schedule"]
    steps, nb, ns = cs["num_inference_steps"], cs["num_blocks"], cs["num_single_blocks"]
    full = np.zeros((steps, nb, 3), dtype=np.bool_)
    single = np.zeros((steps, ns, 3), dtype=np.bool_)
    for s_key, per_block in cs["schedule"].items():
        for b_key, comp in per_block.items():
            if b_key.startswith("single_"):
                for c, name in enumerate(FLUX_SINGLE):
                    single[int(s_key), int(b_key[7:]), c] = bool(comp[name])
            else:
                for c, name in enumerate(FLUX_FULL):
                    full[int(s_key), int(b_key), c] = bool(comp[name])
    return full, single


def flux_flatten(full: np.ndarray, single: np.ndarray) -> np.ndarray:
    """The flat vector ``FluxCacheSchedule.to_numpy`` returns: per step, all full-block bits then all single-block bits
    (``flux_cache_schedule.py:63-88``)."""
    steps = full.shape[0]
    return np.concatenate([full.reshape(steps, -1), single.reshape(steps, -1)], axis=1).reshape(-1)


def flux_unflatten(vec: np.ndarray, steps: int, num_blocks: int, num_single_blocks: int):
    v = np.asarray(vec, dtype=np.bool_).reshape(steps, (num_blocks + num_single_blocks) * 3)
    return v[:, : num_blocks * 3].reshape(steps, num_blocks, 3), v[:, num_blocks * 3 :].reshape(steps, num_single_blocks, 3)


def flux_mac_model(
    executed_full: np.ndarray,
    executed_single: np.ndarray,
    tokens: int,
    dim: int = 3072,
    text_tokens: int = 512,
    in_channels: int = 64,
    joint_attention_dim: int = 4096,
    pooled_dim: int = 768,
    batch: int = 2,
) -> np.ndarray:
    """Per-step Linear MACs exactly as calflops reports them at batch 2 (``ecad/benchmark/compute_macs.py``):
    pinned against every FLUX schedule JSON that carries ``metrics.by_inference_step`` (38 files)."""
    n, t, d = tokens, text_tokens, dim
    s = n + t
    nb, ns = executed_full.shape[1], executed_single.shape[1]
    cost_full = np.array([4 * s * d * d, 8 * n * d * d, 8 * t * d * d], dtype=np.int64)
    cost_single = np.array([3 * s * d * d, 4 * s * d * d, 5 * s * d * d], dtype=np.int64)
    overhead = (
        n * in_channels * d  # x_embedder
        + t * joint_attention_dim * d  # context_embedder
        + 2 * (256 * d + d * d)  # timestep + guidance embedders
        + (pooled_dim * d + d * d)  # pooled text embedder
        + nb * 12 * d * d  # norm1 + norm1_context adaLN linears (never cached)
        + ns * 3 * d * d  # single-block adaLN linears
        + 2 * d * d  # norm_out
        + n * d * in_channels  # proj_out
    )
    ef, es = np.asarray(executed_full, dtype=np.int64), np.asarray(executed_single, dtype=np.int64)
    per_step = overhead + (ef * cost_full[None, None, :]).sum(axis=(1, 2)) + (es * cost_single[None, None, :]).sum(axis=(1, 2))
    return batch * per_step


def flux_attention_core_macs(executed_full, executed_single, tokens: int, dim: int = 3072, text_tokens: int = 512, batch: int = 2):
    """QK^T + PV MACs of the joint attention (S = tokens + text_tokens), not counted by calflops."""
    s = tokens + text_tokens
    n_attn = int(np.asarray(executed_full)[:, :, 0].sum()) + int(np.asarray(executed_single)[:, :, 0].sum())
    return batch * n_attn * 2 * s * s * dim
