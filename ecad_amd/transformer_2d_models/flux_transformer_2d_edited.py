"""``FluxTransformer2DEdited``: host-side facade over the native FLUX engine.

Keeps the reference class's surface (``ecad/transformer_2d_models/flux_transformer_2d_edited.py:35-326``): constructor
keywords, ``from_pretrained(..., dit_scheduler=, cache_schedule=)``, ``forward(hidden_states, encoder_hidden_states,
pooled_projections, timestep, img_ids, txt_ids, guidance, joint_attention_kwargs, return_dict)``, ``reset_cache()``,
``.config``, ``.dtype``, ``.device``.  No torch modules inside: weights live in the engine (fused / stacked), ``forward`` is one
``ecad_flux_step`` at schedule row ``cache_schedule.curr_step``.  No CPU fallback.
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Optional

import torch

from ecad_amd.flux_engine import EcadFluxEngine
from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule

KNOWN_CONFIGS: dict[str, dict[str, Any]] = {
    "black-forest-labs/FLUX.1-dev": dict(guidance_embeds=True),
    "black-forest-labs/FLUX.1-schnell": dict(guidance_embeds=False),
}


def state_dict_shapes(cfg) -> dict[str, tuple[int, ...]]:
    """diffusers ``FluxTransformer2DModel`` state-dict keys and shapes [ext]."""
    d, dh, m = cfg.num_attention_heads * cfg.attention_head_dim, cfg.attention_head_dim, 4
    s: dict[str, tuple[int, ...]] = {}

    def lin(name, out, inp):
        s[name + ".weight"] = (out, inp)
        s[name + ".bias"] = (out,)

    lin("x_embedder", d, cfg.in_channels)
    lin("context_embedder", d, cfg.joint_attention_dim)
    embedders = [("timestep_embedder", 256), ("text_embedder", cfg.pooled_projection_dim)]
    if cfg.guidance_embeds:
        embedders.insert(1, ("guidance_embedder", 256))
    for e, k in embedders:
        lin(f"time_text_embed.{e}.linear_1", d, k)
        lin(f"time_text_embed.{e}.linear_2", d, d)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        lin(b + "norm1.linear", 6 * d, d)
        lin(b + "norm1_context.linear", 6 * d, d)
        for p in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(b + "attn." + p, d, d)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[b + f"attn.{n}.weight"] = (dh,)
        for ff in ("ff", "ff_context"):
            lin(b + ff + ".net.0.proj", m * d, d)
            lin(b + ff + ".net.2", d, m * d)
    for i in range(cfg.num_single_layers):
        b = f"single_transformer_blocks.{i}."
        lin(b + "norm.linear", 3 * d, d)
        lin(b + "proj_mlp", m * d, d)
        lin(b + "proj_out", d, d + m * d)
        for p in ("to_q", "to_k", "to_v"):
            lin(b + "attn." + p, d, d)
        for n in ("norm_q", "norm_k"):
            s[b + f"attn.{n}.weight"] = (dh,)
    lin("norm_out.linear", 2 * d, d)
    lin("proj_out", cfg.in_channels, d)
    return s


def synthetic_state_dict(cfg, seed: int = 0, std: float = 0.02, bias_std: float = 0.02, device: str = "cpu", dtype: torch.dtype = torch.float32):
    """Seeded random-init weights with the checkpoint's shapes (no network => no real weights).  On ``device='cpu'`` the
    recipe and generator order equal the oracle's ``make_weights`` (both sides see identical tensors); the 12 B-parameter
    full model is generated per tensor on the GPU instead (``device='cuda'``), which only the benchmark needs."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in state_dict_shapes(cfg).items():
        r = torch.randn(shape, generator=g, device=device)
        if ".norm_" in name:
            out[name] = (1.0 + 0.1 * r).to(dtype)
        elif name.endswith("bias"):
            out[name] = (r * bias_std).to(dtype)
        else:
            out[name] = (r * std).to(dtype)
    return out


class FluxTransformer2DEdited:
    def __init__(
        self,
        patch_size: int = 1,
        in_channels: int = 64,
        num_layers: int = 19,
        num_single_layers: int = 38,
        attention_head_dim: int = 128,
        num_attention_heads: int = 24,
        joint_attention_dim: int = 4096,
        pooled_projection_dim: int = 768,
        guidance_embeds: bool = False,
        axes_dims_rope: list[int] = [16, 56, 56],
        dit_scheduler=None,
        cache_schedule: FluxCacheSchedule | None = None,
        # engine sizing (not in the reference)
        max_batch: int = 2,
        max_image_tokens: int = 4096,
        max_text_tokens: int = 512,
        torch_dtype: torch.dtype = torch.bfloat16,
        device: int | None = None,
    ):
        if patch_size != 1:
            raise ValueError("FLUX uses patch_size 1 on 2x2-packed latents")
        self.config = SimpleNamespace(
            patch_size=patch_size, in_channels=in_channels, num_layers=num_layers, num_single_layers=num_single_layers,
            attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
            pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds, axes_dims_rope=list(axes_dims_rope),
        )
        self.inner_dim = num_attention_heads * attention_head_dim
        self.dtype = torch_dtype
        dev = torch.cuda.current_device() if device is None else device
        self.device = torch.device("cuda", dev)
        self.engine = EcadFluxEngine(
            num_layers=num_layers, num_single_layers=num_single_layers, num_heads=num_attention_heads, head_dim=attention_head_dim,
            in_channels=in_channels, joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim,
            guidance_embeds=guidance_embeds, axes_dims_rope=tuple(axes_dims_rope), max_batch=max_batch, max_image_tokens=max_image_tokens,
            max_text_tokens=max_text_tokens, dtype=torch_dtype, device=dev,
        )
        self.dit_scheduler = dit_scheduler
        self.cache_schedule: FluxCacheSchedule | None = None
        self._begun_key = None
        if cache_schedule is not None and dit_scheduler is not None:
            self._post_init(dit_scheduler, cache_schedule)

    # -- reference: _post_init (:69-108) ------------------------------------------------------------
    def _post_init(self, dit_scheduler, cache_schedule: FluxCacheSchedule | None, skip_transformer_block_init: bool = False):
        if dit_scheduler is None:
            raise ValueError("A DiTScheduler object must be provided.")
        if cache_schedule is None:
            raise ValueError("A cache schedule is required: the native engine always runs the cached block path.")
        if cache_schedule.num_blocks != self.config.num_layers or cache_schedule.num_single_blocks != self.config.num_single_layers:
            raise ValueError(
                f"schedule has {cache_schedule.num_blocks}+{cache_schedule.num_single_blocks} blocks, model has {self.config.num_layers}+{self.config.num_single_layers}")
        self.dit_scheduler = dit_scheduler
        self.cache_schedule = cache_schedule
        self.engine.set_schedule(*cache_schedule.to_device_bits())
        self._begun_key = None

    def load_state_dict(self, state: dict[str, torch.Tensor], strict: bool = True):
        self.engine.load_state_dict(state)
        return self

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: Optional[str | os.PathLike], dit_scheduler=None, cache_schedule: FluxCacheSchedule | None = None,
                        skip_transformer_block_init: bool = False, subfolder: str | None = None, torch_dtype: torch.dtype = torch.bfloat16,
                        use_safetensors: bool = True, synthetic_seed: int | None = None, **kwargs) -> "FluxTransformer2DEdited":
        """Local checkpoint directory (config.json + *.safetensors) or a known model name (seeded synthetic weights + warning)."""
        path = Path(str(pretrained_model_name_or_path))
        if subfolder and (path / subfolder).exists():
            path = path / subfolder
        state = None
        cfg_kw: dict[str, Any] = {}
        if (path / "config.json").exists():
            with open(path / "config.json") as f:
                cfg_kw = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
            from safetensors.torch import load_file

            state = {}
            for st_file in sorted(path.glob("*.safetensors")):
                state.update(load_file(str(st_file)))
        elif str(pretrained_model_name_or_path) in KNOWN_CONFIGS:
            cfg_kw = dict(KNOWN_CONFIGS[str(pretrained_model_name_or_path)])
        else:
            raise ValueError(f"{pretrained_model_name_or_path!r}: neither a local checkpoint directory nor a known FLUX model name")
        known = set(cls.__init__.__code__.co_varnames)
        cfg_kw = {k: v for k, v in cfg_kw.items() if k in known}
        cfg_kw.update({k: v for k, v in kwargs.items() if k in known})
        model = cls(**cfg_kw, torch_dtype=torch_dtype)
        if state is None:
            seed = synthetic_seed if synthetic_seed is not None else int(os.environ.get("ECAD_SYNTHETIC_SEED", "0"))
            print(f"WARNING: no checkpoint for {pretrained_model_name_or_path!r} on disk; using seeded synthetic weights (seed {seed}).")
            big = model.inner_dim * model.config.num_single_layers > 20000
            state = synthetic_state_dict(model.config, seed, device="cuda" if big else "cpu", dtype=torch_dtype if big else torch.float32)
        model.load_state_dict(state)
        model._post_init(dit_scheduler, cache_schedule, skip_transformer_block_init)
        return model

    # -- reference: reset_cache (:188-193) -------------------------------------------------------------
    def reset_cache(self) -> None:
        self.engine.reset_cache()
        self._begun_key = None

    def to(self, *args, **kwargs):
        return self

    def eval(self):
        return self

    # -- reference: forward (:216-326) ------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None, pooled_projections: torch.Tensor = None,
                timestep: torch.Tensor = None, img_ids: torch.Tensor = None, txt_ids: torch.Tensor = None, guidance: torch.Tensor = None,
                joint_attention_kwargs: Optional[dict[str, Any]] = None, return_dict: bool = True):
        if self.cache_schedule is None:
            raise ValueError("no cache schedule attached (call _post_init / from_pretrained with cache_schedule=)")
        if encoder_hidden_states is None or pooled_projections is None or timestep is None or img_ids is None:
            raise ValueError("encoder_hidden_states, pooled_projections, timestep and img_ids are required")
        if self.config.guidance_embeds and guidance is None:
            raise ValueError("guidance is required for a guidance-distilled checkpoint")
        if txt_ids is not None and bool((txt_ids != 0).any()):
            raise ValueError("non-zero txt_ids are not supported (FluxPipeline passes zeros)")
        ids = img_ids[0] if img_ids.ndim == 3 else img_ids
        gh, gw = int(ids[:, 1].max().item()) + 1, int(ids[:, 2].max().item()) + 1
        if gh * gw != hidden_states.shape[1]:
            raise ValueError("img_ids do not describe a dense (row, col) grid over the packed latents")
        # the values the model sees after its `* 1000` in the model dtype (:270-274); one timestep per batch
        t = float((timestep.to(self.dtype) * 1000).float().flatten()[0])
        g = float((guidance.to(self.dtype) * 1000).float().flatten()[0]) if guidance is not None else 0.0
        step = self.cache_schedule.curr_step
        key = (encoder_hidden_states.data_ptr(), tuple(encoder_hidden_states.shape), tuple(hidden_states.shape))
        if step == 0 or self._begun_key != key:
            self.engine.begin_image_batch(encoder_hidden_states.contiguous(), pooled_projections.contiguous(), gh, gw)
            self._begun_key = key
        out = self.engine.step(step, hidden_states.contiguous(), t, g)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)

    __call__ = forward
