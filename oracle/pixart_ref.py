"""Oracle: fp32 PyTorch restatement of the PixArt-alpha cached transformer.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Never imported by ``ecad_amd``.

Data flow follows
* ``ecad/transformer_2d_models/pixart_transformer_2d_edited.py:160-376``
  (mask->bias :255-291, _process_input :293-330, _create_output :332-376)
* ``ecad/transformer_blocks/cached_transformer_block.py:167-391``
  (ada_norm_single branches :207-215,243-246,264-267,306-310,317-320; cache
  gate :326-391)
* default DiT graph == ``for b in range(28)`` (``ecad/graph/pixart_builder.py:199-204``).

Layer internals are diffusers==0.30.3 [ext] restated from its published source
(not importable here => "restated, not verified against diffusers"):
LayerNorm(eps=1e-6, no affine); AdaLayerNormSingle; PatchEmbed + 2-D sincos table;
PixArtAlphaTextProjection (GELU tanh); Attention/AttnProcessor2_0 (scale
1/sqrt(d_head), additive -10000 mask bias broadcast over heads and queries);
FeedForward("gelu-approximate").

Weights use the diffusers state-dict key names so a real checkpoint would drop in.
``round_dtype`` (optional) rounds activations to fp16/bf16 at the points where the
reference's half-precision modules materialise a tensor, to separate rounding
noise from algorithmic differences in the parity tests.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class PixArtConfig:
    num_attention_heads: int = 16
    attention_head_dim: int = 72
    in_channels: int = 4
    out_channels: int = 8
    num_layers: int = 28
    sample_size: int = 32  # latent side; 32 for the 256x256 checkpoints
    patch_size: int = 2
    caption_channels: int = 4096
    norm_eps: float = 1e-6
    interpolation_scale: float | None = None
    use_additional_conditions: bool = False  # True for PixArt-XL-2-1024-MS (sample_size 128)

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def weight_shapes(cfg: PixArtConfig) -> dict[str, tuple[int, ...]]:
    d, p = cfg.dim, cfg.patch_size
    s: dict[str, tuple[int, ...]] = {
        "pos_embed.proj.weight": (d, cfg.in_channels, p, p),
        "pos_embed.proj.bias": (d,),
        "adaln_single.emb.timestep_embedder.linear_1.weight": (d, 256),
        "adaln_single.emb.timestep_embedder.linear_1.bias": (d,),
        "adaln_single.emb.timestep_embedder.linear_2.weight": (d, d),
        "adaln_single.emb.timestep_embedder.linear_2.bias": (d,),
        "adaln_single.linear.weight": (6 * d, d),
        "adaln_single.linear.bias": (6 * d,),
        "caption_projection.linear_1.weight": (d, cfg.caption_channels),
        "caption_projection.linear_1.bias": (d,),
        "caption_projection.linear_2.weight": (d, d),
        "caption_projection.linear_2.bias": (d,),
        "scale_shift_table": (2, d),
        "proj_out.weight": (p * p * cfg.out_channels, d),
        "proj_out.bias": (p * p * cfg.out_channels,),
    }
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        s[b + "scale_shift_table"] = (6, d)
        for a in ("attn1", "attn2"):
            for proj in ("to_q", "to_k", "to_v", "to_out.0"):
                s[b + f"{a}.{proj}.weight"] = (d, d)
                s[b + f"{a}.{proj}.bias"] = (d,)
        s[b + "ff.net.0.proj.weight"] = (4 * d, d)
        s[b + "ff.net.0.proj.bias"] = (4 * d,)
        s[b + "ff.net.2.weight"] = (d, 4 * d)
        s[b + "ff.net.2.bias"] = (d,)
    if getattr(cfg, "use_additional_conditions", False):  # 1024-MS alpha: PixArtAlphaCombinedTimestepSizeEmbeddings [ext], size_emb_dim = D // 3
        e = d // 3
        for emb in ("resolution_embedder", "aspect_ratio_embedder"):
            q = f"adaln_single.emb.{emb}."
            s[q + "linear_1.weight"], s[q + "linear_1.bias"] = (e, 256), (e,)
            s[q + "linear_2.weight"], s[q + "linear_2.bias"] = (e, e), (e,)
    return s


def make_weights(cfg: PixArtConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.02) -> dict[str, torch.Tensor]:
    """Seeded synthetic weights with the PixArt-XL-2 shapes.

    Linear W ~ N(0, std^2); biases ~ N(0, bias_std^2) (non-zero so that a dropped
    bias shows up in parity tests); scale_shift_table ~ N(0,1)/sqrt(D) as
    diffusers initialises it [ext].
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    w: dict[str, torch.Tensor] = {}
    for name, shape in weight_shapes(cfg).items():
        if name.endswith("scale_shift_table"):
            w[name] = torch.randn(shape, generator=g) / math.sqrt(cfg.dim)
        elif name.endswith("bias"):
            w[name] = torch.randn(shape, generator=g) * bias_std
        else:
            w[name] = torch.randn(shape, generator=g) * std
    return w


# --------------------------------------------------------------------------
# embeddings  (diffusers.models.embeddings [ext])
# --------------------------------------------------------------------------
def sincos_pos_embed_2d(embed_dim: int, grid_size, base_size: int, interpolation_scale: float = 1.0) -> torch.Tensor:
    """get_2d_sincos_pos_embed [ext]: fp32 [grid_h*grid_w, embed_dim]; ``grid_size`` an int or (grid_h, grid_w) -- PatchEmbed.forward [ext]
    rebuilds the table with the actual (height, width) grid when the latent is not the checkpoint's sample size, each axis scaled by its own
    ``grid / base_size``.

    First half of the channels encodes the column index, second half the row
    index (the meshgrid(w, h) quirk of the original MAE code).
    """
    gh, gw = (grid_size, grid_size) if isinstance(grid_size, int) else grid_size
    coords_h = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    coords_w = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale

    def one_d(dim: int, pos: np.ndarray) -> np.ndarray:
        omega = np.arange(dim // 2, dtype=np.float64)
        omega /= dim / 2.0
        omega = 1.0 / 10000**omega
        out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    col = np.tile(coords_w[None, :], (gh, 1))  # [i, j] -> coords_w[j]
    row = np.tile(coords_h[:, None], (1, gw))  # [i, j] -> coords_h[i]
    emb = np.concatenate([one_d(embed_dim // 2, col), one_d(embed_dim // 2, row)], axis=1)
    return torch.from_numpy(emb).float()


def timestep_sinusoid(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0) [ext] -> [B, 256] = [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    arg = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
@dataclass
class CacheState:
    """Per-block caches: cached_attn1/attn2/ff_output (cached_transformer_block.py:116-123)."""

    entries: dict[tuple[int, str], torch.Tensor] = field(default_factory=dict)
    executed_log: list[tuple[int, int, str]] = field(default_factory=list)

    def reset(self) -> None:  # reset_cache (:120-123)
        self.entries.clear()


class PixArtOracle:
    def __init__(self, cfg: PixArtConfig, weights: dict[str, torch.Tensor], round_dtype: torch.dtype | None = None):
        self.cfg = cfg
        self.w = {k: v.float() for k, v in weights.items()}
        if round_dtype is not None:  # the reference holds fp16 weights
            self.w = {k: v.to(round_dtype).float() for k, v in self.w.items()}
        self.round_dtype = round_dtype
        grid = cfg.sample_size // cfg.patch_size
        interp = cfg.interpolation_scale if cfg.interpolation_scale is not None else max(cfg.sample_size // 64, 1)
        self.pos_embed = sincos_pos_embed_2d(cfg.dim, grid, base_size=cfg.sample_size // cfg.patch_size, interpolation_scale=interp)
        self._pos_cache = {(grid, grid): self.pos_embed}
        self._interp = interp
        self.cache = CacheState()

    # -- helpers ---------------------------------------------------------
    def q(self, x: torch.Tensor) -> torch.Tensor:
        return x if self.round_dtype is None else x.to(self.round_dtype).float()

    def lin(self, x: torch.Tensor, name: str) -> torch.Tensor:
        return self.q(F.linear(x, self.w[name + ".weight"], self.w[name + ".bias"]))

    def layer_norm(self, x: torch.Tensor) -> torch.Tensor:
        return self.q(F.layer_norm(x, (x.shape[-1],), eps=self.cfg.norm_eps))

    # -- _process_input pieces (pixart_transformer_2d_edited.py:293-330) ---
    def patch_embed(self, latents: torch.Tensor) -> torch.Tensor:
        x = F.conv2d(latents.float(), self.w["pos_embed.proj.weight"], self.w["pos_embed.proj.bias"], stride=self.cfg.patch_size)
        gh, gw = x.shape[-2], x.shape[-1]
        x = x.flatten(2).transpose(1, 2)  # [B, N, D], token = row*W + col
        if (gh, gw) not in self._pos_cache:  # PatchEmbed.forward [ext]: table rebuilt for the actual grid
            self._pos_cache[(gh, gw)] = sincos_pos_embed_2d(self.cfg.dim, (gh, gw), base_size=self.cfg.sample_size // self.cfg.patch_size, interpolation_scale=self._interp)
        return self.q(x + self._pos_cache[(gh, gw)][None])

    def _timestep_mlp(self, values: torch.Tensor, prefix: str) -> torch.Tensor:
        """TimestepEmbedding(256 -> width) on Timesteps(256) of ``values`` [ext]."""
        e = self.q(timestep_sinusoid(values))
        e = self.q(F.silu(self.lin(e, prefix + "linear_1")))
        return self.lin(e, prefix + "linear_2")

    def adaln_single(self, timestep: torch.Tensor, added_cond: dict | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        embedded = self._timestep_mlp(timestep, "adaln_single.emb.timestep_embedder.")
        if self.cfg.use_additional_conditions:
            # PixArtAlphaCombinedTimestepSizeEmbeddings.forward [ext]: resolution [B, 2] is flattened, embedded per scalar and reshaped to
            # [B, 2E]; aspect_ratio [B, 1] -> [B, E]; conditioning = timestep_emb + cat(resolution_emb, aspect_ratio_emb)
            if added_cond is None or added_cond.get("resolution") is None or added_cond.get("aspect_ratio") is None:
                raise ValueError("`added_cond_kwargs` cannot be None when using additional conditions for `adaln_single`.")  # reference :208-211
            b = timestep.shape[0]
            res = self._timestep_mlp(added_cond["resolution"].float().reshape(-1), "adaln_single.emb.resolution_embedder.").reshape(b, -1)
            ar = self._timestep_mlp(added_cond["aspect_ratio"].float().reshape(-1), "adaln_single.emb.aspect_ratio_embedder.").reshape(b, -1)
            embedded = self.q(embedded + torch.cat([res, ar], dim=1))
        mod = self.lin(self.q(F.silu(embedded)), "adaln_single.linear")
        return mod, embedded  # [B, 6D], [B, D]

    def caption_projection(self, enc: torch.Tensor) -> torch.Tensor:
        h = self.lin(enc.float(), "caption_projection.linear_1")
        h = self.q(F.gelu(h, approximate="tanh"))
        return self.lin(h, "caption_projection.linear_2")

    # -- sub-layers (diffusers Attention / FeedForward [ext]) ---------------
    def attention(self, prefix: str, x: torch.Tensor, ctx: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
        B, Nq, D = x.shape
        H, dh = self.cfg.num_attention_heads, self.cfg.attention_head_dim
        qh = self.lin(x, prefix + ".to_q").view(B, Nq, H, dh).transpose(1, 2)
        kh = self.lin(ctx, prefix + ".to_k").view(B, -1, H, dh).transpose(1, 2)
        vh = self.lin(ctx, prefix + ".to_v").view(B, -1, H, dh).transpose(1, 2)
        scores = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
        if bias is not None:
            scores = scores + bias[:, None, :, :]  # [B,1,1,T]
        o = self.q(torch.softmax(scores, dim=-1) @ vh)
        o = o.transpose(1, 2).reshape(B, Nq, D)
        return self.lin(o, prefix + ".to_out.0")

    def feed_forward(self, prefix: str, x: torch.Tensor) -> torch.Tensor:
        h = self.lin(x, prefix + ".net.0.proj")
        h = self.q(F.gelu(h, approximate="tanh"))
        return self.lin(h, prefix + ".net.2")

    # -- the cache gate (cached_transformer_block.py:326-391) ----------------
    def _gated(self, step: int, block: int, comp: str, recompute: bool, fn):
        # user compute plug-in for this (block, component) (cached_transformer_block.py:125-165): called with the default gated evaluation and
        # the bare sub-layer, e.g. lambda cached, fresh: 2 * cached().  Set by tests through ``self.plugins[(block, comp)]``.
        plug = getattr(self, "plugins", {}).get((block, comp))
        if plug is not None:
            return self.q(plug(lambda: self._gated_default(step, block, comp, recompute, fn), fn))
        return self._gated_default(step, block, comp, recompute, fn)

    def _gated_default(self, step: int, block: int, comp: str, recompute: bool, fn):
        key = (block, comp)
        if recompute or key not in self.cache.entries:
            out = fn()
            self.cache.executed_log.append((step, block, comp))
        else:
            out = self.cache.entries[key]
        self.cache.entries[key] = out
        return out

    def block(self, i: int, step: int, flags, h, enc, enc_bias, mod, self_bias=None):
        """CachedTransformerBlock.forward, ada_norm_single (:167-324)."""
        b = f"transformer_blocks.{i}"
        B = h.shape[0]
        tbl = self.w[b + ".scale_shift_table"][None] + mod.reshape(B, 6, -1)
        tbl = self.q(tbl)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = tbl.chunk(6, dim=1)

        nh = self.q(self.layer_norm(h) * (1 + scale_msa) + shift_msa)
        a1 = self._gated(step, i, "attn1", bool(flags[0]), lambda: self.attention(b + ".attn1", nh, nh, self_bias))
        h = self.q(self.q(gate_msa * a1) + h)

        a2 = self._gated(step, i, "attn2", bool(flags[1]), lambda: self.attention(b + ".attn2", h, enc, enc_bias))
        h = self.q(a2 + h)

        nh = self.q(self.layer_norm(h) * (1 + scale_mlp) + shift_mlp)
        ff = self._gated(step, i, "ff", bool(flags[2]), lambda: self.feed_forward(b + ".ff", nh))
        h = self.q(self.q(gate_mlp * ff) + h)
        return h

    # -- forward (pixart_transformer_2d_edited.py:160-253) -------------------
    def _mask_to_bias(self, mask):
        """_create_attention_mask (:255-291): a 2-D mask (1 keep / 0 discard) becomes the additive bias (1 - m) * -10000 with a singleton query dimension;
        a 3-D tensor is taken to be such a bias already and passes through unchanged."""
        if mask is None:
            return None
        if mask.ndim == 2:
            return self.q(((1 - mask.float()) * -10000.0)[:, None, :])
        return mask.float()

    def forward(self, latents, enc, enc_mask, timestep, step: int, step_flags, added_cond: dict | None = None, attention_mask=None) -> torch.Tensor:
        """One noise prediction.  ``step_flags``: [blocks, 3] bools for this step.  ``enc_mask`` / ``attention_mask``: [B, keys] masks or [B, 1, keys] biases."""
        cfg = self.cfg
        enc_bias = self._mask_to_bias(enc_mask)
        self_bias = self._mask_to_bias(attention_mask)
        h = self.patch_embed(self.q(latents.float()))
        mod, embedded = self.adaln_single(timestep, added_cond)
        ctx = self.caption_projection(self.q(enc.float()))
        for i in range(cfg.num_layers):
            h = self.block(i, step, step_flags[i], h, ctx, enc_bias, mod, self_bias)
        return self.create_output(h, embedded, latents.shape[-2] // cfg.patch_size, latents.shape[-1] // cfg.patch_size)

    def create_output(self, h, embedded, gh: int | None = None, gw: int | None = None):  # :332-376; (height, width) of the token grid from :301-304
        cfg = self.cfg
        tbl = self.w["scale_shift_table"][None] + embedded[:, None]
        shift, scale = tbl.chunk(2, dim=1)
        h = self.q(self.layer_norm(h) * (1 + scale) + shift)
        h = self.lin(h, "proj_out")
        B = h.shape[0]
        p, c = cfg.patch_size, cfg.out_channels
        gh = gh if gh is not None else cfg.sample_size // p
        gw = gw if gw is not None else cfg.sample_size // p
        h = h.reshape(B, gh, gw, p, p, c)
        h = torch.einsum("nhwpqc->nchpwq", h)
        return h.reshape(B, c, gh * p, gw * p)

    def reset_cache(self) -> None:
        self.cache.reset()
