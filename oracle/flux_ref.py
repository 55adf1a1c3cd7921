"""Oracle: fp32 PyTorch restatement of the FLUX.1 cached MMDiT transformer.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Never imported by ``ecad_amd``.

Data flow follows
* ``ecad/transformer_2d_models/flux_transformer_2d_edited.py:191-214`` (_dit_scheduler_forward_short_circuit:
  19 joint blocks, cat([text, image]), 38 single blocks) and ``:216-326`` (forward: x_embedder, time/guidance/pooled
  embedding, context_embedder, RoPE ids, blocks, drop the text rows, norm_out, proj_out)
* ``ecad/transformer_blocks/cached_flux_transformer_block.py:233-291`` (joint block: adaLN-Zero on both streams,
  gated joint attention, gated FF per stream; cache gate :161-231)
* ``ecad/transformer_blocks/cached_flux_transformer_block.py:103-131`` (single block: adaLN-Zero-single,
  proj_mlp -> GELU, attention, cat, gated proj_out; cache gate :50-101)
* component names / bit layout: ``ecad/schedulers/cache_scheduler/flux_cache_schedule.py:51-88``

Layer internals are diffusers==0.30.3 [ext] restated from its published source (not importable here =>
"restated, not verified against diffusers"): AdaLayerNormZero / AdaLayerNormZeroSingle / AdaLayerNormContinuous
(LayerNorm eps 1e-6, no affine), FluxAttnProcessor2_0 / FluxSingleAttnProcessor2_0 (per-head RMSNorm eps 1e-6 with
weight on q and k, text rows first in the joint sequence, RoPE on q and k, SDPA scale 1/sqrt(128)),
EmbedND / rope (theta 10000, axes (16, 56, 56)), CombinedTimestepGuidanceTextProjEmbeddings, FeedForward
("gelu-approximate"), FlowMatchEulerDiscreteScheduler with dynamic shifting, FluxPipeline latent packing.

Weights use the diffusers state-dict key names so a real checkpoint would drop in.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from .pixart_ref import timestep_sinusoid

FLUX_COMPONENTS_FULL = ("full_attn", "full_ff", "full_ff_context")
FLUX_COMPONENTS_SINGLE = ("single_attn", "single_proj_mlp", "single_proj_out")


@dataclass
class FluxConfig:
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    in_channels: int = 64
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: tuple[int, ...] = (16, 56, 56)
    mlp_ratio: int = 4
    norm_eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def weight_shapes(cfg: FluxConfig) -> dict[str, tuple[int, ...]]:
    d, dh, m = cfg.dim, cfg.attention_head_dim, cfg.mlp_ratio
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


def make_weights(cfg: FluxConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.02) -> dict[str, torch.Tensor]:
    """Seeded synthetic weights with the FLUX shapes.  Linear W ~ N(0, std^2) (the adaLN linears and the gate paths are
    therefore non-zero, unlike a zero-init checkpoint, so every data path shows up in the parity tests);
    RMSNorm weights ~ 1 + 0.1 N(0, 1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    w: dict[str, torch.Tensor] = {}
    for name, shape in weight_shapes(cfg).items():
        if ".norm_" in name:
            w[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            w[name] = torch.randn(shape, generator=g) * bias_std
        else:
            w[name] = torch.randn(shape, generator=g) * std
    return w


# --------------------------------------------------------------------------
# RoPE  (diffusers.models.embeddings.rope / EmbedND, attention_processor.apply_rope [ext])
# --------------------------------------------------------------------------
def latent_image_ids(grid_h: int, grid_w: int) -> torch.Tensor:
    """FluxPipeline._prepare_latent_image_ids [ext]: [grid_h*grid_w, 3] = (0, row, col)."""
    ids = torch.zeros(grid_h, grid_w, 3)
    ids[..., 1] += torch.arange(grid_h)[:, None]
    ids[..., 2] += torch.arange(grid_w)[None, :]
    return ids.reshape(grid_h * grid_w, 3)


def rope_table(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0) -> tuple[torch.Tensor, torch.Tensor]:
    """(cos, sin), each [S, sum(axes_dim) / 2] fp32, for position ids [S, 3]: pair p of axis a rotates by
    ids[:, a] * theta**(-2p / axes_dim[a]) (computed in float64 like the reference, stored fp32)."""
    cos, sin = [], []
    for a, dim in enumerate(axes_dim):
        scale = torch.arange(0, dim, 2, dtype=torch.float64) / dim
        omega = 1.0 / (theta**scale)
        out = ids[:, a].double()[:, None] * omega[None, :]
        cos.append(torch.cos(out))
        sin.append(torch.sin(out))
    return torch.cat(cos, dim=1).float(), torch.cat(sin, dim=1).float()


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B, H, S, dh]; consecutive pairs (x[2p], x[2p+1]) rotate: (c x0 - s x1, s x0 + c x1)."""
    xf = x.float().reshape(*x.shape[:-1], -1, 2)
    x0, x1 = xf[..., 0], xf[..., 1]
    c, s = cos[None, None], sin[None, None]
    return torch.stack([c * x0 - s * x1, s * x0 + c * x1], dim=-1).reshape(x.shape)


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
@dataclass
class FluxCacheState:
    entries: dict[tuple[str, str], object] = field(default_factory=dict)
    executed_log: list[tuple[int, str, str]] = field(default_factory=list)

    def reset(self) -> None:
        self.entries.clear()


class FluxOracle:
    def __init__(self, cfg: FluxConfig, weights: dict[str, torch.Tensor], round_dtype: torch.dtype | None = None):
        self.cfg = cfg
        self.w = {k: v.float() for k, v in weights.items()}
        if round_dtype is not None:  # the reference holds bf16 weights
            self.w = {k: v.to(round_dtype).float() for k, v in self.w.items()}
        self.round_dtype = round_dtype
        self.cache = FluxCacheState()

    # -- helpers ---------------------------------------------------------
    def q(self, x: torch.Tensor) -> torch.Tensor:
        return x if self.round_dtype is None else x.to(self.round_dtype).float()

    def lin(self, x: torch.Tensor, name: str) -> torch.Tensor:
        return self.q(F.linear(x, self.w[name + ".weight"], self.w[name + ".bias"]))

    clip_fp16 = True  # tests switch it off to show that an input really overflows

    def clip(self, x: torch.Tensor) -> torch.Tensor:
        """`if hidden_states.dtype == torch.float16: hidden_states = hidden_states.clip(-65504, 65504)` (cached_flux_transformer_block.py:127-128,288-289):
        the fp16 sum has overflowed to +-inf by then (self.q rounds to the storage type), the clip brings it back to the largest half."""
        return x.clamp(-65504.0, 65504.0) if (self.round_dtype == torch.float16 and self.clip_fp16) else x

    def layer_norm(self, x: torch.Tensor) -> torch.Tensor:
        return self.q(F.layer_norm(x, (x.shape[-1],), eps=self.cfg.norm_eps))

    def rms_norm(self, x: torch.Tensor, name: str) -> torch.Tensor:  # RMSNorm(dh, eps 1e-6, weight) [ext]
        var = x.pow(2).mean(-1, keepdim=True)
        return self.q(self.q(x * torch.rsqrt(var + self.cfg.norm_eps)) * self.w[name + ".weight"])

    def heads(self, x: torch.Tensor) -> torch.Tensor:
        B, S, _ = x.shape
        return x.view(B, S, self.cfg.num_attention_heads, self.cfg.attention_head_dim).transpose(1, 2)

    def sdpa(self, qh, kh, vh) -> torch.Tensor:
        scores = qh @ kh.transpose(-1, -2) / math.sqrt(self.cfg.attention_head_dim)
        o = self.q(torch.softmax(scores, dim=-1) @ vh)
        B, H, S, dh = o.shape
        return o.transpose(1, 2).reshape(B, S, H * dh)

    # -- embeddings (flux_transformer_2d_edited.py:268-282) -----------------
    def time_text_embed(self, timestep_x1000: torch.Tensor, guidance_x1000: torch.Tensor | None, pooled: torch.Tensor) -> torch.Tensor:
        """CombinedTimestep(Guidance)TextProjEmbeddings [ext]; inputs are the values the model sees AFTER its `* 1000`."""

        def mlp(x, p):
            return self.lin(self.q(F.silu(self.lin(x, p + ".linear_1"))), p + ".linear_2")

        t = mlp(self.q(timestep_sinusoid(timestep_x1000)), "time_text_embed.timestep_embedder")
        if self.cfg.guidance_embeds:
            t = self.q(t + mlp(self.q(timestep_sinusoid(guidance_x1000)), "time_text_embed.guidance_embedder"))
        return self.q(t + mlp(self.q(pooled.float()), "time_text_embed.text_embedder"))

    # -- the cache gate (cached_flux_transformer_block.py:50-101,161-231) ----
    def _gated(self, step: int, block: str, comp: str, recompute: bool, fn):
        key = (block, comp)
        if recompute or key not in self.cache.entries:
            out = fn()
            self.cache.executed_log.append((step, block, comp))
        else:
            out = self.cache.entries[key]
        self.cache.entries[key] = out
        return out

    # -- joint block (:233-291) -------------------------------------------
    def joint_attention(self, p: str, x, ctx, cos, sin):
        T = ctx.shape[1]
        qh = self.rms_norm(self.heads(self.lin(x, p + ".to_q")), p + ".norm_q")
        kh = self.rms_norm(self.heads(self.lin(x, p + ".to_k")), p + ".norm_k")
        vh = self.heads(self.lin(x, p + ".to_v"))
        cq = self.rms_norm(self.heads(self.lin(ctx, p + ".add_q_proj")), p + ".norm_added_q")
        ck = self.rms_norm(self.heads(self.lin(ctx, p + ".add_k_proj")), p + ".norm_added_k")
        cv = self.heads(self.lin(ctx, p + ".add_v_proj"))
        qh, kh, vh = torch.cat([cq, qh], dim=2), torch.cat([ck, kh], dim=2), torch.cat([cv, vh], dim=2)
        qh, kh = self.q(apply_rope(qh, cos, sin)), self.q(apply_rope(kh, cos, sin))
        o = self.sdpa(qh, kh, vh)
        return self.lin(o[:, T:], p + ".to_out.0"), self.lin(o[:, :T], p + ".to_add_out")

    def feed_forward(self, p: str, x):
        return self.lin(self.q(F.gelu(self.lin(x, p + ".net.0.proj"), approximate="tanh")), p + ".net.2")

    def joint_block(self, i: int, step: int, flags, h, ctx, temb_silu, cos, sin):
        b = f"transformer_blocks.{i}"
        m = self.lin(temb_silu, b + ".norm1.linear")  # AdaLayerNormZero [ext]
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = [t[:, None] for t in m.chunk(6, dim=1)]
        c = self.lin(temb_silu, b + ".norm1_context.linear")
        c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = [t[:, None] for t in c.chunk(6, dim=1)]
        nh = self.q(self.layer_norm(h) * (1 + scale_msa) + shift_msa)
        nc = self.q(self.layer_norm(ctx) * (1 + c_scale_msa) + c_shift_msa)

        a, ca = self._gated(step, str(i), "full_attn", bool(flags[0]), lambda: self.joint_attention(b + ".attn", nh, nc, cos, sin))
        h = self.q(h + self.q(gate_msa * a))
        nh = self.q(self.q(self.layer_norm(h) * (1 + scale_mlp)) + shift_mlp)
        ff = self._gated(step, str(i), "full_ff", bool(flags[1]), lambda: self.feed_forward(b + ".ff", nh))
        h = self.q(h + self.q(gate_mlp * ff))

        ctx = self.q(ctx + self.q(c_gate_msa * ca))
        nc = self.q(self.q(self.layer_norm(ctx) * (1 + c_scale_mlp)) + c_shift_mlp)
        cff = self._gated(step, str(i), "full_ff_context", bool(flags[2]), lambda: self.feed_forward(b + ".ff_context", nc))
        ctx = self.clip(self.q(ctx + self.q(c_gate_mlp * cff)))  # :285-289 (only the text stream is clipped in the joint block)
        return ctx, h

    # -- single block (:103-131) ------------------------------------------
    def single_attention(self, p: str, x, cos, sin):
        qh = self.rms_norm(self.heads(self.lin(x, p + ".to_q")), p + ".norm_q")
        kh = self.rms_norm(self.heads(self.lin(x, p + ".to_k")), p + ".norm_k")
        vh = self.heads(self.lin(x, p + ".to_v"))
        qh, kh = self.q(apply_rope(qh, cos, sin)), self.q(apply_rope(kh, cos, sin))
        return self.sdpa(qh, kh, vh)

    def single_block(self, i: int, step: int, flags, h, temb_silu, cos, sin):
        b = f"single_transformer_blocks.{i}"
        name = f"single_{i}"
        m = self.lin(temb_silu, b + ".norm.linear")  # AdaLayerNormZeroSingle [ext]
        shift, scale, gate = [t[:, None] for t in m.chunk(3, dim=1)]
        nh = self.q(self.layer_norm(h) * (1 + scale) + shift)
        mlp = self._gated(step, name, "single_proj_mlp", bool(flags[1]), lambda: self.lin(nh, b + ".proj_mlp"))
        mlp = self.q(F.gelu(mlp, approximate="tanh"))
        a = self._gated(step, name, "single_attn", bool(flags[0]), lambda: self.single_attention(b + ".attn", nh, cos, sin))
        cat = torch.cat([a, mlp], dim=2)
        out = self._gated(step, name, "single_proj_out", bool(flags[2]), lambda: self.lin(cat, b + ".proj_out"))
        return self.clip(self.q(h + self.q(gate * out)))  # :126-128

    # -- forward (flux_transformer_2d_edited.py:216-326) ---------------------
    def forward(self, latents, enc, pooled, timestep_x1000, guidance_x1000, ids, step: int, flags_full, flags_single) -> torch.Tensor:
        """One velocity prediction.  latents [B, N, 64] (packed), enc [B, T, 4096], pooled [B, 768], ids [T+N, 3];
        flags_full [num_layers, 3] (full_attn, full_ff, full_ff_context), flags_single [num_single_layers, 3]
        (single_attn, single_proj_mlp, single_proj_out).  Returns [B, N, 64]."""
        cfg = self.cfg
        T = enc.shape[1]
        h = self.lin(self.q(latents.float()), "x_embedder")
        temb = self.time_text_embed(timestep_x1000, guidance_x1000, pooled)
        temb_silu = self.q(F.silu(temb))
        ctx = self.lin(self.q(enc.float()), "context_embedder")
        cos, sin = rope_table(ids, cfg.axes_dims_rope)
        for i in range(cfg.num_layers):
            ctx, h = self.joint_block(i, step, flags_full[i], h, ctx, temb_silu, cos, sin)
        x = torch.cat([ctx, h], dim=1)
        for i in range(cfg.num_single_layers):
            x = self.single_block(i, step, flags_single[i], x, temb_silu, cos, sin)
        x = x[:, T:]
        e = self.lin(temb_silu, "norm_out.linear")  # AdaLayerNormContinuous [ext]: chunk order (scale, shift)
        scale, shift = [t[:, None] for t in e.chunk(2, dim=1)]
        x = self.q(self.layer_norm(x) * (1 + scale) + shift)
        return self.lin(x, "proj_out")

    def reset_cache(self) -> None:
        self.cache.reset()


# --------------------------------------------------------------------------
# FlowMatchEulerDiscreteScheduler with dynamic shifting + the FluxPipeline loop [ext]
# (the loop the reference drives from ecad/image_generators/flux_image_generator.py:330-360)
# --------------------------------------------------------------------------
def flow_match_sigmas(num_steps: int, image_seq_len: int, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15) -> np.ndarray:
    """sigmas[0..num_steps] (last = 0): linspace(1, 1/n, n) shifted by mu(image_seq_len) (calculate_shift, time_shift).  Defaults = FLUX.1-dev's
    scheduler_config.json [ext] (max_shift 1.15), which is what diffusers 0.30.3's FluxPipeline passes; 1.16 is only its helper's unused default."""
    sig = np.linspace(1.0, 1.0 / num_steps, num_steps)
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    mu = image_seq_len * m + (base_shift - m * base_seq_len)
    sig = math.exp(mu) / (math.exp(mu) + (1.0 / sig - 1.0))
    return np.concatenate([sig, [0.0]]).astype(np.float32)


def model_scalars(sigmas: np.ndarray, guidance_scale: float, dtype: torch.dtype | None):
    """The (timestep * 1000, guidance * 1000) values the transformer sees.  diffusers 0.30.3 FluxPipeline.__call__ [ext] casts the scheduler's
    timestep to the latents' dtype FIRST (`timestep = t.expand(B).to(latents.dtype)`) and hands over `timestep / 1000`, still in that dtype; the model
    multiplies by 1000 in that dtype (flux_transformer_2d_edited.py:270-274): three roundings, reproduced here one by one."""
    t = torch.from_numpy(sigmas[:-1].copy()) * 1000.0  # scheduler.timesteps (fp32)
    g = torch.tensor([guidance_scale], dtype=torch.float32)
    if dtype is None:
        return t.clone(), g * 1000.0
    t_in = t.to(dtype) / 1000.0  # computed and rounded in `dtype`
    return (t_in * 1000).float(), (g.to(dtype) * 1000).float()


def synthetic_inputs(cfg: FluxConfig, batch: int, grid_h: int, grid_w: int, text_tokens: int, seed: int = 0):
    g = torch.Generator(device="cpu").manual_seed(seed + 100)
    N = grid_h * grid_w
    latents = torch.randn(batch, N, cfg.in_channels, generator=g)
    enc = torch.randn(batch, text_tokens, cfg.joint_attention_dim, generator=g) * 0.1
    pooled = torch.randn(batch, cfg.pooled_projection_dim, generator=g) * 0.5
    ids = torch.cat([torch.zeros(text_tokens, 3), latent_image_ids(grid_h, grid_w)], dim=0)
    return latents, enc, pooled, ids


def denoise(oracle: FluxOracle, latents, enc, pooled, ids, bits_full, bits_single, guidance_scale: float = 5.0, trace: list | None = None):
    """bits_full [steps, num_layers, 3], bits_single [steps, num_single_layers, 3].  Latents are kept in the model
    dtype between steps (scheduler.step casts back to model_output.dtype [ext])."""
    steps = bits_full.shape[0]
    N = latents.shape[1]
    sig = flow_match_sigmas(steps, N)
    t_model, g_model = model_scalars(sig, guidance_scale, oracle.round_dtype)
    B = latents.shape[0]
    x = oracle.q(latents.float())
    oracle.reset_cache()
    for s in range(steps):
        v = oracle.forward(x, enc, pooled, t_model[s].expand(B), g_model.expand(B), ids, s, bits_full[s], bits_single[s])
        x = oracle.q(x + (float(sig[s + 1]) - float(sig[s])) * v)
        if trace is not None:
            trace.append((v.clone(), x.clone()))
    oracle.reset_cache()
    return x
