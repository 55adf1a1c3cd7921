"""Oracle: the PixArt denoising loop with the ECAD callbacks.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Follows the loop mirrored at ``ecad/pipelines/pass_through.py:297-380``
(verbatim diffusers ``PixArtAlphaPipeline.__call__`` [ext]):
CFG concat (uncond first, :239-244), timestep broadcast, transformer call,
``u + g*(t-u)``, learned-sigma drop, scheduler.step, callback(step) which advances
the schedule counters and resets counters + caches after the last step
(``ecad/image_generators/image_generator.py:193-213``).
"""
from __future__ import annotations

import numpy as np
import torch

from .dpm_solver_ref import DPMSolverPP2M
from .pixart_ref import PixArtOracle
from .schedule_ref import StepCounter


def synthetic_inputs(batch: int, text_tokens: int = 120, caption_channels: int = 4096, start_seed: int = 0, latent_side: int = 32):
    """Seeded synthetic prompt embeddings / masks / latents.

    prompt_embeds ~ 0.1*N(0,1) seed 1; mask lengths ~ U{8..T} seed 2; one shared
    negative embedding with a 2-token mask (the reference uses the "" prompt for
    every negative, pixart_image_generator.py:114-126,237-242); latents from a CPU
    generator seeded with ``start_seed`` (pixart_image_generator.py:349-351).
    """
    g1 = torch.Generator(device="cpu").manual_seed(1)
    g2 = torch.Generator(device="cpu").manual_seed(2)
    g3 = torch.Generator(device="cpu").manual_seed(3)
    prompt = torch.randn(batch, text_tokens, caption_channels, generator=g1) * 0.1
    lens = torch.randint(8, text_tokens + 1, (batch,), generator=g2)
    mask = (torch.arange(text_tokens)[None, :] < lens[:, None]).to(torch.int64)
    neg_row = torch.randn(1, text_tokens, caption_channels, generator=g3) * 0.1
    neg = neg_row.repeat(batch, 1, 1)
    neg_mask = torch.zeros(batch, text_tokens, dtype=torch.int64)
    neg_mask[:, :2] = 1
    gl = torch.Generator(device="cpu").manual_seed(start_seed)
    latents = torch.randn(batch, 4, latent_side, latent_side, generator=gl)
    return {
        "prompt_embeds": prompt,
        "prompt_attention_mask": mask,
        "negative_prompt_embeds": neg,
        "negative_prompt_attention_mask": neg_mask,
        "latents": latents,
    }


@torch.no_grad()
def denoise(
    model: PixArtOracle,
    bits: np.ndarray,
    inputs: dict,
    guidance_scale: float = 4.5,
    record: bool = False,
):
    """Run the full schedule; returns final latents (and per-step noise preds if ``record``)."""
    steps = bits.shape[0]
    solver = DPMSolverPP2M(steps)
    counter = StepCounter(steps)
    enc = torch.cat([inputs["negative_prompt_embeds"], inputs["prompt_embeds"]], dim=0)
    mask = torch.cat([inputs["negative_prompt_attention_mask"], inputs["prompt_attention_mask"]], dim=0)
    latents = model.q(inputs["latents"].float())
    trace = []
    for i, t in enumerate(solver.timesteps):
        x_in = torch.cat([latents, latents], dim=0)
        tt = torch.full((x_in.shape[0],), float(t))
        noise = model.forward(x_in, enc, mask, tt, counter.curr_step, bits[counter.curr_step], added_cond=inputs.get("added_cond"))
        u, c = noise.chunk(2)
        eps = u + guidance_scale * (c - u)
        eps = model.q(eps[:, : model.cfg.in_channels])  # learned-sigma drop
        if record:
            trace.append(noise.clone())
        latents = model.q(solver.step(eps, latents))
        counter.per_step_callback(i)
        if counter.curr_step == 0:  # reset fired
            model.reset_cache()
    return (latents, trace) if record else latents
