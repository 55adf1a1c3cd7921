"""Minimal PixArt pipeline around the native transformer (diffusers is not a dependency).

Mirrors the call the reference's image generators make
(``ecad/image_generators/pixart_image_generator.py:358-383``) and the loop at
``ecad/pipelines/pass_through.py:169-404`` (verbatim diffusers ``PixArtAlphaPipeline.__call__``):
CFG concat with the negative half first, 20 DPM-Solver++ steps, ``callback(step, t, latents)``
after every step.  Two execution paths over the same engine:

* ``native_loop=True`` (default when the only callbacks are the schedule counters): the whole loop
  runs as ONE ``ecad_run_schedule`` call (hipGraph replay); the callbacks are then fired in order so
  the host-side counters end in the same state.
* per-step: Python loop, one ``ecad_step`` per step, solver math in torch -- the drop-in for code
  that needs intermediate latents.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from ecad_amd.pipelines.dpm_solver import DPMSolverMultistepScheduler

# 4-channel SD latent -> RGB preview factors (only used when no VAE is attached)
_LATENT_RGB = torch.tensor([[0.298, 0.207, 0.208], [0.187, 0.286, 0.173], [-0.158, 0.189, 0.264], [-0.184, -0.271, -0.473]])


class PixArtAlphaPipeline:
    vae_scale_factor = 8

    def __init__(self, transformer=None, scheduler=None, vae=None, text_encoder=None, tokenizer=None):
        self.transformer = transformer
        self.scheduler = scheduler or DPMSolverMultistepScheduler()
        self.vae = vae
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.native_loop = True
        self.contexts = 1  # execution contexts for submit() / finish(): image batches in flight on this GPU (EngineContexts)
        self._pool = None
        self._device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, transformer=None, text_encoder=None, vae=None, scheduler=None, **kwargs):
        return cls(transformer=transformer, scheduler=scheduler, vae=vae, text_encoder=text_encoder)

    def to(self, device):
        self._device = torch.device(device)
        return self

    def encode_prompt(self, *a, **k):
        raise NotImplementedError("text encoding stays in PyTorch/transformers (T5) and needs the encoder weights; out of scope for the hot path")

    def prepare_latents(self, batch, channels, height, width, dtype, device, generator, latents=None):
        shape = (batch, channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            # randn_tensor [ext]: CPU generator => sample on CPU, then move
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        return latents.to(device) * self.scheduler.init_noise_sigma

    @torch.no_grad()
    def __call__(
        self,
        prompt=None,
        negative_prompt=None,
        prompt_embeds: torch.Tensor | None = None,
        prompt_attention_mask: torch.Tensor | None = None,
        negative_prompt_embeds: torch.Tensor | None = None,
        negative_prompt_attention_mask: torch.Tensor | None = None,
        num_images_per_prompt: int = 1,
        num_inference_steps: int = 20,
        generator: torch.Generator | None = None,
        guidance_scale: float = 4.5,
        height: Optional[int] = None,
        width: Optional[int] = None,
        latents: torch.Tensor | None = None,
        output_type: str = "pil",
        return_dict: bool = True,
        callback: Callable[[int, int, torch.Tensor], None] | None = None,
        callback_steps: int = 1,
        **kwargs,
    ):
        if prompt is not None or prompt_embeds is None:
            raise ValueError("pass pre-computed prompt_embeds (as the reference's image generators do)")
        if num_images_per_prompt != 1:
            raise ValueError("num_images_per_prompt must be 1 (the reference always passes 1)")
        tr = self.transformer
        sample = tr.config.sample_size
        height = height or sample * self.vae_scale_factor
        width = width or sample * self.vae_scale_factor
        device = tr.device
        B = prompt_embeds.shape[0]
        cfg_on = guidance_scale > 1.0
        reps = 2 if cfg_on else 1  # guidance_scale <= 1: no unconditional half (diffusers do_classifier_free_guidance); per-step path only
        if cfg_on:
            enc = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0).to(device)
            mask = torch.cat([negative_prompt_attention_mask, prompt_attention_mask], dim=0).to(device)
        else:
            enc, mask = prompt_embeds.to(device), prompt_attention_mask.to(device)

        self.scheduler.set_timesteps(num_inference_steps, device=None)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(B, tr.config.in_channels, height, width, tr.dtype, device, generator, latents)

        # micro-conditioning of the sample_size-128 alpha checkpoints (diffusers PixArtAlphaPipeline.__call__ step 6.1, mirrored at
        # reference pass_through.py:268-290): one (height, width) / aspect ratio for the whole CFG-doubled batch
        added_cond_kwargs = {"resolution": None, "aspect_ratio": None}
        if getattr(tr, "use_additional_conditions", False):
            added_cond_kwargs = {"resolution": torch.tensor([[float(height), float(width)]]).repeat(reps * B, 1),
                                 "aspect_ratio": torch.tensor([[float(height / width)]]).repeat(reps * B, 1)}
            tr.engine.set_added_cond(added_cond_kwargs["resolution"], added_cond_kwargs["aspect_ratio"])

        if self.native_loop and cfg_on and not getattr(tr, "uses_plugins", False):  # a schedule with compute plug-ins calls back into Python: per-step path
            tr.engine.set_attention_bias(None, None)  # the pipeline passes a 0/1 text mask and no self-attention mask (pass_through.py:332-339)
            lat = tr.engine.run_schedule(enc, mask, latents.float(), timesteps.numpy().astype(np.float32), self.scheduler.coefficient_table(), guidance_scale, use_graph=True)
            latents = lat.to(tr.dtype)
            if callback is not None:
                for i, t in enumerate(timesteps):
                    if i % callback_steps == 0:
                        callback(i, t, latents)
            tr._begun_key = None
        else:
            for i, t in enumerate(timesteps):
                x_in = torch.cat([latents] * 2) if cfg_on else latents
                tt = torch.full((x_in.shape[0],), float(t), device=device)
                noise = tr(x_in, encoder_hidden_states=enc, encoder_attention_mask=mask, timestep=tt, added_cond_kwargs=added_cond_kwargs, return_dict=False)[0]
                if cfg_on:
                    u, c = noise.chunk(2)
                    noise = u + guidance_scale * (c - u)
                if tr.config.out_channels // 2 == tr.config.in_channels:
                    noise = noise.chunk(2, dim=1)[0]
                latents = self.scheduler.step(noise, t, latents)[0]
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)

        if output_type == "latent":
            image = latents
        else:
            image = self.decode(latents, output_type)
        return (image,) if not return_dict else {"images": image}

    # -- several batches in flight (round 3) ----------------------------------------------------------------------------------------------
    def can_submit(self, guidance_scale: float = 4.5) -> bool:
        """submit() / finish() cover what the whole-loop path covers: CFG on, no compute plug-ins, no consumer of intermediate latents."""
        return bool(self.native_loop and guidance_scale > 1.0 and not getattr(self.transformer, "uses_plugins", False))

    def _contexts(self):
        from ecad_amd.engine import EngineContexts

        n = max(1, int(self.contexts))
        if self._pool is None or len(self._pool) != n or self._pool.engines[0] is not self.transformer.engine:
            if self._pool is not None:
                self._pool.close()
            self._pool = EngineContexts(self.transformer.engine, n)
        return self._pool

    @torch.no_grad()
    def submit(self, prompt_embeds, prompt_attention_mask, negative_prompt_embeds, negative_prompt_attention_mask, num_inference_steps: int = 20, generator=None,
               guidance_scale: float = 4.5, height=None, width=None, latents=None, output_type: str = "pil", callback=None, callback_steps: int = 1, **_ignored) -> dict:
        """The first half of ``__call__`` on the next execution context: stages the batch and launches its whole denoising loop, without waiting.  ``finish``
        (in submission order) fires the step callbacks and decodes.  With ``contexts`` = n up to n batches overlap on the device; each batch's latents are bit for
        bit those of ``__call__``."""
        if not self.can_submit(guidance_scale):
            raise ValueError("submit() needs the whole-loop path: guidance_scale > 1, no compute plug-ins, native_loop")
        tr = self.transformer
        height = height or tr.config.sample_size * self.vae_scale_factor
        width = width or tr.config.sample_size * self.vae_scale_factor
        device = tr.device
        B = prompt_embeds.shape[0]
        enc = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0).to(device)
        mask = torch.cat([negative_prompt_attention_mask, prompt_attention_mask], dim=0).to(device)
        self.scheduler.set_timesteps(num_inference_steps, device=None)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(B, tr.config.in_channels, height, width, tr.dtype, device, generator, latents)
        pool = self._contexts()
        if getattr(tr, "use_additional_conditions", False):
            res, ar = torch.tensor([[float(height), float(width)]]).repeat(2 * B, 1), torch.tensor([[float(height / width)]]).repeat(2 * B, 1)
            for e in pool.engines:
                e.set_added_cond(res, ar)
        for e in pool.engines:
            e.set_attention_bias(None, None)
        lat, stream = pool.submit(enc, mask, latents.float(), timesteps.numpy().astype(np.float32), self.scheduler.coefficient_table(), guidance_scale, use_graph=True)
        tr._begun_key = None
        return {"latents": lat, "stream": stream, "output_type": output_type, "callback": callback, "callback_steps": callback_steps, "timesteps": timesteps}

    def finish(self, handle: dict):
        """Second half: wait for that batch only, fire the per-step callbacks in order (the schedule counters end where ``__call__`` leaves them), decode."""
        torch.cuda.current_stream().wait_stream(handle["stream"])
        handle["latents"].record_stream(torch.cuda.current_stream())  # allocated on the context's stream, read here on the caller's: keep the block until this read is done
        latents = handle["latents"].to(self.transformer.dtype)
        if handle["callback"] is not None:
            for i, t in enumerate(handle["timesteps"]):
                if i % handle["callback_steps"] == 0:
                    handle["callback"](i, t, latents)
        return latents if handle["output_type"] == "latent" else self.decode(latents, handle["output_type"])

    def decode(self, latents: torch.Tensor, output_type: str = "pil"):
        """VAE decode when a VAE module is attached (stays PyTorch/MIOpen); otherwise a linear latent->RGB preview."""
        if self.vae is not None and hasattr(self.vae, "decode_to_uint8"):  # native decoder: latents -> uint8 NHWC in one call
            if getattr(self, "_pool", None) is not None and latents.is_cuda:
                # several batches in flight: the decode of a finished batch runs on a high-priority stream, in front of the other batches' denoising launches
                # (bench.py, same box: PNG-ready 158 images/s with the decode queued behind them on the default stream, 174 this way, 166.6 one batch at a time)
                if getattr(self, "_decode_stream", None) is None:
                    self._decode_stream = torch.cuda.Stream(device=latents.device, priority=-1)
                self._decode_stream.wait_stream(torch.cuda.current_stream(latents.device))
                latents.record_stream(self._decode_stream)
                with torch.cuda.stream(self._decode_stream):
                    arr = self.vae.decode_to_uint8(latents, stream=self._decode_stream).cpu().numpy()
            else:
                arr = self.vae.decode_to_uint8(latents).cpu().numpy()
            if output_type == "np":
                return arr
            if output_type == "pt":
                return torch.from_numpy(arr).permute(0, 3, 1, 2).float() / 255.0
            from PIL import Image

            return [Image.fromarray(a) for a in arr]
        if self.vae is not None:
            img = self.vae.decode(latents / getattr(self.vae.config, "scaling_factor", 0.13025), return_dict=False)[0]
        else:
            img = torch.einsum("bchw,cr->brhw", latents.float().cpu(), _LATENT_RGB)
            img = torch.nn.functional.interpolate(img, scale_factor=self.vae_scale_factor, mode="nearest")
        img = (img / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return img
        arr = (img.permute(0, 2, 3, 1).float().cpu().numpy() * 255).round().astype("uint8")
        if output_type == "np":
            return arr
        from PIL import Image

        return [Image.fromarray(a) for a in arr]


class PixArtSigmaPipeline(PixArtAlphaPipeline):
    pass
