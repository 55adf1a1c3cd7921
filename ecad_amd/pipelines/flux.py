"""Minimal FLUX pipeline around the native transformer (diffusers is not a dependency).

Mirrors the call the reference's image generator makes (``ecad/image_generators/flux_image_generator.py:330-360``) into
diffusers' ``FluxPipeline.__call__`` [ext]: 16-channel latents packed 2x2 into [B, (H/16)(W/16), 64] tokens, (0, row, col)
image ids, FlowMatchEuler sigmas with the resolution-dependent shift, the guidance scale embedded (no CFG), and
``callback_on_step_end(pipe, step, t, {"latents", "prompt_embeds"})`` after every step.  Two execution paths:

* ``native_loop=True`` (default when the only callbacks are the schedule counters): ONE ``ecad_flux_run_schedule`` call
  (hipGraph replay); the callbacks are then fired in order so the host-side counters end in the same state.
* per-step: Python loop, one ``ecad_flux_step`` per step, Euler update in torch.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from ecad_amd.pipelines.flow_match import flow_match_sigmas, load_scheduler_config, model_scalars


class FluxPipeline:
    vae_scale_factor = 16  # 8x VAE, 2x2 packing

    def __init__(self, transformer=None, scheduler=None, vae=None, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None):
        self.transformer = transformer
        self.vae = vae
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.native_loop = True
        self.scheduler_config = scheduler if isinstance(scheduler, dict) else load_scheduler_config(None)  # base/max shift and sequence lengths
        self._device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, transformer=None, text_encoder=None, text_encoder_2=None, vae=None, **kwargs):
        # scheduler/scheduler_config.json of the checkpoint directory when there is one (FLUX.1-dev's values otherwise)
        return cls(transformer=transformer, scheduler=load_scheduler_config(pretrained_model_name_or_path), vae=vae, text_encoder=text_encoder, text_encoder_2=text_encoder_2)

    def to(self, device):
        self._device = torch.device(device)
        return self

    def encode_prompt(self, *a, **k):
        raise NotImplementedError("CLIP / T5 prompt encoding needs the encoder checkpoints and stays in PyTorch; use pre-computed embeddings")

    @staticmethod
    def _pack_latents(latents: torch.Tensor) -> torch.Tensor:  # [B, C, H, W] -> [B, (H/2)(W/2), 4C]   (FluxPipeline._pack_latents [ext])
        B, C, H, W = latents.shape
        x = latents.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(B, (H // 2) * (W // 2), C * 4)

    @staticmethod
    def _unpack_latents(latents: torch.Tensor, height: int, width: int) -> torch.Tensor:  # -> [B, C, H/8, W/8]
        B, _, ch = latents.shape
        h, w = height // 16, width // 16
        x = latents.view(B, h, w, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
        return x.reshape(B, ch // 4, h * 2, w * 2)

    @staticmethod
    def _prepare_latent_image_ids(gh: int, gw: int, device, dtype) -> torch.Tensor:
        ids = torch.zeros(gh, gw, 3)
        ids[..., 1] += torch.arange(gh)[:, None]
        ids[..., 2] += torch.arange(gw)[None, :]
        return ids.reshape(gh * gw, 3).to(device=device, dtype=dtype)

    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 20,
                 guidance_scale: float = 5.0, num_images_per_prompt: int = 1, generator: torch.Generator | None = None, latents: torch.Tensor | None = None,
                 prompt_embeds: torch.Tensor | None = None, pooled_prompt_embeds: torch.Tensor | None = None, output_type: str = "pil", return_dict: bool = True,
                 callback_on_step_end: Callable | None = None, callback_on_step_end_tensor_inputs=("latents",), **kwargs):
        if prompt is not None or prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("pass pre-computed prompt_embeds and pooled_prompt_embeds (as the reference's image generator does)")
        if num_images_per_prompt != 1:
            raise ValueError("num_images_per_prompt must be 1 (the reference always passes 1)")
        tr = self.transformer
        height, width = height or 1024, width or 1024
        if height % 16 or width % 16:
            raise ValueError("height and width must be multiples of 16")
        gh, gw = height // 16, width // 16
        device, dtype = tr.device, tr.dtype
        B = prompt_embeds.shape[0]
        enc = prompt_embeds.to(device=device, dtype=dtype).contiguous()
        pooled = pooled_prompt_embeds.to(device=device, dtype=dtype).contiguous()
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            lat = torch.randn((B, tr.config.in_channels // 4, 2 * gh, 2 * gw), generator=generator, device=gdev, dtype=dtype).to(device)  # randn_tensor [ext]
            latents = self._pack_latents(lat)
        sig = flow_match_sigmas(num_inference_steps, gh * gw, self.scheduler_config)
        t_model, g_model = model_scalars(sig, guidance_scale, dtype)
        timesteps = sig[:-1] * 1000.0

        def fire(i, lat_now):
            if callback_on_step_end is not None:
                callback_on_step_end(self, i, timesteps[i], {"latents": lat_now, "prompt_embeds": enc})

        if self.native_loop:
            x = latents.float().contiguous()
            tr.engine.run_schedule(x, enc, pooled, gh, gw, t_model, g_model, sig, use_graph=True)
            latents = x.to(dtype)
            for i in range(num_inference_steps):
                fire(i, latents)
            tr._begun_key = None
        else:
            img_ids = self._prepare_latent_image_ids(gh, gw, device, dtype)
            txt_ids = torch.zeros(enc.shape[1], 3, device=device, dtype=dtype)
            guidance = torch.full((B,), guidance_scale, device=device, dtype=torch.float32) if tr.config.guidance_embeds else None
            for i in range(num_inference_steps):
                t = torch.full((B,), float(timesteps[i]), device=device).to(dtype)
                v = tr(hidden_states=latents, timestep=t / 1000, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=enc, txt_ids=txt_ids, img_ids=img_ids,
                       return_dict=False)[0]
                latents = (latents.float() + float(sig[i + 1] - sig[i]) * v.float()).to(dtype)  # FlowMatchEulerDiscreteScheduler.step [ext]
                fire(i, latents)
        if output_type == "latent":
            image = latents
        else:
            image = self.decode(self._unpack_latents(latents, height, width), output_type)
        return (image,) if not return_dict else {"images": image}

    def decode(self, latents: torch.Tensor, output_type: str = "pil"):
        """16-channel FLUX VAE decode: the native decoder (``HipVaeDecoder`` with the FLUX geometry) or any torch module with ``.decode``;
        without a VAE a labelled three-channel preview of the latents."""
        if self.vae is not None and hasattr(self.vae, "decode_to_uint8"):
            arr = self.vae.decode_to_uint8(latents).cpu().numpy()
            if output_type == "np":
                return arr
            if output_type == "pt":
                return torch.from_numpy(arr).permute(0, 3, 1, 2).float() / 255.0
            from PIL import Image

            return [Image.fromarray(a) for a in arr]
        if self.vae is not None:
            cfg = self.vae.config
            img = self.vae.decode(latents / getattr(cfg, "scaling_factor", 0.3611) + getattr(cfg, "shift_factor", 0.1159), return_dict=False)[0]
        else:
            img = latents.float().cpu()[:, :3]
            img = torch.nn.functional.interpolate(img, scale_factor=8, mode="nearest")
        img = (img / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return img
        arr = (img.permute(0, 2, 3, 1).float().cpu().numpy() * 255).round().astype("uint8")
        if output_type == "np":
            return arr
        from PIL import Image

        return [Image.fromarray(a) for a in arr]
