"""CHECKER ONLY (test infrastructure): AutoencoderKL decoder (the SD / sd-vae-ft-ema architecture PixArt-alpha ships with; FLUX geometry via
``.flux()``) in plain PyTorch.  Only ``tests/`` import it, as the fp32 reference the HIP decoder (``ecad_vae_*``, ``ecad_amd/pipelines/vae_decoder.py::
HipVaeDecoder``) is compared with; the product never does (``tests/test_cpu_host.py::test_product_never_imports_oracle``).

Restates diffusers' ``AutoencoderKL.decode`` [ext] (``diffusers==0.30.3``, absent here: "restated, not verified against diffusers" -- PARITY UNPINNED
for the layer arithmetic, as for the transformer).  The reference calls it at ``ecad/pipelines/pass_through.py:382-396``
(``self.vae.decode(latents / self.vae.config.scaling_factor)`` + ``image_processor.postprocess``).  Module and parameter names follow diffusers
(``decoder.conv_in``, ``decoder.mid_block.resnets.0.norm1``, ``decoder.up_blocks.1.upsamplers.0.conv`` ...), so a real
``vae/diffusion_pytorch_model.safetensors`` would load by key, and so the product's key / shape table (``vae_state_dict_shapes``) can be checked
against this module's ``state_dict()``.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class AttentionBlock(nn.Module):
    def __init__(self, ch: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        o = F.scaled_dot_product_attention(self.to_q(t)[:, None], self.to_k(t)[:, None], self.to_v(t)[:, None])[:, 0]
        return x + self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)


class _Mid(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch), ResnetBlock2D(ch, ch)])
        self.attentions = nn.ModuleList([AttentionBlock(ch)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Upsampler(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, upsample: bool, n: int = 3):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(n)])
        self.upsamplers = nn.ModuleList([_Upsampler(cout)]) if upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class Decoder(nn.Module):
    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512)):
        super().__init__()
        rev = list(reversed(block_out_channels))  # 512, 512, 256, 128
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0])
        blocks, prev = [], rev[0]
        for i, ch in enumerate(rev):
            blocks.append(_UpBlock(prev, ch, upsample=i < len(rev) - 1))
            prev = ch
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(32, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLDecoderOnly(nn.Module):
    """``vae.decode(latents / scaling_factor, return_dict=False)[0]`` surface the pipeline uses."""

    def __init__(self, scaling_factor: float = 0.18215, latent_channels: int = 4, use_post_quant_conv: bool = True, shift_factor: float = 0.0):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1) if use_post_quant_conv else None
        self.decoder = Decoder(latent_channels)
        self.config = SimpleNamespace(scaling_factor=scaling_factor, latent_channels=latent_channels, use_post_quant_conv=use_post_quant_conv, shift_factor=shift_factor)

    @classmethod
    def flux(cls) -> "AutoencoderKLDecoderOnly":
        """Geometry of the FLUX.1 VAE [ext]: 16 latent channels, no post_quant_conv, scaling 0.3611, shift 0.1159."""
        return cls(scaling_factor=0.3611, latent_channels=16, use_post_quant_conv=False, shift_factor=0.1159)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False):
        p = next(self.parameters())
        z = z.to(p.dtype).contiguous(memory_format=torch.channels_last)
        x = self.decoder(self.post_quant_conv(z) if self.post_quant_conv is not None else z)
        return (x,)

    @classmethod
    def random_init(cls, seed: int = 0, device="cuda", dtype=torch.float16, **kw) -> "AutoencoderKLDecoderOnly":
        torch.manual_seed(seed)
        return cls(**kw).to(device=device, dtype=dtype).to(memory_format=torch.channels_last).eval()
