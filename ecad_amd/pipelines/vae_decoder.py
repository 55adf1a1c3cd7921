"""AutoencoderKL decoder (the SD / sd-vae-ft-ema architecture PixArt-alpha ships with), in plain PyTorch.

The decode tail is OUTSIDE the native hot path ("stays PyTorch/MIOpen"); it exists so the
"PNG-ready" latency -- what the reference's 84.09 ms/image contains (``pipelines/pass_through.py:382-396`` inside the
timed region of ``pixart_image_generator.py:405-438``) -- can be reported next to the latents-out number.  Module and
parameter names follow diffusers' ``AutoencoderKL`` [ext] (``decoder.conv_in``, ``decoder.mid_block.resnets.0.norm1``,
``decoder.up_blocks.1.upsamplers.0.conv`` ...), so a real ``vae/diffusion_pytorch_model.safetensors`` loads by key.
Without a checkpoint the weights are random-init: timing is representative, pixels are not.
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


class HipVaeDecoder:
    """The same decoder on the native engine (``ecad_vae_*`` in include/ecad_hip.h): latents in, uint8 NHWC RGB out.

    ``decode_to_uint8`` is what the pipeline uses for ``output_type="pil"``; ``from_torch`` copies the weights of a
    PyTorch ``AutoencoderKLDecoderOnly`` (or any state dict with diffusers' AutoencoderKL keys).
    """

    def __init__(self, scaling_factor: float = 0.18215, dtype: torch.dtype = torch.float16, device: int = 0, latent_channels: int = 4,
                 use_post_quant_conv: bool = True, shift_factor: float = 0.0):
        import ctypes as C

        from ecad_amd import _lib as L

        self._C, self._L = C, L
        self.lib = L.load()
        self.dtype, self.device_index = dtype, device
        self.config = SimpleNamespace(scaling_factor=scaling_factor, latent_channels=latent_channels, use_post_quant_conv=use_post_quant_conv, shift_factor=shift_factor)
        self._h = C.c_void_p()
        rc = self.lib.ecad_vae_create_ex(device, L.ECAD_DTYPE_F16 if dtype == torch.float16 else L.ECAD_DTYPE_BF16, latent_channels, int(use_post_quant_conv), C.byref(self._h))
        if rc != 0:
            raise RuntimeError((self.lib.ecad_vae_last_error(None) or b"ecad_vae_create failed").decode())

    def _chk(self, rc: int):
        if rc != 0:
            msg = (self.lib.ecad_vae_last_error(self._h) or b"error").decode()
            raise ValueError(msg) if rc in (self._L.ECAD_E_INVALID, self._L.ECAD_E_UNSUPPORTED) else RuntimeError(msg)

    def load_state_dict(self, state) -> "HipVaeDecoder":
        C, L = self._C, self._L
        codes = {torch.float16: L.ECAD_DTYPE_F16, torch.bfloat16: L.ECAD_DTYPE_BF16, torch.float32: L.ECAD_DTYPE_F32}
        for name, t in state.items():
            t = t.detach().contiguous()
            if t.dtype not in codes:
                t = t.float()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self._chk(self.lib.ecad_vae_load_weights(self._h, name.encode(), t.data_ptr(), codes[t.dtype], shape, t.dim()))
        self._chk(self.lib.ecad_vae_finalize_weights(self._h))
        return self

    @classmethod
    def from_torch(cls, module: "AutoencoderKLDecoderOnly", device: int = 0, dtype: torch.dtype = torch.float16) -> "HipVaeDecoder":
        c = module.config
        return cls(c.scaling_factor, dtype, device, c.latent_channels, getattr(c, "use_post_quant_conv", True), getattr(c, "shift_factor", 0.0) or 0.0).load_state_dict(module.state_dict())

    def decode_to_uint8(self, latents: torch.Tensor, stream=None) -> torch.Tensor:
        """latents: [B, latent_channels, h, w] (un-scaled pipeline latents) -> uint8 [B, 8h, 8w, 3] on the device."""
        lat = latents.to(device=torch.device("cuda", self.device_index), dtype=torch.float32).contiguous()
        B, _, h, w = lat.shape
        out = torch.empty((B, 8 * h, 8 * w, 3), dtype=torch.uint8, device=lat.device)
        st = torch.cuda.current_stream() if stream is None else stream
        if lat.shape[1] != self.config.latent_channels:
            raise ValueError(f"latents have {lat.shape[1]} channels, decoder expects {self.config.latent_channels}")
        self._chk(self.lib.ecad_vae_decode_ex(self._h, lat.data_ptr(), B, h, w, float(self.config.scaling_factor), float(self.config.shift_factor), out.data_ptr(), int(st.cuda_stream)))
        return out

    def close(self):
        if self._h and self._h.value:
            self.lib.ecad_vae_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
