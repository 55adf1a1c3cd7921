"""Decode tail on the native engine: ``HipVaeDecoder`` (``ecad_vae_*`` in include/ecad_hip.h, ``csrc/vae.hip``) -- AutoencoderKL decoder + uint8
postprocess, what ``self.vae.decode(latents / scaling_factor)`` + ``image_processor.postprocess(.., "pil")`` are in the reference's timed region
(``pipelines/pass_through.py:382-396`` inside ``pixart_image_generator.py:405-438``).  Weights load by diffusers' ``AutoencoderKL`` key names
(``decoder.conv_in.weight``, ``decoder.mid_block.resnets.0.norm1.weight``, ``decoder.up_blocks.1.upsamplers.0.conv.weight`` ...), so a real
``vae/diffusion_pytorch_model.safetensors`` loads as is; without a checkpoint ``synthetic_vae_state_dict`` fills the same table with random values
(timing is representative, pixels are not).  The PyTorch module this decoder is checked against lives in ``oracle/vae_ref.py`` (tests only).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch


def vae_state_dict_shapes(latent_channels: int = 4, use_post_quant_conv: bool = True, block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2) -> dict[str, tuple]:
    """name -> shape of every decoder-side parameter of diffusers' AutoencoderKL [ext] (SD VAE: 49.5 M parameters with post_quant_conv)."""
    out: dict[str, tuple] = {}

    def conv(name, cin, cout, k):
        out[name + ".weight"], out[name + ".bias"] = (cout, cin, k, k), (cout,)

    def norm(name, ch):
        out[name + ".weight"], out[name + ".bias"] = (ch,), (ch,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    if use_post_quant_conv:
        conv("post_quant_conv", latent_channels, latent_channels, 1)
    rev = list(reversed(block_out_channels))
    conv("decoder.conv_in", latent_channels, rev[0], 3)
    resnet("decoder.mid_block.resnets.0", rev[0], rev[0])
    norm("decoder.mid_block.attentions.0.group_norm", rev[0])
    for lin in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"decoder.mid_block.attentions.0.{lin}.weight"], out[f"decoder.mid_block.attentions.0.{lin}.bias"] = (rev[0], rev[0]), (rev[0],)
    resnet("decoder.mid_block.resnets.1", rev[0], rev[0])
    prev = rev[0]
    for i, ch in enumerate(rev):
        for j in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch)
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
        prev = ch
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", rev[-1], 3, 3)
    return out


def synthetic_vae_state_dict(seed: int = 0, latent_channels: int = 4, use_post_quant_conv: bool = True, device="cpu", dtype=torch.float32) -> dict[str, torch.Tensor]:
    """Random weights with torch's default Conv / Linear statistics (uniform in +-1/sqrt(fan_in)); GroupNorm affine = identity."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape in vae_state_dict_shapes(latent_channels, use_post_quant_conv).items():
        if "norm" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        else:
            wshape = shape if name.endswith("weight") else vae_state_dict_shapes(latent_channels, use_post_quant_conv)[name[: -len("bias")] + "weight"]
            bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
        sd[name] = t.to(device=device, dtype=dtype)
    return sd


class HipVaeDecoder:
    """The same decoder on the native engine (``ecad_vae_*`` in include/ecad_hip.h): latents in, uint8 NHWC RGB out.

    ``decode_to_uint8`` is what the pipeline uses for ``output_type="pil"``; ``from_torch`` copies the weights of any module that has diffusers'
    AutoencoderKL ``config`` fields and ``state_dict()`` keys; ``random_init`` is the no-checkpoint stand-in (bench.py's PNG-ready rate).
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
    def random_init(cls, seed: int = 0, device: int = 0, dtype: torch.dtype = torch.float16, flux: bool = False) -> "HipVaeDecoder":
        """SD-VAE geometry (PixArt) or the FLUX.1 VAE geometry [ext] (16 latent channels, no post_quant_conv, scaling 0.3611, shift 0.1159)."""
        kw = dict(scaling_factor=0.3611, latent_channels=16, use_post_quant_conv=False, shift_factor=0.1159) if flux else {}
        dec = cls(dtype=dtype, device=device, **kw)
        return dec.load_state_dict(synthetic_vae_state_dict(seed, dec.config.latent_channels, dec.config.use_post_quant_conv))

    @classmethod
    def from_torch(cls, module, device: int = 0, dtype: torch.dtype = torch.float16) -> "HipVaeDecoder":
        c = module.config
        return cls(c.scaling_factor, dtype, device, c.latent_channels, getattr(c, "use_post_quant_conv", True), getattr(c, "shift_factor", 0.0) or 0.0).load_state_dict(module.state_dict())

    def decode_stats(self) -> dict:
        """Roofline accounting of the last decode: flops of its convolution GEMMs, their launch count, how many ran on the 256 x 128 / 3-stage kernel
        (`conv3s_kernel`) and how many on the generated four-wave K loop over the zero-bordered layout (`conv4_kernel`)."""
        C = self._C
        fl, n, n3, n4, fx = C.c_double(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_double(0)
        self._chk(self.lib.ecad_vae_decode_stats_ex(self._h, C.byref(fl), C.byref(n), C.byref(n3), C.byref(n4), C.byref(fx)))
        # conv_flops: the reference's arithmetic (nine taps per 3 x 3 convolution); conv_flops_executed: what the launches multiplied (upsamplers in their phase form)
        return {"conv_flops": fl.value, "conv_launches": n.value, "conv3s_launches": n3.value, "conv4_launches": n4.value, "conv_flops_executed": fx.value}

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
