"""Python handle on the native FLUX engine (include/ecad_flux.h).  Thin: tensors -> raw device pointers, C return codes ->
the reference's exception types.  All compute is in libecad_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Mapping

import numpy as np
import torch

from ecad_amd import _lib as L
from ecad_amd.engine import _TORCH2ECAD, _ptr, _stream_ptr


class EcadFluxEngine:
    def __init__(
        self,
        num_layers=19,
        num_single_layers=38,
        num_heads=24,
        head_dim=128,
        in_channels=64,
        joint_attention_dim=4096,
        pooled_projection_dim=768,
        guidance_embeds=True,
        axes_dims_rope=(16, 56, 56),
        max_batch=2,
        max_image_tokens=4096,
        max_text_tokens=512,
        dtype: torch.dtype = torch.bfloat16,
        norm_eps=1e-6,
        device: int = 0,
        fp8: bool | None = None,
    ):
        self.lib = L.load()
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("engine dtype must be torch.float16 or torch.bfloat16")
        self.dtype = dtype
        self.device_index = device
        self.device = torch.device("cuda", device)
        self.cfg = L.EcadFluxCfg(
            num_layers, num_single_layers, num_heads, head_dim, in_channels, joint_attention_dim, pooled_projection_dim, int(bool(guidance_embeds)),
            (C.c_int32 * 3)(*axes_dims_rope), max_batch, max_image_tokens, max_text_tokens, _TORCH2ECAD[dtype], norm_eps,
        )
        self.dim = num_heads * head_dim
        self.num_layers, self.num_single_layers, self.in_channels = num_layers, num_single_layers, in_channels
        self._h = C.c_void_p()
        rc = self.lib.ecad_flux_create(C.byref(self.cfg), device, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            self._raise(rc, handle=None)
        self.steps = 0
        if fp8 is not None:  # None: the library default (env ECAD_FLUX_FP8)
            self._chk(self.lib.ecad_flux_set_fp8(self._h, int(fp8)))
        self.fp8 = fp8

    def _raise(self, rc: int, handle="self"):
        msg = self.lib.ecad_flux_last_error(self._h if handle == "self" else None)
        msg = msg.decode() if msg else f"error {rc}"
        if rc in (L.ECAD_E_INVALID, L.ECAD_E_UNSUPPORTED):
            raise ValueError(msg)
        raise RuntimeError(msg)

    def _chk(self, rc: int):
        if rc != 0:
            self._raise(rc)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ecad_flux_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights / schedule -----------------------------------------------------------------
    def load_state_dict(self, state: Mapping[str, torch.Tensor]) -> "EcadFluxEngine":
        """diffusers FluxTransformer2DModel key names; host or device tensors in fp16 / bf16 / fp32."""
        for name, t in state.items():
            t = t.detach().contiguous()
            if t.dtype not in _TORCH2ECAD:
                t = t.float()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self._chk(self.lib.ecad_flux_load_weights(self._h, name.encode(), t.data_ptr(), _TORCH2ECAD[t.dtype], shape, t.dim()))
        self._chk(self.lib.ecad_flux_finalize_weights(self._h))
        return self

    def set_schedule(self, bits_full: np.ndarray, bits_single: np.ndarray) -> None:
        bf = np.ascontiguousarray(np.asarray(bits_full), dtype=np.uint8)
        bs = np.ascontiguousarray(np.asarray(bits_single), dtype=np.uint8)
        if bf.ndim != 3 or bs.ndim != 3 or bf.shape[0] != bs.shape[0] or bf.shape[1:] != (self.num_layers, 3) or bs.shape[1:] != (self.num_single_layers, 3):
            raise ValueError(f"schedule bits must be [steps, {self.num_layers}, 3] and [steps, {self.num_single_layers}, 3]; got {bf.shape}, {bs.shape}")
        self._chk(self.lib.ecad_flux_set_schedule(self._h, bf.ctypes.data_as(C.POINTER(C.c_uint8)), bs.ctypes.data_as(C.POINTER(C.c_uint8)), bf.shape[0]))
        self.steps = bf.shape[0]

    # -- per image batch -----------------------------------------------------------------------
    def _check_io(self, enc, pooled):
        if enc.dtype != self.dtype or pooled.dtype != self.dtype or not enc.is_cuda or not pooled.is_cuda:
            raise ValueError("enc / pooled must be CUDA tensors of the engine dtype")
        if not enc.is_contiguous() or not pooled.is_contiguous():
            raise ValueError("enc / pooled must be contiguous")

    def begin_image_batch(self, enc: torch.Tensor, pooled: torch.Tensor, grid_h: int, grid_w: int, stream=None) -> None:
        self._check_io(enc, pooled)
        B, T, _ = enc.shape
        self._B, self._N = B, grid_h * grid_w
        self._chk(self.lib.ecad_flux_begin_image_batch(self._h, B, grid_h, grid_w, enc.data_ptr(), pooled.data_ptr(), T, _stream_ptr(stream)))

    def step(self, step_idx: int, latents: torch.Tensor, timestep_x1000: float, guidance_x1000: float, stream=None) -> torch.Tensor:
        if latents.dtype != self.dtype or not latents.is_cuda or not latents.is_contiguous():
            raise ValueError("latents must be a contiguous CUDA tensor of the engine dtype")
        out = torch.empty_like(latents)
        self._chk(self.lib.ecad_flux_step(self._h, step_idx, latents.data_ptr(), float(timestep_x1000), float(guidance_x1000), out.data_ptr(), _stream_ptr(stream)))
        return out

    def run_schedule(self, latents: torch.Tensor, enc: torch.Tensor, pooled: torch.Tensor, grid_h: int, grid_w: int, timesteps_x1000, guidance_x1000: float,
                     sigmas, use_graph: bool = True, stream=None) -> torch.Tensor:
        """latents [B, N, C] float32 CUDA, updated in place and returned."""
        self._check_io(enc, pooled)
        if latents.dtype != torch.float32 or not latents.is_cuda or not latents.is_contiguous():
            raise ValueError("latents must be a contiguous float32 CUDA tensor")
        ts = np.ascontiguousarray(np.asarray(timesteps_x1000, dtype=np.float32))
        sg = np.ascontiguousarray(np.asarray(sigmas, dtype=np.float32))
        if ts.shape != (self.steps,) or sg.shape != (self.steps + 1,):
            raise ValueError(f"timesteps must be [{self.steps}] and sigmas [{self.steps + 1}]")
        B, T, _ = enc.shape
        self._chk(self.lib.ecad_flux_run_schedule(
            self._h, B, grid_h, grid_w, enc.data_ptr(), pooled.data_ptr(), T, latents.data_ptr(), ts.ctypes.data_as(C.POINTER(C.c_float)), float(guidance_x1000),
            sg.ctypes.data_as(C.POINTER(C.c_float)), int(use_graph), _stream_ptr(stream)))
        return latents

    def reset_cache(self) -> None:
        self._chk(self.lib.ecad_flux_reset_cache(self._h))

    def stats(self) -> dict:
        s = L.EcadFluxStats()
        self._chk(self.lib.ecad_flux_stats(self._h, C.byref(s)))
        return s.as_dict()

    def set_profiling(self, on: bool) -> None:
        self._chk(self.lib.ecad_flux_set_profiling(self._h, int(on)))

    def get_profile(self) -> dict:
        n = 32
        names = (C.c_char_p * n)()
        ms = (C.c_double * n)()
        launches = (C.c_int64 * n)()
        flops = (C.c_double * n)()
        nbytes = (C.c_double * n)()
        k = self.lib.ecad_flux_get_profile(self._h, n, names, ms, launches, flops, nbytes)
        return {names[i].decode(): {"ms": ms[i], "launches": launches[i], "flops": flops[i], "bytes": nbytes[i]} for i in range(k)}

    # -- single ops (tests, micro-benchmarks) -----------------------------------------------------
    def op_gemm(self, epi: int, bn: int, A, W, bias=None, A2=None, resid=None, gate=None, seg0=0, rows_per_batch0=1, rows_per_batch1=1, want_out=True, stream=None):
        M = A.shape[0]
        K = A.shape[1] + (A2.shape[1] if A2 is not None else 0)
        N = W.shape[0]
        out = torch.empty(M, N, dtype=self.dtype, device=A.device) if want_out else None
        self._chk(self.lib.ecad_flux_op_gemm(
            self._h, epi, bn, A.data_ptr(), _ptr(A2), A.shape[1], W.data_ptr(), _ptr(bias), _ptr(out), M, N, K, _ptr(resid), _ptr(gate),
            gate.stride(0) if gate is not None else 0, seg0, rows_per_batch0, rows_per_batch1, _stream_ptr(stream)))
        return out

    def op_gemm_fp8(self, epi: int, bn: int, A, W, bias=None, iters: int = 0, stream=None):
        """(out, avg_us): the OCP-fp8 GEMM on device-quantised copies of A and W; avg_us over `iters` repeats of the GEMM alone."""
        M, K = A.shape
        N = W.shape[0]
        out = torch.empty(M, N, dtype=self.dtype, device=A.device)
        us = C.c_float(0.0)
        self._chk(self.lib.ecad_flux_op_gemm_fp8(self._h, epi, bn, A.data_ptr(), W.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, iters, C.byref(us), _stream_ptr(stream)))
        return out, us.value

    def op_qk_norm_rope(self, qkv, B, T, N, wq, wk, rope, wq_txt=None, wk_txt=None, stream=None):
        self._chk(self.lib.ecad_flux_op_qk_norm_rope(self._h, qkv.data_ptr(), B, T, N, wq.data_ptr(), wk.data_ptr(), _ptr(wq_txt), _ptr(wk_txt), rope.data_ptr(), _stream_ptr(stream)))
        return qkv

    def op_attention(self, qkv, B, T, N, stream=None):
        out = torch.empty(qkv.shape[0], self.dim, dtype=self.dtype, device=qkv.device)
        self._chk(self.lib.ecad_flux_op_attention(self._h, qkv.data_ptr(), out.data_ptr(), B, T, N, _stream_ptr(stream)))
        return out

    def op_ln_modulate(self, x, B, T, N, shift, scale, stream=None):
        y = torch.empty_like(x)
        self._chk(self.lib.ecad_flux_op_ln_modulate(self._h, x.data_ptr(), y.data_ptr(), B, T, N, shift.data_ptr(), scale.data_ptr(), shift.stride(0), _stream_ptr(stream)))
        return y
