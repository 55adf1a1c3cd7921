"""Python handle on one native engine (one per process and GPU).

Thin: converts torch tensors to raw device pointers and raises the reference's exception
types (ValueError for bad arguments, RuntimeError for device failures).  All compute is in
libecad_hip.so; torch only owns the I/O tensors and the stream.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Mapping

import numpy as np
import torch

from ecad_amd import _lib as L

_TORCH2ECAD = {torch.float16: L.ECAD_DTYPE_F16, torch.bfloat16: L.ECAD_DTYPE_BF16, torch.float32: L.ECAD_DTYPE_F32}


def _stream_ptr(stream=None) -> int:
    if stream is None:
        stream = torch.cuda.current_stream()
    return int(stream.cuda_stream)


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


class EcadEngine:
    def __init__(
        self,
        num_layers=28,
        num_heads=16,
        head_dim=72,
        in_channels=4,
        out_channels=8,
        patch_size=2,
        sample_size=32,
        caption_channels=4096,
        max_batch=32,
        max_tokens=256,
        max_text_tokens=120,
        dtype: torch.dtype = torch.float16,
        norm_eps=1e-6,
        interpolation_scale=None,
        device: int = 0,
        fp8: bool | None = None,
        use_additional_conditions: bool = False,
    ):
        self.lib = L.load()
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("engine dtype must be torch.float16 or torch.bfloat16")
        self._ctor = dict(num_layers=num_layers, num_heads=num_heads, head_dim=head_dim, in_channels=in_channels, out_channels=out_channels, patch_size=patch_size,
                          sample_size=sample_size, caption_channels=caption_channels, max_batch=max_batch, max_tokens=max_tokens, max_text_tokens=max_text_tokens,
                          dtype=dtype, norm_eps=norm_eps, interpolation_scale=interpolation_scale, device=device, fp8=fp8, use_additional_conditions=use_additional_conditions)
        self._donor = None  # the engine whose weights this context reads (kept alive as long as this one)
        self._bits = None
        self.dtype = dtype
        self.device_index = device
        self.device = torch.device("cuda", device)
        self.cfg = L.EcadModelCfg(
            num_layers, num_heads, head_dim, in_channels, out_channels, patch_size, sample_size, caption_channels,
            max_batch, max_tokens, max_text_tokens, _TORCH2ECAD[dtype], norm_eps, float(interpolation_scale or 0.0),
            int(bool(use_additional_conditions)),
        )
        self.dim = num_heads * head_dim
        self._h = C.c_void_p()
        rc = self.lib.ecad_create(C.byref(self.cfg), device, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            self._raise(rc, handle=None)
        self.steps = 0
        if fp8 is not None:  # None: the library default (env ECAD_FP8)
            self._chk(self.lib.ecad_set_fp8(self._h, int(fp8)))
        self.fp8 = fp8
        self.ln_fold = os.environ.get("ECAD_LN_FOLD", "1") != "0"  # the library's default (set_ln_fold changes it)

    # -- errors ---------------------------------------------------------------------------
    def _raise(self, rc: int, handle="self"):
        msg = self.lib.ecad_last_error(self._h if handle == "self" else None)
        msg = msg.decode() if msg else f"error {rc}"
        if rc in (L.ECAD_E_INVALID, L.ECAD_E_UNSUPPORTED):
            raise ValueError(msg)
        raise RuntimeError(msg)

    def _chk(self, rc: int):
        if rc != 0:
            self._raise(rc)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ecad_destroy(self._h)
            self._h = C.c_void_p()
        self._donor = None

    def set_ln_fold(self, on: bool) -> None:
        """LayerNorm folded into the consumer GEMMs of the whole-loop plan (default on): switch it off where every schedule runs for one batch only."""
        self._chk(self.lib.ecad_set_ln_fold(self._h, int(bool(on))))
        self.ln_fold = bool(on)

    def set_batches_in_flight(self, n: int) -> None:
        """How many image batches the caller keeps in flight on this GPU: the few-rows rule (64-row latency tiles while a launch would leave half the chip idle)
        counts the other batches' tiles too.  ``EngineContexts`` sets it for its engines."""
        self._chk(self.lib.ecad_set_batches_in_flight(self._h, int(n)))

    def set_split_k(self, on: bool) -> None:
        """Split-K for the few-rows FF2 (default on; ``ECAD_SPLIT_K``): off restores bit-identity of a sample across batch sizes (``ecad_set_split_k``)."""
        self._chk(self.lib.ecad_set_split_k(self._h, int(bool(on))))

    NATIVE_DEFAULT, NATIVE_RECOMPUTE, NATIVE_CFG_MEAN_CACHE = 0, 1, 2

    def set_native_variants(self, kinds: np.ndarray | None) -> None:
        """kinds [steps, blocks, 3] (attn1, attn2, ff) of ``NATIVE_*`` ids: gate rules the engine lowers into its own kernels, so a schedule that names them stays in
        the whole-loop hipGraph (``ecad_set_native_variants``).  Call after ``set_schedule``; None / all zero: every component uses the default rule."""
        if kinds is None or not np.asarray(kinds).any():
            self._chk(self.lib.ecad_set_native_variants(self._h, None, 0, 0))
            return
        kinds = np.ascontiguousarray(np.asarray(kinds).astype(np.uint8))
        if kinds.ndim != 3 or kinds.shape[2] != 3:
            raise ValueError(f"native variants must have shape (steps, blocks, 3), got {kinds.shape}")
        self._chk(self.lib.ecad_set_native_variants(self._h, kinds.ctypes.data_as(C.POINTER(C.c_uint8)), kinds.shape[0], kinds.shape[1]))

    def new_context(self) -> "EcadEngine":
        """A further execution context on the same GPU: same configuration, the SAME weights (``ecad_share_weights``: no copy, this engine must outlive
        it), the same schedule; its own workspaces, cache arena, plans and hipGraph.  Batches submitted to different contexts on different streams overlap
        on the device (``EngineContexts``)."""
        ctx = EcadEngine(**{**self._ctor, "fp8": None})  # fp8 mode travels with the weights: a context multiplies the donor's e4m3 copies (ecad_share_weights)
        ctx.fp8 = self.fp8
        ctx._chk(self.lib.ecad_share_weights(ctx._h, self._h))
        ctx._donor = self  # (the weight arena itself is owned jointly on the C side: closing this engine first leaves the context usable)
        if self._bits is not None:
            ctx.set_schedule(self._bits)
        return ctx

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights --------------------------------------------------------------------------
    def load_state_dict(self, state: Mapping[str, torch.Tensor]) -> None:
        for name, t in state.items():
            if name == "pos_embed.pos_embed":
                continue
            t = t.detach().contiguous()
            if t.dtype not in _TORCH2ECAD:
                t = t.float()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self._chk(self.lib.ecad_load_weights(self._h, name.encode(), t.data_ptr(), _TORCH2ECAD[t.dtype], shape, t.dim()))
        self._chk(self.lib.ecad_finalize_weights(self._h))

    # -- schedule -------------------------------------------------------------------------
    def set_schedule(self, bits: np.ndarray) -> None:
        bits = np.ascontiguousarray(np.asarray(bits).astype(np.uint8))
        if bits.ndim != 3 or bits.shape[2] != 3:
            raise ValueError(f"schedule bits must have shape (steps, blocks, 3), got {bits.shape}")
        self.steps = int(bits.shape[0])
        self._chk(self.lib.ecad_set_schedule(self._h, bits.ctypes.data_as(C.POINTER(C.c_uint8)), bits.shape[0], bits.shape[1]))
        self._bits = bits

    # -- micro-conditioning (1024-MS alpha checkpoints) ---------------------------------------
    def set_added_cond(self, resolution, aspect_ratio) -> None:
        """``added_cond_kwargs`` of the reference forward (pixart_transformer_2d_edited.py:208-211): resolution [Bt, 2] (height, width) and
        aspect_ratio [Bt] or [Bt, 1]; must precede ``begin_image_batch`` / ``run_schedule`` for a batch of that size."""
        if resolution is None or aspect_ratio is None:
            raise ValueError("`added_cond_kwargs` cannot be None when using additional conditions for `adaln_single`.")
        res = np.ascontiguousarray(torch.as_tensor(resolution).detach().float().cpu().numpy().reshape(-1, 2), dtype=np.float32)
        ar = np.ascontiguousarray(torch.as_tensor(aspect_ratio).detach().float().cpu().numpy().reshape(-1), dtype=np.float32)
        if res.shape[0] != ar.shape[0]:
            raise ValueError(f"resolution has {res.shape[0]} rows but aspect_ratio has {ar.shape[0]}")
        fp = C.POINTER(C.c_float)
        self._chk(self.lib.ecad_set_added_cond(self._h, res.ctypes.data_as(fp), ar.ctypes.data_as(fp), res.shape[0]))

    # -- attention biases ---------------------------------------------------------------------
    def set_attention_bias(self, self_bias: torch.Tensor | None, enc_bias: torch.Tensor | None, stream=None) -> None:
        """Additive attention biases, one float per (sample, key), broadcast over heads and queries: ``self_bias`` [Bt, N] for the self-attention of every block
        (the reference's ``attention_mask`` after ``_create_attention_mask``, pixart_transformer_2d_edited.py:255-279), ``enc_bias`` [Bt, T] in place of the bias
        derived from the 0/1 ``encoder_attention_mask`` (a 3-D mask is a bias already and is added as given, :281-291).  ``None`` switches a bias off.  Must
        precede ``begin_image_batch`` / ``run_schedule`` of the batch it applies to; stays in force until changed."""
        def prep(b):
            if b is None:
                return None
            b = b.to(device=self.device, dtype=torch.float32).contiguous()
            if b.dim() != 2:
                raise ValueError(f"attention bias must be [batch, keys], got {tuple(b.shape)}")
            return b
        sb, eb = prep(self_bias), prep(enc_bias)
        Bt = sb.shape[0] if sb is not None else eb.shape[0] if eb is not None else 1
        if sb is not None and eb is not None and sb.shape[0] != eb.shape[0]:
            raise ValueError("self- and encoder attention biases disagree on the batch size")
        self._keep_bias = (sb, eb)
        self._chk(self.lib.ecad_set_attention_bias(self._h, _ptr(sb), _ptr(eb), int(Bt), int(sb.shape[1]) if sb is not None else 0,
                                                   int(eb.shape[1]) if eb is not None else 0, _stream_ptr(stream)))

    # -- per batch ------------------------------------------------------------------------
    def _prep_enc(self, enc: torch.Tensor, mask: torch.Tensor | None):
        enc = enc.to(device=self.device, dtype=self.dtype).contiguous()
        if mask is not None:
            mask = (mask.to(self.device) != 0).to(torch.uint8).contiguous()
        return enc, mask

    def begin_image_batch(self, enc: torch.Tensor, mask: torch.Tensor | None, height: int, width: int, stream=None) -> None:
        enc, mask = self._prep_enc(enc, mask)
        Bt, T, _ = enc.shape
        self._keep = (enc, mask)
        self._chk(self.lib.ecad_begin_image_batch(self._h, Bt, height, width, enc.data_ptr(), _ptr(mask), T, _stream_ptr(stream)))

    def rebind_batch_inputs(self, enc: torch.Tensor, mask: torch.Tensor | None, height: int, width: int, stream=None) -> None:
        """``begin_image_batch`` for new input tensors of the batch that has begun, leaving the block caches as they are (``ecad_rebind_batch_inputs``)."""
        enc, mask = self._prep_enc(enc, mask)
        Bt, T, _ = enc.shape
        self._keep = (enc, mask)
        self._chk(self.lib.ecad_rebind_batch_inputs(self._h, Bt, height, width, enc.data_ptr(), _ptr(mask), T, _stream_ptr(stream)))

    def step(self, step_idx: int, latents: torch.Tensor, timestep: torch.Tensor, stream=None) -> torch.Tensor:
        latents = latents.to(device=self.device, dtype=self.dtype).contiguous()
        Bt = latents.shape[0]
        ts = timestep.to(device=self.device, dtype=torch.float32).reshape(-1)
        if ts.numel() == 1:
            ts = ts.expand(Bt)
        ts = ts.contiguous()
        out = torch.empty((Bt, self.cfg.out_channels, latents.shape[2], latents.shape[3]), device=self.device, dtype=self.dtype)
        rc = self.lib.ecad_step(self._h, int(step_idx), latents.data_ptr(), ts.data_ptr(), out.data_ptr(), _stream_ptr(stream))
        custom = getattr(self, "_custom", None)
        if rc != 0 and custom is not None and custom[3]["error"] is not None:  # a plug-in raised: the user's exception, not the engine's wrapper
            err, custom[3]["error"] = custom[3]["error"], None
            raise err
        self._chk(rc)
        return out

    def run_schedule(self, enc, mask, latents: torch.Tensor, timesteps, solver_coef, guidance_scale: float = 4.5, use_graph: bool = True, stream=None) -> torch.Tensor:
        """Whole denoising loop for one image batch; returns the final latents (fp32, new tensor)."""
        enc, mask = self._prep_enc(enc, mask)
        Bt, T, _ = enc.shape
        lat = latents.to(device=self.device, dtype=torch.float32).contiguous().clone()
        if lat.shape[0] * 2 != Bt:
            raise ValueError(f"latents batch {lat.shape[0]} must be half of the CFG batch {Bt}")
        ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.float32))
        coef = np.ascontiguousarray(np.asarray(solver_coef, dtype=np.float32))
        if ts.shape[0] != self.steps or coef.shape != (self.steps, 5):
            raise ValueError(f"need {self.steps} timesteps and [{self.steps}, 5] solver coefficients")
        self._chk(
            self.lib.ecad_run_schedule(
                self._h, Bt, lat.shape[2], lat.shape[3], enc.data_ptr(), _ptr(mask), T, lat.data_ptr(),
                ts.ctypes.data_as(C.POINTER(C.c_float)), coef.ctypes.data_as(C.POINTER(C.c_float)),
                float(guidance_scale), int(use_graph), _stream_ptr(stream),
            )
        )
        return lat

    def reset_cache(self) -> None:
        self._chk(self.lib.ecad_reset_cache(self._h))

    # -- user compute plug-ins (ecad_set_custom_components) -------------------------------------
    def set_custom_components(self, mask: np.ndarray | None, callback=None) -> None:
        """mask [blocks, 3] (attn1, attn2, ff) of components routed to ``callback(step, block, component, x) -> Tensor`` during ``step``: ``x`` is the
        sub-layer input [Bt, N, D] (a view of an engine-filled buffer), the return value the sub-layer output before gate and residual."""
        if mask is None or not np.asarray(mask).any():
            self._chk(self.lib.ecad_set_custom_components(self._h, None, 0, None, None, None, None))
            self._custom = None
            return
        mask = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if mask.ndim != 2 or mask.shape[1] != 3:
            raise ValueError(f"plug-in mask must have shape (blocks, 3), got {mask.shape}")
        rows = self.cfg.max_batch * self.cfg.max_tokens
        io_in = torch.empty((rows, self.dim), device=self.device, dtype=self.dtype)
        io_out = torch.empty((rows, self.dim), device=self.device, dtype=self.dtype)
        state = {"error": None}

        def trampoline(_user, step, block, comp, _in, _out, Bt, N, D, _stream):
            try:
                # The engine filled io_in and will consume io_out on the stream step() was given; everything the plug-in launches (its own torch ops,
                # block_attn / block_ff / batch_context, the copy into io_out) has to be ordered on THAT stream.  Normally it is torch's current stream;
                # when the caller passed another one, the callback runs with that stream made current.  The plug-in sees a fresh tensor, as the
                # reference's do: the next component overwrites io_in.
                handle = int(_stream or 0)
                same = handle == int(torch.cuda.current_stream(self.device).cuda_stream)
                with contextlib.nullcontext() if same else torch.cuda.stream(torch.cuda.ExternalStream(handle, device=self.device)):
                    out = callback(step, block, comp, io_in[: Bt * N].view(Bt, N, D).clone())
                    io_out[: Bt * N].copy_(out.reshape(Bt * N, D))
                return 0
            except BaseException as e:  # noqa: BLE001  (re-raised by step() on this thread)
                state["error"] = e
                return 1

        fn = L.COMPONENT_FN(trampoline)
        self._custom = (fn, io_in, io_out, state)  # keep the ctypes thunk and the buffers alive as long as the engine may call them
        self._chk(self.lib.ecad_set_custom_components(self._h, mask.ctypes.data_as(C.POINTER(C.c_uint8)), mask.shape[0], C.cast(fn, C.c_void_p), None,
                                                      io_in.data_ptr(), io_out.data_ptr()))

    def block_attn(self, block: int, which: int, x: torch.Tensor) -> torch.Tensor:
        """The block's own ``attn1(x)`` (which 1) / ``attn2(x, text of the batch)`` (which 2): [Bt, N, D] -> [Bt, N, D], bias only."""
        x = x.to(device=self.device, dtype=self.dtype).contiguous()
        out = torch.empty_like(x)
        self._chk(self.lib.ecad_block_attn(self._h, int(block), int(which), x.data_ptr(), out.data_ptr(), _stream_ptr()))
        return out

    def block_ff(self, block: int, x: torch.Tensor) -> torch.Tensor:
        x = x.to(device=self.device, dtype=self.dtype).contiguous()
        out = torch.empty_like(x)
        self._chk(self.lib.ecad_block_ff(self._h, int(block), x.data_ptr(), out.data_ptr(), _stream_ptr()))
        return out

    def batch_context(self) -> tuple[torch.Tensor, torch.Tensor]:
        """(caption-projected text [Bt, T, D], additive mask bias [Bt, 1, T]) of the batch that has begun -- the block's ``encoder_hidden_states`` / mask."""
        Bt, N, T = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.ecad_copy_batch_context(self._h, None, None, C.byref(Bt), C.byref(N), C.byref(T), _stream_ptr()))
        ctx = torch.empty((Bt.value, T.value, self.dim), device=self.device, dtype=self.dtype)
        bias = torch.empty((Bt.value, 1, T.value), device=self.device, dtype=torch.float32)
        self._chk(self.lib.ecad_copy_batch_context(self._h, ctx.data_ptr(), bias.data_ptr(), None, None, None, _stream_ptr()))
        return ctx, bias.to(self.dtype)

    def stats(self) -> dict:
        s = L.EcadStats()
        self._chk(self.lib.ecad_cache_stats(self._h, C.byref(s)))
        return s.as_dict()

    # -- profiling ------------------------------------------------------------------------
    def set_profiling(self, on: bool) -> None:
        self._chk(self.lib.ecad_set_profiling(self._h, int(on)))

    def get_profile(self) -> dict[str, dict]:
        n = 32
        names = (C.c_char_p * n)()
        ms = (C.c_double * n)()
        launches = (C.c_int64 * n)()
        flops = (C.c_double * n)()
        nbytes = (C.c_double * n)()
        k = self.lib.ecad_get_profile(self._h, n, names, ms, launches, flops, nbytes)
        return {names[i].decode(): {"ms": ms[i], "launches": launches[i], "flops": flops[i], "bytes": nbytes[i]} for i in range(k)}

    # -- single ops (tests / micro-benchmarks) ----------------------------------------------
    def op_gemm(self, epi: int, a, w, bias=None, resid=None, gate_table=None, gate_mod=None, gate_mod_stride=0, rows_per_batch=1, out=None):
        M, K = a.shape
        N = w.shape[0]
        if out is None and epi != 3:
            out = torch.empty((M, N), device=self.device, dtype=self.dtype)
        self._chk(
            self.lib.ecad_op_gemm(self._h, epi, a.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(out), M, N, K, _ptr(resid), _ptr(gate_table), _ptr(gate_mod),
                                  int(gate_mod_stride), int(rows_per_batch), _stream_ptr())
        )
        return out

    GEMM_KERNELS = {"tile144": 0, "wide288": 2, "tile128": 3, "latency64": 5, "latency64_splitk": 6, "narrow4": 7}
    GEMM_SITES = {"qkv": 0, "attn1_out": 1, "attn2_q": 2, "attn2_out": 3, "ff1": 4, "ff2": 5, "text_kv": 6, "caption": 7, "small": 8, "attn1_kv": 9}

    def op_gemm_ex(self, kernel: str, site: str, epi: int, a, w, bias=None, resid=None, gate_table=None, gate_mod=None, gate_mod_stride=0, rows_per_batch=1, out=None,
                   sched: int | None = None):
        """``op_gemm`` on a named kernel / call-site instantiation (``ecad_op_gemm_ex``); ``a`` and ``out`` may be column slices (row stride > width).
        ``sched``: force K-loop schedule variant 0 / 1 of the sequential-order kernels (default 1: staggered LDS-DMA issue); + 4: gemm2w's direct (register)
        epilogue for every epilogue kind, + 8: its staged epilogue (default: direct for the GELU epilogue)."""
        M, K = a.shape
        N = w.shape[0]
        if out is None and epi != 3:
            out = torch.empty((M, N), device=self.device, dtype=self.dtype)
        self._chk(
            self.lib.ecad_op_gemm_ex(self._h, self.GEMM_KERNELS[kernel] | (0 if sched is None else (int(sched) + 1) << 8), self.GEMM_SITES[site], epi, a.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(out), M, N, K,
                                     a.stride(0), out.stride(0) if out is not None else 0, _ptr(resid), _ptr(gate_table), _ptr(gate_mod),
                                     int(gate_mod_stride), int(rows_per_batch), _stream_ptr())
        )
        return out

    def op_ln_modulate(self, x, table_shift, table_scale, mod_shift, mod_scale, mod_stride, rows_per_batch):
        M, D = x.shape
        y = torch.empty_like(x)
        self._chk(self.lib.ecad_op_ln_modulate(self._h, x.data_ptr(), y.data_ptr(), M, D, table_shift.data_ptr(), table_scale.data_ptr(), mod_shift.data_ptr(),
                                               mod_scale.data_ptr(), int(mod_stride), int(rows_per_batch), _stream_ptr()))
        return y

    def op_cache_accumulate(self, h, srcs, gate_tables, gate_mods, mod_stride, rows_per_batch):
        M, D = h.shape
        n = len(srcs)
        arr = lambda xs: (C.c_void_p * n)(*[None if x is None else x.data_ptr() for x in xs])
        self._chk(self.lib.ecad_op_cache_accumulate(self._h, h.data_ptr(), M, D, n, arr(srcs), arr(gate_tables), arr(gate_mods), int(mod_stride), int(rows_per_batch), _stream_ptr()))
        return h

    def op_qproj_attention(self, site: str, x, wq, bq, k, v, Bt: int, n_tokens: int, n_keys: int, key_bias=None):
        """``ecad_op_qproj_attention``: the query projection of a block with the attention of its rows in the GEMM tail (the production launch on the
        256 / 1024 / 4096-token geometries).  site: "qkv" (self-attention) | "attn2_q" (cross-attention, additive ``key_bias`` [Bt, n_keys]).  k / v: [Bt*n_keys, ld]
        views (column slices of a fused buffer allowed: head h at column h * 72 of the view)."""
        out = torch.empty((Bt * n_tokens, wq.shape[0]), device=self.device, dtype=self.dtype)
        assert k.stride(0) == v.stride(0)
        self._chk(self.lib.ecad_op_qproj_attention(self._h, self.GEMM_SITES[site], x.data_ptr(), wq.data_ptr(), _ptr(bq), k.data_ptr(), v.data_ptr(), k.stride(0), _ptr(key_bias),
                                                   out.data_ptr(), Bt, n_tokens, n_keys, _stream_ptr()))
        return out

    def op_attention(self, q, k, v, B, H, Nq, Nk, dh, bias=None, kernel: str = "auto"):
        """q: [B*Nq, ldq] view, k/v: [B*Nk, ld] views (may be column slices of a fused buffer).  kernel: auto | generic (attention.h) | shared (attention128.h) | wide4 (attention_dma.h, four query fragments per wave) | w472 (attention_w472.h: one wave per SIMD, generated stream)."""
        out = torch.empty((B * Nq, H * dh), device=self.device, dtype=self.dtype)
        if kernel != "auto":
            self._chk(self.lib.ecad_op_attention_ex(self._h, {"generic": 1, "shared": 2, "wide4": 3, "w472": 4}[kernel], q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                                    out.data_ptr(), out.stride(0), B, H, Nq, Nk, dh, _ptr(bias), _stream_ptr()))
            return out
        self._chk(self.lib.ecad_op_attention(self._h, q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0),
                                             B, H, Nq, Nk, dh, _ptr(bias), _stream_ptr()))
        return out


class EngineContexts:
    """Several image batches in flight on one GPU: ``n`` execution contexts over ONE set of weights (``EcadEngine.new_context``), each with its own HIP
    stream; ``run_schedule`` calls go to the contexts in turn and return at once (the engine's whole-loop path is asynchronous: staging copies, one hipGraph
    launch, the output copy), so consecutive batches overlap on the device.  What overlaps: a batch's ~2 000 kernel boundaries (launch gap + write drain) and its
    HBM-bound kernels (LayerNorm, cache replays) run beside the other batches' GEMMs.  Measured on the PixArt-alpha 256 `ours_fast` workload:
    batch 16 +3.5 % with three contexts; batch 4 x 4 contexts 69 -> 185 images/s; batch 1 x 4 contexts 17.9 -> 67 images/s.  The reference runs one batch at a
    time (``generate_from_saved_prompts``, image_generator.py:161-186): results per batch are unchanged, bit for bit.

    ``wait()`` (or reading a result on the host) synchronises; results are ordinary tensors produced on the context's stream, so a consumer on another stream
    must wait for that stream (``torch.cuda.current_stream().wait_stream(ctx_stream)``) -- ``submit`` returns ``(tensor, stream)`` for that purpose."""

    def __init__(self, engine: EcadEngine, n: int, adaptive: bool = False, idle_turns: int = 3):
        """``adaptive`` (opt-in): a context whose batch finds every other stream idle on ``idle_turns`` consecutive submissions is told it runs alone
        (``set_batches_in_flight(1)``: full-chip launches instead of half-chip ones that wait for a neighbour) and is told ``n`` again as soon as a peer is
        busy.  Off by default: the caller built the pool to keep ``n`` batches in flight, every (schedule, context) then has ONE plan set, graph and LayerNorm-fold
        arena, and the kernel selection does not depend on timing.  With it on, a context may hold two plan sets per schedule (both count against ``ECAD_PLAN_CACHE``);
        the results are the same bits either way."""
        if n < 1:
            raise ValueError("need at least one context")
        if idle_turns < 1:
            raise ValueError("idle_turns must be at least 1")
        self.engines = [engine] + [engine.new_context() for _ in range(n - 1)]
        self.streams = [torch.cuda.Stream(device=engine.device) for _ in range(n)]
        self._turn = 0
        self.adaptive, self.idle_turns = bool(adaptive), int(idle_turns)
        for e in self.engines:  # the few-rows rule counts all the batches in flight: latency tiles only while the chip would otherwise idle
            e.set_batches_in_flight(n)
        self._declared = [n] * n  # what each context was last told
        self._idle_seen = [0] * n  # consecutive submissions of a context that found every peer idle (adaptive mode)

    def declared_in_flight(self) -> list[int]:
        """What each context currently plans for (``n``, or 1 after ``idle_turns`` lonely submissions in adaptive mode)."""
        return list(self._declared)

    def __len__(self) -> int:
        return len(self.engines)

    def set_schedule(self, bits) -> None:
        for e in self.engines:
            e.set_schedule(bits)

    def submit(self, enc, mask, latents, timesteps, solver_coef, guidance_scale: float = 4.5, use_graph: bool = True):
        """``run_schedule`` of the next context in turn, on that context's stream; returns (final latents, stream) without waiting."""
        k = self._turn % len(self.engines)
        self._turn += 1
        st = self.streams[k]
        if self.adaptive and len(self.engines) > 1:
            # what this batch will run beside: with another context's batch still on the device the engine may launch half-chip kernels that share the chip with it
            # (DESIGN 5a); a context that keeps finding the GPU to itself gets the full-chip ones -- after `idle_turns` such submissions in a row, never on the first
            # turn of a pool (whose peers are idle only because they have not been fed yet), and back to n on the first busy peer
            busy = any(not other.query() for j, other in enumerate(self.streams) if j != k)
            self._idle_seen[k] = 0 if busy else self._idle_seen[k] + 1
            want = 1 if self._idle_seen[k] >= self.idle_turns else len(self.engines)
            if self._declared[k] != want:
                self.engines[k].set_batches_in_flight(want)
                self._declared[k] = want
        st.wait_stream(torch.cuda.current_stream(self.engines[k].device))  # inputs produced on the caller's stream are ready
        for t in (enc, mask, latents):  # and stay allocated until this stream has read them, even if the caller drops them right away
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        with torch.cuda.stream(st):
            out = self.engines[k].run_schedule(enc, mask, latents, timesteps, solver_coef, guidance_scale, use_graph=use_graph)
        return out, st

    def wait(self) -> None:
        for st in self.streams:
            st.synchronize()

    def close(self) -> None:
        for st in self.streams:  # a context's graph may still be running: its engine (hipGraphExec, workspaces) goes only after its stream has drained
            st.synchronize()
        for e in self.engines[1:]:
            e.close()
        self.engines = self.engines[:1]
        if self.engines[0]._h and self.engines[0]._h.value:
            self.engines[0].set_batches_in_flight(1)
