"""ctypes binding of libecad_hip.so (the C ABI in include/ecad_hip.h).

There is NO fallback: if the shared library is missing or does not export the ABI,
importing a symbol raises immediately (``EcadLibraryError``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_NAME = "libecad_hip.so"
ABI_VERSION = 7

ECAD_DTYPE_F16, ECAD_DTYPE_BF16, ECAD_DTYPE_F32 = 0, 1, 2
ECAD_E_INVALID, ECAD_E_HIP, ECAD_E_NOT_READY, ECAD_E_UNSUPPORTED = -1, -2, -3, -4


class EcadLibraryError(RuntimeError):
    pass


class EcadModelCfg(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("patch_size", C.c_int32),
        ("sample_size", C.c_int32),
        ("caption_channels", C.c_int32),
        ("max_batch", C.c_int32),
        ("max_tokens", C.c_int32),
        ("max_text_tokens", C.c_int32),
        ("dtype", C.c_int32),
        ("norm_eps", C.c_float),
        ("interpolation_scale", C.c_float),
        ("use_additional_conditions", C.c_int32),
    ]


class EcadStats(C.Structure):
    _fields_ = [
        ("components_executed", C.c_int64),
        ("components_cached", C.c_int64),
        ("linear_macs_reference", C.c_int64),
        ("linear_macs_executed", C.c_int64),
        ("attn_core_macs", C.c_int64),
        ("cache_bytes_read", C.c_int64),
        ("cache_bytes_written", C.c_int64),
        ("cache_arena_bytes", C.c_int64),
        ("weight_bytes", C.c_int64),
        ("workspace_bytes", C.c_int64),
        ("graph_nodes", C.c_int32),
        ("launches_per_batch", C.c_int32),
        ("plan_builds", C.c_int32),
        ("graph_captures", C.c_int32),
        ("dynamic_step_plans", C.c_int32),
        ("plan_sets_cached", C.c_int32),
        ("ln_folded", C.c_int32),
        ("fold_preps", C.c_int32),
        ("fold_bytes", C.c_int64),
        ("fold_arena_bytes", C.c_int64),
        ("fold_refused", C.c_int32),
        ("replays_folded", C.c_int32),
        ("rmw_launches_wide", C.c_int64),
        ("rmw_launches_narrow4", C.c_int64),
        ("rmw_launches_gemm3s", C.c_int64),
        ("split_k", C.c_int32),
        ("batches_in_flight", C.c_int32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class EcadFluxCfg(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32),
        ("num_single_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("in_channels", C.c_int32),
        ("joint_attention_dim", C.c_int32),
        ("pooled_projection_dim", C.c_int32),
        ("guidance_embeds", C.c_int32),
        ("axes_dims_rope", C.c_int32 * 3),
        ("max_batch", C.c_int32),
        ("max_image_tokens", C.c_int32),
        ("max_text_tokens", C.c_int32),
        ("dtype", C.c_int32),
        ("norm_eps", C.c_float),
    ]


class EcadFluxStats(C.Structure):
    _fields_ = [
        ("components_executed", C.c_int64),
        ("components_cached", C.c_int64),
        ("components_dead", C.c_int64),
        ("linear_macs_reference", C.c_int64),
        ("linear_macs_executed", C.c_int64),
        ("attn_core_macs", C.c_int64),
        ("cache_arena_bytes", C.c_int64),
        ("weight_bytes", C.c_int64),
        ("workspace_bytes", C.c_int64),
        ("graph_nodes", C.c_int32),
        ("launches_per_batch", C.c_int32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


_P = C.c_void_p
_I = C.c_int
_I64 = C.c_int64
_F = C.c_float

# name -> (restype, argtypes); every symbol include/ecad_hip.h and include/ecad_flux.h declare
SIGNATURES = {
    "ecad_abi_version": (_I, []),
    "ecad_create": (_I, [C.POINTER(EcadModelCfg), _I, C.POINTER(_P)]),
    "ecad_destroy": (None, [_P]),
    "ecad_last_error": (C.c_char_p, [_P]),
    "ecad_load_weights": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_I64), _I]),
    "ecad_finalize_weights": (_I, [_P]),
    "ecad_share_weights": (_I, [_P, _P]),
    "ecad_set_ln_fold": (_I, [_P, _I]),
    "ecad_set_batches_in_flight": (_I, [_P, _I]),
    "ecad_set_split_k": (_I, [_P, _I]),
    "ecad_set_native_variants": (_I, [_P, C.POINTER(C.c_uint8), _I, _I]),
    "ecad_set_fp8": (_I, [_P, _I]),
    "ecad_set_added_cond": (_I, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), _I]),
    "ecad_set_attention_bias": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ecad_set_schedule": (_I, [_P, C.POINTER(C.c_uint8), _I, _I]),
    "ecad_begin_image_batch": (_I, [_P, _I, _I, _I, _P, _P, _I, _P]),
    "ecad_rebind_batch_inputs": (_I, [_P, _I, _I, _I, _P, _P, _I, _P]),
    "ecad_step": (_I, [_P, _I, _P, _P, _P, _P]),
    "ecad_run_schedule": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, _I, _P]),
    "ecad_reset_cache": (_I, [_P]),
    "ecad_set_custom_components": (_I, [_P, C.POINTER(C.c_uint8), _I, _P, _P, _P, _P]),
    "ecad_block_attn": (_I, [_P, _I, _I, _P, _P, _P]),
    "ecad_block_ff": (_I, [_P, _I, _P, _P, _P]),
    "ecad_copy_batch_context": (_I, [_P, _P, _P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _P]),
    "ecad_cache_stats": (_I, [_P, C.POINTER(EcadStats)]),
    "ecad_set_profiling": (_I, [_P, _I]),
    "ecad_get_profile": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ecad_op_gemm": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I64, _I, _P]),
    "ecad_op_gemm_ex": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I64, _I64, _P, _P, _P, _I64, _I, _P]),
    "ecad_op_attention_ex": (_I, [_P, _I, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _I, _I, _I, _I, _P, _P]),
    "ecad_op_qproj_attention": (_I, [_P, _I, _P, _P, _P, _P, _P, _I64, _P, _P, _I, _I, _I, _P]),
    "ecad_op_ln_modulate": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _I64, _I, _P]),
    "ecad_op_cache_accumulate": (_I, [_P, _P, _I, _I, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _I64, _I, _P]),
    "ecad_op_attention": (_I, [_P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _I, _I, _I, _I, _P, _P]),
    "ecad_vae_create": (_I, [_I, _I, C.POINTER(_P)]),
    "ecad_vae_destroy": (None, [_P]),
    "ecad_vae_last_error": (C.c_char_p, [_P]),
    "ecad_vae_load_weights": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_I64), _I]),
    "ecad_vae_finalize_weights": (_I, [_P]),
    "ecad_vae_decode": (_I, [_P, _P, _I, _I, _I, C.c_float, _P, _P]),
    "ecad_vae_create_ex": (_I, [_I, _I, _I, _I, C.POINTER(_P)]),
    "ecad_vae_decode_ex": (_I, [_P, _P, _I, _I, _I, C.c_float, C.c_float, _P, _P]),
    "ecad_vae_decode_stats": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_I), C.POINTER(_I)]),
    "ecad_vae_decode_stats_ex": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(C.c_double)]),
    # include/ecad_flux.h
    "ecad_flux_create": (_I, [C.POINTER(EcadFluxCfg), _I, C.POINTER(_P)]),
    "ecad_flux_destroy": (None, [_P]),
    "ecad_flux_last_error": (C.c_char_p, [_P]),
    "ecad_flux_load_weights": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_I64), _I]),
    "ecad_flux_finalize_weights": (_I, [_P]),
    "ecad_flux_set_fp8": (_I, [_P, _I]),
    "ecad_flux_set_schedule": (_I, [_P, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), _I]),
    "ecad_flux_begin_image_batch": (_I, [_P, _I, _I, _I, _P, _P, _I, _P]),
    "ecad_flux_step": (_I, [_P, _I, _P, _F, _F, _P, _P]),
    "ecad_flux_run_schedule": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, C.POINTER(_F), _F, C.POINTER(_F), _I, _P]),
    "ecad_flux_reset_cache": (_I, [_P]),
    "ecad_flux_stats": (_I, [_P, C.POINTER(EcadFluxStats)]),
    "ecad_flux_set_profiling": (_I, [_P, _I]),
    "ecad_flux_get_profile": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ecad_flux_op_gemm": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _I64, _I, _I, _I, _P]),
    "ecad_flux_op_gemm_fp8": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(_F), _P]),
    "ecad_flux_op_qk_norm_rope": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "ecad_flux_op_attention": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ecad_flux_op_ln_modulate": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _I64, _P]),
}

# EcadComponentFn (include/ecad_hip.h): int fn(void* user, int step, int block, int component, const void* in, void* out, int Bt, int N, int D, void* stream)
COMPONENT_FN = C.CFUNCTYPE(_I, _P, _I, _I, _I, _P, _P, _I, _I, _I, _P)

_lib = None


def lib_path() -> Path:
    env = os.environ.get("ECAD_HIP_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / LIB_NAME


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol; raise loudly otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise EcadLibraryError(
            f"{path} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C ecad_amd/csrc`. There is no CPU fallback for the hot path."
        )
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # e.g. libamdhip64 missing
        raise EcadLibraryError(f"failed to load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EcadLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ecad_abi_version() != ABI_VERSION:
        raise EcadLibraryError(f"ABI version mismatch: library {lib.ecad_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib
