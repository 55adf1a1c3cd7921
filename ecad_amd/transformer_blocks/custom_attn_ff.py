"""Per-block compute plug-in registries.

Same surface as the reference (``ecad/transformer_blocks/custom_attn_ff.py:6-59``): ``ComputeAttnRegistry.register`` /
``.get(name)`` keyed by lower-cased function name, defaults ``compute_attn_cached`` / ``compute_ff_cached`` (registered by
``ecad_amd.transformer_blocks.cached_transformer_block``, as the reference's block module registers them).

A registered function has the reference's signature and receives the block view of the native engine
(``CachedTransformerBlock``: ``block_num``, ``cache_schedule``, ``cached_*_output``, and ``attn1`` / ``attn2`` / ``ff`` that run the
HIP sub-layers on device tensors):

    @ComputeFFRegistry.register
    def my_ff(block, norm_hidden_states, **kwargs) -> torch.Tensor: ...
    @ComputeAttnRegistry.register
    def my_attn(block, attn, hidden_states, encoder_hidden_states, attention_mask, **kwargs) -> torch.Tensor: ...

A schedule that names only the defaults never leaves the device (the gate is compiled into the engine's op lists); a block whose
schedule names anything else is driven through these Python callables in per-step mode (``ecad_set_custom_components``).
There is no CPU path: the sub-layers the callables can reach are the engine's.

Native variants (round 6): a registered function may carry ``native_variant = <ECAD_NATIVE_* id>`` -- a gate rule the engine lowers
into its own kernels (``ecad_set_native_variants``), so a schedule that names it runs in per-step mode AND inside the whole-loop
hipGraph, where a Python callable cannot be called.  ``compute_attn_recompute`` / ``compute_ff_recompute`` (ignore the flag: always
evaluate) and ``compute_attn_cfg_mean_cache`` (attn2 caches the mean of the CFG halves -- TGATE's caching rule) ship registered; the
callables themselves stay real (another plug-in may call them, and a block that also names a host-only function runs them from Python).
"""
from __future__ import annotations

from typing import Callable


class ComputeRegistry:
    _registry: dict[str, Callable] = {}

    @classmethod
    def register(cls, func: Callable) -> Callable:
        cls._registry[func.__name__.lower()] = func
        return func

    @classmethod
    def get(cls, key: str | None, none_if_not_found: bool = False) -> Callable | None:
        if key is not None and key.lower() in cls._registry:
            return cls._registry[key.lower()]
        return None if none_if_not_found else cls.default()

    @classmethod
    def default(cls) -> Callable:
        if not hasattr(cls, "DEFAULT"):
            raise NotImplementedError("Subclasses must define a DEFAULT attribute.")
        func = cls._registry.get(getattr(cls, "DEFAULT"))
        if func is None:
            raise ValueError(f"Default function {getattr(cls, 'DEFAULT')!r} not registered.")
        return func

    @classmethod
    def is_native(cls, key: str | None) -> bool:
        """True for the names the engine compiles into its op lists: absent name, the cached default, or a function that declares a native variant."""
        return cls.native_variant(key) is not None

    @classmethod
    def native_variant(cls, key: str | None) -> int | None:
        """The ECAD_NATIVE_* id the engine runs for ``key`` (0 for an absent name / the default), or None for a function only Python can run."""
        if key is None or key.lower() == getattr(cls, "DEFAULT"):
            return 0
        func = cls._registry.get(key.lower())
        kind = getattr(func, "native_variant", None) if func is not None else None
        return int(kind) if kind is not None else None


class ComputeAttnRegistry(ComputeRegistry):
    _registry: dict[str, Callable] = {}
    DEFAULT = "compute_attn_cached"


class ComputeFFRegistry(ComputeRegistry):
    _registry: dict[str, Callable] = {}
    DEFAULT = "compute_ff_cached"


from ecad_amd.transformer_blocks import cached_transformer_block as _defaults  # noqa: E402

_defaults._register_defaults()
