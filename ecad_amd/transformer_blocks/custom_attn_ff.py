"""Per-block compute plug-in registries.

Same surface as the reference (``ecad/transformer_blocks/custom_attn_ff.py:6-59``):
``ComputeAttnRegistry.register`` / ``.get(name)`` keyed by lower-cased function name,
defaults ``compute_attn_cached`` / ``compute_ff_cached``.  In this engine the two
defaults are NATIVE: they are what the HIP engine implements for every block, and the
registered Python callables are markers (calling one raises -- there is no CPU path).
A schedule that names any other function cannot run on the native path and is
rejected at ``ecad_set_schedule`` time by the host wrapper.
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
        return key is None or key.lower() == getattr(cls, "DEFAULT")


class ComputeAttnRegistry(ComputeRegistry):
    _registry: dict[str, Callable] = {}
    DEFAULT = "compute_attn_cached"


class ComputeFFRegistry(ComputeRegistry):
    _registry: dict[str, Callable] = {}
    DEFAULT = "compute_ff_cached"


def _native_only(name: str):
    def marker(*args, **kwargs):
        raise RuntimeError(
            f"{name} is implemented natively inside the HIP engine (libecad_hip.so); "
            "it is not callable from Python and has no CPU fallback."
        )

    marker.__name__ = name
    return marker


ComputeAttnRegistry.register(_native_only("compute_attn_cached"))
ComputeFFRegistry.register(_native_only("compute_ff_cached"))
