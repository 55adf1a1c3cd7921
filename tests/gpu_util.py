"""Shared helpers for the -m gpu parity tests (everything goes through the C ABI)."""
import functools

import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@functools.lru_cache(maxsize=None)
def small_engine(dtype_name: str = "float16", layers: int = 2, max_batch: int = 4, max_tokens: int = 256, max_text: int = 120):
    from ecad_amd.engine import EcadEngine

    return EcadEngine(num_layers=layers, max_batch=max_batch, max_tokens=max_tokens, max_text_tokens=max_text, dtype=getattr(torch, dtype_name))


def within(value: float, tol: float) -> bool:
    """value < tol, and the measured value on stdout (`pytest -s`): the tolerances in these files are kept at <= 2x what this prints on an MI355X."""
    import inspect

    fr = inspect.stack()[1]
    print(f"[parity] {fr.function}:{fr.lineno} measured {value:.3e} tol {tol:.1e}")
    return value < tol
