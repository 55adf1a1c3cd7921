"""Closed-form MAC accounting for PixArt cache schedules.

Replaces the reference's per-schedule calflops pass (``ecad/benchmark/compute_macs.py:255-303``: build zero inputs at
batch 2, hook every nn.Linear/nn.Conv, run 20 forwards) with the closed form that reproduces every per-step value in
the reference's 1090 PixArt-256 schedule JSONs exactly (tests/test_cpu_host.py).  Writes the same ``metrics`` block
(``by_inference_step``, ``total_macs``, ``total_macs_T``, ``total_flops`` omitted: calflops-specific) into the JSON.
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path

import numpy as np

from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule

BATCH_SIZE = 2  # one image with CFG, as the reference counts (compute_macs.py:40)


def executed_components(bits: np.ndarray) -> np.ndarray:
    """recompute-or-cache-empty (cached_transformer_block.py:340-347), resolved over a whole image batch."""
    bits = np.asarray(bits, dtype=np.bool_)
    ex = np.zeros_like(bits)
    have = np.zeros(bits.shape[1:], dtype=np.bool_)
    for s in range(bits.shape[0]):
        ex[s] = bits[s] | ~have
        have |= ex[s]
    return ex


def macs_per_step(bits: np.ndarray, tokens: int = 256, dim: int = 1152, text_tokens: int = 120, caption_channels: int = 4096,
                  size_embedder_macs: int = 0, batch: int = BATCH_SIZE) -> np.ndarray:
    n, d, t = tokens, dim, text_tokens
    cost = np.array([4 * n * d * d, 2 * n * d * d + 2 * t * d * d, 8 * n * d * d], dtype=np.int64)
    overhead = t * (caption_channels * d + d * d) + n * 16 * d + (256 * d + 7 * d * d) + n * d * 32 + size_embedder_macs
    ex = executed_components(bits).astype(np.int64)
    return batch * (overhead + (ex * cost[None, None, :]).sum(axis=(1, 2)))


def total_macs_T(bits: np.ndarray, **kw) -> float:
    return float(macs_per_step(bits, **kw).sum()) / 1e12


def annotate_schedule_file(path: Path, recompute_existing: bool = False) -> bool:
    with open(path) as f:
        data = json.load(f)
    metrics = data.get("metrics", {}) or {}
    if not recompute_existing and metrics.get("total_macs_T") is not None:
        return False
    sched = PixArtCacheSchedule.from_json(path)
    per_step = macs_per_step(sched.to_numpy())
    metrics["by_inference_step"] = {f"{s:03}": {"macs": int(m)} for s, m in enumerate(per_step)}
    metrics["total_macs"] = int(per_step.sum())
    metrics["total_macs_T"] = float(per_step.sum()) / 1e12
    data["metrics"] = metrics
    with open(path, "w") as f:
        json.dump(data, f, indent=4)
    return True


def main():
    ap = argparse.ArgumentParser(description="Closed-form MACs for PixArt cache schedules (writes metrics into the JSON).")
    ap.add_argument("--input-dir", type=Path, required=True, help="schedule file or directory of schedule JSONs")
    ap.add_argument("--recompute-existing", action="store_true")
    args = ap.parse_args()
    files = [args.input_dir] if args.input_dir.is_file() else sorted(args.input_dir.glob("**/*.json"))
    for f in files:
        print(("updated " if annotate_schedule_file(f, args.recompute_existing) else "skipped ") + str(f))


if __name__ == "__main__":
    main()
