"""Closed-form MAC accounting for PixArt and FLUX cache schedules.

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


def flux_macs_per_step(bits_full: np.ndarray, bits_single: np.ndarray, tokens: int, dim: int = 3072, text_tokens: int = 512, in_channels: int = 64,
                       joint_attention_dim: int = 4096, pooled_dim: int = 768, guidance_embeds: bool = True, batch: int = BATCH_SIZE) -> np.ndarray:
    """FLUX (6 component kinds; flux_cache_schedule.py:51-60): reproduces the per-step MACs of all 38 FLUX schedule JSONs of the
    reference that carry metrics (tests/test_cpu_flux.py).  tokens = (height / 16) * (width / 16)."""
    n, t, d = tokens, text_tokens, dim
    s = n + t
    nb, ns = bits_full.shape[1], bits_single.shape[1]
    cost_full = np.array([4 * s * d * d, 8 * n * d * d, 8 * t * d * d], dtype=np.int64)
    cost_single = np.array([3 * s * d * d, 4 * s * d * d, 5 * s * d * d], dtype=np.int64)
    overhead = (n * in_channels * d + t * joint_attention_dim * d + (2 if guidance_embeds else 1) * (256 * d + d * d) + (pooled_dim * d + d * d)
                + nb * 12 * d * d + ns * 3 * d * d + 2 * d * d + n * d * in_channels)
    ef, es = executed_components(bits_full).astype(np.int64), executed_components(bits_single).astype(np.int64)
    return batch * (overhead + (ef * cost_full[None, None, :]).sum(axis=(1, 2)) + (es * cost_single[None, None, :]).sum(axis=(1, 2)))


# geometry per checkpoint name, as the reference's WEIGHTS_TO_SHAPES table has it (compute_macs.py:22-39): latent side multiplier over 32 x 32
# and whether the micro-conditioning embedders run (text length is 120 for all three there, SEQ_LEN :42)
WEIGHTS_TO_GEOMETRY = {
    "PixArt-alpha/PixArt-XL-2-256x256": dict(tokens=256, size_embedder_macs=0),
    "PixArt-alpha/PixArt-Sigma-XL-2-256x256": dict(tokens=256, size_embedder_macs=0),
    "PixArt-alpha/PixArt-XL-2-1024-MS": dict(tokens=4096, size_embedder_macs=3 * (256 * 384 + 384 * 384)),
    # not in the reference's table (it never annotated a Sigma-1024 file): same convention -- latent side x4, text length SEQ_LEN = 120, and the
    # Sigma 1024-MS checkpoint has no micro-conditioning embedders (use_additional_conditions False)
    "PixArt-alpha/PixArt-Sigma-XL-2-1024-MS": dict(tokens=4096, size_embedder_macs=0),
}
DEFAULT_WEIGHTS = {"PixArtAlphaImageGenerator": "PixArt-alpha/PixArt-XL-2-256x256", "PixArtSigmaImageGenerator": "PixArt-alpha/PixArt-Sigma-XL-2-256x256"}


def annotate_schedule_file(path: Path, recompute_existing: bool = False, image_generator: str | None = None, print_detailed: bool = False) -> bool:
    with open(path) as f:
        data = json.load(f)
    metrics = data.get("metrics", {}) or {}
    if not recompute_existing and metrics.get("total_macs_T") is not None:
        return False
    if "num_single_blocks" in data.get("cache_schedule", {}):
        from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule

        fs = FluxCacheSchedule.from_json(path)
        cfg = fs.top_level_config or {}
        per_step = flux_macs_per_step(*fs.to_device_bits(), tokens=(cfg.get("height", 256) // 16) * (cfg.get("width", 256) // 16))
    else:
        sched = PixArtCacheSchedule.from_json(path)
        weights = (sched.top_level_config or {}).get("transformer_weights") or DEFAULT_WEIGHTS.get(image_generator or "PixArtAlphaImageGenerator")
        if weights not in WEIGHTS_TO_GEOMETRY:
            raise ValueError(f"Shapes for weights {weights} not found.")
        per_step = macs_per_step(sched.to_numpy(), **WEIGHTS_TO_GEOMETRY[weights])
    if print_detailed:
        for step, macs in enumerate(per_step):
            print(f"  step {step:03}: {int(macs):>16d} MACs")
    metrics["by_inference_step"] = {f"{s:03}": {"macs": int(m)} for s, m in enumerate(per_step)}
    metrics["total_macs"] = int(per_step.sum())
    metrics["total_macs_T"] = float(per_step.sum()) / 1e12
    data["metrics"] = metrics
    with open(path, "w") as f:
        json.dump(data, f, indent=4)
    return True


def get_argparser() -> argparse.ArgumentParser:
    """Flags of the reference's compute_macs.py (:371-397)."""
    ap = argparse.ArgumentParser(description="Closed-form MACs for PixArt / FLUX cache schedules (writes the metrics block into each JSON).")
    ap.add_argument("--image-generator", type=str, required=False, choices=["PixArtAlphaImageGenerator", "PixArtSigmaImageGenerator", "FluxImageGenerator"],
                    help="model family for schedule files whose config block does not name the weights")
    ap.add_argument("--input-dir", "-i", type=Path, required=True, help="schedule file or directory of schedule JSONs (searched recursively)")
    ap.add_argument("--recompute-existing", action="store_true", help="also files that already carry metrics")
    ap.add_argument("--print-detailed", action="store_true", help="print the MACs of every step")
    return ap


def main():
    args = get_argparser().parse_args()
    files = [args.input_dir] if args.input_dir.is_file() else sorted(args.input_dir.glob("**/*.json"))
    for f in files:
        print(("updated " if annotate_schedule_file(f, args.recompute_existing, args.image_generator, args.print_detailed) else "skipped ") + str(f))


if __name__ == "__main__":
    main()
