"""Latency harness: same flags, same warm-up / sample protocol and the same ``metrics.latency`` block written into the
schedule JSON as ``ecad/benchmark/compute_latency.py:21-241`` (ms per image around one full pipeline call).

``--output-type latent`` (default here) times transformer + solver only; ``pil`` includes the decode tail the reference
times too (an attached VAE when available, else the labelled preview projection).
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path

import torch

from ecad_amd.image_generators.load_image_generator import ImageGeneratorRegistry, get_image_generator_type


def time_for_schedule(image_generator_type, schedule_file: Path, embedding_dir: Path, batch_size: int, warmup_steps: int, num_samples: int,
                      seed: int, recompute_existing: bool, output_type: str = "latent"):
    with open(schedule_file) as f:
        data = json.load(f)
    if not recompute_existing and data.get("metrics", {}).get("latency") is not None:
        print(f"Metrics already computed for {schedule_file}. Skipping.")
        return None
    gen = image_generator_type(start_seed=seed, seed_step=0, schedule_path=schedule_file, max_batch=2 * batch_size)
    times = gen.time_image_generation(embedding_dir, batch_size=batch_size, num_batches=warmup_steps + num_samples, free_after=True, output_type=output_type)
    warmups, latencies = times[:warmup_steps], times[warmup_steps:]
    out = {"avg": sum(latencies) / len(latencies), "batch_size": batch_size, "num_samples": num_samples, "warmup_steps": warmup_steps,
           "gpu": torch.cuda.get_device_name(torch.cuda.current_device()), "warmups": warmups, "latencies": latencies, "output_type": output_type}
    with open(schedule_file) as f:  # reopen in case it changed
        data = json.load(f)
    data.setdefault("metrics", {})["latency"] = out
    with open(schedule_file, "w") as f:
        json.dump(data, f, indent=4)
    print(f"{schedule_file.stem}: {out['avg']:.3f} ms/image ({1000.0 / out['avg']:.1f} images/s)")
    return out


def main():
    p = argparse.ArgumentParser(description="Compute the latency of a model on a given schedule.")
    p.add_argument("--image-generator", type=str, required=True, choices=list(ImageGeneratorRegistry.registry.keys()))
    p.add_argument("-i", "--embedding-dir", type=Path, required=True)
    p.add_argument("-d", "--schedule-dir", type=Path)
    p.add_argument("--schedule-list", nargs="+", required=False)
    p.add_argument("-b", "--batch-size", type=int, default=100)
    p.add_argument("-w", "--warmup-steps", type=int, default=10)
    p.add_argument("-n", "--num-samples", type=int, default=5)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--recompute-existing", action="store_true")
    p.add_argument("--output-type", default="latent", choices=["latent", "pil"])
    a = p.parse_args()
    if not a.embedding_dir.exists():
        raise FileNotFoundError(f"Input directory {a.embedding_dir} not found.")
    gen_type = get_image_generator_type(a.image_generator)
    files: list[Path] = []
    if a.schedule_dir is not None:
        files += [a.schedule_dir] if a.schedule_dir.is_file() else sorted(a.schedule_dir.glob("**/*.json"))
    for s in a.schedule_list or []:
        files += [Path(x) for x in s.split(",") if x]
    if not files:
        raise ValueError("no schedules given (--schedule-dir / --schedule-list)")
    for f in files:
        time_for_schedule(gen_type, f, a.embedding_dir, a.batch_size, a.warmup_steps, a.num_samples, a.seed, a.recompute_existing, a.output_type)
    print("Done.")


if __name__ == "__main__":
    main()
