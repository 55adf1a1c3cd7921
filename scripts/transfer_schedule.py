#!/usr/bin/env python3
"""Resolution transfer of a cache schedule: the (step, block, component) bits found at 256 x 256 are kept, the top-level `config` names the
high-resolution checkpoint / size, and the MAC metrics are re-annotated for the new geometry -- what the reference's paper schedules under
`schedules/schedules_in_paper/flux_256_to_1024/` are (same `cache_schedule` body as `flux_256/*`, `config` {height, width 1024}, metrics of the
1024 run).  The reference ships no PixArt-Sigma 1024 file; BASELINE config 3 names "sigma_cache_schedules best" at 1024 x 1024, so this authors it:

    python scripts/transfer_schedule.py schedules/schedules_in_paper/pixart_sigma_256/ours_fast.json \\
        schedules/sigma_cache_schedules/transfer_256_to_1024/ours_fast_256_to_1024.json \\
        --weights PixArt-alpha/PixArt-Sigma-XL-2-1024-MS --size 1024 --image-generator PixArtSigmaImageGenerator

The `config` keys are the ones `ImageGenerator._load_config` / `PixArtImageGenerator._load_subclass_config_defaults` read
(reference image_generator.py:172-186, pixart_image_generator.py: height / width; `image_generator` is what
`get_image_generator_type_from_config` looks up)."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src", type=Path)
    ap.add_argument("dst", type=Path)
    ap.add_argument("--weights", required=True)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--image-generator", default=None)
    args = ap.parse_args()
    from ecad_amd.benchmark.compute_macs import annotate_schedule_file

    data = json.loads(args.src.read_text())
    cfg = dict(data.get("config") or {})
    cfg.update(transformer_weights=args.weights, height=args.size, width=args.size)
    if args.image_generator:
        cfg["image_generator"] = args.image_generator
    data["config"] = cfg
    data["metrics"] = {"transferred_from": str(args.src), "source_total_macs_T": (data.get("metrics") or {}).get("total_macs_T")}
    args.dst.parent.mkdir(parents=True, exist_ok=True)
    args.dst.write_text(json.dumps(data, indent=4))
    annotate_schedule_file(args.dst, recompute_existing=True)
    m = json.loads(args.dst.read_text())["metrics"]
    print(f"{args.dst}: {m['total_macs_T']:.3f} TMACs per image at {args.size} x {args.size} (source: {m['source_total_macs_T']})")


if __name__ == "__main__":
    main()
