"""One-shot inference CLI with the reference's flags (``ecad/inference/inference.py:18-224``).

Two stated differences: (1) ``--schedule`` IS forwarded to the image generator (the reference parses it at :33-39 but
never passes it at :168, so it always runs uncached); (2) ``--prompt`` / ``--prompt-file`` need the T5 text encoder,
whose checkpoint is not available offline -- use ``--input-embeddings`` (reference ``.pt`` format).
"""
from __future__ import annotations

import argparse
from datetime import datetime
from pathlib import Path

from ecad_amd.image_generators.load_image_generator import ImageGeneratorRegistry, get_image_generator_type

DEFAULT_OUTPUT_DIR = Path("results/inference")


def _check_guidance(gen_type, guidance_scale) -> None:
    """Reference inference.py:208-214: FLUX takes ``--guidance-scale``; the PixArt generators always run CFG 4.5 and refuse the flag."""
    from ecad_amd.image_generators.pixart_image_generator import PixArtImageGenerator

    if guidance_scale is not None and issubclass(gen_type, PixArtImageGenerator):
        raise ValueError("PixArtImageGenerator does not support guidance scale. CFG is always enabled.")


def main(argv=None) -> None:
    p = argparse.ArgumentParser(description="Inference with prompts or embeddings.", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("image_generator", metavar="image-generator", type=str, choices=list(ImageGeneratorRegistry.registry.keys()))
    p.add_argument("-s", "--schedule", type=Path, required=False)
    g = p.add_mutually_exclusive_group(required=True)
    g.add_argument("-p", "--prompt", type=str)
    g.add_argument("-f", "--prompt-file", type=Path)
    g.add_argument("-i", "--input-embeddings", type=Path)
    p.add_argument("-e", "--embedding-name", type=Path, default=Path("embedding.pt"))
    p.add_argument("-o", "--output-dir", type=Path, default=DEFAULT_OUTPUT_DIR / datetime.now().replace(microsecond=0).isoformat())
    p.add_argument("-n", "--num-images-per-prompt", type=int, default=10)
    p.add_argument("--start-seed", type=int, default=0)
    p.add_argument("--seed-step", type=int, default=1)
    p.add_argument("--batch-size-embedding", type=int, default=1)
    p.add_argument("--batch-size-generate", type=int, default=1)
    p.add_argument("--height", type=int)
    p.add_argument("--width", type=int)
    p.add_argument("--guidance-scale", type=float)
    a = p.parse_args(argv)
    if a.schedule is not None and not a.schedule.exists():
        raise ValueError(f"Schedule file {a.schedule} not found.")
    gen_type = get_image_generator_type(a.image_generator)
    _check_guidance(gen_type, a.guidance_scale)  # before any engine is created
    gen = gen_type(start_seed=a.start_seed, seed_step=a.seed_step, schedule_path=a.schedule, max_batch=2 * a.batch_size_generate)
    if a.input_embeddings is None:
        gen.create_encoder_pipeline()  # raises with the explanation above
    kw = {k: v for k, v in (("height", a.height), ("width", a.width)) if v is not None}
    if a.guidance_scale is not None:
        kw["guidance_scale"] = a.guidance_scale
    gen.generate_from_saved_prompts(a.input_embeddings, a.output_dir / "images", batch_size=a.batch_size_generate,
                                    images_per_prompt=a.num_images_per_prompt, free_after=True, **kw)
    print(f"Images saved under {a.output_dir / 'images'}")


if __name__ == "__main__":
    main()
