"""schedule-dir x embedding-dir -> PNG tree (flags and skip/regenerate rules of ``ecad/benchmark/generate_images.py:13-232``)."""
from __future__ import annotations

import argparse
import shutil
from pathlib import Path

from ecad_amd.image_generators.load_image_generator import ImageGeneratorRegistry, get_image_generator_type


def generate_for_schedule(gen_type, schedule_file: Path, input_dir: Path, output_dir: Path, batch_size: int, n_per_prompt: int, seed: int,
                          seed_step: int, regen_if_not_n: int | None, dont_include_seed: bool) -> None:
    out = output_dir / schedule_file.stem
    if out.exists():
        n_png = len(list(out.glob("**/*.png")))
        if regen_if_not_n is None or n_png == regen_if_not_n:
            print(f"Skipping {schedule_file.stem}: {n_png} images already present.")
            return
        shutil.rmtree(out)  # wrong count: wipe and regenerate (reference :27-43)
    gen = gen_type(start_seed=seed, seed_step=seed_step, schedule_path=schedule_file, max_batch=2 * batch_size)
    gen.generate_from_saved_prompts(input_dir, out, batch_size=batch_size, images_per_prompt=n_per_prompt, free_after=True,
                                    include_seed_in_name=not dont_include_seed)


def main():
    p = argparse.ArgumentParser(description="Use prompt embeddings to generate images; one sub-directory per schedule JSON.")
    p.add_argument("--image-generator", type=str, required=True, choices=list(ImageGeneratorRegistry.registry.keys()))
    p.add_argument("-i", "--input-dir", type=Path, required=True)
    p.add_argument("-o", "--output-dir", type=Path, required=True)
    p.add_argument("-d", "--schedule-dir", type=Path, required=True)
    p.add_argument("-n", "--num-images-per-prompt", type=int, default=10)
    p.add_argument("-b", "--batch-size", type=int, default=100)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--seed-step", type=int, default=1)
    p.add_argument("--regen-if-not-n-images", type=int, required=False)
    p.add_argument("--dont-include-seed", action="store_true")
    a = p.parse_args()
    if not a.input_dir.exists():
        raise FileNotFoundError(f"Input directory {a.input_dir} not found.")
    if not a.schedule_dir.exists():
        raise FileNotFoundError(f"Directory {a.schedule_dir} for schedules not found.")
    a.output_dir.mkdir(parents=True, exist_ok=True)
    gen_type = get_image_generator_type(a.image_generator)
    files = [a.schedule_dir] if a.schedule_dir.is_file() else sorted(a.schedule_dir.glob("**/*.json"))
    for f in files:
        generate_for_schedule(gen_type, f, a.input_dir, a.output_dir, a.batch_size, a.num_images_per_prompt, a.seed, a.seed_step,
                              a.regen_if_not_n_images, a.dont_include_seed)
    print("Done.")


if __name__ == "__main__":
    main()
