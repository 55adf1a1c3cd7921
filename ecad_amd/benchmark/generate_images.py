"""schedule-dir x embedding-dir -> PNG tree.  Flags, directory mirroring and the skip / regenerate rule of ``ecad/benchmark/generate_images.py:13-232``:

* a schedule's directory is skipped only when it already holds at least one ``*.png`` directly inside it (``:27-43``) -- a directory a crashed run left empty is
  generated again;
* with ``--regen-if-not-n-images N`` a directory holding another number of PNGs has those PNGs removed (the files, not the directory) and is generated again;
* a schedule directory that contains only directories is mirrored below ``output_dir / <its name>`` (``:94-111``); otherwise every ``**/*.json`` goes to
  ``output_dir / <stem>``.
"""
from __future__ import annotations

import argparse
from pathlib import Path

from ecad_amd.image_generators.load_image_generator import ImageGeneratorRegistry, get_image_generator_type


def pngs_present(directory: Path) -> list[Path]:
    """The PNG files directly inside ``directory`` (the reference counts ``output_dir.glob("*.png")``: not recursive)."""
    return sorted(directory.glob("*.png")) if directory.exists() else []


def needs_generation(out: Path, regen_if_not_n: int | None) -> bool:
    """Skip rule of the reference (``generate_images.py:27-43``); removes the stale PNGs when the directory has to be redone."""
    found = pngs_present(out)
    if not found:
        return True  # missing, or present without a single image (what a crashed run leaves): generate
    if regen_if_not_n is not None and len(found) != regen_if_not_n:
        print(f"Regenerating images for schedule {out.name}; expected {regen_if_not_n} images, found {len(found)}.")
        for png in found:
            png.unlink()
        return True
    print(f"Skipping schedule {out.name} because output directory already exists. "
          f"Expected {regen_if_not_n if regen_if_not_n is not None else 'any number'}, found {len(found)} images.")
    return False


def generate_for_schedule(gen_type, schedule_file: Path, input_dir: Path, out: Path, batch_size: int, n_per_prompt: int, seed: int,
                          seed_step: int, regen_if_not_n: int | None = None, dont_include_seed: bool = False) -> bool:
    """Images of one schedule into ``out`` (the schedule's own directory); returns whether anything was generated."""
    if not needs_generation(out, regen_if_not_n):
        return False
    out.mkdir(parents=True, exist_ok=True)
    print(f"\n\n\nGenerating images for schedule: {schedule_file.stem}.\n")
    try:
        gen = gen_type(start_seed=seed, seed_step=seed_step, schedule_path=schedule_file, max_batch=2 * batch_size)
    except Exception as e:  # the reference's message (:51-55)
        raise ValueError("Error creating image generator, check arguments passed during init in the code.") from e
    gen.generate_from_saved_prompts(input_dir, out, batch_size=batch_size, images_per_prompt=n_per_prompt, free_after=True,
                                    include_seed_in_name=not dont_include_seed)
    return True


def generate_all_schedules(gen_type, schedule_dir: Path, input_dir: Path, output_dir: Path, batch_size: int, n_per_prompt: int, seed: int, seed_step: int,
                           regen_if_not_n: int | None = None, dont_include_seed: bool = False) -> int:
    """Walk ``schedule_dir`` as the reference does (``:66-123``); returns the number of schedules generated."""
    common = (batch_size, n_per_prompt, seed, seed_step, regen_if_not_n, dont_include_seed)
    if schedule_dir.is_file() and schedule_dir.suffix == ".json":
        return int(generate_for_schedule(gen_type, schedule_dir, input_dir, output_dir / schedule_dir.stem, *common))
    children = list(schedule_dir.iterdir())
    if all(c.is_dir() for c in children):  # only directories: mirror the tree below output_dir / <name>
        return sum(generate_all_schedules(gen_type, c, input_dir, output_dir / schedule_dir.name, *common) for c in sorted(children))
    return sum(int(generate_for_schedule(gen_type, f, input_dir, output_dir / f.stem, *common)) for f in sorted(schedule_dir.glob("**/*.json")))


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
    if not a.output_dir.exists():
        print(f"Output directory {a.output_dir} not found. Creating.")
        a.output_dir.mkdir(parents=True, exist_ok=True)
    gen_type = get_image_generator_type(a.image_generator)
    generate_all_schedules(gen_type, a.schedule_dir, a.input_dir, a.output_dir, a.batch_size, a.num_images_per_prompt, a.seed, a.seed_step,
                           a.regen_if_not_n_images, a.dont_include_seed)
    print("Done.")


if __name__ == "__main__":
    main()
