"""``ImageGenerator`` base class: schedule loading, callback chain, seeding, batching, timing.

Host-side mirror of ``ecad/image_generators/image_generator.py:29-495`` -- same constructor
keywords, attributes (``callbacks``, ``config``, ``transformer_weights``, ``dit_scheduler``,
``cache_schedule``, ``num_inference_steps``, ``random_generator``) and methods.  The reset rule is
the reference's: ``_reset_schedules_callback`` is LAST in the chain and, after the final step,
resets both step counters and the transformer's caches (:153-159,193-202).
"""
from __future__ import annotations

import gc
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Callable, Sequence

import torch
from torch.utils.data import DataLoader

from ecad_amd.dataset_utils.prompt_embedding_dataset import PromptEmbeddingDataset
from ecad_amd.pipelines.load_pipeline import pipeline_from_pretrained
from ecad_amd.schedulers.cache_scheduler.cache_schedule import CacheSchedule
from ecad_amd.schedulers.dit_scheduler.dit_scheduler import DiTScheduler


def _from_file_or_none(schedule_type, path):
    """``schedule_type.from_json(path)``; None when there is no path or the file has no block of that kind (KeyError)."""
    if path is None:
        return None
    try:
        return schedule_type.from_json(path)
    except KeyError:
        return None


def _release_device_memory() -> None:
    gc.collect()
    torch.cuda.empty_cache()


class ImageGenerator(ABC):
    def __init__(self, default_transformer_weights: str = None, default_pipeline_weights: str = None, default_pipeline_name: str = None,
                 schedule_path: Path | None = None, start_seed: int = 0, seed_step: int = 1, device: str = "cuda",
                 additional_callbacks: list[Callable] | None = None, dit_schedule_type: type = None, cache_schedule_type: type[CacheSchedule] = None):
        self.encoder_pipeline = self.diffusion_pipeline = None
        self.device = torch.device(device)
        self.schedule_path = None if schedule_path is None else Path(schedule_path)
        self.start_seed, self.seed_step = start_seed, seed_step
        self.additional_callbacks: list[Callable] = list(additional_callbacks or ())
        self.dit_schedule_type, self.cache_schedule_type = dit_schedule_type, cache_schedule_type
        self.default_transformer_weights, self.default_pipeline_weights = default_transformer_weights, default_pipeline_weights
        self.default_pipeline_name = default_pipeline_name
        self._initialize_random_generator()
        self._load_schedule_file()

    def _initialize_random_generator(self):
        # a CPU generator whatever the device: the same seed gives the same noise everywhere (reference :89-97)
        self.random_generator = torch.Generator(device="cpu").manual_seed(self.start_seed)
        print(f"seed: start {self.start_seed}, +{self.seed_step} per successive image of a prompt")

    def _load_schedule_file(self):
        from_file = {"dit": _from_file_or_none(self.dit_schedule_type, self.schedule_path),
                     "cache": _from_file_or_none(self.cache_schedule_type, self.schedule_path)}
        dit = self._default_dit_schedule() if from_file["dit"] is None else from_file["dit"]
        cache = self._default_cache_schedule() if from_file["cache"] is None else from_file["cache"]
        self.dit_scheduler, self.cache_schedule = DiTScheduler(dit), cache
        print(f"schedules: DiT {dit.name!r}, cache {cache.name!r}")

        # step count: a schedule read from the file outranks a default one; two of equal standing must agree
        both_or_neither = (from_file["dit"] is None) == (from_file["cache"] is None)
        if both_or_neither and dit.num_inference_steps != cache.num_inference_steps:
            raise ValueError("DiT and cache schedules have different numbers of inference steps.")
        self.num_inference_steps = (cache if both_or_neither or from_file["cache"] is not None else dit).num_inference_steps
        print(f"inference steps: {self.num_inference_steps}")

        # order matters: step counters, user callbacks, then the reset (which must see the final step last)
        self.callbacks = [self.dit_scheduler.per_step_callback, cache.per_step_callback, *self.additional_callbacks, self._reset_schedules_callback]

        configs = [c for c in (dit.top_level_config, cache.top_level_config) if c]
        if len(configs) == 2 and configs[0] != configs[1]:
            raise ValueError("DiT and cache schedules have different top level configs.")
        self._load_config(configs[0] if configs else {})

    def _load_config(self, config: dict) -> None:
        self.config = config
        for key in ("transformer_weights", "pipeline_weights"):
            setattr(self, key, config.get(key, getattr(self, f"default_{key}")))
        self.pipeline_from_pretrained = pipeline_from_pretrained(config.get("pipeline", {}), self.default_pipeline_name)
        self._load_subclass_config_defaults(config)

    def _load_subclass_config_defaults(self, config: dict) -> None:
        """Hook: height / width / guidance defaults of the model family."""

    def _reset_schedules_callback(self, step: int, timestep: int, **kwargs: Any) -> None:
        if step < self.num_inference_steps - 1:
            return
        for counter in (self.dit_scheduler, self.cache_schedule):
            counter.reset_step()
        if self.diffusion_pipeline is not None:
            self.diffusion_pipeline.transformer.reset_cache()

    def _call_callbacks(self, step: int, timestep: int, **kwargs: Any) -> None:
        for fn in self.callbacks:
            fn(step, timestep, **kwargs)

    @abstractmethod
    def _default_dit_schedule(self): ...

    @abstractmethod
    def _default_cache_schedule(self) -> CacheSchedule: ...

    @abstractmethod
    def create_encoder_pipeline(self): ...

    @abstractmethod
    def create_diffusion_pipeline(self): ...

    def free_encoder_pipeline(self):
        if self.encoder_pipeline is None:
            print("WARNING: No encoder pipeline to free.")
            return
        self.encoder_pipeline = None
        _release_device_memory()

    def free_diffusion_pipeline(self):
        if self.diffusion_pipeline is None:
            print("WARNING: No diffusion pipeline to free.")
            return
        pool = getattr(self.diffusion_pipeline, "_pool", None)
        if pool is not None:
            pool.close()  # the extra execution contexts borrow the first engine's weights: they go first
            self.diffusion_pipeline._pool = None
        engine = getattr(getattr(self.diffusion_pipeline, "transformer", None), "engine", None)
        if engine is not None:
            engine.close()  # device arenas are owned by the native engine, not by torch's allocator
        self.diffusion_pipeline = None
        _release_device_memory()

    def load_and_batch_embeddings(self, embedding_dir: Path, batch_size: int, shuffle: bool = False) -> DataLoader:
        return DataLoader(PromptEmbeddingDataset(embedding_dir), batch_size, shuffle)

    @abstractmethod
    def encode_prompts(self, prompts: list[str], negative_prompts: list[str] | None = None, batch_size: int | None = None, **kwargs) -> Sequence[dict]: ...

    @abstractmethod
    def encode_and_save_prompts(self, name_to_prompt, output_dir: Path, free_after: bool = False, batch_size: int | None = None): ...

    @abstractmethod
    def generate_images(self, prompt_embeds, images_per_prompt, **kwargs): ...

    @abstractmethod
    def generate_images_timed(self, prompt_embeds, **kwargs) -> float: ...

    def _ensure_diffusion_pipeline(self) -> None:
        if self.diffusion_pipeline is None:
            print("Creating diffusion pipeline.")
            self.create_diffusion_pipeline()

    def _image_file_stem(self, name: str, k: int, include_seed_in_name: bool) -> str:
        return f"{name}__image_seed:{self.start_seed + k * self.seed_step:03}" if include_seed_in_name else name

    @torch.inference_mode()
    def generate_from_saved_prompts(self, input_dir: Path, output_dir: Path, batch_size: int = 1, images_per_prompt: int = 1,
                                    free_after: bool = False, include_seed_in_name: bool = True, **kwargs: Any) -> None:
        self._ensure_diffusion_pipeline()
        batches = self.load_and_batch_embeddings(input_dir, batch_size, False)
        for index, embeds in enumerate(batches):
            print(f"batch {index + 1} / {len(batches)}")
            per_prompt_images = self.generate_images(embeds, images_per_prompt, **kwargs)
            for name, rel_path, images in zip(embeds["name"], embeds["relative_path"], per_prompt_images):
                for k, image in enumerate(images):
                    self.save_image(image, Path(output_dir) / rel_path / self._image_file_stem(name, k, include_seed_in_name))
        if free_after:
            self.free_diffusion_pipeline()

    def save_image(self, image, output_path: Path) -> None:
        target = Path(output_path)
        if target.suffix != ".png":
            target = target.with_name(target.name + ".png")
        target.parent.mkdir(parents=True, exist_ok=True)
        image.save(target)

    @torch.inference_mode()
    def time_image_generation(self, input_dir: Path, batch_size: int = 1, num_batches: int = 1, free_after: bool = False, **kwargs) -> list[float]:
        """ms per image for ``num_batches`` batches (reference :442-487; metric of compute_latency.py).  The loader is cycled when it
        holds fewer batches than asked for."""
        import itertools

        self._ensure_diffusion_pipeline()
        batches = self.load_and_batch_embeddings(input_dir, batch_size, False)
        all_times = [self.generate_images_timed(embeds, **kwargs) for embeds in itertools.islice(itertools.cycle(batches), num_batches)]
        if free_after:
            self.free_diffusion_pipeline()
        return all_times
