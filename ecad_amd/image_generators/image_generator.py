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


class ImageGenerator(ABC):
    def __init__(
        self,
        default_transformer_weights: str = None,
        default_pipeline_weights: str = None,
        default_pipeline_name: str = None,
        schedule_path: Path | None = None,
        start_seed: int = 0,
        seed_step: int = 1,
        device: str = "cuda",
        additional_callbacks: list[Callable] | None = None,
        dit_schedule_type: type = None,
        cache_schedule_type: type[CacheSchedule] = None,
    ):
        self.device = torch.device(device)
        self.default_transformer_weights = default_transformer_weights
        self.default_pipeline_weights = default_pipeline_weights
        self.default_pipeline_name = default_pipeline_name
        self.schedule_path = Path(schedule_path) if schedule_path is not None else None
        self.start_seed = start_seed
        self.seed_step = seed_step
        self.additional_callbacks: list[Callable] = additional_callbacks or []
        self.dit_schedule_type = dit_schedule_type
        self.cache_schedule_type = cache_schedule_type
        self.encoder_pipeline = None
        self.diffusion_pipeline = None
        self._initialize_random_generator()
        self._load_schedule_file()

    def _initialize_random_generator(self):
        # always a CPU generator, for reproducibility across devices (reference :89-97)
        self.random_generator = torch.Generator(device="cpu")
        self.random_generator.manual_seed(self.start_seed)
        print(f"Using random start seed: {self.start_seed}. Stepping seed by {self.seed_step} for each successive image per prompt.")

    def _load_schedule_file(self):
        dit_schedule = cache_schedule = None
        dit_from_file = cache_from_file = False
        if self.schedule_path is not None:
            try:
                dit_schedule = self.dit_schedule_type.from_json(self.schedule_path)
                dit_from_file = True
            except KeyError:
                dit_schedule = None
            try:
                cache_schedule = self.cache_schedule_type.from_json(self.schedule_path)
                cache_from_file = True
            except KeyError:
                cache_schedule = None
        if dit_schedule is None:
            dit_schedule = self._default_dit_schedule()
        if cache_schedule is None:
            cache_schedule = self._default_cache_schedule()
        self.dit_scheduler = DiTScheduler(dit_schedule)
        self.cache_schedule = cache_schedule
        print(f"Using DiT schedule: {dit_schedule.name}.")
        print(f"Using cache schedule: {cache_schedule.name}.")

        dit_n, cache_n = dit_schedule.num_inference_steps, cache_schedule.num_inference_steps
        if dit_from_file == cache_from_file and dit_n != cache_n:
            raise ValueError("DiT and cache schedules have different numbers of inference steps.")
        self.num_inference_steps = cache_n if (cache_from_file or not dit_from_file) else dit_n
        print(f"Using {self.num_inference_steps} inference steps.")

        self.callbacks = [self.dit_scheduler.per_step_callback, self.cache_schedule.per_step_callback]
        self.callbacks.extend(self.additional_callbacks)
        self.callbacks.append(self._reset_schedules_callback)  # MUST be last

        dit_config, cache_config = dit_schedule.top_level_config, cache_schedule.top_level_config
        if dit_config and cache_config and dit_config != cache_config:
            raise ValueError("DiT and cache schedules have different top level configs.")
        self._load_config(dit_config if dit_config else cache_config)

    def _load_config(self, config: dict) -> None:
        self.config = config
        self.transformer_weights = config.get("transformer_weights", self.default_transformer_weights)
        self.pipeline_weights = config.get("pipeline_weights", self.default_pipeline_weights)
        self.pipeline_from_pretrained = pipeline_from_pretrained(config.get("pipeline", {}), self.default_pipeline_name)
        self._load_subclass_config_defaults(config)

    def _load_subclass_config_defaults(self, config: dict) -> None:
        pass

    def _reset_schedules_callback(self, step: int, timestep: int, **kwargs: Any) -> None:
        if step >= self.num_inference_steps - 1:
            self.dit_scheduler.reset_step()
            self.cache_schedule.reset_step()
            if self.diffusion_pipeline is not None:
                self.diffusion_pipeline.transformer.reset_cache()

    def _call_callbacks(self, step: int, timestep: int, **kwargs: Any) -> None:
        for callback in self.callbacks:
            callback(step, timestep, **kwargs)

    @abstractmethod
    def _default_dit_schedule(self): ...

    @abstractmethod
    def _default_cache_schedule(self) -> CacheSchedule: ...

    @abstractmethod
    def create_encoder_pipeline(self): ...

    @abstractmethod
    def create_diffusion_pipeline(self): ...

    def free_encoder_pipeline(self):
        if self.encoder_pipeline is not None:
            self.encoder_pipeline = None
            gc.collect()
            torch.cuda.empty_cache()
        else:
            print("WARNING: No encoder pipeline to free.")

    def free_diffusion_pipeline(self):
        if self.diffusion_pipeline is not None:
            tr = getattr(self.diffusion_pipeline, "transformer", None)
            if tr is not None and hasattr(tr, "engine"):
                tr.engine.close()
            self.diffusion_pipeline = None
            gc.collect()
            torch.cuda.empty_cache()
        else:
            print("WARNING: No diffusion pipeline to free.")

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

    @torch.inference_mode()
    def generate_from_saved_prompts(self, input_dir: Path, output_dir: Path, batch_size: int = 1, images_per_prompt: int = 1,
                                    free_after: bool = False, include_seed_in_name: bool = True, **kwargs: Any) -> None:
        if self.diffusion_pipeline is None:
            print("Creating diffusion pipeline.")
            self.create_diffusion_pipeline()
        loader = self.load_and_batch_embeddings(input_dir, batch_size, False)
        for i, embeds in enumerate(loader):
            print(f"Running batch {i} of {len(loader) - 1}.")
            images = self.generate_images(embeds, images_per_prompt, **kwargs)
            for name, rel_path, per_prompt in zip(embeds["name"], embeds["relative_path"], images):
                for k, image in enumerate(per_prompt):
                    image_seed = self.start_seed + k * self.seed_step
                    fname = f"{name}__image_seed:{image_seed:03}" if include_seed_in_name else name
                    self.save_image(image, Path(output_dir) / rel_path / fname)
        if free_after:
            self.free_diffusion_pipeline()

    def save_image(self, image, output_path: Path) -> None:
        output_path = Path(output_path)
        if output_path.suffix != ".png":
            output_path = Path(f"{output_path}.png")
        output_path.parent.mkdir(parents=True, exist_ok=True)
        image.save(output_path)

    @torch.inference_mode()
    def time_image_generation(self, input_dir: Path, batch_size: int = 1, num_batches: int = 1, free_after: bool = False, **kwargs) -> list[float]:
        """ms per image for ``num_batches`` batches (reference :442-487; metric of compute_latency.py)."""
        if self.diffusion_pipeline is None:
            self.create_diffusion_pipeline()
        all_times: list[float] = []
        while len(all_times) < num_batches:
            for embeds in self.load_and_batch_embeddings(input_dir, batch_size, False):
                all_times.append(self.generate_images_timed(embeds, **kwargs))
                if len(all_times) >= num_batches:
                    break
        if free_after:
            self.free_diffusion_pipeline()
        return all_times
