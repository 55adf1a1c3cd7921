"""FLUX image generator (reference: ``ecad/image_generators/flux_image_generator.py:30-418``): same constructor, defaults,
config keys (height / width / guidance_scale), callback wrapper and generate / timed entry points; the transformer is the native
engine facade."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Sequence

import torch

from ecad_amd.image_generators.image_generator import ImageGenerator
from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule
from ecad_amd.schedulers.cache_scheduler.generators.flux_schedule_generators import gen_default as cache_gen_default
from ecad_amd.schedulers.dit_scheduler.dit_scheduler import FluxDiTSchedule
from ecad_amd.transformer_2d_models.flux_transformer_2d_edited import FluxTransformer2DEdited


class FluxImageGenerator(ImageGenerator):
    DEFAULT_WEIGHTS = "black-forest-labs/FLUX.1-dev"
    PIPELINE_WEIGHTS = "black-forest-labs/FLUX.1-dev"
    DEFAULT_HEIGHT = 256
    DEFAULT_WIDTH = 256
    DEFAULT_GUIDANCE_SCALE = 5
    DEFAULT_NUM_INFERENCE_STEPS = 20
    MAX_TEXT_TOKENS = 512

    def __init__(self, weights_name: str | None = None, start_seed: int = 0, seed_step: int = 1, schedule_path: Path | None = None, max_batch: int = 2,
                 transformer_kwargs: dict | None = None):
        if not torch.cuda.is_available():
            raise ValueError("CUDA is required for image generation.")  # same hard requirement as the reference (:46-47)
        self.max_batch = max_batch
        self.transformer_kwargs = transformer_kwargs or {}  # e.g. a reduced geometry for tests
        super().__init__(
            default_transformer_weights=weights_name or self.DEFAULT_WEIGHTS,
            default_pipeline_weights=self.PIPELINE_WEIGHTS,
            default_pipeline_name="flux",
            schedule_path=schedule_path,
            start_seed=start_seed,
            seed_step=seed_step,
            device="cuda",
            dit_schedule_type=FluxDiTSchedule,
            cache_schedule_type=FluxCacheSchedule,
        )

    def _load_subclass_config_defaults(self, config: dict) -> None:
        self.height = config.get("height", self.DEFAULT_HEIGHT)
        self.width = config.get("width", self.DEFAULT_WIDTH)
        self.guidance_scale = config.get("guidance_scale", self.DEFAULT_GUIDANCE_SCALE)

    def _blocks(self) -> tuple[int, int]:
        kw = getattr(self, "transformer_kwargs", None) or {}
        return kw.get("num_layers", 19), kw.get("num_single_layers", 38)

    def _default_dit_schedule(self) -> FluxDiTSchedule:
        nb, ns = self._blocks()
        return FluxDiTSchedule(nb, self.DEFAULT_NUM_INFERENCE_STEPS, "default", num_single_blocks=ns)

    def _default_cache_schedule(self) -> FluxCacheSchedule:
        nb, ns = self._blocks()
        return next(cache_gen_default(nb, ns, self.DEFAULT_NUM_INFERENCE_STEPS))

    def _call_callbacks_wrapper(self, _pipeline, step: int, timestep: int, callback_kwargs: dict[str, Any]) -> dict[str, torch.Tensor]:
        if "latents" not in callback_kwargs or "prompt_embeds" not in callback_kwargs:
            print("WARNING: Callback kwargs missing latents or prompt embeds. This will likley cause a crash.")
        self._call_callbacks(step, timestep)
        return {"latents": callback_kwargs["latents"], "prompt_embeds": callback_kwargs["prompt_embeds"]}

    def create_encoder_pipeline(self):
        raise NotImplementedError(
            "CLIP / T5 prompt encoding needs the text-encoder checkpoints (no network here) and stays outside the native hot path; "
            "use pre-computed embeddings (.pt files in the reference's FluxPromptEmbedding format).")

    def create_diffusion_pipeline(self, skip_transformer_block_init: bool = False) -> None:
        tokens = (max(self.height, 16) // 16) * (max(self.width, 16) // 16)
        transformer = FluxTransformer2DEdited.from_pretrained(
            self.transformer_weights, subfolder="transformer", torch_dtype=torch.bfloat16, use_safetensors=True, dit_scheduler=self.dit_scheduler,
            cache_schedule=self.cache_schedule, skip_transformer_block_init=skip_transformer_block_init, max_batch=self.max_batch, max_image_tokens=tokens,
            max_text_tokens=self.MAX_TEXT_TOKENS, **self.transformer_kwargs)
        pipeline = self.pipeline_from_pretrained(self.pipeline_weights, transformer=transformer, text_encoder=None, text_encoder_2=None, torch_dtype=torch.bfloat16,
                                                 use_safetensors=True).to(self.device)
        pipeline.native_loop = len(self.additional_callbacks) == 0  # whole-loop graph replay only when nobody needs intermediate latents
        self.diffusion_pipeline = pipeline

    def encode_prompts(self, prompts, negative_prompts=None, batch_size=None, **kwargs) -> Sequence[dict]:
        self.create_encoder_pipeline()
        return []

    def encode_and_save_prompts(self, name_to_prompt, output_dir: Path, free_after: bool = False, batch_size: int | None = None):
        self.create_encoder_pipeline()

    def _pipeline_kwargs(self, prompt_embeds: dict, height, width, guidance_scale, output_type="pil"):
        return dict(
            prompt=None, prompt_2=None,
            prompt_embeds=prompt_embeds["prompt_embeds"].to(self.device),
            pooled_prompt_embeds=prompt_embeds["pooled_prompt_embeds"].to(self.device),
            num_images_per_prompt=1, num_inference_steps=self.num_inference_steps, generator=self.random_generator, return_dict=False,
            height=height, width=width, guidance_scale=guidance_scale, output_type=output_type,
            callback_on_step_end=self._call_callbacks_wrapper,  # drives the schedule counters
            callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"],
        )

    @torch.inference_mode()
    def generate_images(self, prompt_embeds: dict, images_per_prompt: int = 1, height: int | None = None, width: int | None = None,
                        guidance_scale: float | None = None, output_type: str = "pil", **kwargs):
        """list (per prompt) of lists (per seed) of images; images of one prompt are generated sequentially (:318-330)."""
        if self.diffusion_pipeline is None:
            print("Diffusion pipeline not initialized. Initializing now.")
            self.create_diffusion_pipeline()
        height, width, guidance_scale = height or self.height, width or self.width, guidance_scale or self.guidance_scale
        all_images = []
        for i in range(images_per_prompt):
            self.random_generator.manual_seed(self.start_seed + i * self.seed_step)
            all_images.append(self.diffusion_pipeline(**self._pipeline_kwargs(prompt_embeds, height, width, guidance_scale, output_type))[0])
        per_prompt = [[] for _ in range(len(all_images[0]))]
        for kth in all_images:
            for j, image in enumerate(kth):
                per_prompt[j].append(image)
        return per_prompt

    @torch.inference_mode()
    def generate_images_timed(self, prompt_embeds: dict, height: int | None = None, width: int | None = None, guidance_scale: float | None = None,
                              output_type: str = "pil", **kwargs) -> float:
        """ms per image around one full pipeline call, device events on the launch stream (:366-418)."""
        if self.diffusion_pipeline is None:
            self.create_diffusion_pipeline()
        height, width, guidance_scale = height or self.height, width or self.width, guidance_scale or self.guidance_scale
        self.random_generator.manual_seed(self.start_seed)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        _ = self.diffusion_pipeline(**self._pipeline_kwargs(prompt_embeds, height, width, guidance_scale, output_type))
        end.record()
        torch.cuda.synchronize()
        return start.elapsed_time(end) / prompt_embeds["prompt_embeds"].shape[0]
