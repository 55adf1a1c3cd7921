"""Pipeline registry (reference: ``ecad/pipelines/load_pipeline.py:16-58``).

``pixart_alpha`` / ``pixart_sigma`` / ``flux`` resolve to the native-engine pipelines; ``tgate`` and
``pass_through`` are outside the hot-path scope and raise.
"""
from __future__ import annotations

from typing import Any, Callable

from ecad_amd.pipelines.flux import FluxPipeline
from ecad_amd.pipelines.pixart_alpha import PixArtAlphaPipeline, PixArtSigmaPipeline


class PipelineRegistry:
    _registry: dict[str, type] = {
        "pixart_alpha": PixArtAlphaPipeline,
        "pixart_sigma": PixArtSigmaPipeline,
        "flux": FluxPipeline,
    }
    _out_of_scope = ("tgate", "pass_through")

    @classmethod
    def get(cls, name: str, default_name: str | None = None) -> type | None:
        if name in cls._out_of_scope:
            raise NotImplementedError(f"pipeline {name!r} is outside the native engine's scope")
        pipeline = cls._registry.get(name)
        if pipeline is None and default_name is not None:
            pipeline = cls._registry.get(default_name)
        return pipeline


def pipeline_from_pretrained(pipeline_config: dict, default_pipeline_name: str | None = None) -> Callable[..., Any]:
    pipeline_class = PipelineRegistry.get(pipeline_config.get("name", default_pipeline_name))
    extra_kwargs = pipeline_config.get("kwargs", {})
    print(f"Using pipeline: {pipeline_class}")

    def from_pretrained(*args: Any, **kwargs: Any):
        return pipeline_class.from_pretrained(*args, **kwargs, **extra_kwargs)

    return from_pretrained
