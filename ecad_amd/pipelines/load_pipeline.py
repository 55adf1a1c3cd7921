"""Pipeline name -> pipeline class, and the ``from_pretrained`` partial the image generators call.

Interface of the reference's ``ecad/pipelines/load_pipeline.py`` (``PipelineRegistry.get(name, default_name)``,
``pipeline_from_pretrained(config_block, default_name)`` returning a callable that forwards to ``cls.from_pretrained`` with the
config's extra kwargs appended).  ``pixart_alpha`` / ``pixart_sigma`` / ``flux`` resolve to the native-engine pipelines; ``tgate`` and
``pass_through`` are outside the hot-path scope and raise ``NotImplementedError``.
"""
from __future__ import annotations

import functools
from typing import Any, Callable


def _resolve(name: str):
    if name == "flux":
        from ecad_amd.pipelines.flux import FluxPipeline

        return FluxPipeline
    from ecad_amd.pipelines import pixart_alpha

    return {"pixart_alpha": pixart_alpha.PixArtAlphaPipeline, "pixart_sigma": pixart_alpha.PixArtSigmaPipeline}.get(name)


class PipelineRegistry:
    names = ("pixart_alpha", "pixart_sigma", "flux")
    unsupported = ("tgate", "pass_through")

    @classmethod
    def get(cls, name: str, default_name: str | None = None) -> type | None:
        if name in cls.unsupported:
            raise NotImplementedError(f"pipeline {name!r} is outside the native engine's scope")
        for candidate in (name, default_name):
            if candidate in cls.names:
                return _resolve(candidate)
        return None


def pipeline_from_pretrained(pipeline_config: dict, default_pipeline_name: str | None = None) -> Callable[..., Any]:
    chosen = PipelineRegistry.get(pipeline_config.get("name", default_pipeline_name))
    print(f"Using pipeline: {chosen}")
    return functools.partial(_call_from_pretrained, chosen, dict(pipeline_config.get("kwargs", {})))


def _call_from_pretrained(pipeline_class, extra_kwargs: dict, *args: Any, **kwargs: Any):
    return pipeline_class.from_pretrained(*args, **kwargs, **extra_kwargs)
