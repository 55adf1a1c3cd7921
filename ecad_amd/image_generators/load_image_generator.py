"""Image-generator registry (reference: ``ecad/image_generators/load_image_generator.py:16-84``)."""
from __future__ import annotations

from ecad_amd.image_generators.flux_image_generator import FluxImageGenerator
from ecad_amd.image_generators.image_generator import ImageGenerator
from ecad_amd.image_generators.pixart_image_generator import PixArtAlphaImageGenerator, PixArtSigmaImageGenerator


class ImageGeneratorRegistry:
    registry: dict[str, type[ImageGenerator]] = {
        "PixArtAlphaImageGenerator": PixArtAlphaImageGenerator,
        "PixArtSigmaImageGenerator": PixArtSigmaImageGenerator,
        "FluxImageGenerator": FluxImageGenerator,
    }

    @classmethod
    def get(cls, name: str, default_name: str | None = None) -> type[ImageGenerator] | None:
        gen = cls.registry.get(name)
        if gen is None and default_name is not None:
            gen = cls.registry.get(default_name)
        return gen


def get_image_generator_type_from_config(config: dict, default_name: str = "PixArtImageGenerator") -> type[ImageGenerator]:
    gen = ImageGeneratorRegistry.get(config["image_generator"], default_name) if "image_generator" in config else None
    if gen is None:
        raise ValueError(f"Image generator not found in config: {config}.")
    return gen


def get_image_generator_type(name: str, default_name: str = "PixArtImageGenerator") -> type[ImageGenerator]:
    gen = ImageGeneratorRegistry.get(name, default_name)
    if gen is None:
        raise ValueError(f"Image generator not found: {name}.")
    return gen
