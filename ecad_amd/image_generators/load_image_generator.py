"""Name -> image-generator class.  Same three names, the same ``ImageGeneratorRegistry.get`` fall-back rule and the same two lookup
helpers (ValueError when nothing matches) as the reference's ``ecad/image_generators/load_image_generator.py``; classes are imported
lazily so that naming a generator does not import every engine binding."""
from __future__ import annotations

import importlib

from ecad_amd.image_generators.image_generator import ImageGenerator

_WHERE = {
    "PixArtAlphaImageGenerator": "ecad_amd.image_generators.pixart_image_generator",
    "PixArtSigmaImageGenerator": "ecad_amd.image_generators.pixart_image_generator",
    "FluxImageGenerator": "ecad_amd.image_generators.flux_image_generator",
}


class _LazyRegistry(dict):
    def get(self, key, default=None):  # dict.get does not call __missing__
        try:
            return self[key]
        except KeyError:
            return default

    def __missing__(self, key):
        if key not in _WHERE:
            raise KeyError(key)
        self[key] = getattr(importlib.import_module(_WHERE[key]), key)
        return self[key]

    def __contains__(self, key):
        return key in _WHERE

    def keys(self):
        return _WHERE.keys()


class ImageGeneratorRegistry:
    registry: dict[str, type[ImageGenerator]] = _LazyRegistry()

    @classmethod
    def get(cls, name: str, default_name: str | None = None) -> type[ImageGenerator] | None:
        for candidate in (name, default_name):
            found = cls.registry.get(candidate) if candidate is not None else None
            if found is not None:
                return found
        return None


def get_image_generator_type(name: str, default_name: str = "PixArtImageGenerator") -> type[ImageGenerator]:
    found = ImageGeneratorRegistry.get(name, default_name)
    if found is None:
        raise ValueError(f"Image generator not found: {name}.")
    return found


def get_image_generator_type_from_config(config: dict, default_name: str = "PixArtImageGenerator") -> type[ImageGenerator]:
    found = ImageGeneratorRegistry.get(config["image_generator"], default_name) if "image_generator" in config else None
    if found is None:
        raise ValueError(f"Image generator not found in config: {config}.")
    return found
