from ecad_amd.image_generators.pixart_image_generator import PixArtAlphaImageGenerator  # noqa: F401
