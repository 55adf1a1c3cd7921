from ecad_amd.image_generators.pixart_image_generator import PixArtSigmaImageGenerator  # noqa: F401
