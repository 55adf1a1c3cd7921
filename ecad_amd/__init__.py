"""ecad_amd: MI355X-native engine for ECAD's cached-diffusion hot path.

Host side (this package) mirrors the reference's plug-in surface; the denoising hot path
is ``ecad_amd/csrc`` (HIP, gfx950) behind the C ABI declared in ``include/ecad_hip.h``.
"""
__version__ = "0.1.0"
