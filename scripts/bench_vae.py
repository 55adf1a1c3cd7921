#!/usr/bin/env python3
"""Decode-tail micro-benchmark: HIP AutoencoderKL decoder, batch 16 of 32x32 latents -> 256x256 uint8."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.pipelines.vae_decoder import HipVaeDecoder
torch.manual_seed(0)
vae = HipVaeDecoder.random_init(0)
lat = torch.randn(16, 4, 32, 32, device="cuda") * 0.18215
for _ in range(3): vae.decode_to_uint8(lat)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): vae.decode_to_uint8(lat)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print(f"HIP VAE decode, 16 x 256x256: {ms:.2f} ms  ({16 * 608e9 / ms / 1e9:.0f} TFLOP/s on ~608 GF/image of conv work)")
