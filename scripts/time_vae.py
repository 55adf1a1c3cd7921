#!/usr/bin/env python3
"""Device time of one HIP VAE decode (batch 16 of 32 x 32 latents unless given), by HIP events over `reps` decodes: python scripts/time_vae.py [B] [reps].
ECAD_VAE_CONV=conv3s times the round-3 path (per-tap gathered A on eight waves) for the A/B against conv4_kernel."""
import sys

import torch

from ecad_amd.pipelines.vae_decoder import HipVaeDecoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
vae = HipVaeDecoder.random_init(3, device=0)
lat = (torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(7)) * 0.36).cuda()
for _ in range(3):
    vae.decode_to_uint8(lat)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    vae.decode_to_uint8(lat)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
st = vae.decode_stats()
print(f"B {B}: {ms:.3f} ms per decode, {st['conv_flops'] / ms / 1e9:.0f} TFLOP/s of the reference's arithmetic ({st['conv_flops_executed'] / ms / 1e9:.0f} executed) over the whole decode; {st}")
