#!/usr/bin/env python3
"""Micro-benchmark of the FLUX joint attention kernel (B = 2, 24 heads x 128, S = 512 + 4096) through the C ABI."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.flux_engine import EcadFluxEngine  # noqa: E402

B, T, N, H = 2, 512, 4096, 24
eng = EcadFluxEngine(num_layers=0, num_single_layers=0, num_heads=H, max_batch=B, max_image_tokens=N, max_text_tokens=T, dtype=torch.bfloat16)
qkv = (torch.randn(B * (T + N), 3 * H * 128, device="cuda") * 0.5).bfloat16()
for _ in range(3):
    out = eng.op_attention(qkv, B, T, N)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    out = eng.op_attention(qkv, B, T, N)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
flops = 4.0 * B * (T + N) ** 2 * H * 128
print(f"joint attention S={T + N}: {us:.1f} us per launch, {flops / us / 1e6:.0f} TFLOP/s  (finite: {bool(torch.isfinite(out.float()).all())})")
