#!/usr/bin/env python3
"""A/B (GPU), one process: attention128.h (register-staged chunks, two barriers per chunk) against attention_dma.h (LDS-DMA into a double-buffered image, one
barrier per chunk; the variants that were measured and removed are listed in that header) on the FLUX joint shape (B 2, 24 heads x 128, S = 512 + 4096), the Sigma-1024 self-attention shape (Bt 8, 16 heads
x 72, N 4096) and the alpha-256 one (Bt 32, N 256).  Outputs must be bit-identical (same fragment layouts, same operation order)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best * 1e3


def flux():
    from ecad_amd.flux_engine import EcadFluxEngine

    B, T, N, H = 2, 512, 4096, 24
    qkv = (torch.randn(B * (T + N), 3 * H * 128, device="cuda") * 0.5).bfloat16()
    engs = {}
    for mode in ("shared", "dma", "dma-split"):
        os.environ["ECAD_FLUX_ATTN"] = mode
        engs[mode] = EcadFluxEngine(num_layers=0, num_single_layers=0, num_heads=H, max_batch=B, max_image_tokens=N, max_text_tokens=T, dtype=torch.bfloat16)
    ref = engs["shared"].op_attention(qkv, B, T, N)
    flops = 4.0 * B * (T + N) ** 2 * H * 128
    for rnd in range(2):
        for mode, eng in engs.items():
            out = eng.op_attention(qkv, B, T, N)
            us = timeit(lambda: eng.op_attention(qkv, B, T, N), 10)
            print(f"FLUX joint S={T + N} {mode:10s}: {us:7.1f} us  {flops / us / 1e6:5.0f} TF  bit-identical to attention128.h: {torch.equal(out, ref)}  finite {bool(torch.isfinite(out.float()).all())}", flush=True)


def pixart(B, N, label):
    from ecad_amd.engine import EcadEngine

    H, dh = 16, 72
    D = H * dh
    qkv = torch.randn(B * N, 3 * D, device="cuda").half()
    engs = {}
    for mode in ("shared", "dma", "dma4"):
        os.environ["ECAD_ATTN"] = mode
        engs[mode] = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
    run = lambda e: e.op_attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, dh)
    ref = run(engs["shared"])
    flops = 4.0 * B * H * N * N * dh
    for rnd in range(2):
        for mode, eng in engs.items():
            out = run(eng)
            us = timeit(lambda: run(eng), 10 if N > 1000 else 50)
            print(f"{label} Bt={B} N={N} {mode:10s}: {us:7.1f} us  {flops / us / 1e6:5.0f} TF  bit-identical: {torch.equal(out, ref)}", flush=True)


if __name__ == "__main__":
    flux()
    pixart(8, 4096, "Sigma-1024 self")
    pixart(32, 1024, "alpha-512 self")
    pixart(32, 256, "alpha-256 self")
