#!/usr/bin/env python3
"""fp8 mode, FF1 -> FF2 hand-over: e4m3 + per-(row, 288-column tile) E8M0 block scales from FF1's epilogue (ECAD_FP8_SEG=1, round 6) against the 16-bit ffmid + per-row
quantiser pass (ECAD_FP8_SEG=0), both measured against the fp16 engine on the same inputs: one transformer forward with every component executed, several seeds.
    python scripts/fp8_handover_accuracy.py        (GPU)"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle.pixart_ref import PixArtConfig, make_weights  # noqa: E402  (weights recipe only)


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def main():
    from ecad_amd.engine import EcadEngine

    layers, B = int(os.environ.get("LAYERS", "1")), 8
    bits = np.ones((1, layers, 3), bool)
    for seed in range(4):
        w = make_weights(PixArtConfig(num_layers=layers), seed=seed)
        g = torch.Generator().manual_seed(100 + seed)
        enc = (torch.randn(2 * B, 120, 4096, generator=g) * 0.1).half().cuda()
        mask = torch.ones(2 * B, 120, dtype=torch.uint8).cuda()
        lat = torch.randn(2 * B, 4, 32, 32, generator=g).half().cuda()
        t = torch.full((2 * B,), 600.0).cuda()
        outs = {}
        for mode in ("f16", "seg", "row"):
            os.environ["ECAD_FP8_SEG"] = "1" if mode == "seg" else "0"
            eng = EcadEngine(num_layers=layers, max_batch=2 * B, max_tokens=256, max_text_tokens=120, dtype=torch.float16, fp8=(mode != "f16"))
            eng.load_state_dict(w)
            eng.set_schedule(bits)
            eng.begin_image_batch(enc, mask, 32, 32)
            outs[mode] = eng.step(0, lat, t).clone()
            eng.close()
        print(f"seed {seed}: fp8 vs fp16 noise prediction, block-scaled hand-over {rel(outs['seg'], outs['f16']):.3e}  row quantiser {rel(outs['row'], outs['f16']):.3e}  between {rel(outs['seg'], outs['row']):.3e}")


if __name__ == "__main__":
    main()
