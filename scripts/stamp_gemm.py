#!/usr/bin/env python3
"""Diagnostic (PG_STAMP build only): in-kernel cycles and clock of the pgemm consumer K loop."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402

eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
for M, N, K in [(8192, 1152, 4608), (8192, 3456, 1152), (8192, 4608, 1152)]:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.03).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    dbg = torch.zeros(max(N, 1024), device="cuda", dtype=torch.float32)  # stamp buffer rides in the bias slot (2 x u64 per WG)
    for _ in range(50):
        eng.op_gemm(0, a, w, dbg, out=out)
    torch.cuda.synchronize()
    dbg.zero_()
    eng.op_gemm(0, a, w, dbg, out=out)
    torch.cuda.synchronize()
    v = dbg.view(torch.int64)[: 2 * 256].reshape(256, 2).cpu().double()
    v = v[v[:, 1] > 0]
    cyc, rt = v[:, 0].median().item(), v[:, 1].median().item()
    steps = K // 64
    print(f"M={M} N={N} K={K}: loop cycles median {cyc:.0f} ({cyc / steps:.0f}/step, ideal 1152), wall {rt / 100:.2f} us, clock {cyc / (rt / 100) / 1e3:.3f} GHz, WGs {len(v)}")
