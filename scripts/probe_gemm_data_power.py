#!/usr/bin/env python3
"""Is a GEMM launch bound by the power cap or by its instruction stream?  Same kernel, same shape, operands either pseudo-random (what bench.py feeds)
or near-constant (few switching bits): a power-bound kernel speeds up with quiet data, an issue- / latency-bound one does not.  Back-to-back launches for
~1 s per point (clocks settled), TFLOP/s by events."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402

eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
CASES = [("wide288", "ff1", 1, 8192, 4608, 1152), ("tile144", "ff2", 3, 8192, 1152, 4608), ("tile144", "attn1_out", 3, 8192, 1152, 1152), ("wide288", "attn1_kv", 0, 8192, 2304, 1152)]
for kernel, site, epi, M, N, K in CASES:
    for data in ("random", "quiet"):
        if data == "random":
            a = torch.randn(M, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * 0.03).half()
        else:
            a = torch.full((M, K), 0.5, device="cuda").half()
            w = torch.full((N, K), 0.03125, device="cuda").half()
        bias = torch.zeros(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        resid = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 3 else None
        kw = dict(resid=resid, out=out)
        n = 6000 if K < 2000 and N < 2000 else 3000
        for _ in range(n // 2):
            eng.op_gemm_ex(kernel, site, epi, a, w, bias, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            eng.op_gemm_ex(kernel, site, epi, a, w, bias, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"{kernel}/{site} {M}x{N}x{K} {data:6s}: {us:7.2f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
