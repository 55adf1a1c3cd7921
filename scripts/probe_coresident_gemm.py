#!/usr/bin/env python3
"""Probe (GPU): does a GEMM built from 4-wave workgroups, two co-resident per CU (gemm.h: 128 x 128 tiles, 64 KB of LDS), keep the matrix pipe busier than
the 8-wave one-workgroup-per-CU kernels?  Same K = 1152, cold operands, plain bias epilogue and the read-modify-write epilogue; shapes chosen so that every
kernel fills the chip in whole rounds (tile128: 8192 x 1024 = 512 tiles = 2 per CU at once; tile144: 8192 x 1152 = 256 tiles; wide288: 8192 x 2304 = 256)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402

eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
CASES = [("tile128", "small", 8192, 1024, 1152), ("tile144", "attn2_q", 8192, 1152, 1152), ("tile144", "attn1_out", 8192, 1152, 1152), ("wide288", "attn1_kv", 8192, 2304, 1152),
         ("tile128", "small", 8192, 2048, 1152), ("tile128", "small", 8192, 4096, 4608), ("tile144", "ff2", 8192, 1152, 4608)]
for epi in (0, 3):
    for kern, site, M, N, K in CASES:
        if epi == 3 and kern == "wide288":
            continue
        nsets = 10
        sets = []
        for i in range(nsets):
            a = torch.randn(M, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * 0.03).half()
            out = torch.empty(M, N, device="cuda", dtype=torch.float16)
            resid = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 3 else None
            sets.append((a, w, out, resid))
        bias = torch.randn(N, device="cuda") * 0.1
        best = 1e9
        for rnd in range(3):
            for i in range(nsets):
                a, w, out, resid = sets[i]
                eng.op_gemm_ex(kern, site, epi, a, w, bias, resid=resid, out=out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            reps = 40
            for i in range(reps):
                a, w, out, resid = sets[i % nsets]
                eng.op_gemm_ex(kern, site, epi, a, w, bias, resid=resid, out=out)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / reps)
        print(f"epi {epi} {kern:8s} {site:5s} M={M} N={N} K={K}: {best * 1e3:7.1f} us  {2.0 * M * N * K / best / 1e9:6.0f} TF", flush=True)
        del sets
