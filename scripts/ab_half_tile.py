#!/usr/bin/env python3
"""A/B (GPU), one process, cold operands, interleaved: gemm3s.h's 256 x 144 tile (8 waves, one workgroup per CU) against its half-tile variant (128 x 144, four
waves, two workgroups per CU) on the N = K = 1152 call sites and FF2, plain and read-modify-write epilogues.  Results must be bit-identical (same k order)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402

eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
CASES = [("attn2_q", 0, 8192, 1152, 1152), ("attn1_out", 3, 8192, 1152, 1152), ("attn2_out", 3, 8192, 1152, 1152), ("ff2", 3, 8192, 1152, 4608), ("small", 0, 512, 1152, 1152),
         ("attn1_out", 3, 512, 1152, 1152), ("ff2", 3, 512, 1152, 4608)]
for site, epi, M, N, K in CASES:
    nsets = 10
    sets = []
    for i in range(nsets):
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * 0.03).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        resid = torch.randn(M, N, device="cuda").half() if epi == 3 else None
        sets.append((a, w, out, resid))
    bias = torch.randn(N, device="cuda") * 0.1
    table = torch.randn(N, device="cuda")
    # bit-identity
    a, w, out, resid = sets[0]
    outs = {}
    for kern in ("tile144", "half144"):
        r = resid.clone() if resid is not None else None
        o = torch.zeros_like(out)
        eng.op_gemm_ex(kern, site, epi, a, w, bias, resid=r, gate_table=table if epi == 3 else None, rows_per_batch=256, out=o)
        outs[kern] = (o, r)
    same = torch.equal(outs["tile144"][0], outs["half144"][0]) and (resid is None or torch.equal(outs["tile144"][1], outs["half144"][1]))
    best = {"tile144": 1e9, "half144": 1e9}
    for rnd in range(4):
        for kern in best:
            for i in range(nsets):
                a, w, out, resid = sets[i]
                eng.op_gemm_ex(kern, site, epi, a, w, bias, resid=resid, gate_table=table if epi == 3 else None, rows_per_batch=256, out=out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            reps = 40
            for i in range(reps):
                a, w, out, resid = sets[i % nsets]
                eng.op_gemm_ex(kern, site, epi, a, w, bias, resid=resid, gate_table=table if epi == 3 else None, rows_per_batch=256, out=out)
            e.record()
            torch.cuda.synchronize()
            best[kern] = min(best[kern], s.elapsed_time(e) / reps)
    fl = 2.0 * M * N * K
    print(f"{site:10s} epi {epi} M={M} K={K}: tile144 {best['tile144'] * 1e3:6.1f} us {fl / best['tile144'] / 1e9:5.0f} TF | half144 {best['half144'] * 1e3:6.1f} us {fl / best['half144'] / 1e9:5.0f} TF | bit-identical {same}", flush=True)
    del sets
