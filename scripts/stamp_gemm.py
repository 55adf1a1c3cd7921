#!/usr/bin/env python3
"""Diagnostic (make -C ecad_amd/csrc diag; ECAD_HIP_LIB=build_tmp/libecad_diag.so): per-workgroup cycles of prologue / K loop / epilogue and the
in-kernel clock of gemm2w / gemm3s on the model's shapes (random data, after >= 1 s of back-to-back launches: MI355X_MICROARCH.md DVFS item 6)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402

eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
import os
CASES = [("wide288", "ff1", 1, 8192, 4608, 1152), ("wide288", "attn1_kv", 0, 8192, 2304, 1152), ("tile144", "qkv", 0, 8192, 1152, 1152),
         ("tile144", "attn1_out", 3, 8192, 1152, 1152), ("tile144", "ff2", 3, 8192, 1152, 4608)]
if os.environ.get("STAMP_LATENCY"):  # the 64-row latency tiles of a batch of 1 (M = 512): how long is a K-step when a workgroup has nobody to hide its DMA latency behind?
    CASES = [("latency64", "qkv", 0, 512, 1152, 1152), ("latency64", "attn1_kv", 0, 512, 2304, 1152), ("latency64", "ff1", 1, 512, 4608, 1152), ("latency64", "attn1_out", 3, 512, 1152, 1152)]
elif os.environ.get("STAMP_SMALL"):  # few workgroups: is the epilogue bound by the chip (HBM / fabric) or by the CU's own store path?
    CASES = [("tile144", "qkv", 0, 256 * m, 1152, 1152) for m in (1, 4, 8, 16, 32)] + [("wide288", "attn1_kv", 0, 256 * m, 2304, 1152) for m in (1, 4, 16, 32)]
for kernel, site, epi, M, N, K in CASES:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.03).half()
    bias = torch.randn(N, device="cuda") * 0.1
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    resid = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 3 else None
    dbg = torch.zeros(8 * 4096, device="cuda", dtype=torch.int64)
    kw = dict(resid=resid, out=out, gate_mod=dbg)
    for _ in range(3000 if K < 2000 else 1500):
        eng.op_gemm_ex(kernel, site, epi, a, w, bias, **kw)
    torch.cuda.synchronize()
    dbg.zero_()
    eng.op_gemm_ex(kernel, site, epi, a, w, bias, **kw)
    torch.cuda.synchronize()
    v = dbg.view(-1, 8).cpu().double()
    v = v[v[:, 3] > 0]
    med = v.median(0).values
    t0 = v[:, 4].min()
    start = (v[:, 4] - t0) / 100.0  # us
    span = ((v[:, 4] + v[:, 3]).max() - t0) / 100.0
    clock = med[5] / (med[3] / 100.0) / 1e3
    steps = K // 64
    if kernel == "wide288" and len(v) > 256:  # a walking workgroup leaves one record per tile: rounds of 256 tiles, timeline in us from the first workgroup's start
        ends = (v[:, 4] + v[:, 3] - t0) / 100.0
        order = torch.argsort(v[:, 4])
        for r in range(len(v) // 256):
            sel = order[r * 256:(r + 1) * 256]
            print(f"   round {r}: tile starts {start[sel].min():.1f} / {start[sel].median():.1f} / {start[sel].max():.1f} us (min / median / max), ends {ends[sel].min():.1f} / {ends[sel].median():.1f} / {ends[sel].max():.1f} us;"
                  f" per tile: fill {v[sel, 0].median():.0f}  loop {v[sel, 1].median():.0f}  epilogue {v[sel, 2].median():.0f} cycles, wall {v[sel, 3].median() / 100:.2f} us")
    print(f"{kernel}/{site} M={M} N={N} K={K} epi={epi}: WGs {len(v)}  prologue {med[0]:.0f}  loop {med[1]:.0f} ({med[1] / steps:.0f}/step)  epilogue {med[2]:.0f} cycles (skew + bias {med[6]:.0f}, acc -> LDS {med[7]:.0f}, LDS -> global {med[2] - med[6] - med[7]:.0f});"
          f"  WG wall {med[3] / 100:.2f} us, clock {clock:.3f} GHz, kernel span {span:.1f} us, starts: median {start.median():.1f} us max {start.max():.1f} us")
