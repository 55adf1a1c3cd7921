#!/usr/bin/env python3
"""In-process A/B of GEMM kernels / K-loop schedules on the model's shapes (random data, interleaved rounds, cold operand rotation):
   python scripts/ab_gemm.py                        every shape: schedule 0 vs 1 on the kernel the engine uses there
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402

CASES = [  # name, kernel, site, epi, M, N, K
    ("q", "tile144", "qkv", 0, 8192, 1152, 1152),
    ("out1", "tile144", "attn1_out", 3, 8192, 1152, 1152),
    ("q2", "tile144", "attn2_q", 0, 8192, 1152, 1152),
    ("out2", "tile144", "attn2_out", 3, 8192, 1152, 1152),
    ("ff2", "tile144", "ff2", 3, 8192, 1152, 4608),
    ("kv", "wide288", "attn1_kv", 0, 8192, 2304, 1152),
    ("ff1", "wide288", "ff1", 1, 8192, 4608, 1152),
]


def main():
    import os
    VAR = tuple(int(v) for v in os.environ.get("AB_SCHEDS", "0,1").split(","))
    only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
    eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
    for name, kernel, site, epi, M, N, K in CASES:
        if only and name not in only:
            continue
        per_set = (M * K + N * K + M * N * (2 if epi == 3 else 1)) * 2
        nsets = max(2, min(12, int(600e6 // per_set) + 1))  # rotate over > 256 MiB so nothing sits in the Infinity Cache, as inside the 28-block pipeline
        sets = []
        for i in range(nsets):
            a = torch.randn(M, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * 0.03).half()
            out = torch.empty(M, N, device="cuda", dtype=torch.float16)
            resid = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 3 else None
            sets.append((a, w, out, resid))
        bias = torch.randn(N, device="cuda") * 0.1
        gate = torch.randn(N, device="cuda") * 0.1 if epi == 3 else None
        reps = max(24, 2 * nsets)
        res = {v: [] for v in VAR}
        # the two schedules must agree bit for bit (same MFMA order per accumulator)
        a, w, out, resid = sets[0]
        o = []
        for v in VAR:
            r = torch.zeros_like(out) if epi == 3 else None
            c = torch.empty_like(out)
            eng.op_gemm_ex(kernel, site, epi, a, w, bias, resid=r, gate_table=gate, rows_per_batch=256, out=c, sched=v)
            o.append((c.clone(), None if r is None else r.clone()))
        same = torch.equal(o[0][0], o[1][0]) and (o[0][1] is None or torch.equal(o[0][1], o[1][1]))
        for rnd in range(4):
            for v in VAR:
                for i in range(nsets):
                    a, w, out, resid = sets[i]
                    eng.op_gemm_ex(kernel, site, epi, a, w, bias, resid=resid, gate_table=gate, rows_per_batch=256, out=out, sched=v)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for i in range(reps):
                    a, w, out, resid = sets[i % nsets]
                    eng.op_gemm_ex(kernel, site, epi, a, w, bias, resid=resid, gate_table=gate, rows_per_batch=256, out=out, sched=v)
                e.record()
                torch.cuda.synchronize()
                res[v].append(s.elapsed_time(e) / reps)
        line = f"{name:5s} {kernel}/{site} M={M} N={N} K={K} epi={epi} sets={nsets} bit-identical={same}:"
        for v in VAR:
            ms = sorted(res[v])[len(res[v]) // 2]
            line += f"  sched{v}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.0f} TF (min {min(res[v]) * 1e3:.1f})"
        print(line, flush=True)
        del sets


if __name__ == "__main__":
    main()
