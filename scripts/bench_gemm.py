#!/usr/bin/env python3
"""GEMM micro-benchmark through the C ABI (ecad_op_gemm) on the model's shapes, random data.

--cold rotates over enough distinct (A, W, out) sets that nothing is resident in the 256 MiB Infinity Cache
(the situation inside the 28-block pipeline, where every block has its own weights)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402

SHAPES = [  # (name, M, N, K, epi)
    ("qkv", 8192, 3456, 1152, 0),
    ("out1", 8192, 1152, 1152, 3),
    ("q2", 8192, 1152, 1152, 0),
    ("ff1", 8192, 4608, 1152, 1),
    ("ff1nogelu", 8192, 4608, 1152, 0),
    ("ff2", 8192, 1152, 4608, 3),
    ("textkv", 3840, 64512, 1152, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import os

    engines = {}
    for impl in ("classic", "wide"):  # both kernels in ONE process, interleaved (guide rule 24)
        os.environ["ECAD_GEMM"] = impl
        engines[impl] = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
    for name, M, N, K, epi in SHAPES:
        if args.only and name not in args.only.split(","):
            continue
        per_set = (M * K + N * K + M * N * (2 if epi == 3 else 1)) * 2
        nsets = max(1, min(12, int(600e6 // per_set) + 1)) if args.cold else 1
        sets = []
        for i in range(nsets):
            a = torch.randn(M, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * 0.03).half()
            out = torch.empty(M, N, device="cuda", dtype=torch.float16)
            resid = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 3 else None
            sets.append((a, w, out, resid))
        bias = torch.randn(N, device="cuda") * 0.1
        reps = max(20, 2 * nsets)
        res = {k: [] for k in engines}
        for rnd in range(3):
            for impl, eng in engines.items():
                for i in range(nsets):
                    a, w, out, resid = sets[i]
                    eng.op_gemm(epi, a, w, bias, resid=resid, out=out)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for i in range(reps):
                    a, w, out, resid = sets[i % nsets]
                    eng.op_gemm(epi, a, w, bias, resid=resid, out=out)
                e.record()
                torch.cuda.synchronize()
                res[impl].append(s.elapsed_time(e) / reps)
        line = f"{name:8s} M={M} N={N} K={K} epi={epi} sets={nsets}:"
        for impl, v in res.items():
            ms = min(v)
            line += f"  {impl}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.0f} TF"
        print(line)
        del sets


if __name__ == "__main__":
    main()
