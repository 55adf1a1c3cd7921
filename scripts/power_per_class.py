#!/usr/bin/env python3
"""Board power x duration per kernel class: joules per flop of the hot kernels, through the C ABI on the product library (no diagnostic build).

If the ceiling of this board is its 1400 W cap, an A/B between two kernels has to be read in energy per flop as well as in microseconds: each class
runs back to back for ~2.5 s on its own (operands rotated through > 256 MiB where the launch is short, as inside the pipeline) while `rocm-smi` is sampled from a
thread; reported per class: us per launch, TFLOP/s, mean board watts while busy, median sclk, and W us per GFLOP (= microjoules per GFLOP).

    python scripts/power_per_class.py [out.json]        # on the GPU box; ~40 s
"""
import json
import os
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine  # noqa: E402
from ecad_amd.flux_engine import EcadFluxEngine  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                continue
            p = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            if p:
                self.rows.append((time.time(), float(p.group(1)), int(c.group(1)) if c else 0))


def measure(name, launch, flops, seconds=2.5):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    smp = Sampler()
    smp.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            launch()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    smp.stop = True
    smp.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    rows = [r for r in smp.rows if r[0] > t0 + 0.5]
    watts = sorted(r[1] for r in rows)
    busy = [w for w in watts if w > 0.6 * watts[-1]] if watts else []
    w = sum(busy) / len(busy) if busy else float("nan")
    clk = sorted(r[2] for r in rows)
    res = {"us_per_launch": us, "tflops": flops / us / 1e6, "watts_busy_mean": w, "watts_max": watts[-1] if watts else None, "sclk_mhz_median": clk[len(clk) // 2] if clk else None,
           "samples": len(rows), "w_us_per_gflop": w * us / (flops / 1e9)}
    print(f"{name:34s} {us:9.1f} us  {res['tflops']:7.0f} TF  {w:6.0f} W  sclk {res['sclk_mhz_median']} MHz  {res['w_us_per_gflop']:.3f} W us / GFLOP  ({len(rows)} samples)", flush=True)
    return res


def main():
    out = {}
    eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
    gem = [("gemm_q (N=K=1152, plain)", "tile144", "qkv", 0, 8192, 1152, 1152), ("gemm_attn1_out (read-modify-write)", "tile144", "attn1_out", 3, 8192, 1152, 1152),
           ("gemm_ff2 (K=4608, read-modify-write)", "tile144", "ff2", 3, 8192, 1152, 4608), ("gemm_attn1_kv_wide", "wide288", "attn1_kv", 0, 8192, 2304, 1152),
           ("gemm_ff1_gelu", "wide288", "ff1", 1, 8192, 4608, 1152),
           # round 5: the read-modify-write epilogue on the wide tile (gemm2w4.h).  Alone on the chip a 128-tile launch leaves half of it idle (the product takes it with batches in
           # flight only); whole rounds at the Sigma-1024 row count
           ("gemm_attn1_out on 256x288 tiles, half the chip", "wide288", "attn1_out", 3, 8192, 1152, 1152), ("gemm_ff2 on 256x288 tiles, half the chip", "wide288", "ff2", 3, 8192, 1152, 4608),
           ("gemm_ff2 M=32768 (256x144 tiles)", "tile144", "ff2", 3, 32768, 1152, 4608), ("gemm_ff2 M=32768 on 256x288 tiles", "wide288", "ff2", 3, 32768, 1152, 4608)]
    os.environ["ECAD_GEMM4"] = "0"
    eng8 = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)   # the 8-wave gemm2w.h behind the same kernel name
    os.environ.pop("ECAD_GEMM4", None)
    gem += [("gemm_attn1_kv_wide (gemm2w.h, 8 waves)", "wide288@8", "attn1_kv", 0, 8192, 2304, 1152), ("gemm_ff1_gelu (gemm2w.h, 8 waves)", "wide288@8", "ff1", 1, 8192, 4608, 1152)]
    for name, kernel, site, epi, M, N, K in gem:
        eng_k = eng8 if kernel.endswith("@8") else eng
        kernel = kernel.split("@")[0]
        per_set = (M * K + N * K + M * N * (2 if epi == 3 else 1)) * 2
        nsets = max(2, min(12, int(600e6 // per_set) + 1))
        sets = []
        for _ in range(nsets):
            a = torch.randn(M, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * 0.03).half()
            o = torch.empty(M, N, device="cuda", dtype=torch.float16)
            r = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 3 else None
            sets.append((a, w, o, r))
        bias = torch.randn(N, device="cuda") * 0.1
        gate = torch.randn(N, device="cuda") * 0.01 if epi == 3 else None
        k = [0]

        def launch():
            a, w, o, r = sets[k[0] % nsets]
            k[0] += 1
            eng_k.op_gemm_ex(kernel, site, epi, a, w, bias, resid=r, gate_table=gate, rows_per_batch=256, out=o)

        out[name] = measure(name, launch, 2.0 * M * N * K)
        del sets
    B, H, N, dh = 8, 16, 4096, 72
    qkv = (torch.randn(B * N, 3 * H * dh, device="cuda") * 0.5).half()
    q, k_, v = qkv[:, : H * dh], qkv[:, H * dh: 2 * H * dh], qkv[:, 2 * H * dh:]
    out["attn_self Sigma-1024 (d_h 72)"] = measure("attn_self Sigma-1024 (d_h 72)", lambda: eng.op_attention(q, k_, v, B, H, N, N, dh), 4.0 * B * H * N * N * dh)
    eng.close()
    eng8.close()
    del qkv
    for label, env in (("attn_joint FLUX (attention_w4.h)", None), ("attn_joint FLUX (attention_dma.h)", "dma")):
        if env:
            os.environ["ECAD_FLUX_ATTN"] = env
        else:
            os.environ.pop("ECAD_FLUX_ATTN", None)
        Bf, T, Nf, Hf = 2, 512, 4096, 24
        fe = EcadFluxEngine(num_layers=0, num_single_layers=0, num_heads=Hf, max_batch=Bf, max_image_tokens=Nf, max_text_tokens=T, dtype=torch.bfloat16)
        x = (torch.randn(Bf * (T + Nf), 3 * Hf * 128, device="cuda") * 0.5).bfloat16()
        out[label] = measure(label, lambda: fe.op_attention(x, Bf, T, Nf), 4.0 * Bf * (T + Nf) ** 2 * Hf * 128)
        fe.close()
        del x
    os.environ.pop("ECAD_FLUX_ATTN", None)
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
