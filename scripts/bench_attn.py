#!/usr/bin/env python3
"""Attention micro-benchmark through the C ABI (shapes of config A), fused-QKV layout, random data."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ecad_amd.engine import EcadEngine
eng = EcadEngine(num_layers=1, max_batch=2, dtype=torch.float16)
B, H, dh, N, T = 32, 16, 72, 256, 120
D = H * dh
qkv = torch.randn(B * N, 3 * D, device="cuda").half()
q2 = torch.randn(B * N, D, device="cuda").half()
kv = torch.randn(B * T, 2 * D * 28, device="cuda").half()
bias = torch.zeros(B, T, device="cuda"); bias[:, 60:] = -10000.0
def run_self(): return eng.op_attention(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], B, H, N, N, dh)
def run_cross(): return eng.op_attention(q2, kv[:, :D], kv[:, D:2*D], B, H, N, T, dh, bias=bias)
for name, fn, fl in (("self", run_self, 4.0*B*H*N*N*dh), ("cross", run_cross, 4.0*B*H*N*T*dh)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 50
    print(f"{name}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s")
