#!/usr/bin/env python3
"""Experiment: one batch of 16 as S concurrent sub-batches on S streams (S engines, separate graphs)."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
from ecad_amd.engine import EcadEngine
from ecad_amd.pipelines.dpm_solver import DPMSolverMultistepScheduler
from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule

dev = torch.device("cuda", 0)
bits = PixArtCacheSchedule.from_json(bench.SCHEDULE).to_numpy()
weights = bench.device_weights(type("C", (), dict(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2, num_layers=28, caption_channels=4096)), 0, dev)
solver = DPMSolverMultistepScheduler(); solver.set_timesteps(20)
ts = solver.timesteps.numpy().astype(np.float32); coefs = solver.coefficient_table()
enc, mask, lat = bench.synthetic_batch(16, 1000, dev)
for S in (1, 2, 4):
    B = 16 // S
    engs, streams, parts = [], [], []
    for s in range(S):
        e = EcadEngine(num_layers=28, max_batch=2 * B, dtype=torch.float16)
        e.load_state_dict(weights); e.set_schedule(bits)
        engs.append(e); streams.append(torch.cuda.Stream())
        idx = torch.arange(s * B, (s + 1) * B, device=dev)
        parts.append((torch.cat([enc[idx], enc[16 + idx]]).contiguous(), torch.cat([mask[idx], mask[16 + idx]]).contiguous(), lat[idx].contiguous()))
    def run():
        outs = []
        for e, st, (en, ma, la) in zip(engs, streams, parts):
            with torch.cuda.stream(st):
                outs.append(e.run_schedule(en, ma, la, ts, coefs, 4.5, use_graph=True, stream=st))
        return outs
    for _ in range(2): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"sub-batches={S} (B={B} each): {dt*1e3:.1f} ms per 16 images -> {16/dt:.1f} img/s")
    del engs
    torch.cuda.empty_cache()
