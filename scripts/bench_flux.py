#!/usr/bin/env python3
"""FLUX.1-dev geometry at 1024x1024 (N = 4096 image tokens + T = 512 text tokens, D = 3072, 19 + 38 blocks), batch 2, 20 steps with
the reference's transferred schedule (schedules_in_paper/flux_256_to_1024/fast_256_to_1024.json); random-init weights.
Prints images/s, the per-kernel-class device-time breakdown and achieved TFLOP/s per class."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule  # noqa: E402
from ecad_amd.transformer_2d_models.flux_transformer_2d_edited import FluxTransformer2DEdited, synthetic_state_dict  # noqa: E402
from ecad_amd.pipelines.flow_match import flow_match_sigmas, model_scalars  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schedule", default="schedules/schedules_in_paper/flux_256_to_1024/fast_256_to_1024.json")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--text", type=int, default=512)
    ap.add_argument("--layers", type=int, default=19)
    ap.add_argument("--single", type=int, default=38)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--json", default=None, help="write the result + per-class profile here")
    ap.add_argument("--fp8", action="store_true", help="opt-in OCP e4m3 block GEMMs (ecad_flux_set_fp8); also reports the distance of the final latents to the bf16 engine")
    ap.add_argument("--decode-tail", action="store_true", help="also time the 16-channel VAE decode + uint8 postprocess on the native decoder")
    a = ap.parse_args()
    t0 = time.time()
    sched = FluxCacheSchedule.from_json(ROOT / a.schedule)
    bf, bs = sched.to_device_bits()
    bf, bs = bf[:, : a.layers], bs[:, : a.single]
    model = FluxTransformer2DEdited(num_layers=a.layers, num_single_layers=a.single, guidance_embeds=True, max_batch=a.batch, max_image_tokens=a.grid * a.grid,
                                    max_text_tokens=a.text, torch_dtype=torch.bfloat16)
    weights = synthetic_state_dict(model.config, 0, device="cuda", dtype=torch.bfloat16)
    eng = model.engine
    if a.fp8:
        eng._chk(eng.lib.ecad_flux_set_fp8(eng._h, 1))
    model.load_state_dict(weights)
    eng.set_schedule(bf, bs)
    print(f"[{time.time() - t0:5.1f}s] engine ready: {eng.stats()}", flush=True)
    B, N, T = a.batch, a.grid * a.grid, a.text
    g = torch.Generator(device="cuda").manual_seed(1)
    enc = (torch.randn(B, T, 4096, generator=g, device="cuda") * 0.1).bfloat16()
    pooled = (torch.randn(B, 768, generator=g, device="cuda") * 0.5).bfloat16()
    steps = bf.shape[0]
    sig = flow_match_sigmas(steps, N)
    tm, gm = model_scalars(sig, 5.0, torch.bfloat16)

    def run(graph):
        x = torch.randn(B, N, 64, generator=g, device="cuda")
        eng.run_schedule(x, enc, pooled, a.grid, a.grid, tm, gm, sig, use_graph=graph)
        return x

    x = run(True)
    torch.cuda.synchronize()
    if a.fp8:  # same inputs through a bf16 engine: how far do 20 cached steps of e4m3 GEMMs move the final latents?
        ref_model = FluxTransformer2DEdited(num_layers=a.layers, num_single_layers=a.single, guidance_embeds=True, max_batch=a.batch, max_image_tokens=a.grid * a.grid,
                                            max_text_tokens=a.text, torch_dtype=torch.bfloat16)
        ref_model.engine._chk(ref_model.engine.lib.ecad_flux_set_fp8(ref_model.engine._h, 0))
        ref_model.load_state_dict(weights)
        ref_model.engine.set_schedule(bf, bs)
        g2 = torch.Generator(device="cuda").manual_seed(77)
        x0 = torch.randn(B, N, 64, generator=g2, device="cuda")
        xa = eng.run_schedule(x0.clone(), enc, pooled, a.grid, a.grid, tm, gm, sig, use_graph=False)
        xb = ref_model.engine.run_schedule(x0.clone(), enc, pooled, a.grid, a.grid, tm, gm, sig, use_graph=False)
        torch.cuda.synchronize()
        d = float((xa.double() - xb.double()).norm() / xb.double().norm())
        print(f"fp8 vs bf16 engine, final latents after {steps} steps: rel-L2 {d:.3e} (cosine {float(torch.nn.functional.cosine_similarity(xa.flatten(), xb.flatten(), dim=0)):.5f})")
        del ref_model
        torch.cuda.empty_cache()
    del weights
    print(f"[{time.time() - t0:5.1f}s] warm; latents finite: {bool(torch.isfinite(x).all())}, std {float(x.std()):.3f}", flush=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        run(True)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.iters
    st = eng.stats()
    tflop = 2 * (st["linear_macs_reference"] + st["attn_core_macs"]) * B / 1e12
    print(f"flux {a.grid * 16}x{a.grid * 16} batch {B}: {ms:.1f} ms per batch = {B / ms * 1e3:.3f} images/s; reference-equivalent {tflop / B:.1f} TFLOP/image -> {tflop / ms * 1e3:.0f} TFLOP/s whole job")
    if a.decode_tail:
        from ecad_amd.pipelines.flux import FluxPipeline
        from ecad_amd.pipelines.vae_decoder import HipVaeDecoder

        torch.manual_seed(0)
        vae = HipVaeDecoder.random_init(0, dtype=torch.bfloat16, flux=True)
        lat = FluxPipeline._unpack_latents(run(True).bfloat16(), a.grid * 16, a.grid * 16)
        for _ in range(2):
            img = vae.decode_to_uint8(lat)
        torch.cuda.synchronize()
        s.record()
        for _ in range(3):
            img = vae.decode_to_uint8(lat)
        e.record()
        torch.cuda.synchronize()
        dms = s.elapsed_time(e) / 3
        print(f"decode tail (HIP FLUX-geometry VAE, random init): {dms:.1f} ms per batch of {B} -> png-ready {B / (ms + dms) * 1e3:.3f} images/s; image {tuple(img.shape)} {img.dtype}")
    if not a.no_profile:
        eng.set_profiling(True)
        run(False)
        torch.cuda.synchronize()
        prof = eng.get_profile()
        eng.set_profiling(False)
        tot = sum(v["ms"] for v in prof.values())
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            if v["launches"]:
                print(f"  {k:24s} {v['ms']:9.2f} ms {100 * v['ms'] / tot:5.1f} %  {v['launches']:5d} launches  {v['flops'] / v['ms'] / 1e9 if v['ms'] else 0:7.0f} TFLOP/s  {v['bytes'] / v['ms'] / 1e6 if v['ms'] else 0:7.0f} GB/s")
        print(f"  sum {tot:.1f} ms")
        if a.json:
            Path(a.json).write_text(json.dumps({"workload": f"flux {a.grid * 16}x{a.grid * 16} batch {B}, {a.schedule}" + (", fp8 e4m3 block GEMMs" if a.fp8 else ""), "ms_per_batch": ms, "images_per_s": B / ms * 1e3,
                                                "reference_equivalent_tflop_per_image": tflop / B, "whole_job_tflops": tflop / ms * 1e3, "stats": st, "profile": prof}, indent=1))


if __name__ == "__main__":
    main()
