"""-m gpu: the reference-shaped host surface end to end (image generator -> pipeline -> engine), on synthetic
embedding files in the reference's ``.pt`` format."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from gpu_util import within

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
FAST = ROOT / "schedules/schedules_in_paper/pixart_alpha_256/ours_fast.json"


@pytest.fixture(scope="module")
def embedding_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("emb")
    g = torch.Generator().manual_seed(0)
    neg = torch.randn(1, 120, 4096, generator=g) * 0.1
    for i in range(4):
        n = int(torch.randint(8, 121, (1,), generator=g))
        mask = torch.zeros(1, 120, dtype=torch.int64)
        mask[:, :n] = 1
        neg_mask = torch.zeros(1, 120, dtype=torch.int64)
        neg_mask[:, :2] = 1
        torch.save({"prompt_embeds": torch.randn(1, 120, 4096, generator=g) * 0.1, "prompt_attention_mask": mask,
                    "negative_prompt_embeds": neg.clone(), "negative_prompt_attention_mask": neg_mask}, d / f"{i:03}__prompt_id:{i}.pt")
    return d


def test_generator_surface_and_both_loops_agree(embedding_dir, tmp_path, capsys):
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type

    gen = get_image_generator_type("PixArtAlphaImageGenerator")(start_seed=7, seed_step=3, schedule_path=FAST, max_batch=8)
    assert gen.num_inference_steps == 20 and gen.cache_schedule.name.startswith("handcrafted")
    assert gen.callbacks[-1] == gen._reset_schedules_callback  # reset MUST be last (image_generator.py:153-159)
    gen.create_diffusion_pipeline()
    batch = next(iter(gen.load_and_batch_embeddings(embedding_dir, 4)))
    a = gen.generate_images(batch, images_per_prompt=1, output_type="latent")  # whole loop in one graph replay
    assert gen.cache_schedule.curr_step == 0 and gen.dit_scheduler.curr_step == 0  # counters reset after step 19
    gen.diffusion_pipeline.native_loop = False  # per-step path: 20 x ecad_step, solver in torch
    b = gen.generate_images(batch, images_per_prompt=1, output_type="latent")
    assert gen.cache_schedule.curr_step == 0
    la, lb = torch.stack([x[0] for x in a]).float(), torch.stack([x[0] for x in b]).float()
    rel = float((la - lb).norm() / lb.norm())
    print(f"native loop vs per-step loop rel-L2: {rel:.2e}")
    assert torch.isfinite(la).all() and within(rel, 3e-3)  # same arithmetic up to where fp16 rounding is applied in the solver
    # PNG path + reference file naming
    gen.diffusion_pipeline.native_loop = True
    gen.generate_from_saved_prompts(embedding_dir, tmp_path / "out", batch_size=2, images_per_prompt=2)
    names = sorted(p.name for p in (tmp_path / "out").glob("*.png"))
    assert len(names) == 8 and names[0] == "000__prompt_id:0__image_seed:007.png" and names[1] == "000__prompt_id:0__image_seed:010.png"


def test_generation_with_three_batches_in_flight_writes_the_same_files(embedding_dir, tmp_path):
    """`contexts=3` (EngineContexts behind pipeline.submit / finish): generate_from_saved_prompts keeps three (batch, seed) units in flight; file names and
    pixels are those of the one-at-a-time run, the schedule counters end at 0, and freeing the pipeline releases the contexts before the donor engine."""
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type

    outs = {}
    for n in (1, 3):
        gen = get_image_generator_type("PixArtAlphaImageGenerator")(start_seed=7, seed_step=3, schedule_path=FAST, max_batch=4, contexts=n)
        gen.generate_from_saved_prompts(embedding_dir, tmp_path / f"c{n}", batch_size=2, images_per_prompt=3)
        assert gen.cache_schedule.curr_step == 0 and gen.dit_scheduler.curr_step == 0
        if n == 3:
            assert len(gen.diffusion_pipeline._pool) == 3
        gen.free_diffusion_pipeline()
        outs[n] = {p.name: p.read_bytes() for p in sorted((tmp_path / f"c{n}").glob("*.png"))}
    assert len(outs[1]) == 12 and outs[1].keys() == outs[3].keys()
    assert all(outs[1][k] == outs[3][k] for k in outs[1])


def test_compute_latency_writes_metrics(embedding_dir, tmp_path):
    from ecad_amd.benchmark.compute_latency import time_for_schedule
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type

    sched = tmp_path / "ours_fast.json"
    data = json.load(open(FAST))
    data["metrics"].pop("latency", None)
    json.dump(data, open(sched, "w"))
    out = time_for_schedule(get_image_generator_type("PixArtAlphaImageGenerator"), sched, embedding_dir, batch_size=4, warmup_steps=1, num_samples=2,
                            seed=0, recompute_existing=False)
    lat = json.load(open(sched))["metrics"]["latency"]
    assert lat["batch_size"] == 4 and len(lat["latencies"]) == 2 and len(lat["warmups"]) == 1 and lat["avg"] == out["avg"] > 0
    assert time_for_schedule(get_image_generator_type("PixArtAlphaImageGenerator"), sched, embedding_dir, 4, 1, 2, 0, False) is None  # skip existing


def test_bad_schedules_raise(tmp_path):
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type
    from ecad_amd.engine import EcadEngine

    data = json.load(open(FAST))
    data["cache_schedule"]["schedule"]["003"]["5"]["custom_compute_attn"] = {"name": "compute_attn_tgate", "kwargs": {"gate_step": 8}}
    p = tmp_path / "tgate.json"
    json.dump(data, open(p, "w"))
    gen = get_image_generator_type("PixArtAlphaImageGenerator")(schedule_path=p, max_batch=2)
    # compute_attn_tgate is the TGATE baseline's function (out of scope, never registered here): naming an unregistered function is an error when the
    # schedule is attached -- registered plug-ins run (tests/test_gpu_plugins.py)
    with pytest.raises(ValueError, match="compute_attn_tgate not found in registry"):
        gen.create_diffusion_pipeline()
    eng = EcadEngine(num_layers=2, max_batch=2)
    with pytest.raises(ValueError, match="blocks"):
        eng.set_schedule(np.ones((20, 28, 3), dtype=bool))
    with pytest.raises(RuntimeError, match="never loaded"):
        eng.load_state_dict({})


def test_alpha_1024_ms_schedule_runs_micro_conditioned(embedding_dir):
    """A schedule file whose config names PixArt-XL-2-1024-MS (sample_size 128 => use_additional_conditions) at 1024 x 1024: the pipeline
    supplies resolution / aspect ratio as diffusers' does (pass_through.py:268-290); the native loop and the per-step forward() path see the
    same conditioning; forward() without added_cond_kwargs raises the reference's ValueError (pixart_transformer_2d_edited.py:208-211)."""
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type_from_config

    path = ROOT / "schedules/alpha_cache_schedules/gen_default_1024x1024/default_1024x1024.json"
    cfg = json.loads(path.read_text())["config"]
    gen = get_image_generator_type_from_config({"image_generator": "PixArtAlphaImageGenerator"})(schedule_path=path, max_batch=2)
    assert gen.transformer_weights == cfg["transformer_weights"] == "PixArt-alpha/PixArt-XL-2-1024-MS" and (gen.height, gen.width) == (1024, 1024)
    gen.create_diffusion_pipeline()
    tr = gen.diffusion_pipeline.transformer
    assert tr.use_additional_conditions and tr.config.sample_size == 128
    batch = next(iter(gen.load_and_batch_embeddings(embedding_dir, 1)))
    a = gen.generate_images(batch, images_per_prompt=1, output_type="latent")[0][0].float()
    assert a.shape == (4, 128, 128) and torch.isfinite(a).all()
    st = tr.engine.stats()
    vec = json.loads((ROOT / "tests/golden/alpha1024_vectors.json").read_text())["alpha_cache_schedules/gen_default_1024x1024/default_1024x1024.json"]
    assert st["linear_macs_reference"] == vec["total_macs"]  # the reference's calflops total for this file (batch 2)
    gen.diffusion_pipeline.native_loop = False
    b = gen.generate_images(batch, images_per_prompt=1, output_type="latent")[0][0].float()
    rel = float((a - b).norm() / b.norm())
    print(f"1024-MS native loop vs per-step forward(): rel-L2 {rel:.2e}")
    assert within(rel, 3.5e-3)
    with pytest.raises(ValueError, match="added_cond_kwargs"):
        tr(torch.zeros(2, 4, 128, 128), encoder_hidden_states=torch.zeros(2, 120, 4096), timestep=torch.zeros(2), added_cond_kwargs=None)
    gen.free_diffusion_pipeline()


def test_sigma_1024_schedule_file_through_the_sigma_generator(tmp_path):
    """BASELINE config 3 as a workload a user can name: the authored schedule file (a 256 -> 1024 transfer of the paper's Sigma `ours_fast`,
    scripts/transfer_schedule.py) carries config.transformer_weights = PixArt-Sigma-XL-2-1024-MS, height / width 1024 and image_generator =
    PixArtSigmaImageGenerator; `get_image_generator_type_from_config` -> PixArtSigmaImageGenerator (pixart_sigma_image_generator.py:1-38) ->
    PixArtSigmaPipeline -> engine at N = 4096, T = 300, sincos interpolation_scale 2, no micro-conditioning.  The whole-loop graph and the per-step
    forward() path agree; the engine's Linear MACs equal the closed form at this run's text length."""
    from ecad_amd.benchmark.compute_macs import macs_per_step
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type_from_config

    path = ROOT / "schedules/sigma_cache_schedules/transfer_256_to_1024/ours_fast_256_to_1024.json"
    data = json.loads(path.read_text())
    src = json.loads((ROOT / "schedules/schedules_in_paper/pixart_sigma_256/ours_fast.json").read_text())
    assert data["cache_schedule"] == src["cache_schedule"]  # a transfer keeps the schedule body (as flux_256_to_1024/* keep flux_256/*'s)
    T = 300
    g = torch.Generator().manual_seed(3)
    neg_mask = torch.zeros(1, T, dtype=torch.int64)
    neg_mask[:, :2] = 1
    mask = torch.zeros(1, T, dtype=torch.int64)
    mask[:, :211] = 1
    torch.save({"prompt_embeds": torch.randn(1, T, 4096, generator=g) * 0.1, "prompt_attention_mask": mask,
                "negative_prompt_embeds": torch.randn(1, T, 4096, generator=g) * 0.1, "negative_prompt_attention_mask": neg_mask}, tmp_path / "000__prompt_id:0.pt")
    gen_type = get_image_generator_type_from_config(data["config"])
    assert gen_type.__name__ == "PixArtSigmaImageGenerator"
    gen = gen_type(schedule_path=path, max_batch=2)
    assert gen.transformer_weights == "PixArt-alpha/PixArt-Sigma-XL-2-1024-MS" and (gen.height, gen.width) == (1024, 1024)
    assert gen.cache_schedule.name == src["cache_schedule"]["name"] and gen.num_inference_steps == 20
    gen.create_diffusion_pipeline()
    tr = gen.diffusion_pipeline.transformer
    assert type(gen.diffusion_pipeline).__name__ == "PixArtSigmaPipeline"
    assert tr.config.sample_size == 128 and tr.config.interpolation_scale == 2 and not tr.use_additional_conditions
    batch = next(iter(gen.load_and_batch_embeddings(tmp_path, 1)))
    a = gen.generate_images(batch, images_per_prompt=1, output_type="latent")[0][0].float()
    assert a.shape == (4, 128, 128) and torch.isfinite(a).all()
    assert gen.cache_schedule.curr_step == 0
    st = tr.engine.stats()
    assert st["linear_macs_reference"] == int(macs_per_step(gen.cache_schedule.to_numpy(), tokens=4096, text_tokens=T, batch=2).sum())
    gen.diffusion_pipeline.native_loop = False
    b = gen.generate_images(batch, images_per_prompt=1, output_type="latent")[0][0].float()
    rel = float((a - b).norm() / b.norm())
    print(f"Sigma-1024 native loop vs per-step forward(): rel-L2 {rel:.2e}")
    assert within(rel, 4e-3)
    imgs = gen.generate_images(batch, images_per_prompt=1, output_type="pil")  # decode tail at 1024 x 1024
    assert imgs[0][0].size == (1024, 1024)
    gen.free_diffusion_pipeline()


def test_pipeline_without_classifier_free_guidance(embedding_dir):
    """guidance_scale <= 1 (diffusers' do_classifier_free_guidance False): only the conditional half runs, through the per-step path.  With
    guidance 1 the CFG formula u + 1 * (c - u) collapses to c, so the CFG-doubled run at guidance 1 + 1e-6 must give the same latents."""
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type

    gen = get_image_generator_type("PixArtAlphaImageGenerator")(schedule_path=FAST, max_batch=4)
    gen.create_diffusion_pipeline()
    batch = next(iter(gen.load_and_batch_embeddings(embedding_dir, 2)))
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(3)).half()
    kw = gen._pipeline_kwargs(batch, 256, 256, "latent")
    a = gen.diffusion_pipeline(**{**kw, "guidance_scale": 1.0, "latents": lat.clone()})[0].float()
    assert gen.cache_schedule.curr_step == 0  # counters were driven and reset
    gen.diffusion_pipeline.native_loop = False
    b = gen.diffusion_pipeline(**{**kw, "guidance_scale": 1.0 + 1e-6, "latents": lat.clone()})[0].float()
    rel = float((a - b).norm() / b.norm())
    print(f"no-CFG vs CFG at guidance 1: rel-L2 {rel:.2e}")
    assert torch.isfinite(a).all() and within(rel, 1e-3)
    gen.free_diffusion_pipeline()
