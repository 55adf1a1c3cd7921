"""CPU tests for the FLUX row: oracle vs the reference's own known answers and captured vectors,
host-side schedule mirror, oracle self-consistency.  No GPU."""
import json
import math
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import flux_ref, schedule_ref

GOLD = Path(__file__).parent / "golden"
REPO = Path(__file__).resolve().parents[1]


def test_flux_mac_known_answers_all_reference_schedules():
    """Every FLUX schedule JSON of the reference that carries per-step MACs (calflops on the real model, batch 2):
    gate semantics + component -> cost mapping + the forced recompute on an empty cache."""
    kat = np.load(GOLD / "flux_mac_kat.npz")
    assert len(kat["names"]) == 38
    for name, packed, macs, geom in zip(kat["names"], kat["bits"], kat["macs"], kat["geom"]):
        h, w, nb, ns, steps = (int(v) for v in geom)
        n = steps * (nb + ns) * 3
        flat = np.unpackbits(packed)[:n].astype(bool)
        full, single = schedule_ref.flux_unflatten(flat, steps, nb, ns)
        assert (schedule_ref.flux_flatten(full, single) == flat).all()
        pred = schedule_ref.flux_mac_model(schedule_ref.resolve_executed(full), schedule_ref.resolve_executed(single), tokens=(h // 16) * (w // 16))
        assert pred.tolist() == macs[:steps].tolist(), name


def test_flux_schedule_vectors_match_reference_classes():
    from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule
    from ecad_amd.schedulers.cache_scheduler.generators.flux_schedule_generators import gen_default

    v = json.load(open(GOLD / "flux_vectors.json"))
    for key, e in v["to_numpy"].items():
        path = REPO / "schedules" / e["path"]
        s = FluxCacheSchedule.from_json(path)
        b = s.to_numpy()
        assert b.size == e["n"] and int(b.sum()) == e["sum"], key
        assert np.packbits(b).tobytes().hex() == e["packed_hex"], key
        full, single = schedule_ref.flux_bits_from_schedule_json(path)
        assert (schedule_ref.flux_flatten(full, single) == b).all()
        df, ds = s.to_device_bits()
        assert (df == full).all() and (ds == single).all()
        assert s.top_level_config == e["config"] and s.name == e["name"]
        # whole-schedule MAC total the reference measured
        h, w = e["config"]["height"], e["config"]["width"]
        tot = schedule_ref.flux_mac_model(schedule_ref.resolve_executed(full), schedule_ref.resolve_executed(single), tokens=(h // 16) * (w // 16)).sum()
        assert int(tot) == e["total_macs"], key
    s = FluxCacheSchedule.from_json(REPO / "schedules" / v["to_numpy"]["flux_1024_fast"]["path"])
    assert s.components == v["components"]
    d = s.to_dict()
    assert list(d.keys()) == v["to_dict_shape"]["top_keys"]
    assert list(d["cache_schedule"].keys()) == v["to_dict_shape"]["cache_schedule_keys"]
    assert list(d["cache_schedule"]["schedule"]["000"].keys()) == v["to_dict_shape"]["block_keys_step0"]
    assert d["cache_schedule"]["schedule"]["002"]["3"] == v["to_dict_shape"]["entry_002_3"]
    assert d["cache_schedule"]["schedule"]["002"]["single_5"] == v["to_dict_shape"]["entry_002_single_5"]
    trace = []
    s.reset_step()
    for i in range(20):
        trace.append([s.curr_step, bool(s.get_recompute("3", "full_ff")), bool(s.get_recompute("single_7", "single_proj_out"))])
        s.per_step_callback(i, 0)
    assert trace == v["step_trace"]
    with pytest.raises(ValueError):
        s.get_recompute("0", "attn1")
    with pytest.raises(ValueError):
        FluxCacheSchedule(19, 20, "x", {}, num_single_blocks=None)
    # population bit vector <-> schedule dict
    pop = v["population"]
    x = np.unpackbits(np.frombuffer(bytes.fromhex(pop["x_packed_hex"]), dtype=np.uint8))[: 20 * 57 * 3].astype(bool)
    rt = FluxCacheSchedule.from_numpy(x)
    assert (rt.to_numpy() == x).all()
    assert rt.schedule[5]["7"] == pop["entry_step5_block7"] and rt.schedule[5]["single_30"] == pop["entry_step5_single_30"]
    d0 = next(gen_default(19, 38, 20, 1024, 1024, 5))
    assert d0.name == v["default_schedule"]["name"] and d0.to_numpy().all() and d0.to_numpy().size == v["default_schedule"]["n"]
    assert d0.top_level_config == v["default_schedule"]["config"]


def _tiny():
    cfg = flux_ref.FluxConfig(num_layers=2, num_single_layers=3, num_attention_heads=2, joint_attention_dim=96, pooled_projection_dim=48)
    return cfg, flux_ref.make_weights(cfg, seed=3)


def test_flux_oracle_gate_log_matches_schedule_semantics():
    cfg, w = _tiny()
    o = flux_ref.FluxOracle(cfg, w)
    lat, enc, pooled, ids = flux_ref.synthetic_inputs(cfg, 2, 4, 6, 8)
    rng = np.random.default_rng(0)
    bf = rng.integers(0, 2, size=(5, 2, 3)).astype(bool)
    bs = rng.integers(0, 2, size=(5, 3, 3)).astype(bool)
    bf[0, 1, 0] = False  # empty cache at step 0 => forced recompute
    trace = []
    out = flux_ref.denoise(o, lat, enc, pooled, ids, bf, bs, trace=trace)
    assert out.shape == lat.shape and torch.isfinite(out).all()
    ef, es = schedule_ref.resolve_executed(bf), schedule_ref.resolve_executed(bs)
    assert ef[0].all() and es[0].all()
    log = set(o.cache.executed_log)
    for s in range(5):
        for b in range(2):
            for c, name in enumerate(schedule_ref.FLUX_FULL):
                assert ((s, str(b), name) in log) == bool(ef[s, b, c])
        for b in range(3):
            for c, name in enumerate(schedule_ref.FLUX_SINGLE):
                assert ((s, f"single_{b}", name) in log) == bool(es[s, b, c])
    assert len(log) == int(ef.sum() + es.sum())


def test_flux_oracle_properties():
    cfg, w = _tiny()
    lat, enc, pooled, ids = flux_ref.synthetic_inputs(cfg, 2, 4, 4, 8)
    # RoPE is a rotation: norms of consecutive pairs are preserved; zero ids = identity
    cos, sin = flux_ref.rope_table(ids, cfg.axes_dims_rope)
    assert cos.shape == (24, 64) and torch.allclose(cos[:8], torch.ones(8, 64)) and torch.allclose(sin[:8], torch.zeros(8, 64))
    x = torch.randn(1, 2, 24, 128)
    y = flux_ref.apply_rope(x, cos, sin)
    assert torch.allclose(x.reshape(1, 2, 24, 64, 2).norm(dim=-1), y.reshape(1, 2, 24, 64, 2).norm(dim=-1), atol=1e-5)
    assert torch.allclose(x[:, :, :8], y[:, :, :8])
    # an all-cached step reuses every sub-layer output: with identical inputs and timestep the prediction repeats exactly
    o = flux_ref.FluxOracle(cfg, w)
    B = 2
    t, g = torch.full((B,), 700.0), torch.full((B,), 5000.0)
    on_f, on_s = np.ones((2, 3), bool), np.ones((3, 3), bool)
    v0 = o.forward(lat, enc, pooled, t, g, ids, 0, on_f, on_s)
    v1 = o.forward(lat, enc, pooled, t, g, ids, 1, ~on_f, ~on_s)
    assert torch.equal(v0, v1)
    # batch independence: sample 0 alone gives the same prediction
    o2 = flux_ref.FluxOracle(cfg, w)
    v_single = o2.forward(lat[:1], enc[:1], pooled[:1], t[:1], g[:1], ids, 0, on_f, on_s)
    assert torch.allclose(v_single, v0[:1], atol=1e-5)
    # scheduler: sigmas decrease from ~1 to 0; mu shift at 4096 tokens is 1.15 (FLUX.1-dev scheduler_config.json)
    sig = flux_ref.flow_match_sigmas(20, 4096)
    assert sig.shape == (21,) and sig[-1] == 0 and (np.diff(sig) < 0).all() and abs(sig[0] - 1.0) < 1e-6
    tm, gm = flux_ref.model_scalars(sig, 5.0, torch.bfloat16)
    assert float(gm[0]) == 4992.0 and tm.shape == (20,)  # 5 * 1000 in bf16


def test_compute_macs_cli_annotates_flux_schedules(tmp_path):
    from ecad_amd.benchmark.compute_macs import annotate_schedule_file

    src = REPO / "schedules/schedules_in_paper/flux_256_to_1024/fast_256_to_1024.json"
    d = json.load(open(src))
    want = d["metrics"]["total_macs"], d["metrics"]["by_inference_step"]["002"]["macs"]
    d["metrics"] = {}
    p = tmp_path / "f.json"
    json.dump(d, open(p, "w"))
    assert annotate_schedule_file(p)
    m = json.load(open(p))["metrics"]
    assert (m["total_macs"], m["by_inference_step"]["002"]["macs"]) == want
    assert not annotate_schedule_file(p)  # already annotated


def test_flux_host_surface_without_gpu():
    """Registries, latent packing, ids, flow-match constants: the host pieces that need no device."""
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type
    from ecad_amd.pipelines.flow_match import calculate_shift, flow_match_sigmas, model_scalars
    from ecad_amd.pipelines.flux import FluxPipeline
    from ecad_amd.pipelines.load_pipeline import PipelineRegistry
    from ecad_amd.transformer_2d_models.flux_transformer_2d_edited import state_dict_shapes, synthetic_state_dict

    assert PipelineRegistry.get("flux") is FluxPipeline
    assert get_image_generator_type("FluxImageGenerator").__name__ == "FluxImageGenerator"
    x = torch.randn(2, 16, 8, 12)
    packed = FluxPipeline._pack_latents(x)
    assert packed.shape == (2, 24, 64) and torch.equal(FluxPipeline._unpack_latents(packed, 64, 96), x)
    assert torch.equal(FluxPipeline._prepare_latent_image_ids(4, 6, "cpu", torch.float32), flux_ref.latent_image_ids(4, 6))
    assert abs(calculate_shift(256) - 0.5) < 1e-12 and abs(calculate_shift(4096) - 1.15) < 1e-12
    sig = flow_match_sigmas(20, 4096)
    assert np.array_equal(sig, flux_ref.flow_match_sigmas(20, 4096))
    # diffusers 0.30.3 FluxPipeline: mu = calculate_shift(seq, scheduler.config.base_image_seq_len, max_image_seq_len, base_shift, max_shift) with FLUX.1-dev's
    # scheduler_config.json (max_shift 1.15) -> sigma_1 at 1024^2 = e^1.15 / (e^1.15 + 1/0.95 - 1); a checkpoint directory's own config wins
    assert abs(float(sig[1]) - math.exp(1.15) / (math.exp(1.15) + 1 / 0.95 - 1)) < 1e-6
    import tempfile

    from ecad_amd.pipelines.flow_match import load_scheduler_config

    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "scheduler"))
        json.dump({"max_shift": 1.3, "base_shift": 0.4, "shift": 3.0}, open(os.path.join(d, "scheduler", "scheduler_config.json"), "w"))
        c = load_scheduler_config(d)
        assert c == {"base_image_seq_len": 256, "max_image_seq_len": 4096, "base_shift": 0.4, "max_shift": 1.3}
        assert abs(float(flow_match_sigmas(20, 4096, c)[1]) - math.exp(1.3) / (math.exp(1.3) + 1 / 0.95 - 1)) < 1e-6
        assert FluxPipeline.from_pretrained(d).scheduler_config == c
    assert load_scheduler_config("black-forest-labs/FLUX.1-dev")["max_shift"] == 1.15
    t, g = model_scalars(sig, 5.0, torch.bfloat16)
    to, go = flux_ref.model_scalars(sig, 5.0, torch.bfloat16)
    assert np.array_equal(t, to.numpy()) and g == float(go[0]) == 4992.0
    cfg = flux_ref.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=64)
    assert state_dict_shapes(cfg) == flux_ref.weight_shapes(cfg)
    a, b = synthetic_state_dict(cfg, 5), flux_ref.make_weights(cfg, 5)
    assert all(torch.equal(a[k], b[k]) for k in b)
    full = flux_ref.FluxConfig()
    n = sum(int(np.prod(s)) for s in state_dict_shapes(full).values())
    assert 11.8e9 < n < 12.0e9  # FLUX.1-dev: 11.9 B parameters
    with pytest.raises(ValueError):
        get_image_generator_type("FluxImageGenerator")()  # CUDA is required (same hard check as the reference)


def test_flow_match_schedule_and_euler_are_exact_on_straight_paths():
    """Properties that do not depend on the source the restatement was written from: the shifted sigma grid starts at 1, falls strictly to
    0 and equals exp(mu) / (exp(mu) + 1/s - 1) of the uniform grid; the Euler update x += (sigma' - sigma) * v integrates a rectified-flow
    (straight) path x_sigma = (1 - sigma) x0 + sigma * eps exactly, for any grid, so a point-mass target is reached to rounding."""
    import math

    from ecad_amd.pipelines.flow_match import calculate_shift

    for n_tok in (256, 1024, 4096):
        sig = flux_ref.flow_match_sigmas(20, n_tok).astype(np.float64)
        assert sig[0] == 1.0 and sig[-1] == 0.0 and np.all(np.diff(sig) < 0)
        mu = calculate_shift(n_tok)
        uniform = np.linspace(1.0, 1.0 / 20, 20)
        assert np.allclose(sig[:-1], math.exp(mu) / (math.exp(mu) + (1.0 / uniform - 1.0)), rtol=1e-6)
        rng = np.random.default_rng(n_tok)
        x0, eps = rng.standard_normal(32), rng.standard_normal(32)
        x = eps.copy()  # sigma = 1: pure noise
        for i in range(20):
            v = (x - x0) / sig[i]  # the exact velocity eps - x0 along the path through (x0, current x)
            x = x + (sig[i + 1] - sig[i]) * v
        assert np.abs(x - x0).max() < 1e-12
    assert abs(calculate_shift(256) - 0.5) < 1e-12 and abs(calculate_shift(4096) - 1.15) < 1e-12  # the two anchors of the linear shift law
