"""CPU suite: host-side mirror of the reference surface + C ABI export check (no compute without a GPU)."""
import ctypes
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"
FAST = ROOT / "schedules/schedules_in_paper/pixart_alpha_256/ours_fast.json"


def test_library_loads_and_exports_every_declared_symbol():
    from ecad_amd import _lib

    lib = _lib.load()
    header = (ROOT / "include" / "ecad_hip.h").read_text() + (ROOT / "include" / "ecad_flux.h").read_text()
    declared = set(re.findall(r"\b(ecad_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ecad_abi_version() == _lib.ABI_VERSION


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ecad_amd import _lib

    lib = _lib.load()
    h = ctypes.c_void_p()
    cfg = _lib.EcadModelCfg(28, 16, 72, 4, 8, 2, 32, 4096, 32, 256, 120, 0, 1e-6, 0.0)
    assert lib.ecad_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == _lib.ECAD_E_HIP
    assert b"no HIP device" in lib.ecad_last_error(None)
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type

    with pytest.raises(ValueError, match="CUDA is required"):
        get_image_generator_type("PixArtAlphaImageGenerator")()


def test_product_never_imports_oracle():
    for py in (ROOT / "ecad_amd").rglob("*.py"):
        txt = py.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, py


def test_cache_schedule_matches_reference_vectors():
    from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule

    vec = json.load(open(GOLD / "schedule_vectors.json"))
    for key, v in vec["to_numpy"].items():
        s = PixArtCacheSchedule.from_json(ROOT / "schedules" / v["path"])
        bits = s.to_numpy(flatten=True)
        assert np.packbits(bits).tobytes().hex() == v["packed_hex"], key
        assert s.name == v["name"] and s.attributes == v["attributes"]
    s = PixArtCacheSchedule.from_json(FAST)
    d = s.to_dict()
    shape = vec["to_dict_shape"]
    assert list(d.keys()) == shape["top_keys"]
    assert list(d["cache_schedule"].keys()) == shape["cache_schedule_keys"]
    assert list(d["cache_schedule"]["schedule"].keys()) == shape["step_keys"]
    assert list(d["cache_schedule"]["schedule"]["000"].keys()) == shape["block_keys_step0"]
    assert d["cache_schedule"]["schedule"]["000"]["17"] == shape["entry_000_17"]
    # step counter + lookups, as traced on the reference object
    s.reset_step()
    for step, (curr, ff3, attn0) in enumerate(vec["step_trace"]["trace"]):
        assert s.curr_step == curr
        assert s.get_recompute("3", "ff") == ff3 and s.get_recompute("0", "attn1") == attn0
        s.per_step_callback(step, 0)
    assert s.curr_step == vec["step_trace"]["curr_step_after_19"]
    s.reset_step()
    assert s.curr_step == 0
    with pytest.raises(ValueError):
        s.get_recompute("0", "bogus")
    assert s.get_custom_compute_attn("0") == vec["custom_compute_default"]["attn"] == {}
    assert s.uses_only_native_compute()


def test_bit_vector_round_trip_matches_population_manager():
    from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule

    pop = json.load(open(GOLD / "schedule_vectors.json"))["population"]
    x = np.unpackbits(np.frombuffer(bytes.fromhex(pop["x_packed_hex"]), dtype=np.uint8))[:1680].astype(bool)
    s = PixArtCacheSchedule.from_numpy(x, "rt")
    assert (s.to_numpy(flatten=True) == x).all() and pop["roundtrip_equal"]
    assert s.schedule[5]["7"] == pop["entry_step5_block7"]


def test_json_round_trip(tmp_path):
    from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule

    s = PixArtCacheSchedule.from_json(FAST)
    s.to_json(tmp_path / "x.json")
    assert json.load(open(tmp_path / "x.json")) == json.load(open(FAST))


def test_registries_and_defaults():
    from ecad_amd.pipelines.load_pipeline import PipelineRegistry
    from ecad_amd.schedulers.cache_scheduler.generators.pixart_schedule_generators import gen_default
    from ecad_amd.transformer_blocks.custom_attn_ff import ComputeAttnRegistry, ComputeFFRegistry

    assert ComputeAttnRegistry.get(None).__name__ == "compute_attn_cached"
    assert ComputeFFRegistry.get("COMPUTE_FF_CACHED").__name__ == "compute_ff_cached"
    assert ComputeAttnRegistry.get("nope", True) is None
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # the defaults are real callables on an engine block (tests/test_gpu_plugins.py); anything else: loud
        ComputeAttnRegistry.get(None)(object(), "attn1", None, None, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ComputeFFRegistry.get(None)(object(), None)

    @ComputeFFRegistry.register
    def My_Plugin_FF(block, norm_hidden_states, **kw):
        return norm_hidden_states

    assert ComputeFFRegistry.get("my_plugin_ff") is My_Plugin_FF and not ComputeFFRegistry.is_native("my_plugin_ff") and ComputeFFRegistry.is_native(None)
    from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule

    sched = PixArtCacheSchedule.from_numpy(np.ones((2, 3, 3), bool), "m", num_blocks=3, num_inference_steps=2)
    assert not sched.custom_component_mask().any() and sched.uses_only_native_compute()
    sched.schedule[1]["2"]["custom_compute_ff"] = {"name": "my_plugin_ff", "kwargs": {"a": 1}}
    sched.schedule[0]["0"]["custom_compute_attn"] = {"name": "COMPUTE_ATTN_CACHED"}  # the default by name is still native
    assert sched.custom_component_mask().tolist() == [[0, 0, 0], [0, 0, 0], [0, 0, 1]] and not sched.uses_only_native_compute()
    sched.schedule[0]["1"]["custom_compute_attn"] = {"name": "other"}
    assert sched.custom_component_mask().tolist() == [[0, 0, 0], [1, 1, 0], [0, 0, 1]]
    assert PipelineRegistry.get("pixart_alpha").__name__ == "PixArtAlphaPipeline"
    with pytest.raises(NotImplementedError):
        PipelineRegistry.get("tgate")
    d = next(gen_default(28, 20))
    ref = json.load(open(GOLD / "schedule_vectors.json"))["default_schedule"]
    assert d.name == ref["name"] and d.to_numpy().all() and ref["all_true"]


def test_solver_matches_oracle_coefficients():
    from ecad_amd.pipelines.dpm_solver import DPMSolverMultistepScheduler
    from oracle.dpm_solver_ref import DPMSolverPP2M

    a = DPMSolverMultistepScheduler()
    a.set_timesteps(20)
    b = DPMSolverPP2M(20)
    assert a.timesteps.tolist() == b.timesteps.tolist()
    tab = a.coefficient_table()
    exp = np.array([b.coefficients(i, b.order_for_step(i)) for i in range(20)], dtype=np.float32)
    assert np.allclose(tab, exp, rtol=1e-6, atol=0)
    # stateful torch path == oracle path on random data
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g)
    xa, xb = x.clone(), x.clone()
    for i in range(20):
        eps = torch.randn(2, 4, 8, 8, generator=g)
        xa = a.step(eps, a.timesteps[i], xa)[0]
        xb = b.step(eps, xb)
    assert torch.allclose(xa, xb, rtol=1e-5, atol=1e-6)


def test_synthetic_weights_identical_to_oracle_recipe():
    from types import SimpleNamespace

    from ecad_amd.transformer_2d_models.pixart_transformer_2d_edited import synthetic_state_dict
    from oracle.pixart_ref import PixArtConfig, make_weights

    cfg = PixArtConfig(num_layers=1)
    ns = SimpleNamespace(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2, num_layers=1, caption_channels=4096)
    a, b = synthetic_state_dict(ns, seed=3), make_weights(cfg, seed=3)
    assert list(a) == list(b)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_ecad_import_alias_resolves_to_native_host_side():
    import ecad.schedulers.cache_scheduler.pixart_cache_schedule as a
    import ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule as b
    from ecad.image_generators.load_image_generator import get_image_generator_type
    from ecad.transformer_blocks.custom_attn_ff import ComputeAttnRegistry

    assert a.PixArtCacheSchedule is b.PixArtCacheSchedule
    assert get_image_generator_type("PixArtAlphaImageGenerator").__module__.startswith("ecad_amd.")
    assert ComputeAttnRegistry.DEFAULT == "compute_attn_cached"


def test_vae_decoder_shapes_and_diffusers_key_names():
    """The product's weight table (names + shapes the HIP decoder loads by) against the checker module's state_dict(), both geometries."""
    from ecad_amd.pipelines.vae_decoder import synthetic_vae_state_dict, vae_state_dict_shapes
    from oracle.vae_ref import AutoencoderKLDecoderOnly

    vae = AutoencoderKLDecoderOnly().eval()
    keys = set(vae.state_dict())
    for k in ("post_quant_conv.weight", "decoder.conv_in.weight", "decoder.mid_block.attentions.0.to_q.weight", "decoder.mid_block.resnets.1.conv2.bias",
              "decoder.up_blocks.0.upsamplers.0.conv.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.weight", "decoder.up_blocks.3.resnets.2.norm2.weight",
              "decoder.conv_norm_out.weight", "decoder.conv_out.bias"):
        assert k in keys, k
    assert vae.state_dict()["decoder.up_blocks.2.resnets.0.conv1.weight"].shape == (256, 512, 3, 3)
    n = sum(p.numel() for p in vae.parameters())
    assert 49e6 < n < 50e6  # SD VAE decoder + post_quant_conv ~49.5 M parameters
    img = vae.decode(torch.randn(1, 4, 8, 8))[0]
    assert img.shape == (1, 3, 64, 64) and torch.isfinite(img).all()
    for module, kw in ((vae, {}), (AutoencoderKLDecoderOnly.flux(), dict(latent_channels=16, use_post_quant_conv=False))):
        assert {k: tuple(v.shape) for k, v in module.state_dict().items()} == vae_state_dict_shapes(**kw)
    sd = synthetic_vae_state_dict(0)
    vae.load_state_dict(sd)  # strict: same keys, same shapes
    assert float(sd["decoder.conv_in.weight"].abs().max()) <= 1.0 / 6.0 and float(sd["decoder.conv_norm_out.weight"].min()) == 1.0


def test_compute_macs_cli_geometry_follows_the_weights(tmp_path, capsys):
    """compute_macs.py picks the geometry from the schedule's transformer_weights as the reference's WEIGHTS_TO_SHAPES does (:22-39,66-70): the
    1024-MS alpha file gets N = 4096 and the size embedders, i.e. exactly the reference's measured numbers; unknown weights are an error."""
    import shutil

    from ecad_amd.benchmark import compute_macs as CM

    vec = json.loads((ROOT / "tests/golden/alpha1024_vectors.json").read_text())["alpha_cache_schedules/gen_default_1024x1024/default_1024x1024.json"]
    f = tmp_path / "s.json"
    shutil.copy(ROOT / "schedules/alpha_cache_schedules/gen_default_1024x1024/default_1024x1024.json", f)
    assert CM.annotate_schedule_file(f) is False  # carries metrics already
    assert CM.annotate_schedule_file(f, recompute_existing=True, print_detailed=True) is True
    m = json.loads(f.read_text())["metrics"]
    assert [m["by_inference_step"][f"{s:03}"]["macs"] for s in range(20)] == vec["macs"] and m["total_macs"] == vec["total_macs"]
    assert "step 019:" in capsys.readouterr().out
    d = json.loads(f.read_text())
    d["config"]["transformer_weights"] = "someone/else"
    f.write_text(json.dumps(d))
    with pytest.raises(ValueError, match="Shapes for weights someone/else not found"):
        CM.annotate_schedule_file(f, recompute_existing=True)
    flags = {a.dest for a in CM.get_argparser()._actions}
    assert {"image_generator", "input_dir", "recompute_existing", "print_detailed"} <= flags


def test_inference_cli_guidance_scale_follows_the_reference(tmp_path):
    """reference inference.py:208-214: ``--guidance-scale`` is forwarded to FLUX and refused for the PixArt generators (CFG 4.5 is fixed there) -- it is
    never silently dropped.  The refusal happens before any engine is created, so it is checkable without a GPU."""
    from ecad_amd.image_generators.load_image_generator import get_image_generator_type
    from ecad_amd.inference import inference

    with pytest.raises(ValueError, match="does not support guidance scale"):
        inference.main(["PixArtAlphaImageGenerator", "-i", str(tmp_path), "--guidance-scale", "3.0"])
    inference._check_guidance(get_image_generator_type("FluxImageGenerator"), 3.0)  # FLUX: accepted
    inference._check_guidance(get_image_generator_type("PixArtSigmaImageGenerator"), None)  # flag absent: fine


def test_generated_instruction_streams_are_what_their_generators_write():
    """`ecad_amd/csrc/gemm4_asm.h`, `attention_w4_asm.h` and `attention_w472_asm.h` are generated (scripts/gen_gemm4.py, gen_attn_w4.py, gen_attn_w4_72.py) and committed: a hand edit of either header, or a
    generator change without regenerating, would leave the tree describing one instruction stream and building another.  `--check` regenerates in memory and compares."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    for gen in ("gen_gemm4.py", "gen_attn_w4.py", "gen_attn_w4_72.py"):
        r = subprocess.run([sys.executable, str(root / "scripts" / gen), "--check"], capture_output=True, text=True)
        assert r.returncode == 0, f"{gen}: {r.stderr[-400:]}"


def test_generate_images_skip_rule_is_the_references(tmp_path):
    """reference generate_images.py:27-43: a schedule's output directory is skipped only when it already holds at least one ``*.png`` directly inside it; an empty
    directory (what a crashed run leaves) is generated again; with ``--regen-if-not-n-images`` a wrong count removes the PNG files (only those, only at the top
    level) and regenerates; a schedule directory that holds only directories contributes its own name to the output path (:94-111)."""
    from PIL import Image

    from ecad_amd.benchmark import generate_images as gi

    made = []

    class FakeGenerator:
        def __init__(self, start_seed, seed_step, schedule_path, max_batch):
            self.schedule_path = schedule_path

        def generate_from_saved_prompts(self, input_dir, out, batch_size, images_per_prompt, free_after, include_seed_in_name):
            made.append(Path(out))
            for k in range(2):
                Image.new("RGB", (2, 2)).save(Path(out) / f"img_{k}.png")

    sched = tmp_path / "schedules"
    (sched / "fam_a").mkdir(parents=True)
    (sched / "fam_b").mkdir()
    for f in ("fam_a/s1.json", "fam_a/s2.json", "fam_b/s3.json"):
        (sched / f).write_text("{}")
    out = tmp_path / "out"
    args = (FakeGenerator, sched, tmp_path, out, 4, 1, 0, 1)
    assert gi.generate_all_schedules(*args) == 3
    assert sorted(p.relative_to(out).as_posix() for p in made) == ["schedules/s1", "schedules/s2", "schedules/s3"]  # a directory of directories adds ITS name; the leaf directories' files go to <stem>
    made.clear()
    assert gi.generate_all_schedules(*args) == 0 and not made  # every directory holds images: skipped
    # a crashed run: the directory exists, no image in it -> generated again (the round-5 build skipped it)
    s1 = out / "schedules" / "s1"
    for p in s1.glob("*.png"):
        p.unlink()
    (s1 / "nested").mkdir()
    Image.new("RGB", (2, 2)).save(s1 / "nested" / "deep.png")  # not counted: the reference's glob is not recursive
    assert gi.generate_all_schedules(*args) == 1 and made == [s1]
    assert (s1 / "nested" / "deep.png").exists()
    made.clear()
    # wrong count with --regen-if-not-n-images: PNG files removed (the directory and everything else in it stay), regenerated
    (s1 / "notes.txt").write_text("kept")
    Image.new("RGB", (2, 2)).save(s1 / "extra.png")
    assert gi.generate_all_schedules(*args, 2) == 1 and made == [s1]
    assert (s1 / "notes.txt").exists() and not (s1 / "extra.png").exists() and len(list(s1.glob("*.png"))) == 2
    made.clear()
    assert gi.generate_all_schedules(*args, 2) == 0  # right count now
    # a single schedule file goes to output_dir / stem; a failing constructor surfaces as the reference's ValueError
    assert gi.generate_all_schedules(FakeGenerator, sched / "fam_b" / "s3.json", tmp_path, tmp_path / "single", 4, 1, 0, 1) == 1
    assert made[-1] == tmp_path / "single" / "s3"

    class Broken:
        def __init__(self, **kw):
            raise RuntimeError("no")

    with pytest.raises(ValueError, match="Error creating image generator"):
        gi.generate_for_schedule(Broken, sched / "fam_b" / "s3.json", tmp_path, tmp_path / "other", 4, 1, 0, 1)


def test_product_build_is_gated_on_the_accumulator_audit():
    """Round-5 ADVICE: the kernels whose K loops are asm statements hand their accumulators over in AGPRs that hipcc treats as free registers behind the statement; a
    spill into them is silent.  The Makefile keeps the device assembly each object was made from and fails on a hazard (scripts/audit_agpr.py --asm); here: the rule is
    in the Makefile, the audit flags a planted hazard, and -- when a build has left its assembly in csrc/build -- that assembly is clean."""
    import subprocess
    import sys

    root = Path(__file__).resolve().parents[1]
    mk = (root / "ecad_amd" / "csrc" / "Makefile").read_text()
    assert "-save-temps" in mk and "audit_agpr.py --asm" in mk and "exit 1" in mk
    sys.path.insert(0, str(root / "scripts"))
    try:
        import audit_agpr
    finally:
        sys.path.pop(0)
    flags = (root / "ecad_amd" / "csrc" / "Makefile").read_text().split("BASEFLAGS := ")[1].split("\n")[0].split()
    assert "-fPIC" in audit_agpr.FLAGS and all(f in audit_agpr.FLAGS for f in flags if not f.startswith("-W") and f != "--offload-arch=$(ARCH)")
    body = ["\t;;#ASMSTART", "\tv_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], a[0:3]", "\t;;#ASMEND", "{planted}", "\t;;#ASMSTART", "\tv_accvgpr_read_b32 v9, a0", "\t;;#ASMEND"]
    kern = "_ZN4ecad14gemm2w4_kernelIDF16_Li1ELi4EEEvNS_8GemmArgsIT_EE:\n{}\n.Lfunc_end0:\n"
    clean = kern.format("\n".join(body).format(planted="\tv_add_f32_e32 v1, v2, v3"))
    spilled = kern.format("\n".join(body).format(planted="\tv_accvgpr_write_b32 a0, v5"))
    assert audit_agpr.audit(clean, "planted-clean") == 0 and audit_agpr.audit(spilled, "planted-hazard") == 1
    kept = sorted((root / "ecad_amd" / "csrc" / "build").glob("*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if kept:
        r = subprocess.run([sys.executable, str(root / "scripts" / "audit_agpr.py"), "--asm", *map(str, kept)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-600:]
