"""-m gpu: the NSGA-II harness with the real in-process evaluator (one resident engine, whole-loop graph replay per candidate) -- row f-4
and the config-4 workload of SURVEY 8(d), at a population small enough for a test."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"


def _embedding_dir(root, n=2):
    d = root / "emb"
    d.mkdir()
    g = torch.Generator().manual_seed(0)
    for i in range(n):
        mask = torch.zeros(1, 120, dtype=torch.int64)
        mask[:, : 20 + 30 * i] = 1
        neg_mask = torch.zeros(1, 120, dtype=torch.int64)
        neg_mask[:, :2] = 1
        torch.save({"prompt_embeds": torch.randn(1, 120, 4096, generator=g) * 0.1, "prompt_attention_mask": mask,
                    "negative_prompt_embeds": torch.randn(1, 120, 4096, generator=g) * 0.1, "negative_prompt_attention_mask": neg_mask}, d / f"{i:03}__prompt_id:{i}.pt")
    return d


def test_single_gpu_trainer_real_evaluator(tmp_path):
    from ecad_amd.benchmark.compute_macs import total_macs_T
    from ecad_amd.genetic import train_nsga2_single_gpu as T
    from ecad_amd.genetic.population_io_manager import PixArtPopulationIOManager

    seed = np.unpackbits(np.load(GOLD / "ga_seed_population.npz")["bits"], axis=1)[:5, :1680].astype(bool)
    seed = np.vstack([np.ones((1, 1680), dtype=bool), seed])  # candidate 0: the uncached schedule itself
    PixArtPopulationIOManager("run", tmp_path / "p", tmp_path / "b", generation_num=0, population_size=6).save_population(seed, 0)
    args = T.parse_args(["--image-generator", "PixArtAlphaImageGenerator", "--name", "run", "--all-populations-dir", str(tmp_path / "p"), "--all-benchmarks-dir",
                         str(tmp_path / "b"), "--population-size", "6", "--embedding-dir", str(_embedding_dir(tmp_path)), "--batch-size", "2", "--num-images-per-prompt", "2",
                         "--min-diff-from-default", "0", "--print-eval-outputs"])
    mgr, algo = T.run(args)
    assert mgr.generation_num == algo.n_gen == 2 and len(list((tmp_path / "p/run/gen_002/candidates").glob("cand_*.json"))) == 6
    scores = [json.loads((tmp_path / f"b/run/gen_001/candidates/cand_{i:03d}/scores.json").read_text())["total_score"] for i in range(6)]
    assert scores[0] == 0.0  # same schedule, same seeds, same graph: bit-identical latents
    assert all(-1.0 < s < 0.0 for s in scores[1:])  # cached candidates drift from the uncached run, by less than its norm
    for i in range(6):
        m = json.loads((tmp_path / f"p/run/gen_001/candidates/cand_{i:03d}.json").read_text())["metrics"]["total_macs_T"]
        assert m == total_macs_T(seed[i].reshape(20, 28, 3))
    assert abs(algo.F[:, 1].max() - 5.71379613696) < 1e-9  # the uncached candidate survives with the reference's all-true TMACs


def test_single_gpu_trainer_takes_the_flux_image_generator(tmp_path):
    """`train_nsga2_single_gpu.py --image-generator FluxImageGenerator` (reference train_nsga2_single_gpu.py:73,115): the FluxPopulationIOManager contract with the
    in-process evaluator on an EcadFluxEngine -- a reduced geometry (1 joint + 2 single blocks, 128 x 128) resumed from a manager_config.json, one generation:
    the uncached candidate scores exactly 0, cached ones drift; scores.json / metrics on disk; generation 2 written."""
    from ecad_amd.genetic import train_nsga2_single_gpu as T
    from ecad_amd.genetic.population_io_manager import FluxPopulationIOManager

    nb, ns, steps, pop = 1, 2, 20, 4
    mgr = FluxPopulationIOManager("fx", tmp_path / "p", tmp_path / "b", generation_num=0, population_size=pop, num_blocks=nb, num_single_blocks=ns, height=128, width=128,
                                  min_diff_from_default=0)
    rng = np.random.default_rng(3)
    seed = rng.random((pop, steps * (nb + ns) * 3)) < 0.6
    seed[0] = True  # candidate 0: the uncached schedule
    mgr.save_population(seed, 0)
    mgr.to_json()
    args = T.parse_args(["--image-generator", "FluxImageGenerator", "--name", "fx", "--all-populations-dir", str(tmp_path / "p"), "--all-benchmarks-dir", str(tmp_path / "b"),
                         "--load-from", str(tmp_path / "p/fx/gen_000/manager_config.json"), "--batch-size", "1", "--num-images-per-prompt", "1", "--print-eval-outputs"])
    out_mgr, algo = T.run(args)
    assert isinstance(out_mgr, FluxPopulationIOManager) and out_mgr.generation_num == algo.n_gen == 2
    assert len(list((tmp_path / "p/fx/gen_002/candidates").glob("cand_*.json"))) == pop
    scores = [json.loads((tmp_path / f"b/fx/gen_001/candidates/cand_{i:03d}/scores.json").read_text())["total_score"] for i in range(pop)]
    assert scores[0] == 0.0 and all(-1.0 < s < 0.0 for s in scores[1:])
    for i in range(pop):
        assert "total_macs_T" in json.loads((tmp_path / f"p/fx/gen_001/candidates/cand_{i:03d}.json").read_text())["metrics"]


def test_multi_gpu_trainer_one_rank(tmp_path, monkeypatch, capsys):
    """train_nsga2_multi_gpu at world size 1 (the same code path every rank runs; the collectives degenerate to identities)."""
    from ecad_amd.genetic import train_nsga2_multi_gpu as M

    monkeypatch.setattr(sys, "argv", ["train_nsga2_multi_gpu", "--name", "mg", "--populations-dir", str(tmp_path / "p"), "--benchmarks-dir", str(tmp_path / "b"),
                                      "--seed-population", str(GOLD / "ga_seed_population.npz"), "--population-size", "4", "--generations", "2", "--images", "2"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    M.main()
    out = capsys.readouterr().out
    assert "generation 2:" in out and "evaluated on 1 rank(s)" in out
    assert len(list((tmp_path / "b/mg/gen_002/candidates").glob("cand_*/scores.json"))) == 4


def test_config4_population_72_one_generation_and_resume(tmp_path, monkeypatch, capsys):
    """BASELINE config 4 at its real population: the 72 seed candidates of schedules/population_initialization (tests/golden/ga_seed_population.npz), full
    depth (28 blocks, 20 steps), evaluated by the sharded trainer -- on the one GPU a test has, i.e. shard size 72 instead of 9; the per-rank code is the
    same.  The uncached reference run and the 72 candidates alternate on one engine: the plan-set cache must serve the reference's plans and graph
    every time (1 + 72 lowerings, not 2 x 72).  Then one more generation resumed from the files."""
    import time

    from ecad_amd.genetic import train_nsga2_multi_gpu as M
    from ecad_amd.genetic.population_io_manager import PixArtPopulationIOManager

    base = ["train_nsga2_multi_gpu", "--name", "c4", "--populations-dir", str(tmp_path / "p"), "--benchmarks-dir", str(tmp_path / "b"), "--population-size", "72",
            "--images", "2"]
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", base + ["--seed-population", str(GOLD / "ga_seed_population.npz"), "--generations", "1"])
    t0 = time.time()
    M.main()
    dt = time.time() - t0
    out = capsys.readouterr().out
    assert "generation 1: 72 candidates" in out
    print(f"population 72, one generation on one rank: {dt:.1f} s ({72 / dt:.1f} candidates/s incl. engine set-up)")
    mgr = PixArtPopulationIOManager.from_json(tmp_path / "p/c4/gen_002/manager_config.json")
    F = mgr.load_evaluation_scores(1)
    assert F.shape == (72, 2) and np.isfinite(F).all() and (F[:, 0] >= 0).all() and F[:, 1].min() < F[:, 1].max() <= 5.71379613696 + 1e-9
    assert (tmp_path / "p/c4/gen_002/checkpoint.pkl").exists() and len(list((tmp_path / "p/c4/gen_002/candidates").glob("cand_*.json"))) == 72
    monkeypatch.setattr(sys, "argv", base + ["--load-from", str(tmp_path / "p/c4/gen_002/manager_config.json"), "--generations", "1"])
    M.main()
    assert "generation 2: 72 candidates" in capsys.readouterr().out and (tmp_path / "p/c4/gen_003/checkpoint.pkl").exists()


def test_candidate_scoring_with_its_prompt_batches_in_flight_gives_the_same_scores():
    """engine_score_fn_batches (round 6; BASELINE config 4: 100 prompts per candidate = several batches): a candidate's prompt batches are kept in flight on execution
    contexts over the rank's weights; every batch's latents are bit-identical to running it alone, so the candidate's score is the same number with one, two or
    three batches in flight."""
    import numpy as np
    import torch

    from ecad_amd.engine import EcadEngine
    from ecad_amd.genetic.evaluate import engine_score_fn_batches
    from oracle.dpm_solver_ref import DPMSolverPP2M
    from oracle.pipeline_ref import synthetic_inputs
    from oracle.pixart_ref import PixArtConfig, make_weights

    layers, steps = 2, 4
    eng = EcadEngine(num_layers=layers, max_batch=8, max_tokens=256, max_text_tokens=120, dtype=torch.float16)
    eng.load_state_dict(make_weights(PixArtConfig(num_layers=layers), seed=0))
    solver = DPMSolverPP2M(steps)
    coefs = np.array([solver.coefficients(i, solver.order_for_step(i)) for i in range(steps)], dtype=np.float32)
    batches = []
    for k, B in enumerate((4, 4, 2, 4, 1)):  # the last batch of a prompt set is the short one
        inp = synthetic_inputs(B, start_seed=50 + k)
        batches.append((torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0).cuda(), torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0).cuda(),
                        inp["latents"].cuda()))
    rng = np.random.default_rng(3)
    cands = [rng.random((steps, layers, 3)) < p for p in (0.3, 0.6, 0.9)]
    got = {}
    for n in (1, 2, 3):
        score = engine_score_fn_batches(eng, batches, solver.timesteps, coefs, 4.5, contexts=n)
        got[n] = [score(i, c) for i, c in enumerate(cands)]
        score.restore()
        score.close()
    assert got[1] == got[2] == got[3]
    assert got[1][0] < got[1][1] < got[1][2] <= 0.0  # more recomputation = closer to the uncached run
    eng.close()
