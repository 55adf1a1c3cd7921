"""CPU suite: NSGA-II bookkeeping + the multi-rank path (gloo, world_size 2)."""
import json
import os
import socket
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"


def test_population_manager_matches_reference_vectors(tmp_path):
    from ecad_amd.genetic.population_io_manager import PixArtPopulationIOManager
    from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule

    ref = json.load(open(GOLD / "schedule_vectors.json"))["population"]
    x = np.unpackbits(np.frombuffer(bytes.fromhex(ref["x_packed_hex"]), dtype=np.uint8))[:1680].astype(bool)
    fast = PixArtCacheSchedule.from_json(ROOT / "schedules/schedules_in_paper/pixart_alpha_256/ours_fast.json")
    mgr = PixArtPopulationIOManager("kat", tmp_path / "p", tmp_path / "b", generation_num=1)
    pop = np.stack([x, np.ones(1680, dtype=bool), fast.to_numpy(flatten=True)])
    assert mgr.get_constraint_violations(pop).reshape(-1).tolist() == ref["G"]
    assert mgr.compute_additional_info(fast.schedule) == ref["additional_info_ours_fast"]
    sd = mgr.binary_vector_to_schedule_dict(x, 20, 28, 3)
    assert mgr.compute_additional_info(sd) == ref["additional_info_x"]
    # on-disk round trip in the reference's layout
    mgr.save_population(pop, 1)
    assert (tmp_path / "p/kat/gen_001/candidates/cand_002.json").exists()
    assert np.array_equal(mgr.load_population_vectors(1), pop)
    assert not mgr.check_offline_eval()
    for i in range(3):
        mgr.write_evaluation(i, score=-0.1 * i, bits=pop[i], generation=1)
    assert mgr.check_offline_eval()
    X, F, G = mgr.ask()
    assert F.shape == (3, 2) and np.allclose(F[:, 0], [1.0, 1.1, 1.2])
    assert abs(F[2, 1] - 2.134989471744) < 1e-12  # TMACs of ours_fast, the reference's own measurement
    assert abs(F[1, 1] - 5.71379613696) < 1e-12  # all-true schedule: 40 * (749223936 + 28 * 5074845696) MACs
    mgr.to_json()
    again = PixArtPopulationIOManager.from_json(tmp_path / "p/kat/gen_001/manager_config.json")
    assert again.to_dict() == mgr.to_dict()


def test_nsga2_sorting_and_loop():
    from ecad_amd.genetic.nsga2 import NSGA2, crowding_distance, fast_non_dominated_sort

    F = np.array([[1.0, 5.0], [2.0, 2.0], [5.0, 1.0], [3.0, 3.0], [6.0, 6.0]])
    G = np.array([0, 0, 0, 0, 0.0])
    fronts = fast_non_dominated_sort(F, G)
    assert sorted(fronts[0]) == [0, 1, 2] and fronts[1] == [3] and fronts[2] == [4]
    G2 = np.array([0, 0, 3.0, 0, 0])  # an infeasible point loses to every feasible one
    assert 2 not in fast_non_dominated_sort(F, G2)[0]
    cd = crowding_distance(F[[0, 1, 2]])
    assert np.isinf(cd[0]) and np.isinf(cd[2]) and np.isfinite(cd[1])
    algo = NSGA2(pop_size=12, n_var=60, seed=0)
    algo.setup()
    for _ in range(3):
        X = algo.ask()
        assert X.shape == (12, 60) and len({x.tobytes() for x in X}) == 12
        Fx = np.stack([X.sum(1), (~X).sum(1) * 0.5 + X[:, 0]], axis=1).astype(float)
        algo.tell(X, Fx, np.zeros(12))
    assert algo.n_gen == 4
    b = NSGA2(pop_size=12, n_var=60, seed=0)
    b.setup()
    c = NSGA2(pop_size=12, n_var=60, seed=0)
    c.setup()
    assert np.array_equal(b.ask(), c.ask())  # deterministic given the seed


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from ecad_amd import distributed as D
    from ecad_amd.genetic.evaluate import evaluate_population

    r, w, _ = D.init("gloo")
    assert (r, w) == (rank, world)
    kat = np.load(GOLD / "ga_seed_population.npz")
    X0 = np.unpackbits(kat["bits"], axis=1)[:, :1680].astype(bool) if rank == 0 else None
    X = D.broadcast_population(X0, (72, 1680))
    seen = []

    def score(c, bits):  # deterministic fake scorer: no GPU in this test
        seen.append(c)
        return float(bits.mean()) - 1.0

    F = evaluate_population(X, score)
    t = D.max_over_ranks(1.0 + rank)
    D.barrier()
    np.savez(Path(out_dir) / f"r{rank}.npz", F=F, seen=np.array(seen), X=X, t=t)
    dist.destroy_process_group()


def test_two_rank_sharded_evaluation_gloo(tmp_path):
    from ecad_amd import distributed as D
    from ecad_amd.benchmark.compute_macs import total_macs_T

    assert list(D.shard_range(72, 8, 3)) == list(range(27, 36))  # 9 per GPU
    assert sum(len(D.shard_range(71, 8, r)) for r in range(8)) == 71  # dedup'd populations: ceil division
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert sorted(r0["seen"].tolist() + r1["seen"].tolist()) == list(range(72))  # every candidate exactly once
    assert r0["seen"].tolist() == list(range(36)) and r1["seen"].tolist() == list(range(36, 72))
    assert np.array_equal(r0["F"], r1["F"]) and np.array_equal(r0["X"], r1["X"])
    assert float(r0["t"]) == 2.0 == float(r1["t"])
    X = r0["X"]
    exp = np.array([[1.0 - (x.mean() - 1.0), total_macs_T(x.reshape(20, 28, 3))] for x in X])
    assert np.allclose(r0["F"], exp, rtol=0, atol=1e-12)


def test_product_mac_model_equals_oracle_and_kat():
    from ecad_amd.benchmark.compute_macs import executed_components, macs_per_step
    from oracle import schedule_ref as S

    kat = np.load(GOLD / "mac_kat.npz")
    for packed, macs in list(zip(kat["bits"], kat["macs"]))[::25]:
        bits = np.unpackbits(packed)[:1680].reshape(20, 28, 3).astype(bool)
        assert np.array_equal(macs_per_step(bits), macs)
        assert np.array_equal(executed_components(bits), S.resolve_executed(bits))


def test_flux_population_manager_contract(tmp_path):
    """FluxPopulationIOManager (reference: ecad/genetic/flux_population_io_manager.py): vector <-> schedule dict as captured from the
    reference, on-disk layout, closed-form MAC objective equal to the reference's measured totals."""
    from ecad_amd.genetic.population_io_manager import FluxPopulationIOManager
    from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule

    v = json.load(open(GOLD / "flux_vectors.json"))
    pop_ref = v["population"]
    n = 20 * 57 * 3
    x = np.unpackbits(np.frombuffer(bytes.fromhex(pop_ref["x_packed_hex"]), dtype=np.uint8))[:n].astype(bool)
    mgr = FluxPopulationIOManager("kat", tmp_path / "p", tmp_path / "b", generation_num=1, height=1024, width=1024)
    sd = mgr.binary_vector_to_schedule_dict(x, 20, 19, 38, 3, 3)
    assert sd[5]["7"] == pop_ref["entry_step5_block7"] and sd[5]["single_30"] == pop_ref["entry_step5_single_30"]
    fast = FluxCacheSchedule.from_json(ROOT / "schedules/schedules_in_paper/flux_256_to_1024/fast_256_to_1024.json")
    pop = np.stack([x, np.ones(n, dtype=bool), fast.to_numpy()])
    assert mgr.get_constraint_violations(pop).reshape(-1).tolist() == [1 - int((~x).sum()), 1, 1 - int((~fast.to_numpy()).sum())]
    mgr.save_population(pop, 1)
    saved = json.load(open(tmp_path / "p/kat/gen_001/candidates/cand_002.json"))
    assert saved["cache_schedule"]["num_single_blocks"] == 38 and saved["config"] == {"height": 1024, "width": 1024, "guidance_scale": 5}
    assert np.array_equal(mgr.load_population_vectors(1), pop)
    for i in range(3):
        mgr.write_evaluation(i, score=0.25, bits=pop[i], generation=1)
    assert mgr.check_offline_eval()
    X, F, G = mgr.ask()
    assert abs(F[2, 1] - v["to_numpy"]["flux_1024_fast"]["total_macs"] / 1e12) < 1e-9  # the reference's measured TMACs of that schedule
    assert abs(F[1, 1] - 1190.24510828544) < 1e-9  # all-true at 1024^2 (1190.2 TMACs)
    info = mgr.compute_additional_info(fast.schedule)
    assert set(info) == {"num_affected_steps", "num_affected_blocks", "num_affected_blocks_single", "total_num_affected_blocks"}
    mgr.to_json()
    again = FluxPopulationIOManager.from_json(tmp_path / "p/kat/gen_001/manager_config.json")
    assert again.to_dict() == mgr.to_dict()


# ---------------------------------------------------------------------------------------------
# the reference-named single-GPU entry point: command line, cycle, checkpoint / resume
# ---------------------------------------------------------------------------------------------
# flags of ecad/genetic/train_nsga2_base.py:43-163 (get_base_argparser) + train_nsga2_single_gpu.py:256-260, with their defaults
REFERENCE_FLAGS = {
    "image_generator": None, "load_from": None, "name": None, "num_cycles": "1", "print_not_submit": False, "embedding_dir": None, "generation_num": None,
    "num_inference_steps": 20, "benchmark_prompts": None, "file_mode": None, "image_naming_mode": None, "min_diff_from_default": 1, "population_size": 72,
    "batch_size": 100, "num_images_per_prompt": 10, "exactly_n_images": 1000, "regen_if_not_n_images": 1000, "maximize_MACs": False, "print_eval_outputs": False,
}


def _stub_score(_index, bits):
    """Deterministic stand-in for the GPU scorer: prefers recomputing early steps (so the two objectives trade off)."""
    w = np.linspace(1.0, 0.1, bits.shape[0])[:, None, None]
    return float((bits * w).sum() / w.sum() / bits[0].size) - 1.0


def test_single_gpu_trainer_command_line_is_the_references(tmp_path):
    from ecad_amd.genetic import train_nsga2_single_gpu as T

    args = T.parse_args(["--image-generator", "PixArtAlphaImageGenerator", "--name", "x"])
    got = vars(args)
    for flag, default in REFERENCE_FLAGS.items():
        if flag in ("image_generator", "name"):
            continue
        assert flag in got, flag
        assert got[flag] == (1 if flag == "num_cycles" else default), flag  # validate_base_args turns "1" into 1
    assert got["all_populations_dir"].parts[-3:] == ("results", "genetic", "populations")
    assert got["all_benchmarks_dir"].parts[-4:] == ("results", "benchmark", "genetic", "populations")
    assert T.get_offline_eval_kwargs(args).keys() == {"embedding_dir", "num_images_per_prompt", "exactly_n_images", "regen_if_not_n_images", "benchmark_prompts",
                                                      "file_mode", "image_naming_mode"}
    assert T.parse_args(["--name", "x", "--num-cycles", "inf"]).num_cycles == "inf"
    with pytest.raises(ValueError, match="name or path"):
        T.parse_args([])
    with pytest.raises(ValueError, match="integer or 'inf'"):
        T.parse_args(["--name", "x", "--num-cycles", "many"])
    with pytest.raises(ValueError, match="print-not-submit"):
        T.parse_args(["--name", "x", "--num-cycles", "3", "--print-not-submit"])
    with pytest.raises(SystemExit):  # argparse: not a registered generator
        T.parse_args(["--name", "x", "--image-generator", "NoSuchGenerator"])


def test_single_gpu_trainer_cycle_checkpoint_and_resume(tmp_path, capsys):
    """Three cycles in one go == one cycle, a fresh process state (manager and algorithm restored from gen_XXX/manager_config.json +
    checkpoint.pkl), two more cycles: same candidates, same files.  Evaluation is the injected stub (the GPU scorer is exercised by the
    multi-GPU trainer's tests); every other line is the shipped loop."""
    from ecad_amd.genetic import train_nsga2_single_gpu as T
    from ecad_amd.genetic.population_io_manager import PixArtPopulationIOManager

    seed = np.unpackbits(np.load(GOLD / "ga_seed_population.npz")["bits"], axis=1)[:8, :1680].astype(bool)

    def common(root):
        return ["--image-generator", "PixArtAlphaImageGenerator", "--all-populations-dir", str(root / "p"), "--all-benchmarks-dir", str(root / "b"), "--population-size", "8"]

    def seeded(root):
        PixArtPopulationIOManager("run", root / "p", root / "b", generation_num=0, population_size=8).save_population(seed, 0)

    # no seed population and nobody to ask: the reference's ValueError
    with pytest.raises(ValueError, match="No initial population found"):
        T.run(T.parse_args(common(tmp_path / "none") + ["--name", "run"]), score_fn=_stub_score, allow_random_seed_population=False)

    # --- uninterrupted
    a = tmp_path / "a"
    seeded(a)
    mgr_a, algo_a = T.run(T.parse_args(common(a) + ["--name", "run", "--num-cycles", "3"]), score_fn=_stub_score)
    assert mgr_a.generation_num == algo_a.n_gen == 4  # generations 1..3 evaluated and told, generation 4 written and waiting
    assert (a / "p/run/gen_004/checkpoint.pkl").exists() and (a / "p/run/gen_004/manager_config.json").exists()
    assert np.array_equal(mgr_a.load_population_vectors(1), seed)  # generation 1 is the seed population (train_nsga2_base.py:236-250)
    assert mgr_a.check_offline_eval() is False and len(list((a / "p/run/gen_004/candidates").glob("cand_*.json"))) == 8
    scores = json.loads((a / "b/run/gen_003/candidates/cand_000/scores.json").read_text())
    cand = json.loads((a / "p/run/gen_003/candidates/cand_000.json").read_text())
    assert set(scores) == {"total_score"} and "total_macs_T" in cand["metrics"] and cand["cache_schedule"]["num_blocks"] == 28

    # --- interrupted after one cycle, resumed from the files alone
    b = tmp_path / "b"
    seeded(b)
    T.run(T.parse_args(common(b) + ["--name", "run", "--num-cycles", "1"]), score_fn=_stub_score)
    capsys.readouterr()
    mgr_b, algo_b = T.run(T.parse_args(common(b) + ["--load-from", str(b / "p/run/gen_002/manager_config.json"), "--num-cycles", "2"]), score_fn=_stub_score)
    assert "algorithm checkpoint <-" in capsys.readouterr().out
    assert mgr_b.generation_num == 4
    for g in (2, 3, 4):
        assert np.array_equal(mgr_a.load_population_vectors(g), mgr_b.load_population_vectors(g)), g
    assert np.array_equal(algo_a.F, algo_b.F) and np.array_equal(algo_a.X, algo_b.X)
    # survivors are non-dominated-sorted on (1 - score, TMACs): the all-recompute-ish end and the cheap end both survive
    assert algo_a.F[:, 1].min() < algo_a.F[:, 1].max()

    # --- resumed on a generation that already has its scores: advance first, then evaluate the new generation (one of each per cycle)
    calls = []
    T.InProcessEvaluator("PixArtAlphaImageGenerator", score_fn=_stub_score)(mgr_b)  # evaluate generation 4 out of band
    mgr_c, _ = T.run(T.parse_args(common(b) + ["--load-from", str(b / "p/run/gen_004/manager_config.json"), "--num-cycles", "1"]),
                     score_fn=lambda i, bits: calls.append(i) or _stub_score(i, bits))
    assert calls == list(range(8)) and mgr_c.generation_num == 5 and mgr_c.check_offline_eval()
    assert not (b / "p/run/gen_006").exists()

    # --- --print-not-submit describes the evaluation of the generation it just wrote and leaves the benchmark directory alone
    T.run(T.parse_args(common(b) + ["--load-from", str(b / "p/run/gen_005/manager_config.json"), "--print-not-submit"]), score_fn=_stub_score)
    out = capsys.readouterr().out
    assert "generation 6: 8 candidate schedule(s)" in out and "stand-in for ImageReward" in out
    assert (b / "p/run/gen_006/candidates/cand_000.json").exists() and not (b / "b/run/gen_006").exists()


def test_bitflip_mutation_follows_pymoo_semantics():
    """BitflipMutation(prob=0.05) in pymoo 0.6.1.3 [ext] (reference train_nsga2_base.py:238): `prob` = chance that an offspring is mutated at all,
    per-bit rate prob_var = min(0.5, 1 / n_var).  Expected flips per child = 0.05 * 1680 / 1680 = 0.05 -- not 0.05 * 1680 = 84."""
    from ecad_amd.genetic.nsga2 import NSGA2

    algo = NSGA2(pop_size=72, n_var=1680, seed=3)
    assert algo.mutation_prob == 0.05 and algo.mutation_prob_var == 1.0 / 1680
    assert NSGA2(n_var=1).mutation_prob_var == 0.5  # pymoo caps the per-bit rate at 0.5
    n = 40000
    flips = np.array([int(algo._mutate(np.zeros(1680, dtype=bool)).sum()) for _ in range(n)])
    touched = (flips > 0).mean()
    # P(mutated and at least one bit flips) = 0.05 * (1 - (1 - 1/1680)^1680) = 0.0316; mean flips = 0.05
    assert abs(flips.mean() - 0.05) < 0.006 and abs(touched - 0.05 * (1 - (1 - 1 / 1680) ** 1680)) < 0.004
    # both rates are exposed, as in pymoo: prob = 1 mutates every child, prob_var sets the per-bit rate
    hot = NSGA2(n_var=1680, mutation_prob=1.0, mutation_prob_var=0.05, seed=1)
    m = np.mean([hot._mutate(np.zeros(1680, dtype=bool)).sum() for _ in range(200)])
    assert 80 < m < 88


def test_checkpoint_without_mutation_prob_var_still_resumes():
    """A checkpoint.pkl dill-pickled before `mutation_prob_var` existed (round-1 code) has no such attribute; resuming it must not lose a generation
    to an AttributeError in the first _offspring() after tell().  It gets pymoo's default and says so."""
    import dill

    from ecad_amd.genetic.nsga2 import NSGA2

    algo = NSGA2(pop_size=6, n_var=24, seed=5)
    algo.setup()
    X = algo.ask()
    algo.tell(X, np.stack([X.sum(1), (~X).sum(1)], 1).astype(float), np.zeros(len(X)))
    state = dict(algo.__dict__)
    del state["mutation_prob_var"]
    old = NSGA2.__new__(NSGA2)
    old.__dict__.update(state)  # what the old class's instances looked like
    blob = dill.dumps(old)
    with pytest.warns(RuntimeWarning, match="predates mutation_prob_var"):
        back = dill.loads(blob)
    assert back.mutation_prob_var == 1.0 / 24
    assert back.ask().shape[1] == 24  # _offspring() -> _mutate() runs
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        dill.loads(dill.dumps(algo))  # a current checkpoint loads silently


def _search_worker(rank: int, world: int, port: int, root: str, mode: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from ecad_amd import distributed as D
    from ecad_amd.genetic import train_nsga2_multi_gpu as M
    from ecad_amd.genetic.evaluate import evaluate_population
    from ecad_amd.genetic.population_io_manager import PixArtPopulationIOManager

    D.init("gloo")
    root = Path(root)
    n_var, pop = 1680, 8
    seed = np.unpackbits(np.load(GOLD / "ga_seed_population.npz")["bits"], axis=1)[:, :n_var].astype(bool)

    def mk(name):
        return lambda: PixArtPopulationIOManager(name, root / "p", root / "b", generation_num=1, population_size=pop)

    def ev(X):
        return evaluate_population(X, _stub_score)

    def search(name, gens, load_from=None, seed_bits=seed):
        return M.sharded_search(rank, world, lambda: M.open_search(mk(name), PixArtPopulationIOManager, n_var, pop, load_from, seed_bits), n_var, gens, ev)

    if mode == "resume":
        search("whole", 3)
        search("parts", 1)
        search("parts", 2, load_from=root / "p/parts/gen_002/manager_config.json")
    elif mode == "short_seed":  # 5 seed rows for a population of 8: generation 1 has 5 candidates and nobody hangs in the broadcast
        search("short", 2, seed_bits=seed[:5])
    elif mode == "rank0_fails":
        try:
            search("broken", 1, load_from=root / "p/none/gen_001/manager_config.json")
        except RuntimeError as e:
            (root / f"err{rank}.txt").write_text(str(e))
    elif mode == "rank1_eval_fails":  # an exception inside evaluate() on a non-zero rank must not leave rank 0 in the all-gather

        def bad_score(c, bits):
            if rank == 1:
                raise FloatingPointError("NaN score on rank 1")
            return _stub_score(c, bits)

        try:
            M.sharded_search(rank, world, lambda: M.open_search(mk("evfail"), PixArtPopulationIOManager, n_var, pop, None, seed), n_var, 1,
                             lambda X: evaluate_population(X, bad_score))
        except RuntimeError as e:
            (root / f"everr{rank}.txt").write_text(str(e))
    elif mode == "rank0_tell_fails":  # rank 0 failing AFTER the evaluation (files, tell, ask, checkpoint) must raise on every rank too
        import ecad_amd.genetic.train_nsga2_single_gpu as S

        if rank == 0:
            real, calls = S.checkpoint_manager_and_algorithm, {"n": 0}

            def flaky(m, a):
                calls["n"] += 1
                if calls["n"] >= 2:  # the first call checkpoints generation 1 in open_search
                    raise OSError("disk full")
                return real(m, a)

            S.checkpoint_manager_and_algorithm = flaky
        try:
            search("tellfail", 2)
        except RuntimeError as e:
            (root / f"tellerr{rank}.txt").write_text(str(e))
    dist.destroy_process_group()


def test_multi_gpu_trainer_resumes_and_never_hangs_gloo(tmp_path):
    """World 2 over gloo.  (1) Three generations in one go == one generation, then a fresh call resumed from gen_002/manager_config.json +
    checkpoint.pkl for two more: same candidates on disk.  (2) A seed population shorter than --population-size runs (the header carries the
    row count).  (3) Rank 0 failing to open the search raises on EVERY rank instead of leaving the others in a collective."""
    from ecad_amd.genetic.population_io_manager import PixArtPopulationIOManager

    for mode in ("resume", "short_seed", "rank0_fails", "rank1_eval_fails", "rank0_tell_fails"):
        mp.spawn(_search_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    a = PixArtPopulationIOManager.from_json(tmp_path / "p/whole/gen_004/manager_config.json")
    b = PixArtPopulationIOManager.from_json(tmp_path / "p/parts/gen_004/manager_config.json")
    assert a.generation_num == b.generation_num == 4
    for g in (1, 2, 3, 4):
        assert np.array_equal(a.load_population_vectors(g), b.load_population_vectors(g)), g
    assert (tmp_path / "p/parts/gen_004/checkpoint.pkl").exists() and (tmp_path / "p/whole/gen_002/checkpoint.pkl").exists()
    assert len(list((tmp_path / "p/short/gen_001/candidates").glob("cand_*.json"))) == 5
    assert len(list((tmp_path / "p/short/gen_003/candidates").glob("cand_*.json"))) == 8
    assert "could not open the search" in (tmp_path / "err0.txt").read_text() and "could not open the search" in (tmp_path / "err1.txt").read_text()
    # (4) evaluate() raising on rank 1 and (5) rank 0 failing after the evaluation: every rank raises, nobody waits in a collective
    assert "this rank (1): FloatingPointError" in (tmp_path / "everr1.txt").read_text() and "another rank" in (tmp_path / "everr0.txt").read_text()
    assert "disk full" in (tmp_path / "tellerr0.txt").read_text() and "rank 0 failed after evaluating generation 1" in (tmp_path / "tellerr1.txt").read_text()
    # the files are the single-GPU trainer's: it continues the sharded run from its checkpoint
    from ecad_amd.genetic import train_nsga2_single_gpu as T

    mgr, algo = T.run(T.parse_args(["--image-generator", "PixArtAlphaImageGenerator", "--all-populations-dir", str(tmp_path / "p"), "--all-benchmarks-dir", str(tmp_path / "b"),
                                    "--population-size", "8", "--load-from", str(tmp_path / "p/parts/gen_004/manager_config.json"), "--num-cycles", "1"]), score_fn=_stub_score)
    assert mgr.generation_num == algo.n_gen == 5
