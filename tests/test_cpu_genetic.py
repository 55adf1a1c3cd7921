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
