#!/usr/bin/env python3
"""NSGA-II over cache schedules with the population sharded across the GPUs of one node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m ecad_amd.genetic.train_nsga2_multi_gpu --name demo --generations 3 --seed-population tests/golden/ga_seed_population.npz

Same on-disk contract and the same resumable cycle as ``train_nsga2_single_gpu.py`` (gen_XXX/candidates/cand_YYY.json, scores.json,
``gen_XXX/manager_config.json`` + ``gen_XXX/checkpoint.pkl`` written after every generation; ``--load-from <manager_config.json>`` resumes, and a run
started by either trainer can be continued by the other); evaluation happens in-process on every rank instead of three serial subprocesses.
Rank 0 owns ask/tell and the files; ranks exchange a 3-integer header (error flag, generation, rows), the bit matrix (broadcast) and 2 float64
per candidate (all-gather).  Failures reach every rank instead of hanging a collective: rank 0's ask / open errors through the header in front of the
bit matrix, an exception inside any rank's scoring through a one-integer MAX all-reduce that every rank enters in front of the all-gather (evaluate.py), rank 0's tell / ask / file errors
through a second header behind them.
"""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np
import torch

from ecad_amd import distributed as D
from ecad_amd.genetic.evaluate import engine_score_fn, evaluate_population
from ecad_amd.genetic.nsga2 import NSGA2
from ecad_amd.genetic.population_io_manager import PixArtPopulationIOManager


def open_search(make_manager, manager_cls, n_var: int, population_size: int, load_from: Path | None, seed_bits: np.ndarray | None):
    """Rank 0: (manager, algorithm) resumed from ``load_from`` (+ the checkpoint.pkl beside it) or started new, with generation 1 on disk and checkpointed."""
    from ecad_amd.genetic.train_nsga2_single_gpu import checkpoint_manager_and_algorithm

    if load_from is not None:
        mgr = manager_cls.from_json(load_from)
        algo = mgr.load_algorithm()
        if algo.n_var != n_var:
            raise ValueError(f"checkpoint searches {algo.n_var}-bit schedules, this model has {n_var}")
        return mgr, algo
    mgr = make_manager()
    algo = NSGA2(pop_size=population_size, n_var=n_var, crossover_prob=0.9, n_points=4, mutation_prob=0.05, seed=0)
    if seed_bits is not None:
        seed_bits = np.asarray(seed_bits, dtype=np.bool_)
        if seed_bits.ndim != 2 or seed_bits.shape[1] != n_var or len(seed_bits) == 0:
            raise ValueError(f"seed population must be [rows >= 1, {n_var}] bits, got {seed_bits.shape}")
        seed_bits = seed_bits[:population_size]
    algo.setup(seed_bits)
    mgr.generation_num = algo.n_gen
    mgr.save_population(algo.ask(), algo.n_gen)
    checkpoint_manager_and_algorithm(mgr, algo)
    return mgr, algo


def sharded_search(rank: int, world: int, open_fn, n_var: int, generations: int, evaluate) -> tuple:
    """The generation loop.  Every generation: rank 0 takes the candidates waiting on disk (fewer than pop_size rows after duplicate elimination
    or a short seed population are fine: the header carries the row count), all ranks evaluate their shard, rank 0 writes the scores, tells,
    asks, writes the next generation and checkpoints -- one evaluation and one advance per generation, like the single-GPU cycle."""
    from ecad_amd.genetic.train_nsga2_single_gpu import checkpoint_manager_and_algorithm

    mgr = algo = None
    hdr, msg = [0, 0, 0], ""
    if rank == 0:
        try:
            mgr, algo = open_fn()
        except Exception as e:  # noqa: BLE001  (reported on every rank below)
            hdr, msg = [1, 0, 0], f"{type(e).__name__}: {e}"
    if D.broadcast_ints(hdr, 3)[0]:
        raise RuntimeError(f"rank 0 could not open the search{': ' + msg if msg else ''}")
    for _ in range(generations):
        X, hdr, msg = None, [0, 0, 0], ""
        if rank == 0:
            try:
                X = np.asarray(algo.ask(), dtype=np.bool_)
                if X.ndim != 2 or X.shape[1] != n_var or len(X) == 0:
                    raise ValueError(f"population has shape {X.shape}, expected [1..{algo.pop_size}, {n_var}]")
                hdr = [0, algo.n_gen, len(X)]
            except Exception as e:  # noqa: BLE001
                hdr, msg = [1, 0, 0], f"{type(e).__name__}: {e}"
        err, gen, rows = D.broadcast_ints(hdr, 3)
        if err:
            raise RuntimeError(f"rank 0 failed before generation broadcast{': ' + msg if msg else ''}")
        X = D.broadcast_population(X, (rows, n_var))
        F = evaluate(X)  # evaluate_population: a scoring failure on any rank raises on every rank (flag exchange in front of its all-gather)
        hdr, msg = [0, gen, rows], ""
        if rank == 0:
            try:
                for i in range(len(X)):
                    mgr.write_evaluation(i, 1.0 - F[i, 0], X[i], gen)
                Xd, Fd, Gd = mgr.ask()
                algo.tell(Xd, Fd, Gd)
                mgr.generation_num = algo.n_gen
                mgr.save_population(algo.ask(), algo.n_gen)
                checkpoint_manager_and_algorithm(mgr, algo)
                print(f"generation {gen}: {rows} candidates, best (1-score) {Fd[:, 0].min():.4f}, min TMACs {Fd[:, 1].min():.3f}, evaluated on {world} rank(s)")
            except Exception as e:  # noqa: BLE001
                hdr, msg = [1, gen, rows], f"{type(e).__name__}: {e}"
        # status of rank 0's tell / ask / file section: without it a failure there left the other ranks in the next generation's broadcast
        if D.broadcast_ints(hdr, 3)[0]:
            raise RuntimeError(f"rank 0 failed after evaluating generation {gen}{': ' + msg if msg else ''}")
    D.barrier()
    return mgr, algo


def main_flux(args, rank, world, local_rank):
    """FLUX search: population default 24 (reference flux_population_io_manager.py:38), candidates are flat FluxCacheSchedule vectors."""
    from types import SimpleNamespace

    from ecad_amd.flux_engine import EcadFluxEngine
    from ecad_amd.genetic.evaluate import flux_engine_score_fn
    from ecad_amd.genetic.population_io_manager import FluxPopulationIOManager
    from ecad_amd.pipelines.flow_match import flow_match_sigmas, model_scalars
    from ecad_amd.transformer_2d_models.flux_transformer_2d_edited import synthetic_state_dict

    nb, ns, heads = (int(v) for v in args.flux_geometry.split(","))
    full = heads == 24
    jd, pd = (4096, 768) if full else (128, 64)
    steps, gh, gw, T, B = 20, args.height // 16, args.width // 16, 512 if full else 32, args.images
    n_var = steps * (nb + ns) * 3
    eng = EcadFluxEngine(num_layers=nb, num_single_layers=ns, num_heads=heads, joint_attention_dim=jd, pooled_projection_dim=pd, max_batch=B, max_image_tokens=gh * gw,
                         max_text_tokens=T, dtype=torch.bfloat16, device=local_rank)
    cfg = SimpleNamespace(num_layers=nb, num_single_layers=ns, num_attention_heads=heads, attention_head_dim=128, in_channels=64, joint_attention_dim=jd,
                          pooled_projection_dim=pd, guidance_embeds=True)
    eng.load_state_dict(synthetic_state_dict(cfg, 0, device=f"cuda:{local_rank}" if full else "cpu", dtype=torch.bfloat16 if full else torch.float32))
    g = torch.Generator().manual_seed(1)
    dev = torch.device("cuda", local_rank)
    enc = (torch.randn(B, T, jd, generator=g) * 0.1).bfloat16().to(dev)
    pooled = (torch.randn(B, pd, generator=g) * 0.5).bfloat16().to(dev)
    lat = torch.randn(B, gh * gw, 64, generator=g).to(dev)
    sig = flow_match_sigmas(steps, gh * gw)
    tm, gm = model_scalars(sig, 5.0, torch.bfloat16)
    score_fn = flux_engine_score_fn(eng, enc, pooled, lat, gh, gw, tm, gm, sig, nb, ns)

    pop = args.population_size
    # every rank needs the closed-form MAC objective; the formula only needs the geometry
    from ecad_amd.benchmark.compute_macs import flux_macs_per_step

    def macs_fn(bits):
        v = np.asarray(bits, dtype=np.bool_).reshape(steps, (nb + ns) * 3)
        return float(flux_macs_per_step(v[:, : nb * 3].reshape(steps, nb, 3), v[:, nb * 3:].reshape(steps, ns, 3), tokens=gh * gw, dim=heads * 128,
                                        text_tokens=T, joint_attention_dim=jd, pooled_dim=pd).sum()) / 1e12

    def make_manager():
        return FluxPopulationIOManager(args.name, args.populations_dir, args.benchmarks_dir, generation_num=1, population_size=pop, num_blocks=nb, num_single_blocks=ns,
                                       height=args.height, width=args.width)

    sharded_search(rank, world, lambda: open_search(make_manager, FluxPopulationIOManager, n_var, pop, args.load_from, None), n_var, args.generations,
                   lambda X: evaluate_population(X, score_fn, macs_fn=macs_fn))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="nsga2_multi_gpu")
    ap.add_argument("--populations-dir", type=Path, default=Path("results/genetic/populations"))
    ap.add_argument("--benchmarks-dir", type=Path, default=Path("results/benchmark/genetic/populations"))
    ap.add_argument("--seed-population", type=Path, default=None, help="npz with packed 'bits' [pop, 210]")
    ap.add_argument("--load-from", type=Path, default=None, help="gen_XXX/manager_config.json of an existing run (either trainer's): resume from its checkpoint.pkl")
    ap.add_argument("--population-size", type=int, default=72)
    ap.add_argument("--generations", type=int, default=1)
    ap.add_argument("--images", type=int, default=4, help="prompts per candidate evaluation (synthetic embeddings)")
    ap.add_argument("--num-layers", type=int, default=28)
    ap.add_argument("--model", default="pixart", choices=["pixart", "flux"], help="flux: FluxPopulationIOManager contract (reference train_nsga2_base.py / flux_problem.py)")
    ap.add_argument("--flux-geometry", default="19,38,24", help="joint blocks, single blocks, heads (reduce for smoke runs, e.g. 2,3,2)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    args = ap.parse_args()

    rank, world, local_rank = D.init()
    if args.model == "flux":
        return main_flux(args, rank, world, local_rank)
    from ecad_amd.engine import EcadEngine
    from ecad_amd.pipelines.dpm_solver import DPMSolverMultistepScheduler
    from ecad_amd.transformer_2d_models.pixart_transformer_2d_edited import synthetic_state_dict
    from types import SimpleNamespace

    steps, blocks, n_var = 20, args.num_layers, 20 * args.num_layers * 3
    eng = EcadEngine(num_layers=blocks, max_batch=2 * args.images, dtype=torch.float16, device=local_rank)
    eng.load_state_dict(synthetic_state_dict(SimpleNamespace(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2,
                                                             num_layers=blocks, caption_channels=4096), seed=0))
    g = torch.Generator().manual_seed(1)
    B, T = args.images, 120
    enc = torch.cat([(torch.randn(1, T, 4096, generator=g) * 0.1).repeat(B, 1, 1), torch.randn(B, T, 4096, generator=g) * 0.1])
    mask = torch.ones(2 * B, T, dtype=torch.uint8)
    lat = torch.randn(B, 4, 32, 32, generator=g)
    solver = DPMSolverMultistepScheduler()
    solver.set_timesteps(steps)
    score_fn = engine_score_fn(eng, enc.cuda(), mask.cuda(), lat.cuda(), solver.timesteps.numpy().astype(np.float32), solver.coefficient_table())

    seed_bits = None
    if rank == 0 and args.seed_population is not None and args.load_from is None:
        seed_bits = np.unpackbits(np.load(args.seed_population)["bits"], axis=1)[:, :n_var].astype(np.bool_)

    def make_manager():
        return PixArtPopulationIOManager(args.name, args.populations_dir, args.benchmarks_dir, generation_num=1, population_size=args.population_size, num_blocks=blocks)

    sharded_search(rank, world, lambda: open_search(make_manager, PixArtPopulationIOManager, n_var, args.population_size, args.load_from, seed_bits), n_var,
                   args.generations, lambda X: evaluate_population(X, score_fn, steps, blocks))


if __name__ == "__main__":
    main()
