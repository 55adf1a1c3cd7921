"""Sharded evaluation of one NSGA-II generation: candidates split in contiguous blocks across ranks (72 over 8 GPUs = 9
per GPU), each rank generates its candidates' images with its own engine, scores them locally and contributes two
float64 per candidate to one all-gather.  Replaces the three serial subprocesses of
``ecad/genetic/train_nsga2_single_gpu.py:198-224`` (generate_images / score_images / compute_macs).

ImageReward (``ecad/benchmark/score_images.py``) is a third-party model that is absent here, so the quality
objective is a labelled STAND-IN: negative relative L2 distance of the final latents to the uncached (all-true
schedule) run with identical seeds.  The MAC objective is exact (closed form).
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from ecad_amd import distributed as D
from ecad_amd.benchmark.compute_macs import total_macs_T


def latent_fidelity_score(latents, reference_latents) -> float:
    """Stand-in quality score in (-inf, 0]: 0 == identical to the uncached run."""
    import torch

    return -float((latents.float() - reference_latents.float()).norm() / reference_latents.float().norm().clamp_min(1e-12))


def evaluate_population(X: np.ndarray, score_fn: Callable[[int, np.ndarray], float], steps: int = 20, blocks: int = 28,
                        macs_fn: Callable[[np.ndarray], float] | None = None) -> np.ndarray:
    """F[pop, 2] = [1 - score, TMACs] for the whole population, computed cooperatively by all ranks.  ``macs_fn`` (flat bit vector ->
    TMACs) overrides the PixArt closed form (FLUX: ``FluxPopulationIOManager._macs_T``)."""
    rank, world = D._world()
    n = len(X)
    mine = D.shard_range(n, world, rank)
    local = np.zeros((len(mine), 2), dtype=np.float64)
    err = ""
    try:
        for k, c in enumerate(mine):
            if macs_fn is None:
                bits = np.asarray(X[c], dtype=np.bool_).reshape(steps, blocks, 3)
                local[k] = [1.0 - score_fn(c, bits), total_macs_T(bits)]
            else:
                bits = np.asarray(X[c], dtype=np.bool_)
                local[k] = [1.0 - score_fn(c, bits), macs_fn(bits)]
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    # one flag exchange that EVERY rank enters before the all-gather: a rank whose scoring raised must not leave the others waiting in it
    if D.any_rank_failed(bool(err)):
        raise RuntimeError("candidate evaluation failed on " + (f"this rank ({rank}): {err}" if err else "another rank"))
    return D.gather_fitness(local, n)


def engine_score_fn(engine, enc, mask, latents, timesteps, coefs, guidance: float = 4.5):
    """score_fn over a fixed prompt/latent batch on this rank's engine (reference latents from the all-true schedule).  The engine's LayerNorm fold is
    switched off for the search (every candidate schedule runs for ONE batch: the fold's per-schedule weight preparation would never be amortised);
    ``score.restore()`` puts the engine's previous setting back -- call it before the engine goes on to serve a fixed schedule."""
    steps = len(timesteps)
    fold_before = getattr(engine, "ln_fold", True)
    engine.set_ln_fold(False)
    engine.set_schedule(np.ones((steps, engine.cfg.num_layers, 3), dtype=np.uint8))
    ref = engine.run_schedule(enc, mask, latents, timesteps, coefs, guidance)

    def score(_idx: int, bits: np.ndarray) -> float:
        engine.set_schedule(bits)
        return latent_fidelity_score(engine.run_schedule(enc, mask, latents, timesteps, coefs, guidance), ref)

    score.restore = lambda: engine.set_ln_fold(fold_before)
    return score


def _pad_batch(enc, mask, latents, full: int):
    """A partial last prompt batch (config 4: 100 prompts = 6 x 16 + 4) filled up to the full batch size by repeating its last prompt / latent: the candidate then needs
    ONE plan set and ONE graph capture per schedule instead of two, at the price of 12 discarded images.  Measured on config 4 (round 6, same box, same scores to the
    last bit): 9.83 s per generation against 9.46 without -- lowering + capture of a second batch size costs 10-25 ms (scripts/ga_time_probe.py), twelve more images of
    these lightly cached seed schedules 70-100 ms -- so it is OFF by default (``pad_partial=True`` for callers whose schedules change every batch).  The rows of the real
    images are what they are in a batch of their own: a row's arithmetic does not depend on its neighbours from batch 4 up (engine tests)."""
    import torch

    n = latents.shape[0]
    if n == full:
        return enc, mask, latents, n
    r = full - n
    enc = torch.cat([enc[:n], enc[n - 1 : n].expand(r, *enc.shape[1:]), enc[n:], enc[2 * n - 1 : 2 * n].expand(r, *enc.shape[1:])], 0)   # [negative | prompt] halves
    mask = torch.cat([mask[:n], mask[n - 1 : n].expand(r, *mask.shape[1:]), mask[n:], mask[2 * n - 1 : 2 * n].expand(r, *mask.shape[1:])], 0)
    latents = torch.cat([latents, latents[n - 1 : n].expand(r, *latents.shape[1:])], 0)
    return enc.contiguous(), mask.contiguous(), latents.contiguous(), n


def engine_score_fn_batches(engine, batches, timesteps, coefs, guidance: float = 4.5, contexts: int = 3, pad_partial: bool = False):
    """score_fn over SEVERAL fixed prompt batches (config 4: 100 prompts per candidate = 6 batches of 16 + one of 4), the batches of a candidate kept in flight on
    ``contexts`` execution contexts over this rank's weights (``EngineContexts``: batch i + 1 .. i + contexts - 1 run while batch i is being scored; every batch's
    latents are bit-identical to running it alone).  Score = mean over the batches of the stand-in fidelity to the uncached run of the same batch (real images only:
    ``pad_partial`` fills a smaller batch up to the largest one, see ``_pad_batch``).  ``score.restore()`` as in ``engine_score_fn``; ``score.close()`` releases the
    extra contexts."""
    import torch

    from ecad_amd.engine import EngineContexts

    full = max(l.shape[0] for _, _, l in batches)
    real = [l.shape[0] for _, _, l in batches]
    if pad_partial and 2 * full <= engine.cfg.max_batch:
        padded = [_pad_batch(e, m, l, full) for e, m, l in batches]
        batches = [(e, m, l) for e, m, l, _ in padded]
    steps = len(timesteps)
    fold_before = getattr(engine, "ln_fold", True)
    engine.set_ln_fold(False)
    engine.set_schedule(np.ones((steps, engine.cfg.num_layers, 3), dtype=np.uint8))
    refs = [engine.run_schedule(e, m, l, timesteps, coefs, guidance)[:n].clone() for (e, m, l), n in zip(batches, real)]
    pool = EngineContexts(engine, max(1, int(contexts))) if contexts > 1 else None
    if pool is not None:
        for ctx in pool.engines[1:]:
            ctx.set_ln_fold(False)

    def score(_idx: int, bits: np.ndarray) -> float:
        if pool is None:
            engine.set_schedule(bits)
            return float(np.mean([latent_fidelity_score(engine.run_schedule(e, m, l, timesteps, coefs, guidance)[: r.shape[0]], r) for (e, m, l), r in zip(batches, refs)]))
        pool.set_schedule(bits)
        pending, scores = [], []

        def finish():
            i, out, st = pending.pop(0)
            torch.cuda.current_stream(engine.device).wait_stream(st)
            scores.append(latent_fidelity_score(out[: refs[i].shape[0]], refs[i]))  # .norm() -> float: waits for that batch only

        for i, (e, m, l) in enumerate(batches):
            out, st = pool.submit(e, m, l, timesteps, coefs, guidance)
            pending.append((i, out, st))
            if len(pending) >= len(pool):
                finish()
        while pending:
            finish()
        return float(np.mean(scores))

    def restore():
        engine.set_ln_fold(fold_before)

    def close():
        if pool is not None:
            pool.close()

    score.restore, score.close, score.references = restore, close, refs
    return score


def flux_engine_score_fn(engine, enc, pooled, latents, grid_h: int, grid_w: int, timesteps_x1000, guidance_x1000: float, sigmas, num_blocks: int, num_single_blocks: int):
    """score_fn for FLUX candidates (flat ``FluxCacheSchedule.to_numpy`` vectors) on this rank's ``EcadFluxEngine``."""
    steps = len(timesteps_x1000)

    def split(bits):
        v = np.asarray(bits, dtype=np.bool_).reshape(steps, (num_blocks + num_single_blocks) * 3)
        return v[:, : num_blocks * 3].reshape(steps, num_blocks, 3), v[:, num_blocks * 3:].reshape(steps, num_single_blocks, 3)

    def run(bits):
        engine.set_schedule(*split(bits))
        return engine.run_schedule(latents.clone(), enc, pooled, grid_h, grid_w, timesteps_x1000, guidance_x1000, sigmas)

    ref = run(np.ones(steps * (num_blocks + num_single_blocks) * 3, dtype=np.bool_))

    def score(_idx: int, bits: np.ndarray) -> float:
        return latent_fidelity_score(run(bits), ref)

    return score
