"""On-disk population bookkeeping for the NSGA-II search (file contracts of the reference:
``ecad/genetic/population_io_manager.py:24-400`` and ``pixart_population_io_manager.py:28-240``).

    <populations>/<name>/gen_XXX/candidates/cand_YYY.json      schedule JSON (+ metrics.total_macs_T)
    <benchmarks>/<name>/gen_XXX/candidates/cand_YYY/scores.json {"total_score": float}
    <populations>/<name>/gen_XXX/manager_config.json

F = [1 - score, TMACs] (``load_evaluation_scores`` :246-273), G = min_diff - hamming(x, default) (:216-225).
"""
from __future__ import annotations

import json
import re
from pathlib import Path
from typing import Any

import numpy as np

from ecad_amd.benchmark.compute_macs import total_macs_T
from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule


class PixArtPopulationIOManager:
    CONFIG_FILENAME = "manager_config.json"
    CHECKPOINT_FILENAME = "checkpoint.pkl"  # reference: population_io_manager.py:27,116-145 (dill dump of the algorithm, per generation dir)
    SCORE_KEY = "total_score"
    METRIC_KEY = "total_macs_T"

    def __init__(self, name: str, all_populations_dir: Path, all_benchmarks_dir: Path, generation_num: int | None = None,
                 num_inference_steps: int = 20, min_diff_from_default: int = 1, population_size: int = 72, num_blocks: int = 28,
                 num_component_types: int = 3, maximize_macs: bool = False, candidate_config: dict | None = None):
        self.name = name
        self.population_dir = Path(all_populations_dir) / name
        self.benchmark_dir = Path(all_benchmarks_dir) / name
        self.population_dir.mkdir(parents=True, exist_ok=True)
        self.benchmark_dir.mkdir(parents=True, exist_ok=True)
        if generation_num is None:
            gens = [int(p.name.split("_")[1]) for p in self.population_dir.iterdir() if p.is_dir() and re.fullmatch(r"gen_\d+", p.name)]
            generation_num = max(gens) if gens else 1
        self.generation_num = generation_num
        self.num_inference_steps = num_inference_steps
        self.min_diff_from_default = min_diff_from_default
        self.population_size = population_size
        self.num_blocks = num_blocks
        self.num_component_types = num_component_types
        self.maximize_macs = maximize_macs
        self.candidate_config: dict[str, Any] = candidate_config or {}
        self.default_bits = np.ones(num_inference_steps * num_blocks * num_component_types, dtype=np.bool_)

    # -- paths ------------------------------------------------------------------------------
    def _gen_dir(self, generation: int | None = None) -> Path:
        g = self.generation_num if generation is None else generation
        d = self.population_dir / f"gen_{g:03d}"
        d.mkdir(parents=True, exist_ok=True)
        return d

    def get_pop_candidates_dir(self, generation: int | None = None) -> Path:
        d = self._gen_dir(generation) / "candidates"
        d.mkdir(parents=True, exist_ok=True)
        return d

    def get_benchmark_gen_dir(self, generation: int | None = None) -> Path:
        g = self.generation_num if generation is None else generation
        return self.benchmark_dir / f"gen_{g:03d}/candidates"

    # -- config -----------------------------------------------------------------------------
    def to_dict(self) -> dict[str, Any]:
        return {
            "name": self.name, "population_dir": str(self.population_dir), "benchmark_dir": str(self.benchmark_dir),
            "generation_num": self.generation_num, "num_inference_steps": self.num_inference_steps,
            "min_diff_from_default": self.min_diff_from_default, "population_size": self.population_size,
            "num_blocks": self.num_blocks, "num_component_types": self.num_component_types,
        }

    def to_json(self, file_path: Path | None = None) -> None:
        file_path = file_path or self._gen_dir() / self.CONFIG_FILENAME
        with open(file_path, "w") as f:
            json.dump(self.to_dict(), f, indent=4)

    @classmethod
    def from_json(cls, file_path: Path) -> "PixArtPopulationIOManager":
        with open(file_path) as f:
            c = json.load(f)
        kw = {k: c[k] for k in ("generation_num", "num_inference_steps", "num_blocks", "num_component_types", "min_diff_from_default", "population_size") if k in c}
        return cls(c["name"], Path(c["population_dir"]).parent, Path(c["benchmark_dir"]).parent, **kw)

    # -- population <-> files -----------------------------------------------------------------
    @staticmethod
    def binary_vector_to_schedule_dict(x: np.ndarray, num_inference_steps: int, num_blocks: int, num_component_types: int) -> dict:
        return PixArtCacheSchedule.from_numpy(x, num_blocks=num_blocks, num_inference_steps=num_inference_steps).schedule

    def compute_additional_info(self, schedule: dict) -> dict[str, Any]:
        bits = PixArtCacheSchedule(self.num_blocks, self.num_inference_steps, "tmp", schedule).to_numpy()
        changed = ~bits.all(axis=2)  # [steps, blocks]: any component differs from the all-true default
        return {"num_affected_steps": int(changed.any(axis=1).sum()), "num_affected_blocks": int(changed.any(axis=0).sum()),
                "total_num_affected_blocks": int(changed.sum())}

    # -- model-specific hooks (overridden by FluxPopulationIOManager) ----------------------------
    def _schedule_from_vector(self, x: np.ndarray, name: str):
        sd = self.binary_vector_to_schedule_dict(x, self.num_inference_steps, self.num_blocks, self.num_component_types)
        return PixArtCacheSchedule(self.num_blocks, self.num_inference_steps, name, sd, attributes=self.compute_additional_info(sd),
                                   top_level_config=self.candidate_config)

    def _schedule_from_json(self, fp: Path):
        return PixArtCacheSchedule.from_json(fp)

    def _macs_T(self, bits: np.ndarray) -> float:
        return total_macs_T(np.asarray(bits).reshape(self.num_inference_steps, self.num_blocks, 3))

    def save_population(self, population: np.ndarray, generation: int | None = None) -> None:
        g = self.generation_num if generation is None else generation
        for i, x in enumerate(np.asarray(population, dtype=np.bool_)):
            self._schedule_from_vector(x, f"{self.name}_gen_{g:03d}_cand_{i:03d}").to_json(self.get_pop_candidates_dir(g) / f"cand_{i:03d}.json")

    def load_population_schedules(self, generation: int | None = None) -> list:
        out = []
        for fp in self.get_pop_candidates_dir(generation).glob("cand_*.json"):
            out.append((int(fp.stem.split("_")[-1]), self._schedule_from_json(fp)))
        return sorted(out, key=lambda t: t[0])

    def load_population_vectors(self, generation: int | None = None) -> np.ndarray:
        return np.array([s.to_numpy(flatten=True) for _, s in self.load_population_schedules(generation)])

    def get_constraint_violations(self, population: np.ndarray) -> np.ndarray:
        diff = np.sum(np.asarray(population, dtype=np.bool_) != self.default_bits[None], axis=1)
        return (self.min_diff_from_default - diff).reshape(-1, 1)

    # -- algorithm checkpoint (resume a search where it stopped) -----------------------------------
    def save_algorithm(self, algorithm: Any) -> None:
        n_gen = getattr(algorithm, "n_gen", None)
        if n_gen is not None:  # the algorithm's generation counter is the authority
            self.generation_num = n_gen
        import dill

        target = self._gen_dir() / self.CHECKPOINT_FILENAME
        with open(target, "wb") as f:
            dill.dump(algorithm, f)
        print(f"algorithm checkpoint -> {target}")

    def load_algorithm(self, generation: int | None = None) -> Any:
        """FileNotFoundError when the generation has no checkpoint: callers start from the seed population then."""
        import dill

        source = self._gen_dir(generation) / self.CHECKPOINT_FILENAME
        with open(source, "rb") as f:
            algorithm = dill.load(f)
        print(f"algorithm checkpoint <- {source} (generation {algorithm.n_gen})")
        return algorithm

    # -- evaluation results -------------------------------------------------------------------
    def write_evaluation(self, index: int, score: float, bits: np.ndarray, generation: int | None = None) -> None:
        """In-process evaluation writes what score_images.py / compute_macs.py would have written."""
        d = self.get_benchmark_gen_dir(generation) / f"cand_{index:03d}"
        d.mkdir(parents=True, exist_ok=True)
        with open(d / "scores.json", "w") as f:
            json.dump({self.SCORE_KEY: float(score)}, f)
        fp = self.get_pop_candidates_dir(generation) / f"cand_{index:03d}.json"
        with open(fp) as f:
            data = json.load(f)
        data.setdefault("metrics", {})[self.METRIC_KEY] = self._macs_T(bits)
        with open(fp, "w") as f:
            json.dump(data, f, indent=4)

    def check_offline_eval(self) -> bool:
        cand_dirs = [d for d in self.get_benchmark_gen_dir().glob("cand_*/") if d.is_dir()]
        if any(not list(d.glob("scores*.json")) for d in cand_dirs):
            return False
        jsons = list(self.get_pop_candidates_dir().glob("cand_*.json"))
        if not jsons or len(cand_dirs) != len(jsons):
            return False
        for fp in jsons:
            with open(fp) as f:
                m = json.load(f).get("metrics", {})
            if self.METRIC_KEY not in m and "total_macs" not in m:
                return False
        return True

    def load_evaluation_scores(self, generation: int | None = None) -> np.ndarray:
        scores, metrics = {}, {}
        for d in self.get_benchmark_gen_dir(generation).glob("cand_*"):
            js = sorted(d.glob("scores*.json"))
            if d.is_dir() and js:
                with open(js[0]) as f:
                    scores[int(d.name.split("_")[1])] = 1.0 - json.load(f)[self.SCORE_KEY]
        for fp in self.get_pop_candidates_dir(generation).glob("cand_*.json"):
            with open(fp) as f:
                m = json.load(f)["metrics"]
            metrics[int(fp.stem.split("_")[1])] = m[self.METRIC_KEY] if self.METRIC_KEY in m else m["total_macs"] / 1000**4
        if scores.keys() != metrics.keys():
            raise ValueError(f"candidate indices in scores and metrics do not match: {sorted(scores)} vs {sorted(metrics)}")
        F = np.zeros((len(scores), 2))
        for i in scores:
            F[i] = [scores[i], -metrics[i] if self.maximize_macs else metrics[i]]
        return F

    def ask(self):
        X = self.load_population_vectors()
        return X, self.load_evaluation_scores(), self.get_constraint_violations(X)


class FluxPopulationIOManager(PixArtPopulationIOManager):
    """FLUX flavour (reference: ``ecad/genetic/flux_population_io_manager.py:28-265``): population 24, 19 + 38 blocks, three component
    kinds each, candidates carry the ``height / width / guidance_scale`` config block; the flat vector is
    ``FluxCacheSchedule.to_numpy`` (per step: full blocks, then single blocks)."""

    def __init__(self, name: str, all_populations_dir: Path, all_benchmarks_dir: Path, generation_num: int | None = None, num_inference_steps: int = 20,
                 min_diff_from_default: int = 1, population_size: int = 24, num_blocks: int = 19, num_single_blocks: int = 38, num_component_types_full: int = 3,
                 num_component_types_single: int = 3, height: int = 256, width: int = 256, guidance_scale: float = 5, maximize_macs: bool = False):
        super().__init__(name, all_populations_dir, all_benchmarks_dir, generation_num, num_inference_steps, min_diff_from_default, population_size, num_blocks, 3,
                         maximize_macs, candidate_config={"height": height, "width": width, "guidance_scale": guidance_scale})
        self.num_single_blocks = num_single_blocks
        self.num_component_types_full, self.num_component_types_single = num_component_types_full, num_component_types_single
        self.height, self.width, self.guidance_scale = height, width, guidance_scale
        self.default_bits = np.ones(num_inference_steps * (num_blocks + num_single_blocks) * 3, dtype=np.bool_)

    def to_dict(self) -> dict[str, Any]:
        d = super().to_dict()
        d.pop("num_component_types", None)
        d.update(num_single_blocks=self.num_single_blocks, num_component_types_full=self.num_component_types_full,
                 num_component_types_single=self.num_component_types_single, height=self.height, width=self.width, guidance_scale=self.guidance_scale)
        return d

    @classmethod
    def from_json(cls, file_path: Path) -> "FluxPopulationIOManager":
        with open(file_path) as f:
            c = json.load(f)
        keys = ("generation_num", "num_inference_steps", "num_blocks", "num_single_blocks", "min_diff_from_default", "population_size", "height", "width", "guidance_scale")
        return cls(c["name"], Path(c["population_dir"]).parent, Path(c["benchmark_dir"]).parent, **{k: c[k] for k in keys if k in c})

    @staticmethod
    def binary_vector_to_schedule_dict(x: np.ndarray, num_inference_steps: int, num_blocks: int, num_single_blocks: int, num_component_types_full: int = 3,
                                       num_component_types_single: int = 3) -> dict:
        from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule

        return FluxCacheSchedule.from_numpy(x, num_blocks=num_blocks, num_single_blocks=num_single_blocks, num_inference_steps=num_inference_steps).schedule

    def compute_additional_info(self, schedule: dict) -> dict[str, Any]:
        steps, full, single, total = 0, set(), set(), 0
        for per_block in schedule.values():
            changed = [b for b, comp in per_block.items() if not all(comp.values())]
            steps += bool(changed)
            total += len(changed)
            for b in changed:
                (single if str(b).startswith("single_") else full).add(b)
        return {"num_affected_steps": steps, "num_affected_blocks": len(full), "num_affected_blocks_single": len(single), "total_num_affected_blocks": total}

    def _schedule_from_vector(self, x: np.ndarray, name: str):
        from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule

        sd = self.binary_vector_to_schedule_dict(x, self.num_inference_steps, self.num_blocks, self.num_single_blocks)
        return FluxCacheSchedule(self.num_blocks, self.num_inference_steps, name, sd, attributes=self.compute_additional_info(sd),
                                 top_level_config=self.candidate_config, num_single_blocks=self.num_single_blocks)

    def _schedule_from_json(self, fp: Path):
        from ecad_amd.schedulers.cache_scheduler.flux_cache_schedule import FluxCacheSchedule

        return FluxCacheSchedule.from_json(fp)

    def _split(self, bits: np.ndarray):
        v = np.asarray(bits, dtype=np.bool_).reshape(self.num_inference_steps, (self.num_blocks + self.num_single_blocks) * 3)
        return v[:, : self.num_blocks * 3].reshape(self.num_inference_steps, self.num_blocks, 3), v[:, self.num_blocks * 3:].reshape(self.num_inference_steps, self.num_single_blocks, 3)

    def _macs_T(self, bits: np.ndarray) -> float:
        from ecad_amd.benchmark.compute_macs import flux_macs_per_step

        full, single = self._split(bits)
        return float(flux_macs_per_step(full, single, tokens=(self.height // 16) * (self.width // 16)).sum()) / 1e12
