#!/usr/bin/env python3
"""NSGA-II over cache schedules on one GPU, resumable: the reference's ``ecad/genetic/train_nsga2_single_gpu.py`` entry point.

Same command line (every flag of ``train_nsga2_base.get_base_argparser`` + ``--print-eval-outputs``), same files
(``gen_XXX/candidates/cand_YYY.json``, ``gen_XXX/manager_config.json``, ``gen_XXX/checkpoint.pkl``, benchmark
``gen_XXX/candidates/cand_YYY/scores.json``, ``metrics.total_macs_T`` inside each candidate file) and the same cycle:

    resume from the checkpoint of the manager's generation, or seed generation 1 from ``gen_000/candidates``;
    per cycle: [tell + ask + write the next candidates, if the generation on disk already has its scores]; checkpoint; evaluate the
               generation on disk; [tell + ask + write + checkpoint, if the first bracket did not run]
               -- one evaluation and one advance per cycle, whichever state the run was resumed in (reference :160-243).

What differs is the evaluation step.  The reference shells out to three scripts per generation (``generate_images.py``,
``score_images.py``, ``compute_macs.py``: train_nsga2_single_gpu.py:28-123,198-224), each of which reloads the model.  Here one
engine stays resident and the generation is evaluated in this process (``InProcessEvaluator``): latents for every candidate from
the native whole-loop path, TMACs from the closed-form model that reproduces the reference's calflops numbers.  The quality
objective is a labelled stand-in (ImageReward is a third-party model that is not available offline): 1 + negative relative L2
distance of the final latents to the uncached run of the same prompts and seeds.  The flags that only parameterise the ImageReward
scorer (``--benchmark-prompts``, ``--file-mode``, ``--image-naming-mode``, ``--exactly-n-images``, ``--regen-if-not-n-images``)
are accepted and echoed by ``--print-not-submit``, nothing more.  For more than one GPU use ``train_nsga2_multi_gpu.py``.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path
from typing import Any, Callable

import numpy as np

from ecad_amd.genetic.nsga2 import NSGA2
from ecad_amd.genetic.population_io_manager import FluxPopulationIOManager, PixArtPopulationIOManager

REPO_ROOT = Path(__file__).resolve().parents[2]
DEFAULT_POPULATIONS_DIR = REPO_ROOT / "results/genetic/populations/"
DEFAULT_BENCHMARKS_DIR = REPO_ROOT / "results/benchmark/genetic/populations/"
GENERATOR_NAMES = ("PixArtAlphaImageGenerator", "PixArtSigmaImageGenerator", "FluxImageGenerator")

ScoreFn = Callable[[int, np.ndarray], float]


# ------------------------------------------------------------------------------------------------
# command line
# ------------------------------------------------------------------------------------------------
def get_base_argparser() -> argparse.ArgumentParser:
    """The flags shared by the reference's single-GPU and SLURM trainers (train_nsga2_base.py:43-163): names, types and defaults."""
    p = argparse.ArgumentParser(add_help=False)
    opt = p.add_argument
    opt("--image-generator", type=str, choices=list(GENERATOR_NAMES), help="image generator (model family) whose schedules are searched")
    opt("--load-from", type=Path, help="manager_config.json of an existing run; without it the other arguments describe a new run")
    opt("--name", type=str, help="name of the run (directory under the populations / benchmarks roots)")
    opt("--num-cycles", default="1", help="cycles to run; 'inf' runs until interrupted")
    opt("--print-not-submit", action="store_true", help="describe the evaluation of the current generation instead of running it (one cycle only)")
    opt("--all-populations-dir", type=Path, default=DEFAULT_POPULATIONS_DIR)
    opt("--all-benchmarks-dir", type=Path, default=DEFAULT_BENCHMARKS_DIR)
    opt("--embedding-dir", type=Path, help="prompt-embedding .pt files to generate from (seeded synthetic embeddings without it)")
    opt("--generation-num", type=int, default=None, help="generation to start from (default: highest on disk)")
    opt("--num-inference-steps", type=int, default=20)
    opt("--benchmark-prompts", type=Path, help="(ImageReward scorer only)")
    opt("--file-mode", choices=["json", "text", "txt", "tsv"], help="(ImageReward scorer only)")
    opt("--image-naming-mode", choices=["image_reward", "parti"], help="(ImageReward scorer only)")
    opt("--min-diff-from-default", type=int, default=1, help="constraint: bits that must differ from the all-recompute schedule")
    opt("--population-size", type=int, default=72)
    opt("--batch-size", type=int, default=100, help="prompts per generation call")
    opt("--num-images-per-prompt", type=int, default=10)
    opt("--exactly-n-images", type=int, default=1000, help="(ImageReward scorer only)")
    opt("--regen-if-not-n-images", type=int, default=1000, help="(ImageReward scorer only)")
    opt("--maximize-MACs", action="store_true", help="flip the sign of the MAC objective")
    return p


def validate_base_args(args: argparse.Namespace) -> None:
    if args.load_from is None and args.name is None:
        raise ValueError("Must provide a name or path to JSON file.")
    cycles = str(args.num_cycles)
    if cycles.isdigit():
        args.num_cycles = int(cycles)
    elif cycles != "inf":
        raise ValueError("num_cycles must be an integer or 'inf'.")
    if args.print_not_submit and args.num_cycles != 1:
        raise ValueError("print-not-submit can only be used with num_cycles=1.")


def parse_args(argv: list[str] | None = None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(description="NSGA-II over cache schedules on one GPU, evaluated in-process; resumable.", parents=[get_base_argparser()])
    parser.add_argument("--print-eval-outputs", action="store_true", help="print the per-candidate objectives of every evaluated generation")
    args = parser.parse_args(argv)
    validate_base_args(args)
    return args


def get_offline_eval_kwargs(args: argparse.Namespace) -> dict[str, Any]:
    keys = ("embedding_dir", "num_images_per_prompt", "exactly_n_images", "regen_if_not_n_images", "benchmark_prompts", "file_mode", "image_naming_mode")
    return {k: getattr(args, k) for k in keys}


# ------------------------------------------------------------------------------------------------
# manager / algorithm life cycle
# ------------------------------------------------------------------------------------------------
def _manager_type(image_generator: str | type):
    name = image_generator if isinstance(image_generator, str) else image_generator.__name__
    if name.startswith("PixArt"):
        return PixArtPopulationIOManager
    if name.startswith("Flux"):
        return FluxPopulationIOManager
    raise ValueError(f"Unsupported Image Generator type {name}.")


def initialize_manager(args: argparse.Namespace, image_generator: str | type):
    cls = _manager_type(image_generator)
    if args.load_from is not None:
        if not Path(args.load_from).exists():
            raise ValueError(f"JSON file {args.load_from} does not exist.")
        return cls.from_json(args.load_from)
    return cls(name=args.name, all_populations_dir=args.all_populations_dir, all_benchmarks_dir=args.all_benchmarks_dir, generation_num=args.generation_num,
               num_inference_steps=args.num_inference_steps, min_diff_from_default=args.min_diff_from_default, population_size=args.population_size,
               maximize_macs=getattr(args, "maximize_MACs", False))


def init_gen_0(manager, image_generator: str | type | None = None, allow_random: bool | None = None) -> NSGA2:
    """New search: generation 1 = the seed population under ``gen_000/candidates`` (reference train_nsga2_base.py:184-252).  Without one the
    reference asks on stdin whether to sample at random; so does this when stdin is a terminal (``allow_random`` overrides the question)."""
    seed = manager.load_population_vectors(0)
    if seed.size == 0:
        if allow_random is None:
            allow_random = sys.stdin.isatty() and input("No initial population found. Randomly seed the population? (y/N) ") == "y"
        if not allow_random:
            raise ValueError("No initial population found")
        seed = None
    n_var = len(manager.default_bits)
    algorithm = NSGA2(pop_size=manager.population_size, n_var=n_var, crossover_prob=0.9, n_points=4, mutation_prob=0.05, seed=0)
    algorithm.setup(None if seed is None else seed.astype(np.bool_))
    manager.generation_num = algorithm.n_gen  # the seed generation is 0 on disk, the algorithm starts counting at 1
    manager.save_population(algorithm.ask(), algorithm.n_gen)
    print(f"Candidates for generation {manager.generation_num} saved.")
    return algorithm


def train_one_cycle(manager, algorithm: NSGA2) -> bool:
    """tell + ask when the manager's generation has been evaluated (scores and MAC metrics on disk); False otherwise."""
    if not manager.check_offline_eval():
        print(f"Offline evaluation not found for generation {manager.generation_num}.")
        return False
    X, F, G = manager.ask()
    algorithm.tell(X, F, G)
    manager.generation_num = algorithm.n_gen
    manager.save_population(algorithm.ask())
    print(f"Candidates for generation {manager.generation_num} saved.")
    return True


def checkpoint_manager_and_algorithm(manager, algorithm: NSGA2) -> None:
    manager.save_algorithm(algorithm)
    manager.to_json()
    print(f"Checkpointed manager config and algorithm for generation {manager.generation_num}")


# ------------------------------------------------------------------------------------------------
# evaluation of one generation, in this process
# ------------------------------------------------------------------------------------------------
class InProcessEvaluator:
    """Scores every candidate of the manager's current generation and writes what the reference's three scripts would have left on disk.

    ``score_fn(index, bits) -> score`` (higher is better) is injected in tests; on a GPU it is built lazily from one resident engine."""

    def __init__(self, image_generator: str, embedding_dir: Path | None = None, batch_size: int = 100, num_images_per_prompt: int = 10,
                 score_fn: ScoreFn | None = None, verbose: bool = False):
        self.image_generator, self.embedding_dir = image_generator, embedding_dir
        self.batch_size, self.num_images_per_prompt = batch_size, num_images_per_prompt
        self.score_fn, self.verbose = score_fn, verbose

    def describe(self, manager) -> str:
        n = len(list(manager.get_pop_candidates_dir().glob("cand_*.json")))
        source = self.embedding_dir if self.embedding_dir is not None else "seeded synthetic embeddings"
        return (f"generation {manager.generation_num}: {n} candidate schedule(s) under {manager.get_pop_candidates_dir()}\n"
                f"  generate: {self.image_generator}, prompts from {source}, batch {self.batch_size}, {self.num_images_per_prompt} image(s) per prompt -> latents, in-process\n"
                f"  score:    latent fidelity to the uncached run (stand-in for ImageReward) -> {manager.get_benchmark_gen_dir()}/cand_XXX/scores.json\n"
                f"  metrics:  closed-form TMACs -> metrics.{manager.METRIC_KEY} in each candidate file")

    def _build_score_fn(self, manager) -> ScoreFn:
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("in-process evaluation needs a GPU: the hot path has no CPU fallback (use --print-not-submit to inspect the plan)")
        if isinstance(manager, FluxPopulationIOManager):
            return self._build_flux_score_fn(manager)
        from types import SimpleNamespace

        from ecad_amd.engine import EcadEngine
        from ecad_amd.genetic.evaluate import engine_score_fn
        from ecad_amd.pipelines.dpm_solver import DPMSolverMultistepScheduler
        from ecad_amd.transformer_2d_models.pixart_transformer_2d_edited import synthetic_state_dict

        prompts = min(self.batch_size, 16)  # one resident engine sized for the paper's bench batch
        T = 300 if "Sigma" in self.image_generator else 120
        engine = EcadEngine(num_layers=manager.num_blocks, max_batch=2 * prompts, max_text_tokens=T, dtype=torch.float16)
        engine.load_state_dict(synthetic_state_dict(SimpleNamespace(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2,
                                                                    num_layers=manager.num_blocks, caption_channels=4096), seed=0))
        g = torch.Generator().manual_seed(1)
        if self.embedding_dir is not None:
            from ecad_amd.dataset_utils.prompt_embedding_dataset import PromptEmbeddingDataset

            data = PromptEmbeddingDataset(self.embedding_dir)
            items = [data[i] for i in range(min(prompts, len(data)))]
            prompts = len(items)
            stack = lambda key: torch.stack([it[key] for it in items])  # noqa: E731  (the dataset squeezes: [T, C] and [T] per prompt)
            enc = torch.cat([stack("negative_prompt_embeds"), stack("prompt_embeds")]).float()
            mask = torch.cat([stack("negative_prompt_attention_mask"), stack("prompt_attention_mask")])
            T = enc.shape[1]
        else:
            enc = torch.cat([(torch.randn(1, T, 4096, generator=g) * 0.1).repeat(prompts, 1, 1), torch.randn(prompts, T, 4096, generator=g) * 0.1])
            mask = torch.ones(2 * prompts, T, dtype=torch.uint8)
        solver = DPMSolverMultistepScheduler()
        solver.set_timesteps(manager.num_inference_steps)
        ts, coefs = solver.timesteps.numpy().astype(np.float32), solver.coefficient_table()
        per_seed = []
        for k in range(self.num_images_per_prompt):  # one latent draw per image of a prompt, seeds as the generators step them
            lat = torch.randn(prompts, 4, 32, 32, generator=torch.Generator().manual_seed(k))
            per_seed.append(engine_score_fn(engine, enc.cuda(), mask.cuda(), lat.cuda(), ts, coefs))
        return lambda index, bits: float(np.mean([fn(index, bits) for fn in per_seed]))

    def _build_flux_score_fn(self, manager) -> ScoreFn:
        """`--image-generator FluxImageGenerator` (reference train_nsga2_single_gpu.py:73,115): a resident ``EcadFluxEngine`` of the manager's geometry scores
        every candidate in process.  The 19 + 38-block geometry is FLUX.1-dev's (24 heads, 4096-wide text, random-init weights of its shapes: there is no
        checkpoint offline); any other block count is a reduced test geometry (2 heads)."""
        import torch
        from types import SimpleNamespace

        from ecad_amd.flux_engine import EcadFluxEngine
        from ecad_amd.genetic.evaluate import flux_engine_score_fn
        from ecad_amd.pipelines.flow_match import flow_match_sigmas, model_scalars
        from ecad_amd.transformer_2d_models.flux_transformer_2d_edited import synthetic_state_dict

        nb, ns, steps = manager.num_blocks, manager.num_single_blocks, manager.num_inference_steps
        full = (nb, ns) == (19, 38)
        heads, jd, pd, T = (24, 4096, 768, 512) if full else (2, 128, 64, 32)
        gh, gw = manager.height // 16, manager.width // 16
        B = max(1, min(self.batch_size, 2))
        eng = EcadFluxEngine(num_layers=nb, num_single_layers=ns, num_heads=heads, joint_attention_dim=jd, pooled_projection_dim=pd, max_batch=B, max_image_tokens=gh * gw,
                             max_text_tokens=T, dtype=torch.bfloat16)
        cfg = SimpleNamespace(num_layers=nb, num_single_layers=ns, num_attention_heads=heads, attention_head_dim=128, in_channels=64, joint_attention_dim=jd,
                              pooled_projection_dim=pd, guidance_embeds=True)
        eng.load_state_dict(synthetic_state_dict(cfg, 0, device="cuda" if full else "cpu", dtype=torch.bfloat16 if full else torch.float32))
        sig = flow_match_sigmas(steps, gh * gw)
        tm, gm = model_scalars(sig, float(manager.guidance_scale), torch.bfloat16)
        per_seed = []
        for k in range(self.num_images_per_prompt):
            g = torch.Generator().manual_seed(1 + k)
            enc = (torch.randn(B, T, jd, generator=g) * 0.1).bfloat16().cuda()
            pooled = (torch.randn(B, pd, generator=g) * 0.5).bfloat16().cuda()
            lat = torch.randn(B, gh * gw, 64, generator=g).cuda()
            per_seed.append(flux_engine_score_fn(eng, enc, pooled, lat, gh, gw, tm, gm, sig, nb, ns))
        return lambda index, bits: float(np.mean([fn(index, bits) for fn in per_seed]))

    def __call__(self, manager) -> np.ndarray:
        if self.score_fn is None:
            self.score_fn = self._build_score_fn(manager)
        rows = []
        for index, schedule in manager.load_population_schedules():
            bits = schedule.to_numpy(flatten=True).astype(np.bool_)
            shaped = bits if isinstance(manager, FluxPopulationIOManager) else bits.reshape(manager.num_inference_steps, manager.num_blocks, manager.num_component_types)
            score = float(self.score_fn(index, shaped))  # stored as total_score; the managers read the objective back as 1 - total_score
            manager.write_evaluation(index, score, bits)
            rows.append((index, score))
            if self.verbose:
                print(f"  cand_{index:03d}: score {score:.5f}")
        return np.array(rows)


# ------------------------------------------------------------------------------------------------
# the loop
# ------------------------------------------------------------------------------------------------
def train_nsga2_single_gpu(manager, algorithm: NSGA2, evaluator: InProcessEvaluator, num_cycles: int | None, print_not_submit: bool = True) -> None:
    while num_cycles is None or num_cycles > 0:
        advanced = train_one_cycle(manager, algorithm)
        checkpoint_manager_and_algorithm(manager, algorithm)
        if print_not_submit:
            print(evaluator.describe(manager))
        else:
            evaluator(manager)  # the generation on disk: the one just written if the step above advanced, else the one waiting for its scores
            if not advanced:
                if not train_one_cycle(manager, algorithm):
                    raise ValueError("Offline evaluation not found.")
                checkpoint_manager_and_algorithm(manager, algorithm)
        if num_cycles is not None:
            num_cycles -= 1


def run(args: argparse.Namespace, score_fn: ScoreFn | None = None, allow_random_seed_population: bool | None = None):
    manager = initialize_manager(args, args.image_generator)
    try:
        print("Attempting to load from checkpoint...")
        algorithm = manager.load_algorithm()
    except FileNotFoundError as err:
        print(f"Checkpoint not found: {err}")
        algorithm = init_gen_0(manager, args.image_generator, allow_random_seed_population)
    evaluator = InProcessEvaluator(args.image_generator, args.embedding_dir, args.batch_size, args.num_images_per_prompt, score_fn=score_fn,
                                   verbose=getattr(args, "print_eval_outputs", False))
    train_nsga2_single_gpu(manager, algorithm, evaluator, None if args.num_cycles == "inf" else args.num_cycles, args.print_not_submit)
    return manager, algorithm


def main() -> None:
    run(parse_args())


if __name__ == "__main__":
    main()
