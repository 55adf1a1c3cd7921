"""FlowMatchEulerDiscreteScheduler with dynamic shifting, as FluxPipeline drives it [ext] (the pipeline the reference calls at
``ecad/image_generators/flux_image_generator.py:330-360``): host-side constants only -- the update ``x <- x + (sigma[i+1] -
sigma[i]) * v`` itself runs inside the engine's captured graph (``ecad_flux_run_schedule``)."""
from __future__ import annotations

import json
import math
from pathlib import Path

import numpy as np
import torch


# FLUX.1-dev's scheduler/scheduler_config.json [ext]: what diffusers 0.30.3's FluxPipeline hands to calculate_shift
# (`scheduler.config.base_image_seq_len, max_image_seq_len, base_shift, max_shift`).  NB max_shift is 1.15 there; 1.16 is only the
# never-used default of the pipeline's helper.
FLUX_DEV_SCHEDULER_CONFIG = {"base_image_seq_len": 256, "max_image_seq_len": 4096, "base_shift": 0.5, "max_shift": 1.15}


def load_scheduler_config(weights_dir=None) -> dict:
    """`<weights_dir>/scheduler/scheduler_config.json` when a checkpoint directory is present (its values win), else FLUX.1-dev's."""
    cfg = dict(FLUX_DEV_SCHEDULER_CONFIG)
    if weights_dir is not None:
        f = Path(weights_dir) / "scheduler" / "scheduler_config.json"
        if f.is_file():
            cfg.update({k: v for k, v in json.load(open(f)).items() if k in cfg})
    return cfg


def calculate_shift(image_seq_len: int, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5, max_shift: float = 1.15) -> float:
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


def flow_match_sigmas(num_steps: int, image_seq_len: int, scheduler_config: dict | None = None) -> np.ndarray:
    """sigmas[0..num_steps], last = 0: linspace(1, 1/n, n) passed through time_shift(mu, 1, .)."""
    sig = np.linspace(1.0, 1.0 / num_steps, num_steps)
    c = scheduler_config or FLUX_DEV_SCHEDULER_CONFIG
    mu = calculate_shift(image_seq_len, c["base_image_seq_len"], c["max_image_seq_len"], c["base_shift"], c["max_shift"])
    sig = math.exp(mu) / (math.exp(mu) + (1.0 / sig - 1.0))
    return np.concatenate([sig, [0.0]]).astype(np.float32)


def model_scalars(sigmas: np.ndarray, guidance_scale: float, dtype: torch.dtype):
    """(timesteps_x1000 [steps] float32 numpy, guidance_x1000 float): what the transformer sees.  FluxPipeline [ext] casts the timestep to the
    latents' dtype first, divides by 1000 in that dtype, and the model multiplies by 1000 in that dtype (flux_transformer_2d_edited.py:270-274)."""
    t = torch.from_numpy(sigmas[:-1].copy()) * 1000.0
    t_in = t.to(dtype) / 1000.0
    g = torch.tensor([guidance_scale], dtype=torch.float32).to(dtype)
    return (t_in * 1000).float().numpy(), float((g * 1000).float()[0])
