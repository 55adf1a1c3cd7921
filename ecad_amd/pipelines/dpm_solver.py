"""DPM-Solver++(2M) host side: timesteps, sigmas and the per-step update coefficients.

Product code (the oracle has its own restatement in ``oracle/dpm_solver_ref.py``).  Matches the
scheduler the reference's pipelines instantiate -- diffusers ``DPMSolverMultistepScheduler`` with the
PixArt scheduler_config [ext]: linear betas 1e-4..2e-2 over 1000 train steps, solver_order 2,
``dpmsolver++``, epsilon prediction, midpoint, linspace spacing, final sigma 0 -- reached through
``self.scheduler.step`` in the loop mirrored at ``ecad/pipelines/pass_through.py:357-370``.

The update is affine in (x, x0_i, x0_{i-1}); the engine takes it as 5 floats per step
(``ecad_run_schedule``), the per-step Python path applies the same numbers with torch ops.
"""
from __future__ import annotations

import math

import numpy as np
import torch


class DPMSolverMultistepScheduler:
    order = 1  # pipeline-loop "order" (one model call per step)
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02, solver_order: int = 2):
        if solver_order != 2:
            raise ValueError("only solver_order=2 (DPM-Solver++ 2M) is implemented")
        self.num_train_timesteps = num_train_timesteps
        betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0, dtype=np.float32)
        self.num_inference_steps = None
        self.timesteps = None
        self.sigmas = None
        self._reset_state()

    def _reset_state(self):
        self.step_index = 0
        self.lower_order_nums = 0
        self._x0_prev = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.num_train_timesteps
        ts = np.linspace(0, n - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        all_sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        sig = np.interp(ts, np.arange(0, n), all_sigmas)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)
        self.num_inference_steps = num_inference_steps
        self._reset_state()

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma: float):
        alpha = 1.0 / math.sqrt(sigma * sigma + 1.0)
        return alpha, sigma * alpha

    def step_coefficients(self, i: int, order: int):
        """(alpha_s, sigma_s, c_x, c_m0, c_m1): x0 = (x - sigma_s*eps)/alpha_s; x' = c_x*x + c_m0*x0 + c_m1*x0_prev."""
        a_t, s_t = self._alpha_sigma(float(self.sigmas[i + 1]))
        a_s, s_s = self._alpha_sigma(float(self.sigmas[i]))
        lam_t = math.inf if s_t == 0.0 else math.log(a_t) - math.log(s_t)
        lam_s = math.log(a_s) - math.log(s_s)
        h = lam_t - lam_s
        a = -a_t * (math.exp(-h) - 1.0)
        c_x = s_t / s_s
        if order == 1:
            return a_s, s_s, c_x, a, 0.0
        a_p, s_p = self._alpha_sigma(float(self.sigmas[i - 1]))
        lam_p = math.log(a_p) - math.log(s_p)
        r0 = (lam_s - lam_p) / h
        return a_s, s_s, c_x, a + 0.5 * a / r0, -0.5 * a / r0

    def order_at(self, i: int) -> int:
        """First step and (final sigma == 0) last step are first order, the rest second order."""
        return 1 if (i == 0 or i == self.num_inference_steps - 1) else 2

    def coefficient_table(self) -> np.ndarray:
        return np.array([self.step_coefficients(i, self.order_at(i)) for i in range(self.num_inference_steps)], dtype=np.float32)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = False, **kw):
        i = self.step_index
        order = 1 if (self.lower_order_nums < 1 or i == self.num_inference_steps - 1) else 2
        a_s, s_s, c_x, c0, c1 = self.step_coefficients(i, order)
        x = sample.float()
        x0 = (x - s_s * model_output.float()) / a_s
        nxt = c_x * x + c0 * x0
        if order == 2:
            nxt = nxt + c1 * self._x0_prev
        self._x0_prev = x0
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.step_index += 1
        return (nxt.to(model_output.dtype),)
