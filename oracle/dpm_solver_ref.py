"""Oracle: DPM-Solver++(2M) as the PixArt pipelines configure it.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Restates diffusers==0.30.3 ``DPMSolverMultistepScheduler`` [ext] with the
PixArt-alpha scheduler_config (beta linear 1e-4..2e-2, 1000 train steps,
solver_order 2, dpmsolver++, epsilon prediction, midpoint, linspace spacing,
lower_order_final, final_sigmas_type "zero", no thresholding / karras).
The reference reaches it through ``self.scheduler.step`` in the loop mirrored at
``ecad/pipelines/pass_through.py:357-370``.
"""
from __future__ import annotations

import numpy as np


class DPMSolverPP2M:
    def __init__(self, num_inference_steps: int, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02):
        betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
        alphas_cumprod = np.cumprod((1.0 - betas).astype(np.float32), axis=0)
        self.num_inference_steps = num_inference_steps
        # set_timesteps, timestep_spacing == "linspace", lambda_min_clipped = -inf
        ts = np.linspace(0, num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig_all = np.array(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5)
        sigmas = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
        self.sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)  # final_sigmas_type == "zero"
        self.timesteps = ts
        self.reset()

    def reset(self) -> None:
        self.step_index = 0
        self.lower_order_nums = 0
        self.model_outputs: list = [None, None]

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1.0 / np.sqrt(sigma * sigma + 1.0)
        return alpha_t, sigma * alpha_t

    def coefficients(self, i: int, order: int):
        """x_next = c_x * x + c_m0 * x0_i + c_m1 * x0_{i-1}; with x0 = (x - sigma_t*eps)/alpha_t.

        Returns (alpha_s, sigma_s, c_x, c_m0, c_m1) as float64.
        """
        with np.errstate(divide="ignore"):
            s_t, s_s0 = np.float64(self.sigmas[i + 1]), np.float64(self.sigmas[i])
            a_t, sg_t = self._alpha_sigma(s_t)
            a_s0, sg_s0 = self._alpha_sigma(s_s0)
            lam_t = np.log(a_t) - np.log(sg_t)
            lam_s0 = np.log(a_s0) - np.log(sg_s0)
            h = lam_t - lam_s0
            a = -a_t * (np.exp(-h) - 1.0)
            c_x = sg_t / sg_s0
            if order == 1:
                return a_s0, sg_s0, c_x, a, 0.0
            s_s1 = np.float64(self.sigmas[i - 1])
            a_s1, sg_s1 = self._alpha_sigma(s_s1)
            lam_s1 = np.log(a_s1) - np.log(sg_s1)
            r0 = (lam_s0 - lam_s1) / h
            return a_s0, sg_s0, c_x, a + 0.5 * a / r0, -0.5 * a / r0

    def order_for_step(self, i: int) -> int:
        """Which update ``step`` picks at index i for a fresh run (solver_order 2)."""
        last = i == self.num_inference_steps - 1  # final_sigmas_type == "zero" => lower_order_final
        if i == 0 or last:
            return 1
        return 2

    def step(self, eps, sample):
        """eps, sample: float arrays/tensors of the same shape; returns next sample."""
        i = self.step_index
        order = 1 if (self.lower_order_nums < 1 or i == self.num_inference_steps - 1) else 2
        a_s, sg_s, c_x, c0, c1 = self.coefficients(i, order)
        x0 = (sample - float(sg_s) * eps) / float(a_s)
        self.model_outputs = [self.model_outputs[1], x0]
        nxt = float(c_x) * sample + float(c0) * x0
        if order == 2:
            nxt = nxt + float(c1) * self.model_outputs[0]
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.step_index += 1
        return nxt
