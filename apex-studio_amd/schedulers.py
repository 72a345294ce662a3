"""Sampler-side scheduler steps.  These stay in Python (BASELINE north_star: "the sampler loop and
latent-preview callback stay in Python"); device-agnostic torch ops like the reference's.

FlowMatchEulerDiscreteScheduler mirrors the diffusers class the Flux / QwenImage manifests name
(reference manifest/image/flux-dev-text-to-image-1.0.0.v1.yml:45; semantics: SURVEY.md App. A; the
in-tree sibling is scheduler/flow.py:293-355): `set_timesteps(sigmas=, mu=)`, `.timesteps`,
`.sigmas`, `.step(model_output, timestep, sample, return_dict=False)`, `.order`, `set_begin_index`.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0,
                 use_dynamic_shifting: bool = False, base_shift: float = 0.5, max_shift: float = 1.15,
                 base_image_seq_len: int = 256, max_image_seq_len: int = 4096,
                 shift_terminal: Optional[float] = None, time_shift_type: str = "exponential"):
        self.config = dict(num_train_timesteps=num_train_timesteps, shift=shift,
                           use_dynamic_shifting=use_dynamic_shifting, base_shift=base_shift,
                           max_shift=max_shift, base_image_seq_len=base_image_seq_len,
                           max_image_seq_len=max_image_seq_len, shift_terminal=shift_terminal,
                           time_shift_type=time_shift_type)
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        # diffusers keeps the ends of the TRAINING schedule, already shifted when the shift is static; set_timesteps
        # without explicit sigmas spaces between them (and then applies the shift again — upstream behaviour)
        tr = torch.linspace(1, num_train_timesteps, num_train_timesteps, dtype=torch.float64).flip(0) / num_train_timesteps
        if not use_dynamic_shifting:
            tr = shift * tr / (1 + (shift - 1) * tr)
        self.sigma_max, self.sigma_min = float(tr[0]), float(tr[-1])
        self.timesteps = None
        self.sigmas = None
        self._step_index = None
        self._begin_index = None

    @classmethod
    def flux_dev(cls):
        """FLUX.1-dev scheduler_config.json values."""
        return cls(shift=3.0, use_dynamic_shifting=True)

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def _time_shift(self, mu: float, sigma: float, t: torch.Tensor) -> torch.Tensor:
        if self.config["time_shift_type"] == "exponential":
            return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)
        return mu / (mu + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None,
                      sigmas: Optional[Sequence[float]] = None, mu: Optional[float] = None):
        n = self.num_train_timesteps
        if sigmas is None:
            sigmas = torch.linspace(self.sigma_max * n, self.sigma_min * n, num_inference_steps, dtype=torch.float64) / n
        else:
            sigmas = torch.as_tensor(list(sigmas) if not torch.is_tensor(sigmas) else sigmas,
                                     dtype=torch.float64)
        if self.config["use_dynamic_shifting"]:
            if mu is None:
                raise ValueError("`mu` must be passed when use_dynamic_shifting is True")
            sigmas = self._time_shift(mu, 1.0, sigmas)
        else:
            sigmas = self.shift * sigmas / (1 + (self.shift - 1) * sigmas)
        if self.config["shift_terminal"]:
            one_minus = 1 - sigmas
            sigmas = 1 - one_minus / (one_minus[-1] / (1 - self.config["shift_terminal"]))
        sigmas = sigmas.to(torch.float32)
        self.timesteps = (sigmas * n).to(device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1)]).to(device)
        self._step_index = None
        return self.timesteps

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True):
        if self._step_index is None:
            self._step_index = self._begin_index if self._begin_index is not None else 0
        i = self._step_index
        dt = self.sigmas[i + 1] - self.sigmas[i]
        prev = sample.to(torch.float32) + dt * model_output.to(torch.float32)
        self._step_index += 1
        prev = prev.to(model_output.dtype)
        return (prev,) if not return_dict else {"prev_sample": prev}


class UniPCMultistepScheduler:
    """Flow-matching UniPC (B(h) form, `bh2`, predict-x0) as Wan 2.x configures it.

    Mirrors the in-tree scheduler the reference ships next to the diffusers one
    (apps/api/src/scheduler/unipc.py: __init__ :60-127, set_timesteps :159-219, convert_model_output
    :278-346, predictor :348-482, corrector :484-622, step :651-737) — same method surface
    (`set_timesteps`, `.timesteps`, `.sigmas`, `step(model_output, timestep, sample, return_dict)`,
    `.order`), restated compactly: with alpha = 1 - sigma, lambda = log(alpha / sigma), h = lambda_t -
    lambda_s, the update is  x_t = (sigma_t / sigma_s) x - alpha_t (e^{-h} - 1) (m0 + rho-weighted
    differences), where the rhos solve the small Vandermonde system R rho = b of UniPC.
    Latents stay in float32 as in the reference engine (engine/wan/shared/__init__.py:569-571)."""

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2,
                 prediction_type: str = "flow_prediction", shift: float = 1.0, predict_x0: bool = True,
                 solver_type: str = "bh2", lower_order_final: bool = True, final_sigmas_type: str = "zero",
                 use_dynamic_shifting: bool = False, disable_corrector=()):
        if prediction_type != "flow_prediction" or not predict_x0 or solver_type != "bh2":
            raise NotImplementedError("only the flow_prediction / predict_x0 / bh2 configuration is on the hot path")
        self.config = dict(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                           prediction_type=prediction_type, shift=shift, predict_x0=predict_x0,
                           solver_type=solver_type, lower_order_final=lower_order_final,
                           final_sigmas_type=final_sigmas_type, use_dynamic_shifting=use_dynamic_shifting)
        self.order = solver_order
        self.disable_corrector = list(disable_corrector)
        n = num_train_timesteps
        s = 1.0 - torch.linspace(1.0, 1.0 / n, n, dtype=torch.float64).flip(0)
        if not use_dynamic_shifting:
            s = shift * s / (1 + (shift - 1) * s)
        s = s.to(torch.float32)
        self.sigma_min, self.sigma_max = float(s[-1]), float(s[0])
        self.timesteps = None
        self.sigmas = None
        self._step_index = None
        self._begin_index = None

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas=None,
                      mu: Optional[float] = None, shift: Optional[float] = None):
        cfg = self.config
        if sigmas is None:
            sigmas = torch.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1,
                                    dtype=torch.float64)[:-1]
        else:
            sigmas = torch.as_tensor(sigmas, dtype=torch.float64)
        if cfg["use_dynamic_shifting"]:
            if mu is None:
                raise ValueError("`mu` must be passed when use_dynamic_shifting is True")
            sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1))
        else:
            sh = cfg["shift"] if shift is None else shift
            sigmas = sh * sigmas / (1 + (sh - 1) * sigmas)
        if cfg["final_sigmas_type"] != "zero":
            raise NotImplementedError("final_sigmas_type must be 'zero'")
        self.timesteps = (sigmas * cfg["num_train_timesteps"]).to(torch.int64).to(device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1, dtype=torch.float64)]).to(torch.float32)  # stays on CPU
        self.num_inference_steps = len(self.timesteps)
        self._m = [None] * self.order          # converted model outputs (x0 predictions), newest last
        self._lower_order_nums = 0
        self._last_sample = None
        self._this_order = 1
        self._step_index = None
        return self.timesteps

    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _coeffs(self, h, rks, order):
        """rho vector of UniPC's B(h) update for the given r_k (last r is 1)."""
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        fact = 1
        R, b = [], []
        r = torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in rks])
        for i in range(1, order + 1):
            R.append(torch.pow(r, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return torch.stack(R), torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in b]), h_phi_1, B_h

    def _update(self, x, m0, older, s_from, s_to, order, corrector_m=None):
        """x at sigma[s_from] -> sigma[s_to]; `older` = earlier x0 predictions (newest first) with the sigma
        index each was taken at; corrector_m = x0 prediction at the target (corrector only)."""
        sig = self.sigmas
        sigma_t, sigma_s = sig[s_to], sig[s_from]
        alpha_t = 1 - sigma_t
        lam_t, lam_s = self._lam(sigma_t), self._lam(sigma_s)
        h = lam_t - lam_s
        rks, D1s = [], []
        for (mi, si) in older[:order - 1]:
            rk = (self._lam(sig[si]) - lam_s) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        R, b, h_phi_1, B_h = self._coeffs(h, rks, order)
        x_t = (sigma_t / sigma_s) * x - alpha_t * h_phi_1 * m0
        if corrector_m is None:                       # predictor (UniP)
            if D1s:
                rhos = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
                res = sum(float(rh) * d for rh, d in zip(rhos, D1s))
                x_t = x_t - alpha_t * B_h * res
        else:                                         # corrector (UniC)
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
            res = sum(float(rh) * d for rh, d in zip(rhos[:-1], D1s)) if D1s else 0
            x_t = x_t - alpha_t * B_h * (res + float(rhos[-1]) * (corrector_m - m0))
        return x_t.to(x.dtype)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True,
             generator=None):
        if self._step_index is None:
            self._step_index = self._begin_index if self._begin_index is not None else 0
        i = self._step_index
        x0 = sample - self.sigmas[i] * model_output      # flow prediction -> x0 (convert_model_output)
        if i > 0 and (i - 1) not in self.disable_corrector and self._last_sample is not None:
            older = [(m, i - 1 - (k + 1)) for k, m in enumerate(reversed(self._m[:-1])) if m is not None]
            sample = self._update(self._last_sample, self._m[-1], older, i - 1, i, self._this_order,
                                  corrector_m=x0)
        self._m = self._m[1:] + [x0]
        order = min(self.order, len(self.timesteps) - i) if self.config["lower_order_final"] else self.order
        self._this_order = min(order, self._lower_order_nums + 1)
        self._last_sample = sample
        older = [(m, i - (k + 1)) for k, m in enumerate(reversed(self._m[:-1])) if m is not None]
        prev = self._update(sample, x0, older, i, i + 1, self._this_order)
        if self._lower_order_nums < self.order:
            self._lower_order_nums += 1
        self._step_index += 1
        return (prev,) if not return_dict else {"prev_sample": prev}
