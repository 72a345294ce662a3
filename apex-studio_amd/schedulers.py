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
            smax, smin = 1.0, 1.0 / n
            sigmas = torch.linspace(smax * n, smin * n, num_inference_steps, dtype=torch.float64) / n
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
