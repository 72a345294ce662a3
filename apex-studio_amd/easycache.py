"""EasyCache step skipping for the Wan transformer — host-side control flow around the HIP forward.

Mirrors the reference's `enable_easy_cache(num_steps, thresh, ret_steps, should_reset_global_cache)` / `disable_easy_cache()` and
the rule of `easycache_forward_` (R/src/transformer/wan/base/model.py:202-520, :1645-1680): calls alternate conditional (even
count) / unconditional (odd); on even calls the input change since the previous even call, scaled by the measured output / input
change rate K and the previous output's magnitude, is accumulated; while the sum stays under `thresh` the PAIR is served from the
cache (`raw_input + (last computed output - its input)`), except during the first `ret_steps` pairs and the last pair, which are
always computed.  The reference keeps this state in module globals shared by whichever transformer is loaded and resets it when it
enables the cache on a newly loaded expert; here the state belongs to the model instance and `enable_easy_cache(...,
should_reset_global_cache=True)` resets it, so the engine enables it on an expert each time that expert takes over (engine_wan.py).

The reductions are a handful of elementwise torch ops over one latent (a few MB) and one host read per conditional call — noise
beside a 5 s transformer forward; the forward itself is the unchanged HIP path.  Outputs are float32, as the reference returns."""
from __future__ import annotations

from typing import Callable, Optional

import torch


class EasyCache:
    def __init__(self, num_steps: int, thresh: float, ret_steps: int = 10):
        self.num_steps = int(num_steps) * 2            # cond / uncond pairs
        self.thresh = float(thresh)
        self.ret_steps = int(ret_steps) * 2
        self.reset()

    def reset(self):
        self.cnt = 0
        self.accumulated = 0.0
        self.should_calc = True
        self.k: Optional[float] = None
        self.prev_in_even = self.prev_out_even = self.prev_out_odd = self.prev_prev_in_even = None
        self.cache_even = self.cache_odd = None
        self.computed = []                            # per call: did the transformer run? (diagnostics / tests)

    @staticmethod
    def _mean_abs(t: torch.Tensor) -> float:
        return float(t.float().abs().mean())

    @torch.no_grad()
    def __call__(self, hidden_states: torch.Tensor, out_channels: int, forward: Callable[[], torch.Tensor]) -> torch.Tensor:
        raw_input = hidden_states[:, :out_channels].float().clone()
        even = self.cnt % 2 == 0
        if even:
            if self.cnt < self.ret_steps or self.cnt >= self.num_steps - 2:
                self.should_calc, self.accumulated = True, 0.0
            elif self.prev_in_even is not None and self.prev_out_even is not None and self.k is not None:
                change = self._mean_abs(raw_input - self.prev_in_even)
                self.accumulated += self.k * (change / self._mean_abs(self.prev_out_even))
                if self.accumulated < self.thresh:
                    self.should_calc = False
                else:
                    self.should_calc, self.accumulated = True, 0.0
            else:
                self.should_calc = True
            self.prev_in_even = raw_input
        cache, prev_out = (self.cache_even, self.prev_out_even) if even else (self.cache_odd, self.prev_out_odd)
        if not self.should_calc and prev_out is not None:
            self.cnt += 1
            self.computed.append(False)
            return raw_input + cache
        output = forward().float()
        if even:
            if self.prev_out_even is not None and self.prev_prev_in_even is not None:
                self.k = self._mean_abs(output - self.prev_out_even) / self._mean_abs(self.prev_in_even - self.prev_prev_in_even)
            self.prev_prev_in_even = self.prev_in_even
            self.prev_out_even = output
            self.cache_even = output - raw_input
        else:
            self.prev_out_odd = output
            self.cache_odd = output - raw_input
        self.cnt += 1
        self.computed.append(True)
        return output
