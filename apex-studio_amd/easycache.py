"""EasyCache step skipping for the Wan transformer — host-side control flow around the HIP forward.

Mirrors the reference's `enable_easy_cache(num_steps, thresh, ret_steps, should_reset_global_cache)` / `disable_easy_cache()` and
the rule of `easycache_forward_` (R/src/transformer/wan/base/model.py:202-520, :1645-1680): calls alternate conditional (even
count) / unconditional (odd); on even calls the input change since the previous even call, scaled by the measured output / input
change rate K and the previous output's magnitude, is accumulated; while the sum stays under `thresh` the PAIR is served from the
cache (`raw_input + (last computed output - its input)`), except during the first `ret_steps` pairs and the last pair, which are
always computed.  The reference keeps this state in module globals shared by whichever transformer is loaded: it resets them when it
enables the cache on the high-noise expert and does NOT when it enables it on the low-noise expert (`should_reset_global_cache=False`,
R/src/engine/wan/shared/__init__.py:372-381 vs :435-444), so the call count, K, the accumulated error and the caches carry across the
expert switch and the last pair of the clip is always computed.  Here the state is an object: `enable_easy_cache(...,
should_reset_global_cache=True)` makes a fresh one, `share_easy_cache_state(other)` + `enable_easy_cache(..., False)` continues
another model's (engine_wan.py does that at the switch) — per clip, not per process, so two engines in one process do not collide.

Dtypes follow the reference: `raw_input`, the caches and the change statistics stay in the dtypes torch gives them from
`hidden_states` and the model's output (bf16 in production), only the returned tensor is cast to float32 — the skip decisions
are then taken on the same rounded statistics as the reference's.  The reductions are a handful of elementwise torch ops over one
latent (a few MB) and one host read per conditional call — noise beside a 5 s transformer forward; the forward itself is the
unchanged HIP path."""
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
        self.k: Optional[torch.Tensor] = None
        self.prev_in_even = self.prev_out_even = self.prev_out_odd = self.prev_prev_in_even = None
        self.cache_even = self.cache_odd = None
        self.computed = []                            # per call: did the transformer run? (diagnostics / tests)

    @staticmethod
    def _mean_abs(t: torch.Tensor) -> torch.Tensor:
        return t.flatten().abs().mean()

    @torch.no_grad()
    def __call__(self, hidden_states: torch.Tensor, out_channels: int, forward: Callable[[], torch.Tensor]) -> torch.Tensor:
        raw_input = hidden_states[:, :out_channels].clone()
        even = self.cnt % 2 == 0
        if even:
            if self.cnt < self.ret_steps or self.cnt >= self.num_steps - 2:
                self.should_calc, self.accumulated = True, 0.0
            elif self.prev_in_even is not None and self.prev_out_even is not None and self.k is not None:
                change = self._mean_abs(raw_input - self.prev_in_even)
                self.accumulated = self.accumulated + self.k * (change / self._mean_abs(self.prev_out_even))
                if bool(self.accumulated < self.thresh):              # the one host read of a conditional call
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
            return (raw_input + cache).float()
        output = forward()
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
        return output.float()
