"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's EasyCache step skipping for the Wan transformer.

Follows `easycache_forward_` (R/src/transformer/wan/base/model.py:202-520) and `enable_easy_cache` / `reset_wantf_global_cache`
(:1645-1680, :170-200): the reference keeps the state in module-level globals; here it is an object.  Calls alternate
conditional (even count) / unconditional (odd count); the decision is taken on even calls only and covers the pair:

  raw_input  = hidden_states[:, :out_channels]                                                      (:246)
  always compute while cnt < 2 ret_steps or cnt >= 2 num_steps - 2                                   (:254-259)
  else, once a previous input / output pair and the rate K exist:
      pred_change = K * mean|raw_input - prev_input_even| / mean|prev_output_even|                  (:265-277)
      accumulated += pred_change;  skip the pair while accumulated < thresh, else compute and reset (:278-283)
  a skipped call returns (raw_input + cache_{even|odd}).float(),  cache = output - raw_input of the last computed call  (:292-310, :501-505)
  after a computed EVEN call:  K = mean|output - prev_output_even| / mean|prev_input_even - prev_prev_input_even|      (:483-499)

Pinned by tests/golden/wan_easycache.pt (the reference function itself run here on the tiny Wan model, float64)."""
from __future__ import annotations

from typing import Callable, Optional

import torch


class EasyCacheState:
    def __init__(self, num_steps: int, thresh: float, ret_steps: int = 10):
        self.num_steps = num_steps * 2            # cond / uncond pairs
        self.thresh = thresh
        self.ret_steps = ret_steps * 2
        self.cnt = 0
        self.accumulated = 0.0
        self.should_calc = True
        self.k: Optional[torch.Tensor] = None
        self.prev_in_even = self.prev_out_even = self.prev_out_odd = self.prev_prev_in_even = None
        self.cache_even = self.cache_odd = None


def easycache_forward(st: EasyCacheState, forward: Callable[[], torch.Tensor], hidden_states: torch.Tensor, out_channels: int):
    """One call of the wrapped model.  `forward()` runs the transformer on the current arguments and returns its output
    [B, out_channels, F, H, W]; returns (output as float32, computed?)."""
    raw_input = hidden_states[:, :out_channels].clone()
    is_even = st.cnt % 2 == 0
    if is_even:
        if st.cnt < st.ret_steps or st.cnt >= st.num_steps - 2:
            st.should_calc = True
            st.accumulated = 0.0
        elif st.prev_in_even is not None and st.prev_out_even is not None:
            change = (raw_input - st.prev_in_even).flatten().abs().mean()
            if st.k is not None:
                st.accumulated = st.accumulated + st.k * (change / st.prev_out_even.flatten().abs().mean())
                if bool(st.accumulated < st.thresh):
                    st.should_calc = False
                else:
                    st.should_calc = True
                    st.accumulated = 0.0
            else:
                st.should_calc = True
        else:
            st.should_calc = True
        st.prev_in_even = raw_input.clone()
    if is_even and not st.should_calc and st.prev_out_even is not None:
        st.cnt += 1
        return (raw_input + st.cache_even).float(), False
    if not is_even and not st.should_calc and st.prev_out_odd is not None:
        st.cnt += 1
        return (raw_input + st.cache_odd).float(), False
    output = forward()
    if is_even:
        if st.prev_out_even is not None:
            out_change = (output - st.prev_out_even).flatten().abs().mean()
            if st.prev_prev_in_even is not None:
                st.k = out_change / (st.prev_in_even - st.prev_prev_in_even).flatten().abs().mean()
        st.prev_prev_in_even = st.prev_in_even
        st.prev_out_even = output.clone()
        st.cache_even = output - raw_input
    else:
        st.prev_out_odd = output.clone()
        st.cache_odd = output - raw_input
    st.cnt += 1
    return output.float(), True
