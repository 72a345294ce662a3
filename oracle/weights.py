"""CPU restatement of FP-scaled weight dequantisation — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows /root/reference/apps/api/src/quantize/scaled_layer.py: `fp8_activation_dequant` (:154-167,
`qdq.to(dtype) * scale.to(dtype)`) for float8 weights and `_scale_and_cast_weight` (:496-549) for the scalar /
per-out-feature broadcast.  PINNED: tests/golden/fp_scaled.pt holds the outputs of those two reference functions
(run here by tests/golden/make_golden.py) on seeded fp8 weights and scales.
"""
from __future__ import annotations

import torch


def dequant(weight: torch.Tensor, scale_weight: torch.Tensor, dtype: torch.dtype = torch.bfloat16,
            per_out_feature_dim: int = 0) -> torch.Tensor:
    if weight.dtype not in (torch.float8_e4m3fn, torch.float8_e5m2):
        # the reference raises for a scaled non-fp8 weight (`physical_dtype in (torch.uint8)`, :525)
        raise TypeError(f"scale_weight with a {weight.dtype} weight")
    # fp8 path: plain broadcasting (:165-167); a per-row scale is stored [out, 1] (or is a scalar)
    return weight.to(dtype) * scale_weight.to(dtype)
