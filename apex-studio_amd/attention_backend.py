"""The "hip_mfma" attention backend — B-op plug-in point.

Honours the calling convention of every backend in the reference's attention_register
(apps/api/src/attention/functions.py:84, e.g. `sdpa` :338-377):

    fn(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, softmax_scale=None, **kwargs) -> Tensor

q:[B,H,Sq,D], k,v:[B,H,Sk,D] (possibly permuted, non-contiguous views); returns [B,H,Sq,D] in q's
dtype without aliasing or modifying the inputs; enqueues on torch's current stream; no host sync.
Masks, dropout and causal attention are not on the hot path: they raise, they do
not fall back.
"""
from __future__ import annotations

import torch

from . import ops
from .lib import ApexMIError

KEY = "hip_mfma"


def hip_mfma(q, k, v, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False,
             softmax_scale=None, **kwargs):
    if attn_mask is not None:
        raise ApexMIError("hip_mfma: attn_mask is not supported (no hot-path call site passes one)")
    if dropout_p:
        raise ApexMIError("hip_mfma: dropout is not supported (inference only)")
    if is_causal:
        raise ApexMIError("hip_mfma: causal attention is not supported")
    return ops.attention(q, k, v, softmax_scale)


def available() -> bool:
    """True when the HIP library is built and a ROCm device is visible."""
    try:
        from . import lib
        lib.load()
    except Exception:
        return False
    return torch.cuda.is_available()


def register(attention_register, set_default: bool = False, overwrite: bool = True):
    """Register under KEY in the given FunctionRegister (the reference's, or register.attention_register)."""
    attention_register(KEY, overwrite=overwrite, available=available())(hip_mfma)
    if set_default:
        attention_register.set_default(KEY)
    return attention_register


def register_models(transformers_registry, vae_registry=None):
    """Register the drop-in component classes (B-model) next to "flux.base" / "wan.base" /
    "qwenimage.base" / "hunyuanvideo15.base" (reference transformer/base.py:3; auto-scan transformer/__init__.py:21-84)."""
    from .flux import FluxTransformer2DModel
    from .hunyuan15 import HunyuanVideo15Transformer3DModel
    from .qwenimage import QwenImageTransformer2DModel
    from .wan import WanTransformer3DModel
    ok = available()
    transformers_registry("flux.mi355", overwrite=True, available=ok)(FluxTransformer2DModel)
    transformers_registry("wan.mi355", overwrite=True, available=ok)(WanTransformer3DModel)
    transformers_registry("qwenimage.mi355", overwrite=True, available=ok)(QwenImageTransformer2DModel)
    transformers_registry("hunyuanvideo15.mi355", overwrite=True, available=ok)(HunyuanVideo15Transformer3DModel)
    if vae_registry is not None:   # reference vae/__init__.py:9-73 (keys "auto" | "wan" | "qwenimage" | "hunyuanvideo15")
        from .vae_flux import AutoencoderKL
        from .vae_hunyuan15 import AutoencoderKLHunyuanVideo15
        from .vae_wan import AutoencoderKLWan
        vae_registry("auto_mi355", overwrite=True, available=ok)(AutoencoderKL)
        vae_registry("wan_mi355", overwrite=True, available=ok)(AutoencoderKLWan)
        vae_registry("qwenimage_mi355", overwrite=True, available=ok)(AutoencoderKLWan)
        vae_registry("hunyuanvideo15_mi355", overwrite=True, available=ok)(AutoencoderKLHunyuanVideo15)
    return transformers_registry
