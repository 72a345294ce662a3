"""The "hip_mfma" attention backend — B-op plug-in point.

Honours the calling convention of every backend in the reference's attention_register
(apps/api/src/attention/functions.py:84, e.g. `sdpa` :338-377):

    fn(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, softmax_scale=None, **kwargs) -> Tensor

q:[B,H,Sq,D], k,v:[B,H,Sk,D] (possibly permuted, non-contiguous views); returns [B,H,Sq,D] in q's
dtype without aliasing or modifying the inputs; enqueues on torch's current stream; no host sync when
`attn_mask` is None (every hot-path call: the reference forwards `attn_mask=attention_mask`, which is None there —
flux/base/attention.py:89-94, wan/base/attention.py:397-399, qwenimage/base/model.py:555-562).

A KEY-PADDING mask (bool keep-mask or additive 0 / -inf mask that does not vary along the query or head dimension:
[B, Sk], [B, 1, 1, Sk], [1, 1, 1, Sk], ...) — what a padded prompt batch brings — is honoured exactly: softmax over the
kept keys only, by running the same kernel over each sample's kept keys (a prefix is a view, anything else one gather).
It costs one host sync (the kept-key count sizes the launch).  Masks that vary along the query dimension, dropout and causal
attention are not on any call site of the path: they raise, they do not fall back.
"""
from __future__ import annotations

import torch

from . import ops
from .lib import ApexMIError

KEY = "hip_mfma"


def hip_mfma(q, k, v, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False,
             softmax_scale=None, **kwargs):
    if dropout_p:
        raise ApexMIError("hip_mfma: dropout is not supported (inference only)")
    if is_causal:
        raise ApexMIError("hip_mfma: causal attention is not supported")
    if attn_mask is None:
        return ops.attention(q, k, v, softmax_scale)
    keep = _key_keep_mask(attn_mask, q.shape[0], k.shape[2])
    if bool(keep.all()):
        return ops.attention(q, k, v, softmax_scale)
    B, H, Sq, D = q.shape
    out = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device).permute(0, 2, 1, 3)
    counts = keep.sum(dim=1).tolist()
    for b in range(B):
        n = int(counts[b])
        if n == 0:
            raise ApexMIError(f"hip_mfma: attn_mask drops every key of sample {b}")
        kb, vb = k[b:b + 1], v[b:b + 1]
        if bool(keep[b, :n].all()):               # a padded tail: the kept keys are a prefix (views, no copy)
            kb, vb = kb[:, :, :n], vb[:, :, :n]
        else:
            idx = keep[b].nonzero().flatten()
            kb, vb = kb.index_select(2, idx), vb.index_select(2, idx)
        out[b:b + 1].copy_(ops.attention(q[b:b + 1], kb, vb, softmax_scale))
    return out


def _key_keep_mask(attn_mask: torch.Tensor, B: int, Sk: int) -> torch.Tensor:
    """attn_mask (bool keep-mask, or additive: finite-and-not-hugely-negative = keep) -> bool [B, Sk]; raises unless the mask
    is constant along the head and query dimensions."""
    m = attn_mask
    if m.dim() == 2:
        if m.shape[-1] != Sk or m.shape[0] not in (1, B):
            raise ApexMIError(f"hip_mfma: attn_mask shape {tuple(m.shape)} is not a [B, Sk] key-padding mask")
        m = m[:, None, None, :]
    if m.dim() == 3:
        m = m[:, None]
    if m.dim() != 4 or m.shape[-1] != Sk or m.shape[1] != 1 or m.shape[2] != 1 or m.shape[0] not in (1, B):
        raise ApexMIError(f"hip_mfma: attn_mask of shape {tuple(attn_mask.shape)} varies along the head / query dimension; only "
                          "key-padding masks ([B, Sk], [B, 1, 1, Sk]) are supported")
    keep = m[:, 0, 0, :]
    if keep.dtype != torch.bool:
        keep = torch.isfinite(keep) & (keep > -1e4)
    return keep.expand(B, Sk)


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
