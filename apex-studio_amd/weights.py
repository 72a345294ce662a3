"""Checkpoint shards -> the packed device layout, without a host-side state dict.

The reference loads every weight file into a full state dict, runs the key converter, optionally patches the
model with FP-scaled layers that dequantise `weight * scale_weight` on EVERY forward, and calls
`model.load_state_dict(sd, strict=False, assign=True)` per file (`R/src/mixins/loader_mixin.py:439-531`;
`R/src/quantize/scaled_layer.py:496-549`, `fp8_activation_dequant` :154-167).

Here tensors are streamed one at a time out of the safetensors shards (`safe_open`, lazy) straight into the model's
parameters on the GPU — which, once `model.pack()` has run, are views of the fused QKV / modulation matrices, so
the "fused/tiled device layout" is written exactly once.  FP8-scaled pairs (`<m>.weight` in float8_e4m3fn / e5m2
plus `<m>.scale_weight`) are dequantised once at load by `apexmi_dequant_fp8_scaled` into the same bf16 values the
reference recomputes per forward, so the denoise kernels are unchanged.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import torch

FP8_DTYPES = (torch.float8_e4m3fn, torch.float8_e5m2)


def iter_checkpoint(files: Sequence[str], converter=None, key_map: Optional[Dict[str, str]] = None,
                    model_keys: Optional[Sequence[str]] = None) -> Iterator[Tuple[str, Callable[[], torch.Tensor]]]:
    """(key, loader) for every tensor of every shard; `loader()` reads that one tensor to the host.

    With a `converter` (converters.get_transformer_converter: original-format Wan / BFL-Flux files), every file's keys go
    through `key_map` and the converter FIRST, as the reference does per weight file (loader_mixin.py:462-473) — but on
    placeholders: no tensor is read to decide the plan, and a target that is a row range of a fused tensor (q / k / v of
    `img_attn.qkv.weight`, the four parts of `linear1`) is read with safetensors' `get_slice`."""
    from .converters import Src
    for path in files:
        if path.endswith(".safetensors"):
            from safetensors import safe_open
            f = safe_open(path, framework="pt", device="cpu")
            if converter is None:
                for k in f.keys():
                    yield remap_key(k, key_map), (lambda f=f, k=k: f.get_tensor(k))
                continue
            plan = {remap_key(k, key_map): Src(k, tuple(f.get_slice(k).get_shape())) for k in f.keys()}
            get, rows = f.get_tensor, (lambda k, a, b, f=f: f.get_slice(k)[a:b])
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True, mmap=True)
            if converter is None:
                for k, v in sd.items():
                    yield remap_key(k, key_map), (lambda v=v: v)
                continue
            plan = {remap_key(k, key_map): Src(k, tuple(v.shape)) for k, v in sd.items()}
            get, rows = sd.__getitem__, None
        converter.convert(plan, list(model_keys) if model_keys is not None else None)
        for k, src in plan.items():
            yield k, (lambda src=src, get=get, rows=rows: src.read(get, rows))


def remap_key(key: str, key_map: Optional[Dict[str, str]]) -> str:
    """`key_map` semantics of the reference loader (loader_mixin.py:462-471): replace a substring."""
    if key_map:
        for src, dst in key_map.items():
            if src in key:
                key = key.replace(src, dst)
    return key


@torch.no_grad()
def load_checkpoint_into(model: torch.nn.Module, files: Sequence[str], key_map: Optional[Dict[str, str]] = None,
                         strict: bool = False, converter="auto", keep_fp8: bool = False) -> Tuple[List[str], List[str]]:
    """Stream `files` into `model` (already on the GPU, bf16).  Returns (missing_keys, unexpected_keys) like
    `load_state_dict(strict=False)`; `strict=True` raises on either.  Shapes must match exactly, except that a
    0-d / 1-element `scale_weight` may pair with any weight (scaled_layer.py:444-493).

    `converter`: the checkpoint key converter the files go through per file, as the reference's loader does
    (loader_mixin.py:473 `converter.convert(state_dict, model_keys)`): "auto" = the table of the model's family
    (`model._converter_base`; original-format Wan / BFL-Flux files are renamed and split on the fly, diffusers-keyed files
    pass through the converter's own already-converted test), None = keys are taken as they are, or a converters.KeyConverter.

    `keep_fp8`: fp8-scaled weights the model can hold RESIDENT (`model._fp8_resident_key(key)`: the block Linears of
    `wan.mi355`) stay float8 + scale in HBM — `ops.Fp8Weight`, dequantised per call as the reference's FPScaledLinear does
    (scaled_layer.py:390-552) — instead of being dequantised once into the bf16 parameter; the parameter's bf16 storage is
    released (`model._fp8_adopt()`).  Forwards are bit-identical to the dequantise-at-load path; half the weight bytes."""
    from . import ops
    resident = getattr(model, "_fp8_resident_key", None) if keep_fp8 else None
    if keep_fp8 and resident is None:
        raise NotImplementedError(f"{type(model).__name__} has no resident-fp8 weight mode (keep_fp8=True): wan.mi355 has")
    targets: Dict[str, torch.Tensor] = dict(model.named_parameters())
    targets.update({k: v for k, v in model.named_buffers() if k not in targets})
    if any(not t.is_cuda for t in targets.values()):
        raise RuntimeError("load_checkpoint_into: move the model to the GPU first (there is no CPU path)")
    if converter == "auto":
        from .converters import NoOpKeyConverter, get_transformer_converter
        converter = get_transformer_converter(getattr(model, "_converter_base", ""))
        if isinstance(converter, NoOpKeyConverter):
            converter = None
    entries = list(iter_checkpoint(files, converter, key_map, list(targets)))
    scales = {k[:-len("scale_weight")]: ld for k, ld in entries if k.endswith("scale_weight")}
    seen, unexpected = set(), []
    for key, ld in entries:
        if key.endswith("scale_weight"):
            if key[:-len("scale_weight")] + "weight" not in targets:
                unexpected.append(key)
            continue
        if key not in targets:
            unexpected.append(key)
            continue
        dst = targets[key]
        src = ld()
        if tuple(src.shape) != tuple(dst.shape):
            raise ValueError(f"{key}: checkpoint shape {tuple(src.shape)} != parameter shape {tuple(dst.shape)}")
        prefix = key[:-len("weight")] if key.endswith("weight") else None
        if src.dtype in FP8_DTYPES:
            if prefix is None or prefix not in scales:
                raise ValueError(f"{key}: fp8 tensor without a '{(prefix or key)}scale_weight' partner")
            if dst.dtype != torch.bfloat16:
                raise TypeError(f"{key}: fp8-scaled weights load into bf16 parameters, not {dst.dtype}")
            q = src.view(torch.uint8).to(dst.device, non_blocking=True).view(src.dtype)
            if resident is not None and dst.dim() == 2 and resident(key):
                dst._fp8 = ops.Fp8Weight(q, scales[prefix]())            # adopted after the loop (model._fp8_adopt)
            else:
                out2d = dst.data.view(dst.shape[0], -1) if dst.dim() != 2 else dst.data
                ops.dequant_fp8_scaled(q, scales[prefix](), out=out2d)
        else:
            if prefix is not None and prefix in scales:
                # the reference raises here too (`physical_dtype in (torch.uint8)`, scaled_layer.py:525)
                raise TypeError(f"{key}: '{prefix}scale_weight' is present but the weight is {src.dtype}, not fp8")
            dst.data.copy_(src.to(dst.device, non_blocking=True))
        seen.add(key)
    # derived state (f32 modulation tables, padded patch-embed operand, packed conv weights) follows the new values
    for m in model.modules():
        hook = getattr(m, "_weights_changed", None)
        if callable(hook):
            hook()
    if resident is not None:
        model._fp8_adopt()
    missing = [k for k in targets if k not in seen]
    if strict and (missing or unexpected):
        raise RuntimeError(f"load_checkpoint_into: missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                           f"unexpected {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
    return missing, unexpected
