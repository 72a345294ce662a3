"""Checkpoint key converters: original-format weight files -> the diffusers-style keys the model classes carry
(VERDICT r2 "missing" 1-2).

The manifests ship ORIGINAL-format files — `Wan-AI/Wan2.2-T2V-A14B/high_noise_model` and the Kijai fp8-scaled Wan files
(`blocks.N.self_attn.q.weight` ...), BFL `flux1-dev` (`double_blocks.N.img_attn.qkv.weight` ...), lightx2v LoRAs keyed
`diffusion_model.blocks.N.self_attn.q.lora_down.weight` — and the reference renames / splits them per weight file before
`load_state_dict` (`R/src/mixins/loader_mixin.py:439-531` -> `converter.convert(state_dict, model_keys)`), LoRAs through
the same table (`R/src/lora/manager.py:633-644`).  This module restates that pipeline
(`R/src/converters/base_converter.py:563-585`: pre-handlers -> ordered substring renames -> post-handlers -> wrapper-prefix
strip, with the "already converted" early exit :333-433) and the two tables of the hot-path families:

    WanKeyConverter    R/src/converters/transformer_converters.py:134-198   (`wan.base`)
    FluxKeyConverter   R/src/converters/transformer_converters.py:1372-1840 (`flux.base`: fused-QKV / linear1 splits,
                       guidance group, `final_layer.adaLN_modulation` with [shift, scale] -> [scale, shift],
                       R/src/converters/utils.py:82-85)
    QwenImage          `qwenimage.base` has no table in the reference (get_transformer_converter falls through to the no-op,
                       R/src/converters/convert.py:71-122): its files are diffusers-keyed.

Pinned by running the reference's own converter classes on seeded original-key state dicts in the build container
(`tests/golden/make_golden.py convert`, `tests/golden/convert_keys.pt`: resulting keys and a checksum per tensor).

Design difference from the reference: conversion never needs the tensors.  The handlers are written against three
layout primitives (`rows3`, `rows_split`, `swap_halves`) that work on real tensors (LoRA state dicts, small) AND on `Src`
placeholders that only remember (file key, row range, half swap).  `weights.load_checkpoint_into(converter=...)` converts a
dict of placeholders per file and then streams every target straight into its (packed) parameter — a fused
`img_attn.qkv.weight` is read as three row ranges with `get_slice`, never materialised and chunked on the host.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, replace
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch

WRAPPERS = ("model.diffusion_model.", "diffusion_model.model.", "model.", "diffusion_model.", "module.", "unet.")
LORA_SEGMENTS = (".lora_A.", ".lora_B.", ".lora_up.", ".lora_down.", ".Lora_A.", ".Lora_B.", ".Lora_up.", ".Lora_down.")


# ---- placeholders: a tensor of a weight file that has not been read ---------------------------------------------------
@dataclass(frozen=True)
class Src:
    key: str                                   # key inside the weight file
    shape: Tuple[int, ...]
    dtype: Any = None
    rows: Optional[Tuple[int, int]] = None      # row range [start, stop) of dim 0
    swap: bool = False                          # exchange the two halves of dim 0 after reading

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def _range(self, a: int, b: int) -> "Src":
        if self.swap:
            raise ValueError(f"{self.key}: a row range of a half-swapped tensor is not representable")
        base = self.rows[0] if self.rows else 0
        return replace(self, shape=(b - a,) + tuple(self.shape[1:]), rows=(base + a, base + b))

    def read(self, get_tensor: Callable[[str], torch.Tensor], get_rows=None) -> torch.Tensor:
        """Materialise: `get_tensor(key)` reads the whole tensor, `get_rows(key, a, b)` a row range (safetensors `get_slice`)."""
        if self.rows is not None:
            t = get_rows(self.key, *self.rows) if get_rows is not None else get_tensor(self.key)[self.rows[0]:self.rows[1]]
        else:
            t = get_tensor(self.key)
        if self.swap:
            a, b = t.chunk(2, dim=0)
            t = torch.cat([b, a], dim=0)
        return t


def rows_split(t, sizes: Sequence[int]):
    """t split along dim 0 into pieces of `sizes` rows (torch.split for tensors, row ranges for placeholders)."""
    if isinstance(t, Src):
        if sum(sizes) != t.shape[0]:
            raise ValueError(f"{t.key}: cannot split {t.shape[0]} rows into {tuple(sizes)}")
        out, a = [], 0
        for n in sizes:
            out.append(t._range(a, a + n))
            a += n
        return out
    return list(torch.split(t, list(sizes), dim=0))


def rows3(t):
    n = t.shape[0]
    if n % 3:
        raise ValueError(f"fused q/k/v tensor with {n} rows")
    return rows_split(t, (n // 3,) * 3)


def swap_halves(t):
    """[shift, scale] -> [scale, shift] along dim 0 (AdaLayerNormContinuous is scale-first)."""
    if isinstance(t, Src):
        if t.rows is not None:
            raise ValueError(f"{t.key}: half swap of a row range is not representable")
        return replace(t, swap=not t.swap)
    a, b = t.chunk(2, dim=0)
    return torch.cat([b, a], dim=0)


def _move(sd: Dict[str, Any], old: str, new: str) -> None:
    if old in sd:
        sd[new] = sd.pop(old)


# ---- the pipeline ---------------------------------------------------------------------------------------------------------
def _is_regex(p: str) -> bool:
    return any(c in p for c in "^$()[]{}|?+\\")


def _lora_variants(key: str) -> Iterable[str]:
    yield key
    for seg in LORA_SEGMENTS:
        if seg in key:
            yield key.replace(seg, ".", 1)


def _under_model(candidate: str, model: set) -> bool:
    """candidate is a model key, or lives under a module name listed in `model`."""
    if candidate in model:
        return True
    parts = candidate.split(".")
    return any(".".join(parts[:i]) in model for i in range(1, len(parts)))


def overlap_score(keys: Iterable[str], model_keys: Sequence[str]) -> int:
    model = set(model_keys)
    return sum(1 for k in keys if any(_under_model(v, model) for v in _lora_variants(k)))


def _strip_for_overlap(keys: List[str], reference: set) -> List[str]:
    out = list(keys)
    changed = True
    while changed:
        changed = False
        for p in WRAPPERS:
            if not out or not all(k.startswith(p) for k in out):
                continue
            cut = [k[len(p):] for k in out]
            if any(not k for k in cut) or len(set(cut)) != len(cut):
                continue
            if sum(k in reference for k in cut) > sum(k in reference for k in out):
                out, changed = cut, True
                break
    return out


class KeyConverter:
    """pre-handlers -> renames -> post-handlers -> prefix strip, in place (`convert`)."""
    GENERIC = {"norm", "norm1", "norm2", "norm3", "weight", "bias"}
    PRIORITY = ("norm2", "norm3", "norm__placeholder")        # the Wan norm swap must see SOURCE keys only

    def __init__(self):
        self.rename: Dict[str, str] = {}
        self.pre: Dict[str, Callable[[str, Dict[str, Any]], None]] = {}
        self.post: Dict[str, Callable[[str, Dict[str, Any]], None]] = {}

    # -- helpers
    @staticmethod
    def drop(key: str, sd: Dict[str, Any]) -> None:
        sd.pop(key, None)

    def _ordered_rename(self) -> List[Tuple[str, str]]:
        first = [(k, self.rename[k]) for k in self.PRIORITY if k in self.rename]
        rest = sorted(((k, v) for k, v in self.rename.items() if k not in self.PRIORITY), key=lambda kv: -len(kv[0]))
        return first + rest       # longest source first: `cross_attn.k_img` before `cross_attn.k`

    def renamed(self, key: str, table: Optional[List[Tuple[str, str]]] = None) -> str:
        for src, dst in (table if table is not None else self._ordered_rename()):
            if "*" in src and not _is_regex(src):          # glob: every '*' captures and is substituted in order
                parts = src.split("*")
                pat = "".join(re.escape(p) + ("(.*?)" if i < len(parts) - 1 else "") for i, p in enumerate(parts))

                def sub(m, dst=dst):
                    r = dst
                    for g in m.groups():
                        r = r.replace("*", g, 1)
                    return r
                key = re.sub(pat, sub, key)
            elif _is_regex(src):
                key = re.sub(src, dst, key)
            else:
                key = key.replace(src, dst)
        return key

    @classmethod
    def _specific(cls, s: str) -> bool:
        return bool(s) and s not in cls.GENERIC and ("." in s or "_" in s or len(s) >= 8)

    def _matches_model(self, keys: List[str], model_keys: Sequence[str]) -> bool:
        state, model = set(keys), set(model_keys)
        model_n = set(_strip_for_overlap(list(model), state))
        state_n = set(_strip_for_overlap(list(state), model_n))
        total, hit = len(state_n), len(state_n & model_n)
        if total == 0 or hit < min(10, total):
            return False
        return hit / total >= 0.98 and total - hit <= max(2, int(0.02 * total))

    def already_converted(self, sd: Dict[str, Any], model_keys: Optional[Sequence[str]] = None) -> bool:
        if not sd:
            return True
        keys = list(sd)
        if model_keys and self._matches_model(keys, model_keys):
            return True
        if any(_is_regex(k) or "*" in k for k in self.rename):
            return False if model_keys else not any(self.renamed(k) != k for k in keys)
        if any("norm__placeholder" in k for k in keys):
            return False
        if any(m in k for m in list(self.pre) + list(self.post) for k in keys):
            return False
        src_marks = [k for k in self.rename if self._specific(k)]
        dst_marks = [v for v in self.rename.values() if self._specific(v)]
        if not dst_marks or not any(m in k for m in dst_marks for k in keys):
            return False
        return not any(m in k for m in src_marks for k in keys)

    def strip_prefixes(self, sd: Dict[str, Any], model_keys: Optional[Sequence[str]] = None) -> None:
        """Wrapper prefixes (`diffusion_model.`, `model.`, `base_model.model.` ...): with `model_keys` the prefix (also when
        only a subset of the keys carries it) whose removal raises the overlap most; without, unanimous known wrappers."""
        if not sd:
            return
        if model_keys:
            seeds = WRAPPERS + ("base_model.model.", "base_model.")
            changed = True
            while changed:
                changed = False
                keys = list(sd)
                score = overlap_score(keys, model_keys)
                split = [k.split(".") for k in keys]
                n = 0
                while n < min(8, len(split[0])) and split[0][n] and all(len(s) > n and s[n] == split[0][n] for s in split[1:]):
                    n += 1
                cands, seen = [], set()
                for p in list(seeds) + [".".join(split[0][:i]) + "." for i in range(1, n + 1)]:
                    if p and p not in seen:
                        seen.add(p)
                        cands.append(p)
                best, best_score = None, score
                for p in cands:
                    if not any(k.startswith(p) for k in keys):
                        continue
                    cut = [k[len(p):] if k.startswith(p) else k for k in keys]
                    if any(not k for k in cut) or len(set(cut)) != len(cut):
                        continue
                    s = overlap_score(cut, model_keys)
                    if s > best_score:
                        best, best_score = p, s
                    elif s == best_score and best is not None and len(p) < len(best):
                        best = p
                if best and best_score > score:
                    for k in list(sd):
                        if k.startswith(best):
                            _move(sd, k, k[len(best):])
                    changed = True
            if overlap_score(list(sd), model_keys) > 0:
                return
        changed = True
        while changed:
            changed = False
            for p in ("model.diffusion_model.", "diffusion_model.model.", "diffusion_model.", "unet.", "base_model.model."):
                keys = list(sd)
                cut = [k[len(p):] for k in keys]
                if all(k.startswith(p) for k in keys) and all(cut) and len(set(cut)) == len(cut):
                    for k, c in zip(keys, cut):
                        _move(sd, k, c)
                    changed = True
                    break

    def convert(self, sd: Dict[str, Any], model_keys: Optional[Sequence[str]] = None) -> Dict[str, Any]:
        if self.already_converted(sd, model_keys):
            return sd
        for key in list(sd):
            for marker, fn in self.pre.items():
                if marker in key:
                    fn(key, sd)
        table = self._ordered_rename()
        for key in list(sd):
            _move(sd, key, self.renamed(key, table))
        for key in list(sd):
            for marker, fn in self.post.items():
                if marker in key:
                    fn(key, sd)
        self.strip_prefixes(sd, model_keys)
        return sd


class NoOpKeyConverter(KeyConverter):
    def convert(self, sd, model_keys=None):
        return sd


class WanKeyConverter(KeyConverter):
    """Original Wan 2.x keys -> diffusers WanTransformer3DModel keys (transformer_converters.py:134-198)."""

    def __init__(self):
        super().__init__()
        attn = {"q": "to_q", "k": "to_k", "v": "to_v", "o": "to_out.0", "norm_q": "norm_q", "norm_k": "norm_k"}
        self.rename = {
            "time_embedding.0": "condition_embedder.time_embedder.linear_1",
            "time_embedding.2": "condition_embedder.time_embedder.linear_2",
            "text_embedding.0": "condition_embedder.text_embedder.linear_1",
            "text_embedding.2": "condition_embedder.text_embedder.linear_2",
            "time_projection.1": "condition_embedder.time_proj",
            "head.modulation": "scale_shift_table",
            "head.head": "proj_out",
            "modulation": "scale_shift_table",
            "ffn.0": "ffn.net.0.proj",
            "ffn.2": "ffn.net.2",
            # the original block calls its norms norm1, norm3, norm2: swap 2 <-> 3 through a placeholder
            "norm2": "norm__placeholder",
            "norm3": "norm2",
            "norm__placeholder": "norm3",
            # image-to-video / first-last-frame / IP variants
            "img_emb.proj.0": "condition_embedder.image_embedder.norm1",
            "img_emb.proj.1": "condition_embedder.image_embedder.ff.net.0.proj",
            "img_emb.proj.3": "condition_embedder.image_embedder.ff.net.2",
            "img_emb.proj.4": "condition_embedder.image_embedder.norm2",
            "img_emb.emb_pos": "condition_embedder.image_embedder.pos_embed",
            "self_attn.q_loras": "attn1.add_q_lora",
            "self_attn.k_loras": "attn1.add_k_lora",
            "self_attn.v_loras": "attn1.add_v_lora",
            "cross_attn.k_img": "attn2.add_k_proj",
            "cross_attn.v_img": "attn2.add_v_proj",
            "cross_attn.norm_k_img": "attn2.norm_added_k",
        }
        for src, dst in attn.items():
            self.rename[f"self_attn.{src}"] = f"attn1.{dst}"
            self.rename[f"cross_attn.{src}"] = f"attn2.{dst}"
        # difference vectors of some LoRA exports and the fp8 marker tensor have no counterpart in the model
        self.pre = {".diff_b": self.drop, ".diff": self.drop, "scaled_fp8": self.drop}


class FluxKeyConverter(KeyConverter):
    """BFL Flux keys -> diffusers FluxTransformer2DModel keys, base weights and LoRA factors alike
    (transformer_converters.py:1372-1840)."""
    _GUIDANCE = {"guidance_in.in_layer.": "time_text_embed.guidance_embedder.linear_1.",
                 "guidance_in.out_layer.": "time_text_embed.guidance_embedder.linear_2."}

    def __init__(self):
        super().__init__()
        self.rename = {
            "time_in.in_layer.": "time_text_embed.timestep_embedder.linear_1.",
            "time_in.out_layer.": "time_text_embed.timestep_embedder.linear_2.",
            "vector_in.in_layer.": "time_text_embed.text_embedder.linear_1.",
            "vector_in.out_layer.": "time_text_embed.text_embedder.linear_2.",
            "txt_in.": "context_embedder.",
            "img_in.": "x_embedder.",
            "double_blocks.": "transformer_blocks.",
            "single_blocks.": "single_transformer_blocks.",
            ".img_mod.lin.": ".norm1.linear.",
            ".txt_mod.lin.": ".norm1_context.linear.",
            "img_attn.norm.query_norm.scale": "attn.norm_q.weight",
            "img_attn.norm.key_norm.scale": "attn.norm_k.weight",
            "txt_attn.norm.query_norm.scale": "attn.norm_added_q.weight",
            "txt_attn.norm.key_norm.scale": "attn.norm_added_k.weight",
            ".img_mlp.0.": ".ff.net.0.proj.",
            ".img_mlp.2.": ".ff.net.2.",
            ".txt_mlp.0.": ".ff_context.net.0.proj.",
            ".txt_mlp.2.": ".ff_context.net.2.",
            ".img_attn.proj.": ".attn.to_out.0.",
            ".txt_attn.proj.": ".attn.to_add_out.",
            ".modulation.lin.": ".norm.linear.",
            ".linear2.": ".proj_out.",
            "final_layer.linear.": "proj_out.",
        }
        self.inner_dim, self.mlp_ratio = 3072, 4.0          # FLUX.1-dev, used when a file does not reveal them

    # -- shape inference (transformer_converters.py:1418-1505): only what the splits need
    def _infer(self, sd: Dict[str, Any]) -> None:
        inner = ratio = None
        for key, t in sd.items():
            shape = tuple(getattr(t, "shape", ()))
            if len(shape) != 2:
                continue
            if (inner is None or ratio is None) and "single_blocks." in key and key.endswith(".linear1.weight"):
                hidden = shape[0] - 3 * shape[1]
                if hidden > 0:
                    inner, ratio = shape[1], hidden / shape[1]
            if inner is None and "double_blocks." in key and "img_attn.qkv.weight" in key and shape[0] // 3 > 0:
                inner = shape[0] // 3
            if ratio is None and "double_blocks." in key and "img_mlp.0.weight" in key:
                inner = inner if inner is not None else shape[1]
                if inner and shape[0] > 0:
                    ratio = shape[0] / inner
        self.inner_dim = inner if inner is not None else self.inner_dim
        self.mlp_ratio = ratio if ratio is not None else self.mlp_ratio

    def _single_sizes(self) -> Tuple[int, int, int, int]:
        d = self.inner_dim
        return (d, d, d, int(d * self.mlp_ratio))

    # -- handlers that run before the renames
    def _guidance(self, _key: str, sd: Dict[str, Any]) -> None:
        names = [s + p for s in self._GUIDANCE for p in ("weight", "bias")]
        if all(n in sd for n in names):                       # all four or nothing
            for s, d in self._GUIDANCE.items():
                for p in ("weight", "bias"):
                    sd[d + p] = sd.pop(s + p)

    @staticmethod
    def _final_mod(key: str, sd: Dict[str, Any]) -> None:
        if key in sd:
            sd["norm_out.linear." + ("weight" if key.endswith(".weight") else "bias")] = swap_halves(sd.pop(key))

    @staticmethod
    def _double_qkv(key: str, sd: Dict[str, Any]) -> None:
        m = re.match(r"double_blocks\.(\d+)\.(img|txt)_attn\.qkv\.(weight|bias)$", key)
        if not m or key not in sd:
            return
        names = ("to_q", "to_k", "to_v") if m.group(2) == "img" else ("add_q_proj", "add_k_proj", "add_v_proj")
        for name, part in zip(names, rows3(sd.pop(key))):
            sd[f"transformer_blocks.{m.group(1)}.attn.{name}.{m.group(3)}"] = part

    @staticmethod
    def _double_qkv_lora(key: str, sd: Dict[str, Any]) -> None:
        m = re.match(r"(?:unet\.)?double_blocks\.(\d+)\.(img_attn|txt_attn)\.qkv\.(lora_down|lora_up|lora_A|lora_B)\.(weight|bias)$", key)
        if not m or key not in sd:
            return
        i, which, kind, suffix = m.groups()
        names = ("to_q", "to_k", "to_v") if which == "img_attn" else ("add_q_proj", "add_k_proj", "add_v_proj")
        t = sd.pop(key)
        parts = [t, t, t] if kind in ("lora_down", "lora_A") else rows3(t)      # the down factor is shared, the up factor split
        for name, part in zip(names, parts):
            sd[f"transformer_blocks.{i}.attn.{name}.{kind}.{suffix}"] = part

    def _single_linear1_lora(self, key: str, sd: Dict[str, Any]) -> None:
        m = re.match(r"(?:unet\.)?single_blocks\.(\d+)\.linear1\.(lora_down|lora_up|lora_A|lora_B)\.(weight|bias)$", key)
        if not m or key not in sd:
            return
        i, kind, suffix = m.groups()
        t = sd.pop(key)
        parts = [t] * 4 if kind in ("lora_down", "lora_A") else rows_split(t, self._single_sizes())
        for name, part in zip(("attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp"), parts):
            sd[f"single_transformer_blocks.{i}.{name}.{kind}.{suffix}"] = part

    def _single_linear1(self, key: str, sd: Dict[str, Any]) -> None:
        m = re.match(r"single_blocks\.(\d+)\.linear1\.(weight|bias)$", key)
        if not m:
            return
        w, b = (f"single_blocks.{m.group(1)}.linear1.{s}" for s in ("weight", "bias"))
        if w in sd and b in sd:
            if not key.endswith(".weight"):
                return
            for suffix, t in (("weight", sd.pop(w)), ("bias", sd.pop(b))):
                for name, part in zip(("attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp"), rows_split(t, self._single_sizes())):
                    sd[f"single_transformer_blocks.{m.group(1)}.{name}.{suffix}"] = part
            return
        for k in (w, b):          # half a pair: keep the fused tensor under its original name, shielded from the renames
            _move(sd, k, k.replace("single_blocks.", "single_blocks__keep.", 1))

    # -- handlers that run after the renames
    @staticmethod
    def _unshield(key: str, sd: Dict[str, Any]) -> None:
        if "single_blocks__keep." in key:
            _move(sd, key, key.replace("single_blocks__keep.", "single_blocks.", 1))

    @staticmethod
    def _single_norm_scale(key: str, sd: Dict[str, Any]) -> None:
        m = re.match(r"^(?:unet\.)?single_transformer_blocks\.(\d+)\.norm\.(query_norm|key_norm)\.scale$", key)
        if m and key in sd:
            _move(sd, key, f"single_transformer_blocks.{m.group(1)}.attn.{'norm_q' if m.group(2) == 'query_norm' else 'norm_k'}.weight")

    def convert(self, sd: Dict[str, Any], model_keys: Optional[Sequence[str]] = None) -> Dict[str, Any]:
        if model_keys is not None and self.already_converted(sd, model_keys):
            return sd
        if not sd:
            return sd
        keys = list(sd)
        target = any(m in k for k in keys for m in ("transformer_blocks.", "single_transformer_blocks.", "time_text_embed."))
        source = any(m in k for k in keys for m in ("double_blocks.", "single_blocks.", "time_in.", "vector_in.", "guidance_in.",
                                                     "txt_in.", "img_in.", "final_layer."))
        if target and not source:
            return sd
        self._infer(sd)
        self.pre = {
            "guidance_in.": self._guidance,
            "final_layer.adaLN_modulation.1.": self._final_mod,
            ".qkv.lora_": self._double_qkv_lora,
            ".img_attn.qkv.weight": self._double_qkv, ".img_attn.qkv.bias": self._double_qkv,
            ".txt_attn.qkv.weight": self._double_qkv, ".txt_attn.qkv.bias": self._double_qkv,
            ".linear1.lora_": self._single_linear1_lora,
            ".linear1.": self._single_linear1,
        }
        self.post = {"single_blocks__keep.": self._unshield, ".norm.query_norm.scale": self._single_norm_scale,
                     ".norm.key_norm.scale": self._single_norm_scale}
        return super().convert(sd)          # NB: the shared pipeline runs WITHOUT model_keys here, as the reference's does


class Hunyuan15KeyConverter(KeyConverter):
    """Original HunyuanVideo-1.5 keys (Tencent release, Comfy-Org repackaging, lightx2v LoRAs keyed on it) -> the diffusers-style
    keys of `HunyuanVideo15Transformer3DModel` (reference `HunyuanVideo15TransformerConverter`, transformer_converters.py:899-1110):
    the rename table, then the fused `*_attn_qkv` / `*_attn.qkv` / refiner `self_attn_qkv` tensors split into q / k / v — base
    weights, biases and LoRA-up factors by thirds along the fused dimension, LoRA-down factors / `.alpha` / scalar fp8 scales
    shared by the three."""
    _QKV = (("img_attn_qkv", ("attn.to_q", "attn.to_k", "attn.to_v")), ("img_attn.qkv", ("attn.to_q", "attn.to_k", "attn.to_v")),
            ("txt_attn_qkv", ("attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj")),
            ("txt_attn.qkv", ("attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj")))

    def __init__(self):
        super().__init__()
        ref = "txt_in.individual_token_refiner.blocks."
        dst = "context_embedder.token_refiner.refiner_blocks."
        self.rename = {
            "double_blocks": "transformer_blocks",
            "txt_in.t_embedder.in_layer": "context_embedder.time_text_embed.timestep_embedder.linear_1",
            "txt_in.t_embedder.out_layer": "context_embedder.time_text_embed.timestep_embedder.linear_2",
            "txt_in.c_embedder.in_layer": "context_embedder.time_text_embed.text_embedder.linear_1",
            "txt_in.c_embedder.out_layer": "context_embedder.time_text_embed.text_embedder.linear_2",
            ref + "*.self_attn.proj": dst + "*.attn.to_out.0",
            ref + "*.mlp.0": dst + "*.ff.net.0.proj",
            ref + "*.mlp.2": dst + "*.ff.net.2",
            "time_in.in_layer": "time_embed.timestep_embedder.linear_1",
            "time_in.out_layer": "time_embed.timestep_embedder.linear_2",
            "byt5_in.fc1": "context_embedder_2.linear_1",
            "byt5_in.fc2": "context_embedder_2.linear_2",
            "byt5_in.fc3": "context_embedder_2.linear_3",
            "byt5_in.layernorm": "context_embedder_2.norm",
            "cond_type_embedding": "cond_type_embed",
            "time_in.mlp.0": "time_embed.timestep_embedder.linear_1",
            "time_in.mlp.2": "time_embed.timestep_embedder.linear_2",
            "time_r_in.mlp.0": "time_embed.timestep_embedder_r.linear_1",
            "time_r_in.mlp.2": "time_embed.timestep_embedder_r.linear_2",
            "final_layer.linear": "proj_out",
            "final_layer.adaLN_modulation.1": "norm_out.linear",
            "img_in.proj": "x_embedder.proj",
            "vision_in.proj.0": "image_embedder.norm_in",
            "vision_in.proj.1": "image_embedder.linear_1",
            "vision_in.proj.3": "image_embedder.linear_2",
            "vision_in.proj.4": "image_embedder.norm_out",
            "txt_in.c_embedder.linear_1": "context_embedder.time_text_embed.text_embedder.linear_1",
            "txt_in.c_embedder.linear_2": "context_embedder.time_text_embed.text_embedder.linear_2",
            "txt_in.input_embedder": "context_embedder.proj_in",
            "txt_in.t_embedder.mlp.0": "context_embedder.time_text_embed.timestep_embedder.linear_1",
            "txt_in.t_embedder.mlp.2": "context_embedder.time_text_embed.timestep_embedder.linear_2",
            ref + "*.adaLN_modulation.1": dst + "*.norm_out.linear",
            ref + "*.norm1": dst + "*.norm1",
            ref + "*.norm2": dst + "*.norm2",
            ref + "*.mlp.fc1": dst + "*.ff.net.0.proj",
            ref + "*.mlp.fc2": dst + "*.ff.net.2",
            ref + "*.self_attn_proj": dst + "*.attn.to_out.0",
            ref: dst,
            ".img_attn_k.": ".attn.to_k.", ".img_attn_k_norm.": ".attn.norm_k.",
            ".img_attn_q.": ".attn.to_q.", ".img_attn_q_norm.": ".attn.norm_q.",
            ".img_attn_v.": ".attn.to_v.", ".img_attn_proj.": ".attn.to_out.0.",
            ".txt_attn_k.": ".attn.add_k_proj.", ".txt_attn_k_norm.": ".attn.norm_added_k.",
            ".txt_attn_q.": ".attn.add_q_proj.", ".txt_attn_q_norm.": ".attn.norm_added_q.",
            ".txt_attn_v.": ".attn.add_v_proj.", ".txt_attn_proj.": ".attn.to_add_out.",
            ".txt_mlp.fc1": ".ff_context.net.0.proj", ".txt_mlp.fc2": ".ff_context.net.2",
            ".img_mlp.fc1": ".ff.net.0.proj", ".img_mlp.fc2": ".ff.net.2",
            ".img_mod.linear": ".norm1.linear", ".txt_mod.linear": ".norm1_context.linear",
            ".img_attn.proj": ".attn.to_out.0", ".txt_attn.proj": ".attn.to_add_out",
            ".img_mod.lin.": ".norm1.linear.", ".txt_mod.lin.": ".norm1_context.linear.",
            ".img_mlp.0": ".ff.net.0.proj", ".img_mlp.2": ".ff.net.2",
            ".txt_mlp.0": ".ff_context.net.0.proj", ".txt_mlp.2": ".ff_context.net.2",
        }
        self.post = {"double_blocks": self._blocks, "transformer_blocks": self._blocks,
                     "self_attn_qkv": self._refiner_qkv, "self_attn.qkv": self._refiner_qkv}

    @staticmethod
    def _shared(key: str) -> bool:          # LoRA "down" factors and alphas belong to q, k and v alike
        return ".lora_down" in key or ".lora_A" in key or key.endswith(".alpha")

    @staticmethod
    def _thirds(key: str, t):
        shape = tuple(t.shape)
        dim = 0 if len(shape) != 2 or shape[0] > shape[1] else 1          # `get_chunk_dim`
        if shape[dim] % 3:
            raise ValueError(f"Expected QKV fused dim divisible by 3 for key='{key}', shape={shape}, chunk_dim={dim}")
        if dim == 0:
            return rows3(t)
        if isinstance(t, Src):
            raise ValueError(f"{key}: a column split of a streamed tensor is not representable")
        return list(torch.chunk(t, 3, dim=1))

    def _write(self, key: str, sd: Dict[str, Any], src: str, names: Tuple[str, str, str]) -> None:
        t = sd.pop(key)
        numel = 1
        for n in tuple(t.shape):
            numel *= n
        if self._shared(key) or len(tuple(t.shape)) == 0 or numel == 1:
            parts = (t, t, t)
        else:
            parts = self._thirds(key, t)
        for name, part in zip(names, parts):
            sd[key.replace(src, name).replace("double_blocks", "transformer_blocks")] = part

    def _blocks(self, key: str, sd: Dict[str, Any]) -> None:
        for src, names in self._QKV:
            if src in key and key in sd:
                self._write(key, sd, src, names)

    def _refiner_qkv(self, key: str, sd: Dict[str, Any]) -> None:
        if key not in sd:
            return
        src = "self_attn_qkv" if "self_attn_qkv" in key else "self_attn.qkv"
        t = sd.pop(key)
        parts = (t, t, t) if self._shared(key) else rows3(t)
        for name, part in zip(("attn.to_q", "attn.to_k", "attn.to_v"), parts):
            sd[key.replace(src, name)] = part


def get_transformer_converter(model_base: str) -> KeyConverter:
    """`get_transformer_converter` (R/src/converters/convert.py:71-122) for the families on the hot path; registry keys of
    this backend ("wan.mi355" ...) select the same tables as the reference's ("wan.base" ...)."""
    # Only the bases whose tables are ported: the reference has DIFFERENT converters for wan.vace / s2v / animate / multitalk /
    # ovi / flashvsr ... (convert.py:71-122); sending those through the wan.base table would silently mis-rename them.
    family, _, variant = model_base.partition(".")
    if variant not in ("", "base", "mi355"):
        raise NotImplementedError(f"key converter for model base {model_base!r} is not ported (have: wan / flux / hunyuanvideo15 "
                                  ".base; qwenimage files are diffusers-keyed)")
    if family == "wan":
        return WanKeyConverter()
    if family == "flux":
        return FluxKeyConverter()
    if family == "hunyuanvideo15":
        return Hunyuan15KeyConverter()
    return NoOpKeyConverter()          # qwenimage files of the manifests are diffusers-keyed


# ---- LoRA state dicts -----------------------------------------------------------------------------------------------------
_KOHYA_KEEP = ("diffusion_model", "double_blocks", "single_blocks", "transformer_blocks", "img_attn", "txt_attn", "img_mod",
               "txt_mod", "img_mlp", "txt_mlp", "query_norm", "key_norm", "time_embed", "time_embedding", "pos_embed",
               "proj_in", "proj_out")


def kohya_unflatten(prefix: str) -> str:
    """Kohya's single-file format flattens the dots of a module path into underscores; module names with REAL underscores
    (`double_blocks`, `img_attn`, `linear_1` ...) are shielded before the remaining underscores become dots
    (R/src/lora/lora_converter.py:16-70)."""
    mark = "\x01"          # never occurs in a state-dict key
    for tok in sorted(_KOHYA_KEEP, key=len, reverse=True):
        if tok in prefix:
            prefix = prefix.replace(tok, tok.replace("_", mark))
    for word in ("linear", "conv", "norm"):
        prefix = re.sub(word + r"_(\d+)", lambda m, word=word: f"{word}{mark}{m.group(1)}", prefix)
    return prefix.replace("_", ".").replace(mark, "_")


def kohya_to_peft(sd: Dict[str, Any]) -> Dict[str, Any]:
    """In place: `lora_unet_double_blocks_0_img_attn_qkv.lora_down.weight` -> `unet.double_blocks.0.img_attn.qkv.lora_A.weight`
    (R/src/lora/lora_converter.py:185-255; `.alpha` entries keep their place next to the factors and are folded later)."""
    items = list(sd.items())
    sd.clear()
    for k, v in items:
        for head, full in (("lora_te2.", "text_encoder_2."), ("lora_te1.", "text_encoder."), ("lora_unet", "unet")):
            if k.startswith(head):
                k = k.replace(head, full, 1)
                break
        k = k.replace("dora_scale", "lora_magnitude_vector")
        last = k.rfind(".")
        if last != -1:
            cut = k.rfind(".", 0, last)
            cut = cut if cut != -1 else last          # keep the last two dotted tokens (`.lora_down.weight`), or `.alpha`
            k = kohya_unflatten(k[:cut]) + k[cut:]
        sd[k.replace(".lora_down", ".lora_A").replace(".lora_up", ".lora_B")] = v
    return sd
