"""LoRA adapters merged into the device weights.

The reference applies LoRAs through diffusers' PEFT backend at transformer load
(`R/src/engine/base_engine.py:1303-1318`, `apply_loras` :2467-2512 -> `LoraManager.load_into`
`R/src/lora/manager.py:454-606`): the state dict is normalised to PEFT keys by `LoraConverter.convert`
(`R/src/lora/lora_converter.py:80-183`, alpha folded into the factors by `scale_alpha` :152-164), injected with
`lora_alpha = r` (manager.py:398-452, so PEFT's own scaling is 1) and activated with
`model.set_adapters(names, weights=scales)`; the runtime then computes `base(x) + scale * B(A(x))` per Linear.

Here the adapters never run as separate GEMMs: with 288 GB of HBM the touched base weights are kept beside the
merged ones, and every `set_adapters` recomputes

    W = base + sum_i scale_i * B_i A_i

as ONE GEMM per weight on the HIP path (`apexmi_gemm_bf16`, gate/residual epilogue: the adapters' factors are
concatenated along the rank, the base weight is the residual), so a weight is rounded to bf16 once no matter how
many adapters are active and the denoise step runs the unchanged kernels.  The model classes mix in
`LoraAdapterMixin`, which mirrors the PEFT surface the reference calls (`load_lora_adapter`, `set_adapters`,
`delete_adapters`, `disable_lora` / `enable_lora`, `unload_lora_weights`).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

PREFIXES = ("transformer.", "diffusion_model.", "model.", "unet.")  # R/src/lora/manager.py:383-396
_DOWN = {"lora_down": "lora_A", "lora_up": "lora_B"}                 # BASE_TO_PEFT, lora_converter.py:11-14


def alpha_scales(rank: int, alpha: float) -> Tuple[float, float]:
    """Split alpha / rank into (scale_down, scale_up) by powers of two, as the reference does
    (`LoraConverter.get_alpha_scales`, lora_converter.py:138-150)."""
    scale = alpha / rank
    scale_down, scale_up = scale, 1.0
    while scale_down * 2 < scale_up:
        scale_down *= 2
        scale_up /= 2
    return scale_down, scale_up


def state_dict_type(keys: Iterable[str]) -> str:
    """Format detection in the reference's order (lora_converter.py:98-136)."""
    keys = list(keys)
    if any(p in k for k in keys for p in ("lora_te1.", "lora_te2.", "lora_unet", "dora_scale")):
        return "kohya_ss"
    if any(p in k for k in keys for p in ("lora_down", "lora_up")):
        return "base"
    if any(p in k for k in keys for p in (".to_q_lora", ".to_k_lora", ".to_v_lora", ".to_out_lora")):
        return "diffusers_old"
    if any(p in k for k in keys for p in (".lora_linear_layer.up", ".lora_linear_layer.down")):
        return "diffusers"
    return "peft"


def normalize_lora_state_dict(sd: Dict[str, torch.Tensor], model_keys: Optional[Sequence[str]] = None
                              ) -> Dict[str, torch.Tensor]:
    """PEFT-keyed copy of `sd` with alpha folded into the factors: what `LoraConverter().convert(sd, model_keys)` returns
    (lora_converter.py:166-183) for the PEFT, lora_down / lora_up ("base") and Kohya single-file formats.  The two legacy
    diffusers formats are rejected: their rename tables (DIFFUSERS_TO_PEFT, DIFFUSERS_OLD_TO_PEFT) live in the absent
    diffusers package."""
    from .converters import KeyConverter, kohya_to_peft
    kind = state_dict_type(sd.keys())
    out: Dict[str, torch.Tensor] = dict(sd)
    if kind == "kohya_ss":
        kohya_to_peft(out)
    elif kind == "base":
        conv = KeyConverter()
        conv.rename = dict(_DOWN)
        conv.post = {".diff_b": conv.drop, ".diff": conv.drop, "scaled_fp8": conv.drop}   # special_keys_map :92-96
        conv.convert(out, model_keys)
    elif kind != "peft":
        raise ValueError(f"LoRA state dict format '{kind}' is not supported by the MI355X backend (its rename table lives in "
                         f"diffusers); convert it to PEFT (lora_A / lora_B) keys first")
    for k in [k for k in out if ".alpha" in k]:                 # scale_alpha :152-164
        down_key, up_key = k.replace(".alpha", ".lora_A.weight"), k.replace(".alpha", ".lora_B.weight")
        if down_key in out and up_key in out:
            sdn, sup = alpha_scales(out[down_key].shape[0], float(out[k].item()))
            out[down_key] = out[down_key] * sdn
            out[up_key] = out[up_key] * sup
    return out


def convert_lora_state_dict(sd: Dict[str, torch.Tensor], model_base: str = "", model_keys: Optional[Sequence[str]] = None
                            ) -> Dict[str, torch.Tensor]:
    """`LoraManager.maybe_convert_state_dict` (manager.py:633-644): normalise to PEFT keys, run the MODEL's key converter
    over the LoRA keys (so `diffusion_model.blocks.N.self_attn.q.lora_down.weight` of the lightx2v files, or a fused
    `double_blocks.N.img_attn.qkv.lora_up` of a BFL-keyed Flux LoRA, land on `blocks.N.attn1.to_q.lora_A.weight` /
    per-projection factors), then strip wrapper prefixes against the model's keys."""
    from .converters import KeyConverter, get_transformer_converter
    out = normalize_lora_state_dict(sd, model_keys)
    get_transformer_converter(model_base).convert(out, model_keys)
    KeyConverter().strip_prefixes(out, model_keys)
    return out


def split_modules(sd: Dict[str, torch.Tensor], prefix: Optional[str] = None
                  ) -> Dict[str, Dict[str, torch.Tensor]]:
    """{module path: {"A": [r, in], "B": [out, r], "bias": [out]?}} from a normalised state dict.  `prefix`
    (e.g. "transformer") is removed as `load_lora_adapter(prefix=...)` does; with prefix=None a unanimous known
    prefix is removed (manager.py:383-396)."""
    keys = [k for k in sd if ".lora_" in k]
    if prefix is None:
        for p in PREFIXES:
            if keys and all(k.startswith(p) for k in keys):
                prefix = p[:-1]
                break
    mods: Dict[str, Dict[str, torch.Tensor]] = {}
    for k in keys:
        v = sd[k]
        name = k[len(prefix) + 1:] if prefix and k.startswith(prefix + ".") else k
        if name.endswith(".lora_A.weight"):
            mods.setdefault(name[:-len(".lora_A.weight")], {})["A"] = v
        elif name.endswith(".lora_B.weight"):
            mods.setdefault(name[:-len(".lora_B.weight")], {})["B"] = v
        elif name.endswith(".lora_B.bias"):
            mods.setdefault(name[:-len(".lora_B.bias")], {})["bias"] = v
        else:
            raise ValueError(f"unsupported LoRA tensor '{k}' (DoRA / embedding adapters are not implemented)")
    for m, d in mods.items():
        if "A" not in d or "B" not in d:
            raise ValueError(f"LoRA module '{m}' lacks lora_A or lora_B")
        if d["A"].dim() != 2 or d["B"].dim() != 2 or d["A"].shape[0] != d["B"].shape[1]:
            raise ValueError(f"LoRA module '{m}': only Linear adapters are supported, got A {tuple(d['A'].shape)} "
                             f"B {tuple(d['B'].shape)}")
    return mods


def merge_weight_(weight: torch.Tensor, base: torch.Tensor, adapters: Sequence[Tuple[torch.Tensor, torch.Tensor, float]]
                  ) -> None:
    """weight[out, in] (bf16, on the GPU, may be a row-range view of a packed matrix) = base + sum_i s_i B_i A_i.
    One `apexmi_gemm_bf16` launch: X = [s_1 B_1 | s_2 B_2 | ..] ([out, R]), W = [A_1; A_2; ..]^T ([in, R]),
    R = sum of ranks padded to 64, epilogue gate (= 1) * X W^T + base."""
    from . import ops
    dev = weight.device
    n_out, n_in = weight.shape
    ranks = [int(a.shape[0]) for a, _, _ in adapters]
    R = max(64, (sum(ranks) + 63) // 64 * 64)
    x = torch.zeros((n_out, R), dtype=torch.float32, device=dev)
    w = torch.zeros((n_in, R), dtype=torch.float32, device=dev)
    c = 0
    for (a, b, s), r in zip(adapters, ranks):
        if tuple(a.shape) != (r, n_in) or tuple(b.shape) != (n_out, r):
            raise ValueError(f"LoRA factors {tuple(b.shape)} x {tuple(a.shape)} do not fit weight {tuple(weight.shape)}")
        x[:, c:c + r] = b.to(dev, torch.float32) * float(s)
        w[:, c:c + r] = a.to(dev, torch.float32).t()
        c += r
    gate = torch.ones(n_in, dtype=torch.float32, device=dev)
    ops.gemm(x.to(torch.bfloat16), w.to(torch.bfloat16), None, out=weight, epilogue="gate_res", gate=gate,
             residual=base)


class LoraAdapterMixin:
    """PEFT-shaped adapter surface of the drop-in model classes; every change re-merges the touched weights."""

    def _lora_state(self):
        if not hasattr(self, "_lora_adapters"):
            self._lora_adapters: Dict[str, Dict[str, Dict[str, torch.Tensor]]] = {}   # name -> module -> factors
            self._lora_scales: Dict[str, float] = {}
            self._lora_base: Dict[str, torch.Tensor] = {}        # parameter name -> untouched copy
            self._lora_enabled = True
        return self._lora_adapters

    def _lora_param(self, module: str, what: str) -> torch.Tensor:
        sd = dict(self.named_parameters())
        key = f"{module}.{what}"
        if key not in sd:
            raise KeyError(f"LoRA targets '{key}', which is not a parameter of {type(self).__name__}")
        return sd[key]

    def load_lora_adapter(self, state_dict: Dict[str, torch.Tensor], adapter_name: str = "default",
                          prefix: Optional[str] = None, metadata: Optional[dict] = None, activate: bool = True,
                          **_ignored) -> None:
        """Register one adapter and (as PEFT does) activate it with weight 1.  Mirrors
        `PeftAdapterMixin.load_lora_adapter` as the reference calls it (manager.py:571-585); `activate=False`
        defers the merge to the `set_adapters` call that follows."""
        ads = self._lora_state()
        if adapter_name in ads:
            raise ValueError(f"adapter '{adapter_name}' is already loaded")
        keys = [k for k, _ in self.named_parameters()]
        mods = split_modules(convert_lora_state_dict(state_dict, getattr(self, "_converter_base", ""), keys), prefix)
        if not mods:
            raise ValueError("no LoRA tensors found in the state dict")
        for m, d in mods.items():                      # validate before touching anything
            w = self._lora_param(m, "weight")
            wshape = getattr(w, "_fp8_shape", None) or tuple(w.shape)      # a resident-fp8 parameter has no bf16 storage (wan.py)
            if tuple(d["B"].shape[:1]) + tuple(d["A"].shape[1:]) != tuple(wshape):
                raise ValueError(f"adapter '{adapter_name}', module '{m}': delta {d['B'].shape[0]}x{d['A'].shape[1]} "
                                 f"does not fit weight {tuple(wshape)}")
        ads[adapter_name] = mods
        self._lora_scales[adapter_name] = 1.0 if activate else 0.0
        if activate:
            self._lora_remerge(mods.keys())

    def set_adapters(self, adapter_names, weights=None) -> None:
        """Activate exactly `adapter_names` with `weights` (default 1.0); others get scale 0 (manager.py:588-597)."""
        ads = self._lora_state()
        names = [adapter_names] if isinstance(adapter_names, str) else list(adapter_names)
        if weights is None:
            weights = [1.0] * len(names)
        elif not isinstance(weights, (list, tuple)):
            weights = [weights] * len(names)
        for n in names:
            if n not in ads:
                raise ValueError(f"adapter '{n}' is not loaded")
        touched = set()
        for n in ads:
            new = float(weights[names.index(n)]) if n in names else 0.0
            if new != self._lora_scales[n]:
                touched.update(ads[n].keys())
            self._lora_scales[n] = new
        self._lora_remerge(touched)

    def delete_adapters(self, adapter_names) -> None:
        ads = self._lora_state()
        names = [adapter_names] if isinstance(adapter_names, str) else list(adapter_names)
        touched = set()
        for n in names:
            touched.update(ads.pop(n).keys())
            self._lora_scales.pop(n)
        self._lora_remerge(touched)

    def disable_lora(self) -> None:
        self._lora_state()
        self._lora_enabled = False
        self._lora_remerge({m for mods in self._lora_adapters.values() for m in mods})

    def enable_lora(self) -> None:
        self._lora_state()
        self._lora_enabled = True
        self._lora_remerge({m for mods in self._lora_adapters.values() for m in mods})

    def unload_lora_weights(self) -> None:
        self.delete_adapters(list(self._lora_state().keys()))

    def active_adapters(self) -> List[str]:
        self._lora_state()
        return [n for n, s in self._lora_scales.items() if s != 0.0]

    @torch.no_grad()
    def _lora_remerge(self, modules: Iterable[str]) -> None:
        for m in sorted(set(modules)):
            w = self._lora_param(m, "weight")
            if not w.is_cuda:
                raise RuntimeError("LoRA merging runs on the GPU: move the model to the device first "
                                   "(there is no CPU path)")
            wkey = f"{m}.weight"
            if wkey not in self._lora_base:
                self._lora_base[wkey] = w.detach().clone()
            active = [(d[m]["A"], d[m]["B"], self._lora_scales[n]) for n, d in self._lora_adapters.items()
                      if m in d and self._lora_scales[n] != 0.0 and self._lora_enabled]
            if active:
                merge_weight_(w.data, self._lora_base[wkey], active)
            else:
                w.data.copy_(self._lora_base[wkey])
            # bias deltas (lora_bias=True adapters): b = base_b + sum s_i * bias_i, f32 sum rounded once
            bias_ad = [(d[m]["bias"], self._lora_scales[n]) for n, d in self._lora_adapters.items()
                       if m in d and "bias" in d[m]]
            if bias_ad:
                b = self._lora_param(m, "bias")
                bkey = f"{m}.bias"
                if bkey not in self._lora_base:
                    self._lora_base[bkey] = b.detach().clone()
                acc = self._lora_base[bkey].float()
                if self._lora_enabled:
                    for bv, s in bias_ad:
                        acc = acc + float(s) * bv.to(b.device, torch.float32)
                b.data.copy_(acc.to(b.dtype))
            if not any(m in d for d in self._lora_adapters.values()):   # nothing refers to it any more
                self._lora_base.pop(wkey, None)
                self._lora_base.pop(f"{m}.bias", None)


def load_lora_file(path: str) -> Dict[str, torch.Tensor]:
    """safetensors / torch checkpoint -> state dict (manager.py:806-810)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu", weights_only=True)


def apply_loras(model, loras: Sequence, adapter_names: Optional[List[str]] = None,
                scales: Optional[List[float]] = None) -> List[str]:
    """Engine-side entry mirroring `BaseEngine.apply_loras` / `LoraManager.load_into`: each entry is a path, a
    state dict, or (path | state dict, scale); `scales` overrides per entry; zero-scale entries are skipped;
    repeated adapter names keep the last scale.  Returns the adapter names in activation order."""
    final: Dict[str, float] = {}
    for i, entry in enumerate(loras):
        src, scale = (entry if isinstance(entry, tuple) else (entry, 1.0))
        if scales is not None and i < len(scales) and scales[i] is not None:
            scale = float(scales[i])
        name = adapter_names[i] if adapter_names and i < len(adapter_names) else f"lora_{i}"
        if float(scale) == 0.0:
            continue
        if name not in model._lora_state():
            sd = load_lora_file(src) if isinstance(src, str) else src
            model.load_lora_adapter(sd, adapter_name=name, activate=False)
        final[name] = float(scale)
    if final:
        model.set_adapters(list(final.keys()), weights=list(final.values()))
    return list(final.keys())


class EngineLoraMixin:
    """`engine.apply_loras(...)` with the reference's signature (`BaseEngine.apply_loras`,
    R/src/engine/base_engine.py:2467-2512): `model_name_or_type` picks the component attribute on the engine
    ("transformer"; Wan's experts are "high_noise_transformer" / "low_noise_transformer", `transformer` and
    `transformer_2` being accepted aliases), or pass the module as `model`."""

    _LORA_ALIASES = {"transformer": ("transformer", "high_noise_transformer"),
                     "transformer_2": ("low_noise_transformer",)}

    def apply_loras(self, loras, adapter_names=None, scales=None, model_name_or_type: str = "transformer", model=None):
        if model is None:
            for attr in self._LORA_ALIASES.get(model_name_or_type, (model_name_or_type,)):
                model = getattr(self, attr, None)
                if model is not None:
                    break
        if model is None:
            raise RuntimeError(f"engine has no component '{model_name_or_type}' to apply LoRAs to")
        names = apply_loras(model, loras, adapter_names=adapter_names, scales=scales)
        self.loaded_loras = getattr(self, "loaded_loras", {})
        for n in names:
            self.loaded_loras[n] = model._lora_scales[n]
        return names
