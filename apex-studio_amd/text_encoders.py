"""Text encoders on the MI355X HIP ops: drop-ins for the `transformers` classes the reference resolves by name
(text_encoder/text_encoder.py:24-82 — manifest `base: T5EncoderModel | UMT5EncoderModel |
CLIPTextModel` with `module_name`, instantiated by LoaderMixin._load_model through `_from_config(config_dict)`,
mixins/loader_mixin.py:176-257, and called at text_encoder.py:335-342 as
`model(input_ids=..., attention_mask=..., output_hidden_states=...)`).

Same state-dict keys as the checkpoints the manifests point at (transformers 4.57 layout: `shared.weight`,
`encoder.block.N.layer.0.SelfAttention.{q,k,v,o}.weight`, ... / `text_model.embeddings.token_embedding.weight`, ...),
same outputs (`last_hidden_state`, `hidden_states`, `pooler_output`).  bf16 on a ROCm device only — no CPU fallback.

Per layer: RMS / Layer norm (`apexmi_ln_modulate`), ONE fused QKV GEMM, per-head attention in three launches
(`apexmi_attn_fwd_bias`: batched scores GEMM with f32 output, row softmax with the relative-position bias / padding /
causal mask, batched P V GEMM), output projection with the residual in the GEMM epilogue, feed-forward GEMMs with the
activation in the epilogue.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict

import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .flux import _Config


def _cfg_dict(config, kwargs) -> dict:
    if config is None:
        cfg = {}
    elif isinstance(config, dict):
        cfg = dict(config)
    elif hasattr(config, "to_dict"):
        cfg = dict(config.to_dict())
    else:
        cfg = dict(vars(config))
    cfg.update(kwargs)
    return cfg


class _W(nn.Module):
    def __init__(self, cout, cin, bias, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, **kw), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(cout, **kw), requires_grad=False)
        else:
            self.bias = None


class _N(nn.Module):
    def __init__(self, dim, bias, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, **kw), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(dim, **kw), requires_grad=False)


class _Emb(nn.Module):
    def __init__(self, n, dim, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim, **kw), requires_grad=False)


class _Base(nn.Module):
    """Common surface: from_config, dtype/device, fused-weight cache invalidation, the no-CPU-fallback check."""

    @classmethod
    def from_config(cls, config=None, **kwargs):
        return cls(config, **kwargs)

    _from_config = from_config

    storage_dtype = torch.bfloat16

    def set_storage_dtype(self, dtype: torch.dtype):
        """Verification mode (flux.py `set_storage_dtype`): float32 keeps every activation between the kernels in f32 (GEMMs through
        the exact bf16 split, attention through the f32 row kernel) so the bf16-weight encoder can be compared with the fp32
        reference at ~1e-6 per element instead of the bf16 rounding floor.  Weights stay bf16.  Not a production path."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("storage dtype is bfloat16 (production) or float32 (verification)")
        self.storage_dtype = dtype
        return self

    def _first(self):
        return next(self.parameters())

    @property
    def dtype(self):
        return self._first().dtype

    @property
    def device(self):
        return self._first().device

    def _apply(self, fn, *a, **k):
        self._fused: Dict[int, tuple] = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._fused = {}
        return super().load_state_dict(*a, **k)

    def _weights_changed(self):
        """Parameters were written in place (`weights.load_checkpoint_into`): drop the fused / padded copies."""
        self._fused = {}

    def _check(self, input_ids):
        if self.device.type != "cuda" or self.dtype != torch.bfloat16:
            raise _l.ApexMIError(f"{type(self).__name__} (mi355) needs bf16 weights on a ROCm device (no CPU fallback)")
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [batch, sequence]")

    def _qkv(self, key, mods):
        """Fused [3 inner, d] projection weight (and bias) of one attention layer, built once."""
        f = self._fused.get(key)
        if f is None:
            w = torch.cat([m.weight.data for m in mods], dim=0).contiguous()
            b = torch.cat([m.bias.data for m in mods], dim=0).contiguous() if mods[0].bias is not None else None
            f = (w, b)
            self._fused[key] = f
        return f

    def _ones(self, n):
        o = self._fused.get(("ones", n))
        if o is None:
            o = (torch.ones(n, dtype=torch.float32, device=self.device),)
            self._fused[("ones", n)] = o
        return o[0]


# ---- T5 / UMT5 ------------------------------------------------------------------------------------------------

def _t5_buckets(S: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """T5Attention._relative_position_bucket (bidirectional) for every distance j - i in [-(S-1), S-1]: host integer /
    log arithmetic over 2S-1 values, in the float32 expression order of modeling_t5.py."""
    rp = torch.arange(-(S - 1), S, dtype=torch.long)
    nb = num_buckets // 2
    buckets = (rp > 0).to(torch.long) * nb
    rp = rp.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (buckets + torch.where(rp < max_exact, rp, large)).to(torch.int32)


class _T5Block(nn.Module):
    def __init__(self, c, has_bias, **kw):
        super().__init__()
        inner = c.d_kv * c.num_heads
        sa, ff = nn.Module(), nn.Module()
        sa.SelfAttention = nn.Module()
        for n in ("q", "k", "v"):
            setattr(sa.SelfAttention, n, _W(inner, c.d_model, False, **kw))
        sa.SelfAttention.o = _W(c.d_model, inner, False, **kw)
        if has_bias:
            sa.SelfAttention.relative_attention_bias = _Emb(c.relative_attention_num_buckets, c.num_heads, **kw)
        sa.layer_norm = _N(c.d_model, False, **kw)
        ff.DenseReluDense = nn.Module()
        ff.DenseReluDense.wi_0 = _W(c.d_ff, c.d_model, False, **kw)
        ff.DenseReluDense.wi_1 = _W(c.d_ff, c.d_model, False, **kw)
        ff.DenseReluDense.wo = _W(c.d_model, c.d_ff, False, **kw)
        ff.layer_norm = _N(c.d_model, False, **kw)
        self.layer = nn.ModuleList([sa, ff])


class T5EncoderModel(_Base):
    """transformers.T5EncoderModel (T5 v1.1 "gated-gelu" feed-forward: Flux's text_encoder_2)."""

    per_layer_bias = False

    def __init__(self, config=None, device=None, dtype=torch.bfloat16, **kwargs):
        super().__init__()
        cfg = _cfg_dict(config, kwargs)
        c = self.config = _Config(
            vocab_size=cfg.get("vocab_size", 32128), d_model=cfg.get("d_model", 4096), d_kv=cfg.get("d_kv", 64),
            d_ff=cfg.get("d_ff", 10240), num_layers=cfg.get("num_layers", 24), num_heads=cfg.get("num_heads", 64),
            relative_attention_num_buckets=cfg.get("relative_attention_num_buckets", 32),
            relative_attention_max_distance=cfg.get("relative_attention_max_distance", 128),
            layer_norm_epsilon=cfg.get("layer_norm_epsilon", 1e-6),
            feed_forward_proj=cfg.get("feed_forward_proj", "gated-gelu"))
        if c.feed_forward_proj != "gated-gelu":
            raise NotImplementedError(f"t5 (mi355): feed_forward_proj={c.feed_forward_proj!r}; the shipped encoders "
                                      "(T5 v1.1 XXL, UMT5 XXL) are gated-gelu")
        kw = dict(device=device, dtype=dtype)
        self.shared = _Emb(c.vocab_size, c.d_model, **kw)
        self.encoder = nn.Module()
        self.encoder.embed_tokens = self.shared                        # tied, as in transformers
        self.encoder.block = nn.ModuleList([_T5Block(c, self.per_layer_bias or i == 0, **kw)
                                            for i in range(c.num_layers)])
        self.encoder.final_layer_norm = _N(c.d_model, False, **kw)
        self._fused = {}

    def _bias(self, att, S):
        c = self.config
        bucket = _t5_buckets(S, c.relative_attention_num_buckets, c.relative_attention_max_distance).to(self.device)
        return ops.relpos_bias(att.relative_attention_bias.weight.data.contiguous(), bucket, S, S)

    @ops.on_model_device
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, output_hidden_states=False, return_dict=True, **_):
        self._check(input_ids)
        c = self.config
        B, S = input_ids.shape
        H, inner, eps = c.num_heads, c.num_heads * c.d_kv, c.layer_norm_epsilon
        ids = input_ids.to(self.device, torch.int64).reshape(-1).contiguous()
        if int(ids.min()) < 0 or int(ids.max()) >= c.vocab_size:
            raise IndexError("input_ids out of range for the embedding table")
        keep = None
        if attention_mask is not None:
            keep = (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous()
        x = ops.gather_rows(self.shared.weight.data, ids, out_dtype=self.storage_dtype)
        ones = self._ones(c.d_model)
        hidden, bias = [], None
        for blk in self.encoder.block:
            if output_hidden_states:
                hidden.append(x.view(B, S, -1))
            sa, ff = blk.layer
            att = sa.SelfAttention
            if hasattr(att, "relative_attention_bias"):
                bias = self._bias(att, S)
            wqkv, _ = self._qkv(id(att), (att.q, att.k, att.v))
            qkv = ops.gemm(ops.ln_modulate(x, gamma=sa.layer_norm.weight.data, rms=True, eps=eps), wqkv)
            a = torch.empty((B * S, inner), dtype=x.dtype, device=x.device)
            for b in range(B):
                r = slice(b * S, (b + 1) * S)
                ops.attention_bias(qkv[r, :inner], qkv[r, inner:2 * inner], qkv[r, 2 * inner:], H, 1.0, bias=bias,
                                   keep=None if keep is None else keep[b], out=a[r])
            x = ops.gemm(a, att.o.weight.data, epilogue="gate_res", gate=ones, residual=x)
            d = ff.DenseReluDense
            h = ops.ln_modulate(x, gamma=ff.layer_norm.weight.data, rms=True, eps=eps)
            h = ops.mul(ops.gemm(h, d.wi_0.weight.data, epilogue="gelu"), ops.gemm(h, d.wi_1.weight.data))
            x = ops.gemm(h, d.wo.weight.data, epilogue="gate_res", gate=ones, residual=x)
        x = ops.ln_modulate(x, gamma=self.encoder.final_layer_norm.weight.data, rms=True, eps=eps).view(B, S, -1)
        if output_hidden_states:
            hidden.append(x)
        out = SimpleNamespace(last_hidden_state=x, hidden_states=tuple(hidden) if output_hidden_states else None)
        return out if return_dict else (x,) + ((out.hidden_states,) if output_hidden_states else ())


class UMT5EncoderModel(T5EncoderModel):
    """transformers.UMT5EncoderModel (Wan's text encoder): every block owns its relative_attention_bias."""

    per_layer_bias = True

    def __init__(self, config=None, device=None, dtype=torch.bfloat16, **kwargs):
        cfg = _cfg_dict(config, kwargs)
        cfg.setdefault("vocab_size", 256384)
        super().__init__(cfg, device=device, dtype=dtype)


# ---- CLIP text ------------------------------------------------------------------------------------------------

class _CLIPLayer(nn.Module):
    def __init__(self, d, inter, **kw):
        super().__init__()
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, _W(d, d, True, **kw))
        self.layer_norm1, self.layer_norm2 = _N(d, True, **kw), _N(d, True, **kw)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = _W(inter, d, True, **kw), _W(d, inter, True, **kw)


class CLIPTextModel(_Base):
    """transformers.CLIPTextModel (Flux's pooled-prompt encoder): causal pre-LN encoder, EOS pooling."""

    def __init__(self, config=None, device=None, dtype=torch.bfloat16, **kwargs):
        super().__init__()
        cfg = _cfg_dict(config, kwargs)
        c = self.config = _Config(
            vocab_size=cfg.get("vocab_size", 49408), hidden_size=cfg.get("hidden_size", 768),
            intermediate_size=cfg.get("intermediate_size", 3072), num_hidden_layers=cfg.get("num_hidden_layers", 12),
            num_attention_heads=cfg.get("num_attention_heads", 12),
            max_position_embeddings=cfg.get("max_position_embeddings", 77), layer_norm_eps=cfg.get("layer_norm_eps", 1e-5),
            hidden_act=cfg.get("hidden_act", "quick_gelu"), eos_token_id=cfg.get("eos_token_id", 2))
        if c.hidden_act not in ("quick_gelu", "gelu"):
            raise NotImplementedError(f"clip (mi355): hidden_act={c.hidden_act!r}")
        kw = dict(device=device, dtype=dtype)
        tm = self.text_model = nn.Module()
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = _Emb(c.vocab_size, c.hidden_size, **kw)
        tm.embeddings.position_embedding = _Emb(c.max_position_embeddings, c.hidden_size, **kw)
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList([_CLIPLayer(c.hidden_size, c.intermediate_size, **kw)
                                           for _ in range(c.num_hidden_layers)])
        tm.final_layer_norm = _N(c.hidden_size, True, **kw)
        self._fused = {}

    @ops.on_model_device
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, output_hidden_states=False, return_dict=True, **_):
        self._check(input_ids)
        c, tm = self.config, self.text_model
        B, S = input_ids.shape
        if S > c.max_position_embeddings:
            raise ValueError(f"sequence length {S} exceeds max_position_embeddings {c.max_position_embeddings}")
        H, d, eps = c.num_attention_heads, c.hidden_size, c.layer_norm_eps
        ids2 = input_ids.to(self.device, torch.int64)
        ids = ids2.reshape(-1).contiguous()
        if int(ids.min()) < 0 or int(ids.max()) >= c.vocab_size:
            raise IndexError("input_ids out of range for the embedding table")
        keep = None
        if attention_mask is not None:
            keep = (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous()
        x = ops.gather_rows(tm.embeddings.token_embedding.weight.data, ids,
                            pos=tm.embeddings.position_embedding.weight.data[:S], out_dtype=self.storage_dtype)
        ones = self._ones(d)
        act = "quick_gelu" if c.hidden_act == "quick_gelu" else "gelu_erf"
        hidden = [x.view(B, S, d)] if output_hidden_states else []
        for layer in tm.encoder.layers:
            at = layer.self_attn
            wqkv, bqkv = self._qkv(id(at), (at.q_proj, at.k_proj, at.v_proj))
            h = ops.ln_modulate(x, gamma=layer.layer_norm1.weight.data, beta=layer.layer_norm1.bias.data, eps=eps)
            qkv = ops.gemm(h, wqkv, bqkv)
            a = torch.empty((B * S, d), dtype=x.dtype, device=x.device)
            for b in range(B):
                r = slice(b * S, (b + 1) * S)
                ops.attention_bias(qkv[r, :d], qkv[r, d:2 * d], qkv[r, 2 * d:], H, (d // H) ** -0.5,
                                   keep=None if keep is None else keep[b], causal=True, out=a[r])
            x = ops.gemm(a, at.out_proj.weight.data, at.out_proj.bias.data, epilogue="gate_res", gate=ones, residual=x)
            h = ops.ln_modulate(x, gamma=layer.layer_norm2.weight.data, beta=layer.layer_norm2.bias.data, eps=eps)
            h = ops.gemm(h, layer.mlp.fc1.weight.data, layer.mlp.fc1.bias.data, epilogue=act)
            x = ops.gemm(h, layer.mlp.fc2.weight.data, layer.mlp.fc2.bias.data, epilogue="gate_res", gate=ones, residual=x)
            if output_hidden_states:
                hidden.append(x.view(B, S, d))
        last = ops.ln_modulate(x, gamma=tm.final_layer_norm.weight.data, beta=tm.final_layer_norm.bias.data,
                               eps=eps).view(B, S, d)
        if c.eos_token_id == 2:           # legacy configs (the shipped CLIP-L): the EOS token has the highest id
            idx = ids2.argmax(dim=-1)
        else:
            idx = (ids2 == c.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=last.device), idx]
        out = SimpleNamespace(last_hidden_state=last, pooler_output=pooled,
                              hidden_states=tuple(hidden) if output_hidden_states else None)
        return out if return_dict else (last, pooled) + ((out.hidden_states,) if output_hidden_states else ())
