"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the text encoders the reference instantiates BY CLASS NAME from the third-party
`transformers` package (pinned transformers==4.57.1, apps/api/requirements/requirements.txt:79; resolved in
apps/api/src/text_encoder/text_encoder.py:24-82 from the manifest `base:` — `T5EncoderModel` and `CLIPTextModel` for
Flux (manifest/image/flux-dev-text-to-image-1.0.0.v1.yml:62,77), `UMT5EncoderModel` for Wan 2.2
(manifest/video/wan-2.2-a14b-text-to-video-1.0.0.v1.yml:77)) and called at text_encoder.py:335-342.

The algorithm lives in that dependency, not in /root/reference; it is restated here from the published modeling code
(models/t5/modeling_t5.py, models/umt5/modeling_umt5.py, models/clip/modeling_clip.py) with the same state-dict keys.
Parity IS pinned: tests/golden/make_golden.py runs the `transformers` build installed in this container (5.15.0 — the
encoder arithmetic of these three classes is unchanged since 4.57) on small configs and saves inputs / outputs
(tests/golden/text_encoders.pt); tests/test_oracle_golden.py checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import FP32, Policy


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """T5Attention._relative_position_bucket, bidirectional (encoder) case."""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rp = relative_position.abs()
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, large)


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


class T5LayerNorm(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        var = x.float().pow(2).mean(-1, keepdim=True)
        return self.weight * (x * torch.rsqrt(var + self.eps))


class T5Attention(nn.Module):
    def __init__(self, d_model, d_kv, heads, has_bias, num_buckets, max_distance):
        super().__init__()
        inner = d_kv * heads
        self.heads, self.d_kv, self.num_buckets, self.max_distance = heads, d_kv, num_buckets, max_distance
        self.q, self.k, self.v = (nn.Linear(d_model, inner, bias=False) for _ in range(3))
        self.o = nn.Linear(inner, d_model, bias=False)
        self.relative_attention_bias = nn.Embedding(num_buckets, heads) if has_bias else None

    def compute_bias(self, S: int) -> torch.Tensor:
        pos = torch.arange(S)
        bucket = relative_position_bucket(pos[None, :] - pos[:, None], self.num_buckets, self.max_distance)
        return self.relative_attention_bias(bucket).permute(2, 0, 1).unsqueeze(0)          # [1, H, S, S]

    def forward(self, x, bias, keep, pol: Policy):
        B, S, _ = x.shape
        q, k, v = (pol.r(m(x)).view(B, S, self.heads, self.d_kv).transpose(1, 2) for m in (self.q, self.k, self.v))
        scores = q @ k.transpose(-1, -2) + bias                                             # no 1/sqrt(d) in T5
        if keep is not None:
            scores = scores.masked_fill(~keep[:, None, None, :], float("-inf"))
        p = pol.r(torch.softmax(scores.float(), dim=-1))
        return pol.r((p @ v).transpose(1, 2).reshape(B, S, -1))


class T5Block(nn.Module):
    def __init__(self, cfg, has_bias: bool):
        super().__init__()
        sa = nn.Module()
        sa.SelfAttention = T5Attention(cfg.d_model, cfg.d_kv, cfg.num_heads, has_bias, cfg.relative_attention_num_buckets,
                                       cfg.relative_attention_max_distance)
        sa.layer_norm = T5LayerNorm(cfg.d_model, cfg.layer_norm_epsilon)
        ff = nn.Module()
        ff.DenseReluDense = nn.Module()
        self.gated = cfg.feed_forward_proj.startswith("gated")
        if self.gated:
            ff.DenseReluDense.wi_0 = nn.Linear(cfg.d_model, cfg.d_ff, bias=False)
            ff.DenseReluDense.wi_1 = nn.Linear(cfg.d_model, cfg.d_ff, bias=False)
        else:
            ff.DenseReluDense.wi = nn.Linear(cfg.d_model, cfg.d_ff, bias=False)
        ff.DenseReluDense.wo = nn.Linear(cfg.d_ff, cfg.d_model, bias=False)
        ff.layer_norm = T5LayerNorm(cfg.d_model, cfg.layer_norm_epsilon)
        self.layer = nn.ModuleList([sa, ff])

    def forward(self, x, bias, keep, pol: Policy):
        sa, ff = self.layer
        a = sa.SelfAttention(pol.r(sa.layer_norm(x)), bias, keep, pol)
        x = pol.r(x + sa.SelfAttention.o(a))
        h = pol.r(ff.layer_norm(x))
        d = ff.DenseReluDense
        if self.gated:
            h = pol.r(pol.r(gelu_new(d.wi_0(h))) * pol.r(d.wi_1(h)))
        else:
            h = pol.r(F.relu(d.wi(h)))
        return pol.r(x + d.wo(h))


class T5EncoderModel(nn.Module):
    """T5EncoderModel (position bias owned by block 0 and shared) / UMT5EncoderModel (`per_layer_bias`: every block has
    its own relative_attention_bias — modeling_umt5.py UMT5Block)."""

    def __init__(self, vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64,
                 relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                 feed_forward_proj="gated-gelu", per_layer_bias=False, **_):
        super().__init__()
        self.cfg = SimpleNamespace(d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_heads=num_heads,
                                   relative_attention_num_buckets=relative_attention_num_buckets,
                                   relative_attention_max_distance=relative_attention_max_distance,
                                   layer_norm_epsilon=layer_norm_epsilon, feed_forward_proj=feed_forward_proj)
        self.per_layer_bias = per_layer_bias
        self.shared = nn.Embedding(vocab_size, d_model)
        self.encoder = nn.Module()
        self.encoder.embed_tokens = self.shared                     # tied, as in transformers
        self.encoder.block = nn.ModuleList([T5Block(self.cfg, per_layer_bias or i == 0) for i in range(num_layers)])
        self.encoder.final_layer_norm = T5LayerNorm(d_model, layer_norm_epsilon)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, policy: Policy = FP32):
        pol = policy
        x = pol.r(self.shared(input_ids))
        S = x.shape[1]
        keep = attention_mask.bool() if attention_mask is not None else None
        hidden = []
        bias = None
        for blk in self.encoder.block:
            hidden.append(x)
            att = blk.layer[0].SelfAttention
            if att.relative_attention_bias is not None:
                bias = att.compute_bias(S)
            x = blk(x, bias, keep, pol)
        x = pol.r(self.encoder.final_layer_norm(x))
        hidden.append(x)
        return SimpleNamespace(last_hidden_state=x, hidden_states=tuple(hidden))


class CLIPEncoderLayer(nn.Module):
    def __init__(self, d, heads, inter, eps, act):
        super().__init__()
        self.heads, self.act = heads, act
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, nn.Linear(d, d))
        self.layer_norm1, self.layer_norm2 = nn.LayerNorm(d, eps=eps), nn.LayerNorm(d, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, inter), nn.Linear(inter, d)

    def forward(self, x, keep, pol: Policy):
        B, S, d = x.shape
        a = self.self_attn
        h = pol.r(self.layer_norm1(x))
        q, k, v = (pol.r(m(h)).view(B, S, self.heads, d // self.heads).transpose(1, 2) for m in (a.q_proj, a.k_proj, a.v_proj))
        scores = q @ k.transpose(-1, -2) * (d // self.heads) ** -0.5
        mask = torch.ones(S, S, dtype=torch.bool).tril()[None, None]
        if keep is not None:
            mask = mask & keep[:, None, None, :]
        p = pol.r(torch.softmax(scores.masked_fill(~mask, float("-inf")).float(), dim=-1))
        o = pol.r((p @ v).transpose(1, 2).reshape(B, S, d))
        x = pol.r(x + a.out_proj(o))
        h = pol.r(self.layer_norm2(x))
        h = pol.r(self.act(self.mlp.fc1(h)))
        return pol.r(x + self.mlp.fc2(h))


class CLIPTextModel(nn.Module):
    """CLIPTextModel: token + learned position embeddings, pre-LN causal encoder, final LayerNorm, pooled output at the
    EOS token (argmax of the ids when eos_token_id == 2, the legacy convention of the shipped CLIP-L config; otherwise
    the first occurrence of eos_token_id) — modeling_clip.py CLIPTextTransformer.forward."""

    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5, hidden_act="quick_gelu",
                 eos_token_id=2, **_):
        super().__init__()
        act = {"quick_gelu": quick_gelu, "gelu": F.gelu}[hidden_act]
        self.eos_token_id = eos_token_id
        tm = self.text_model = nn.Module()
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = nn.Embedding(vocab_size, hidden_size)
        tm.embeddings.position_embedding = nn.Embedding(max_position_embeddings, hidden_size)
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList([CLIPEncoderLayer(hidden_size, num_attention_heads, intermediate_size,
                                                            layer_norm_eps, act) for _ in range(num_hidden_layers)])
        tm.final_layer_norm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, policy: Policy = FP32):
        pol, tm = policy, self.text_model
        B, S = input_ids.shape
        x = pol.r(tm.embeddings.token_embedding(input_ids) + tm.embeddings.position_embedding(torch.arange(S))[None])
        keep = attention_mask.bool() if attention_mask is not None else None
        hidden = [x]
        for layer in tm.encoder.layers:
            x = layer(x, keep, pol)
            hidden.append(x)
        last = pol.r(tm.final_layer_norm(x))
        if self.eos_token_id == 2:
            idx = input_ids.argmax(dim=-1)
        else:
            idx = (input_ids == self.eos_token_id).int().argmax(dim=-1)
        return SimpleNamespace(last_hidden_state=last, pooler_output=last[torch.arange(B), idx], hidden_states=tuple(hidden))
