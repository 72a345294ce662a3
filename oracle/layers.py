"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement (plain PyTorch) of the leaf layers the reference's denoise path takes from
the un-vendored `diffusers` dependency (reference pins `diffusers @ git+https://github.com/
huggingface/diffusers.git`, un-versioned main, apps/api/requirements/requirements.txt:15), written
from the published diffusers semantics summarised in SURVEY.md Appendix A and cross-checked against
the in-tree corroborating code cited per function.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.

Parity status: diffusers itself is absent from the container, but the reference tree carries its own copies of nearly
every leaf restated here, and those copies RUN here.  tests/golden/leaf_pins.pt (tests/golden/make_golden.py
`gen_leaf_pins`, file:line per leaf in its docstring) holds their outputs; tests/test_oracle_leaf_pins.py checks this
module against them: get_timestep_embedding (bit-identical), TimestepEmbedding, PixArtAlphaTextProjection, FeedForward /
GELU-tanh, RMSNorm, get_1d_rotary_pos_embed, apply_rotary_emb (both sequence dims), AdaLayerNormZero (projection, chunk
order, formula), the chunk orders of the Zero / ZeroSingle forms, and the [scale, shift] order of
AdaLayerNormContinuous (through the reference's checkpoint converter).  Also pinned by reference-run fixtures: the
attention operator (reference `sdpa`), the in-tree efficiency ops, and the reference's own Flux / Wan / Qwen block
wiring executed on top of these leaves; FP32LayerNorm is pinned to torch's own `F.layer_norm` in f32 and to the written-out
float64 definition (tests/test_oracle_leaf_pins.py::test_fp32_layernorm_is_torch_layer_norm_in_f32).  Left "parity unpinned"
(no copy in the tree): the CombinedTimestep*Embeddings containers (sums of pinned parts), `LinearActivation`.

`emulate_bf16`: the GPU path stores activations in bf16 between kernels and accumulates in f32.
`Policy.r(x)` rounds to bf16 at exactly those storage points so a like-for-like comparison is
possible (SURVEY.md §7 "tolerance vs precision policy"); with emulate_bf16=False it is the pure
fp32 restatement.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class Policy:
    emulate_bf16: bool = False

    def r(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.bfloat16).to(torch.float32) if self.emulate_bf16 else x


FP32 = Policy(False)
BF16_STORAGE = Policy(True)


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int, flip_sin_to_cos: bool = False,
                           downscale_freq_shift: float = 1.0, scale: float = 1.0,
                           max_period: int = 10000) -> torch.Tensor:
    """diffusers get_timestep_embedding; in-tree copy: reference
    transformer/qwenimage/base/model.py:46-97."""
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float,
                 scale: float = 1.0):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = (num_channels, flip_sin_to_cos,
                                                                downscale_freq_shift, scale)

    def forward(self, t):
        return get_timestep_embedding(t, self.num_channels, self.flip, self.shift, self.scale)


class TimestepEmbedding(nn.Module):
    """linear_2(silu(linear_1(x))) — MLX restatement: reference mlx/modules/embedding.py:8-139."""

    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class PixArtAlphaTextProjection(nn.Module):
    """linear_2(act(linear_1(x))); act = silu (Flux pooled text) or gelu-tanh (Wan text).
    Reference use: wan model.py:764-766; MLX restatement mlx/modules/layers.py:7-95."""

    def __init__(self, in_features: int, hidden_size: int, act_fn: str = "silu",
                 out_features: Optional[int] = None):
        super().__init__()
        out_features = out_features or hidden_size
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, out_features)
        self.act_fn = act_fn

    def forward(self, x):
        h = self.linear_1(x)
        h = F.silu(h) if self.act_fn == "silu" else F.gelu(h, approximate="tanh")
        return self.linear_2(h)


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    """time_text_embed of Flux-dev (guidance_embeds=True); call site flux model.py:430-437,539-543."""

    def __init__(self, embedding_dim: int, pooled_projection_dim: int):
        super().__init__()
        self.time_proj = Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.guidance_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, "silu")

    def forward(self, timestep, guidance, pooled_projection):
        t_emb = self.timestep_embedder(self.time_proj(timestep).to(pooled_projection.dtype))
        g_emb = self.guidance_embedder(self.time_proj(guidance).to(pooled_projection.dtype))
        return t_emb + g_emb + self.text_embedder(pooled_projection)


class CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim: int, pooled_projection_dim: int):
        super().__init__()
        self.time_proj = Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, "silu")

    def forward(self, timestep, pooled_projection):
        t_emb = self.timestep_embedder(self.time_proj(timestep).to(pooled_projection.dtype))
        return t_emb + self.text_embedder(pooled_projection)


class AdaLayerNormZero(nn.Module):
    """emb = linear(silu(temb)); shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp =
    chunk(6); x = LN(x)(1+scale_msa)+shift_msa.  Corroboration: reference
    transformer/chroma/base/model.py:59-135, transformer/hunyuanvideo/base/model.py:98-145."""

    def __init__(self, embedding_dim: int, num_embeddings=None, norm_type: str = "layer_norm", bias: bool = True):
        super().__init__()
        assert num_embeddings is None and norm_type == "layer_norm" and bias
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        emb = self.linear(F.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    def __init__(self, embedding_dim: int):
        super().__init__()
        self.linear = nn.Linear(embedding_dim, 3 * embedding_dim)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        emb = self.linear(F.silu(emb))
        shift_msa, scale_msa, gate_msa = emb.chunk(3, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa


class AdaLayerNormContinuous(nn.Module):
    """scale FIRST then shift (reference converters/utils.py:82-85 `swap_scale_shift`)."""

    def __init__(self, embedding_dim: int, conditioning_embedding_dim: int,
                 elementwise_affine: bool = False, eps: float = 1e-6):
        super().__init__()
        self.linear = nn.Linear(conditioning_embedding_dim, 2 * embedding_dim)
        self.norm = nn.LayerNorm(embedding_dim, eps=eps, elementwise_affine=elementwise_affine)

    def forward(self, x, conditioning_embedding):
        emb = self.linear(F.silu(conditioning_embedding).to(x.dtype))
        scale, shift = emb.chunk(2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class GELUProj(nn.Module):
    """diffusers activations.GELU(dim_in, dim_out, approximate='tanh'): proj then gelu-tanh.
    MLX restatement: reference mlx/modules/act.py:32-41."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class LinearActivation(nn.Module):
    """diffusers `LinearActivation(dim_in, dim_out, activation="silu")`: act(proj(x)); the "linear-silu" FeedForward
    of the HunyuanVideo-1.5 token refiner (reference transformer/hunyuanvideo15/base/model.py:296-301)."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.silu(self.proj(x))


class FeedForward(nn.Module):
    """net.0 = GELU-tanh proj ("gelu-approximate") or Linear+SiLU ("linear-silu"), net.1 = Dropout(0),
    net.2 = Linear (biases on)."""

    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4,
                 inner_dim: Optional[int] = None, activation_fn: str = "gelu-approximate", dropout: float = 0.0):
        super().__init__()
        assert activation_fn in ("gelu-approximate", "linear-silu") and dropout == 0.0
        inner = inner_dim if inner_dim is not None else int(dim * mult)
        act = GELUProj(dim, inner) if activation_fn == "gelu-approximate" else LinearActivation(dim, inner)
        self.net = nn.ModuleList([act, nn.Dropout(0.0),
                                  nn.Linear(inner, dim_out if dim_out is not None else dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class FP32LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return F.layer_norm(x.float(), self.normalized_shape,
                            self.weight.float() if self.weight is not None else None,
                            self.bias.float() if self.bias is not None else None, self.eps).to(x.dtype)


class RMSNorm(nn.Module):
    """diffusers RMSNorm: x * rsqrt(mean(x^2) + eps) [* weight], statistics in f32."""

    def __init__(self, dim: int, eps: float, elementwise_affine: bool = True):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim)) if elementwise_affine else None

    def forward(self, x):
        var = x.float().pow(2).mean(-1, keepdim=True)
        y = x.float() * torch.rsqrt(var + self.eps)
        if self.weight is not None:
            y = y * self.weight.float()
        return y.to(x.dtype)


def get_1d_rotary_pos_embed(dim: int, pos: torch.Tensor, theta: float = 10000.0, use_real: bool = True,
                            repeat_interleave_real: bool = True,
                            freqs_dtype=torch.float64) -> Tuple[torch.Tensor, torch.Tensor]:
    """MLX restatement: reference mlx/modules/rotary.py:6-78; call site flux model.py:347-354."""
    assert use_real and repeat_interleave_real
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=freqs_dtype)[: dim // 2] / dim))
    ang = torch.outer(pos.to(freqs_dtype), freqs)
    cos = ang.cos().repeat_interleave(2, dim=1).float()
    sin = ang.sin().repeat_interleave(2, dim=1).float()
    return cos, sin


def apply_rotary_emb(x: torch.Tensor, freqs_cis, sequence_dim: int = 1) -> torch.Tensor:
    """use_real=True, use_real_unbind_dim=-1; x [B,S,H,D] for sequence_dim=1.
    Same arithmetic as the in-tree apply_rotary_emb_qwen (qwenimage/base/model.py:100-151)."""
    cos, sin = freqs_cis
    if sequence_dim == 1:
        cos, sin = cos[None, :, None, :], sin[None, :, None, :]
    else:
        cos, sin = cos[None, None, :, :], sin[None, None, :, :]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


LOG2E_F32 = torch.tensor(1.4426950408889634, dtype=torch.float32)


def _exp2_scaled(s: torch.Tensor, c: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """2^(fma(s, c, -m)) as the kernels evaluate it: one f32 rounding of s*c - m (the product of two f32 values is
    exact in f64), then exp2."""
    return torch.exp2((s.double() * c.double() - m.double()).float())


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, softmax_scale: Optional[float] = None,
         policy: "Policy" = None) -> torch.Tensor:
    """softmax(q k^T * scale) v in f32 — what reference attention/functions.py:338-377 (`sdpa`,
    the default backend) computes; pinned against that function in tests/golden.

    With the bf16 storage policy the rounding points of the flash kernels (csrc/attention.hip) are reproduced:
    base-2 softmax with c = f32(scale) * f32(log2 e), probabilities p = 2^(s c - M) against an INTEGER row maximum
    M = ceil(max_j(s_j) * c), the row sum taken over the UNROUNDED p in f32, p rounded to bf16 as the P V operand,
    O = (bf16(p) v) / sum.  Because M is an integer the bf16 rounding of p commutes with every rescale the kernel
    performs, so this one-pass form equals the kernel's online (tiled, deferred-rescale, key-split) evaluation up
    to f32 summation order.  The caller rounds the output for storage."""
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(q.shape[-1])
    if policy is None or not policy.emulate_bf16:
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
        p = torch.softmax(s, dim=-1)
        return torch.matmul(p, v.float()).to(q.dtype)
    c = torch.tensor(scale, dtype=torch.float32) * LOG2E_F32
    qf, kf, vf = q.float(), k.float(), v.float()
    lead = qf.shape[:-2]
    qf, kf, vf = qf.reshape(-1, *qf.shape[-2:]), kf.reshape(-1, *kf.shape[-2:]), vf.reshape(-1, *vf.shape[-2:])
    out = torch.empty(qf.shape[0], qf.shape[1], vf.shape[-1], dtype=torch.float32)
    for h in range(qf.shape[0]):                     # per head: bounds the f64 temporaries at long sequences
        for r0 in range(0, qf.shape[1], 4096):
            s = qf[h, r0:r0 + 4096] @ kf[h].transpose(0, 1)
            m = torch.ceil(s.max(dim=-1, keepdim=True).values * c)
            pr = _exp2_scaled(s, c, m)
            l = pr.sum(dim=-1, keepdim=True)
            out[h, r0:r0 + 4096] = (pr.to(torch.bfloat16).float() @ vf[h]) / l
    return out.reshape(*lead, *out.shape[-2:]).to(q.dtype)


def sdpa_materialized(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, softmax_scale: Optional[float] = None,
                      policy: "Policy" = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Single-head attention of the VAE mid blocks.  fp32 policy: plain softmax(q k^T scale [+ mask]) v.  bf16 storage
    policy: the rounding points of the materialised HIP path (GEMM -> softmax_rows_kernel -> GEMM): f32 scores,
    p = 2^(s c - max(s) c) / sum rounded to bf16 AFTER normalisation, out = p v accumulated in f32; `mask` is a
    boolean keep-mask (masked probabilities are exact zeros).  The caller rounds the output."""
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(q.shape[-1])
    s = torch.matmul(q.float(), k.float().transpose(-1, -2))
    if policy is None or not policy.emulate_bf16:
        s = s * scale
        if mask is not None:
            s = s.masked_fill(~mask, float("-inf"))
        return torch.matmul(torch.softmax(s, dim=-1), v.float()).to(q.dtype)
    c = torch.tensor(scale, dtype=torch.float32) * LOG2E_F32
    if mask is not None:
        s = s.masked_fill(~mask, -1.0e30)
    m = s.max(dim=-1, keepdim=True).values * c
    pr = _exp2_scaled(s, c, m)
    if mask is not None:
        pr = pr * mask
    pr = pr * (1.0 / pr.sum(dim=-1, keepdim=True))
    return torch.matmul(pr.to(torch.bfloat16).float(), v.float()).to(q.dtype)


def sdpa_dispatch(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, policy: "Policy" = None,
                  mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention as the VAE mid blocks call it ([..., S, D], one head).  fp32 policy: plain softmax attention.  bf16 storage
    policy: the rounding points of whichever kernel `apexmi_attn_fwd` dispatches to for the shape (csrc/attention.hip,
    `attn_fwd_impl`): head dim 128 -> the flash kernels (`sdpa`); a multiple of 128 up to 1024 with Sq*Sk >= 256*256, or
    any frame-causal call -> the materialised path (`sdpa_materialized`); otherwise the generic kernel, whose
    probabilities stay f32."""
    if policy is None or not policy.emulate_bf16:
        return sdpa_materialized(q, k, v, policy=None, mask=mask)
    D, Sq, Sk = q.shape[-1], q.shape[-2], k.shape[-2]
    if mask is None and D == 128:
        return sdpa(q, k, v, policy=policy)
    if D % 128 == 0 and D <= 1024 and (mask is not None or Sq * Sk >= 256 * 256):
        return sdpa_materialized(q, k, v, policy=policy, mask=mask)
    return sdpa_materialized(q, k, v, policy=None, mask=mask)


class DiffusersAttention(nn.Module):
    """diffusers `Attention` as the Qwen block configures it (reference
    transformer/qwenimage/base/model.py:606-618): a parameter container whose forward
    hands itself to the processor.  to_q/to_k/to_v and add_q/k/v_proj are Linear(dim, heads*dim_head,
    bias); norm_q/norm_k/norm_added_q/norm_added_k = RMSNorm(dim_head, eps); to_out = [Linear, Dropout];
    to_add_out = Linear."""

    def __init__(self, query_dim: int, cross_attention_dim=None, added_kv_proj_dim=None, dim_head: int = 64,
                 heads: int = 8, out_dim=None, context_pre_only=None, bias: bool = False, processor=None,
                 qk_norm=None, eps: float = 1e-5):
        super().__init__()
        inner = out_dim if out_dim is not None else dim_head * heads
        self.heads = inner // dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.norm_q = RMSNorm(dim_head, eps) if qk_norm == "rms_norm" else None
        self.norm_k = RMSNorm(dim_head, eps) if qk_norm == "rms_norm" else None
        self.norm_added_q = self.norm_added_k = None
        if added_kv_proj_dim is not None:
            self.add_q_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_k_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            if qk_norm == "rms_norm":
                self.norm_added_q = RMSNorm(dim_head, eps)
                self.norm_added_k = RMSNorm(dim_head, eps)
            self.to_add_out = nn.Linear(inner, query_dim, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(inner, out_dim if out_dim is not None else query_dim, bias=True),
                                     nn.Dropout(0.0)])
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        if self.processor is None:
            return attn_processor_2_0(self, hidden_states, attention_mask)
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kwargs)


def attn_processor_2_0(attn: "DiffusersAttention", hidden_states: torch.Tensor, attention_mask=None) -> torch.Tensor:
    """diffusers' default `AttnProcessor2_0` for plain self-attention (the HunyuanVideo-1.5 token refiner,
    reference transformer/hunyuanvideo15/base/model.py:288-294, 322-326): q/k/v Linear, heads split,
    scaled_dot_product_attention with an additive mask broadcast over heads and queries, to_out[0], Dropout(0)."""
    B, S, _ = hidden_states.shape
    h = attn.heads
    q = attn.to_q(hidden_states).view(B, S, h, -1).transpose(1, 2)
    k = attn.to_k(hidden_states).view(B, S, h, -1).transpose(1, 2)
    v = attn.to_v(hidden_states).view(B, S, h, -1).transpose(1, 2)
    mask = None
    if attention_mask is not None:
        mask = attention_mask.reshape(B, 1, -1, attention_mask.shape[-1]).to(q.dtype)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, S, -1).to(q.dtype)
    return attn.to_out[1](attn.to_out[0](o))


def attn_processor_2_0_policy(attn: "DiffusersAttention", hidden_states: torch.Tensor, attention_mask, pol) -> torch.Tensor:
    """`attn_processor_2_0` with the storage policy's rounding points (q, k, v and the attention output are
    stored; the projection output is rounded by the caller)."""
    B, S, _ = hidden_states.shape
    h = attn.heads
    q = pol.r(attn.to_q(hidden_states)).view(B, S, h, -1).transpose(1, 2)
    k = pol.r(attn.to_k(hidden_states)).view(B, S, h, -1).transpose(1, 2)
    v = pol.r(attn.to_v(hidden_states)).view(B, S, h, -1).transpose(1, 2)
    mask = None
    if attention_mask is not None:
        mask = attention_mask.reshape(B, 1, -1, attention_mask.shape[-1]).to(q.dtype)
    if pol.emulate_bf16 and (mask is None or (B == 1 and mask.shape[2] == 1)):
        # the flash kernel's rounding points (P in bf16 against an integer running max); a key-padding mask drops the
        # masked keys, which is also how the HIP token refiner applies it
        if mask is not None:
            keep = torch.nonzero(mask[0, 0, 0] == 0, as_tuple=False).flatten()
            k, v = k.index_select(2, keep), v.index_select(2, keep)
        o = sdpa(q, k, v, policy=pol)
    else:
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=False)
    o = pol.r(o.transpose(1, 2).reshape(B, S, -1))
    return attn.to_out[0](o)
