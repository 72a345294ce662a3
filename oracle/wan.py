"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the Wan 2.x DiT forward the reference runs per denoise step:
  WanTransformer3DModel.forward       reference transformer/wan/base/model.py:1684-1891
  WanTransformerBlock.forward         :1101-1333  (modulation :1117-1128, gates :1196-1204, :1316-1324)
  WanAttnProcessor2_0.__call__        transformer/wan/base/attention.py:305-413
  WanTimeTextImageEmbedding.forward   model.py:773-823
  WanRotaryPosEmbed                   :847-945 (t/h/w split 44/42/42 for head_dim 128, complex pairs)
  InplaceRMSNorm (INTENDED semantics) transformer/efficiency/mod.py:24-35 — the fp32 path of the
      reference aliases its input; the oracle implements x*rsqrt(mean(x^2)+eps)*w.
  apply_wan_rope_inplace              transformer/efficiency/ops.py:112-160
Text-to-video scope: no image conditioning (`added_kv_proj_dim=None`), no IP adapter, no EasyCache.
Pinned by tests/golden/wan_hybrid.pt: the reference's own classes run in float64 (which avoids the
fp32 aliasing defect) on top of oracle.layers leaves.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .layers import Policy, FP32


def wan_rope_table(grid: Tuple[int, int, int], head_dim: int = 128, theta: float = 10000.0):
    """cos/sin [S, head_dim/2] for a (frames, height, width) token grid; positions start at 0 on
    every axis (the time table's sentinel row at t = -1 is skipped by the reference, model.py:934)."""
    f, h, w = grid
    h_dim = w_dim = 2 * (head_dim // 6)
    t_dim = head_dim - h_dim - w_dim

    def ang(dim, n):
        base = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
        return torch.outer(torch.arange(n, dtype=torch.float64), base)

    at, ah, aw = ang(t_dim, f), ang(h_dim, h), ang(w_dim, w)
    a = torch.cat([at.view(f, 1, 1, -1).expand(f, h, w, -1), ah.view(1, h, 1, -1).expand(f, h, w, -1),
                   aw.view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, head_dim // 2)
    return a.cos().float(), a.sin().float()


def apply_wan_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B, H, S, D]; complex multiply of the (2i, 2i+1) pairs."""
    xr, xi = x.float().unflatten(3, (-1, 2)).unbind(-1)
    outr = xr * cos - xi * sin
    outi = xi * cos + xr * sin
    return torch.stack([outr, outi], dim=-1).flatten(3).to(x.dtype)


class WanAttention(nn.Module):
    def __init__(self, dim: int, heads: int, eps: float):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        self.norm_q = L.RMSNorm(dim, eps)   # across ALL heads*head_dim channels
        self.norm_k = L.RMSNorm(dim, eps)

    def forward(self, x, ctx, rope, pol: Policy):
        src = x if ctx is None else ctx
        # the across-heads RMSNorm is its own pass writing bf16 in place, in the reference (InplaceRMSNorm on bf16 tensors,
        # transformer/efficiency/mod.py:24-35) as in the HIP path: a storage point under the bf16 policy
        q = pol.r(self.norm_q(pol.r(self.to_q(x))))
        k = pol.r(self.norm_k(pol.r(self.to_k(src))))
        v = pol.r(self.to_v(src))
        q = q.unflatten(2, (self.heads, -1)).transpose(1, 2)
        k = k.unflatten(2, (self.heads, -1)).transpose(1, 2)
        v = v.unflatten(2, (self.heads, -1)).transpose(1, 2)
        if rope is not None:
            q, k = apply_wan_rope(q, *rope), apply_wan_rope(k, *rope)
        q, k = pol.r(q), pol.r(k)
        o = pol.r(L.sdpa(q, k, v, policy=pol).transpose(1, 2).flatten(2, 3))
        return self.to_out[0](o)


class WanTransformerBlock(nn.Module):
    def __init__(self, dim: int, ffn_dim: int, heads: int, cross_attn_norm: bool = True, eps: float = 1e-6):
        super().__init__()
        self.norm1 = L.FP32LayerNorm(dim, eps, elementwise_affine=False)
        self.attn1 = WanAttention(dim, heads, eps)
        self.attn2 = WanAttention(dim, heads, eps)
        self.norm2 = L.FP32LayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.ffn = L.FeedForward(dim, inner_dim=ffn_dim)
        self.norm3 = L.FP32LayerNorm(dim, eps, elementwise_affine=False)
        self.scale_shift_table = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, x, ctx, temb6, rope, pol: Policy):
        shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = (
            self.scale_shift_table + temb6.float()).chunk(6, dim=1)
        n = pol.r(self.norm1(x) * (1 + scale_msa) + shift_msa)
        x = pol.r(x + self.attn1(n, None, rope, pol) * gate_msa)
        n = pol.r(self.norm2(x))
        x = pol.r(x + self.attn2(n, ctx, None, pol))
        n = pol.r(self.norm3(x) * (1 + c_scale) + c_shift)
        h = pol.r(self.ffn.net[0](n))
        x = pol.r(x + self.ffn.net[2](h) * c_gate)
        return x


class WanTimeTextEmbedding(nn.Module):
    def __init__(self, dim: int, time_freq_dim: int, time_proj_dim: int, text_embed_dim: int):
        super().__init__()
        self.timesteps_proj = L.Timesteps(time_freq_dim, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.time_embedder = L.TimestepEmbedding(time_freq_dim, dim)
        self.time_proj = nn.Linear(dim, time_proj_dim)
        self.text_embedder = L.PixArtAlphaTextProjection(text_embed_dim, dim, act_fn="gelu_tanh")

    def forward(self, timestep, text, pol: Policy):
        temb = self.time_embedder(self.timesteps_proj(timestep))
        timestep_proj = self.time_proj(F.silu(temb))
        h = pol.r(F.gelu(self.text_embedder.linear_1(text), approximate="tanh"))
        return temb, timestep_proj, pol.r(self.text_embedder.linear_2(h))


class WanTransformer3DModel(nn.Module):
    def __init__(self, patch_size=(1, 2, 2), num_attention_heads: int = 40, attention_head_dim: int = 128,
                 in_channels: int = 16, out_channels: int = 16, text_dim: int = 4096, freq_dim: int = 256,
                 ffn_dim: int = 13824, num_layers: int = 40, cross_attn_norm: bool = True,
                 eps: float = 1e-6):
        super().__init__()
        dim = num_attention_heads * attention_head_dim
        self.patch_size, self.heads, self.head_dim = tuple(patch_size), num_attention_heads, attention_head_dim
        self.out_channels = out_channels
        self.patch_embedding = nn.Conv3d(in_channels, dim, kernel_size=patch_size, stride=patch_size)
        self.condition_embedder = WanTimeTextEmbedding(dim, freq_dim, dim * 6, text_dim)
        self.blocks = nn.ModuleList([WanTransformerBlock(dim, ffn_dim, num_attention_heads, cross_attn_norm, eps)
                                     for _ in range(num_layers)])
        self.norm_out = L.FP32LayerNorm(dim, eps, elementwise_affine=False)
        self.proj_out = nn.Linear(dim, out_channels * math.prod(patch_size))
        self.scale_shift_table = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    @torch.no_grad()
    def forward(self, hidden_states, timestep, encoder_hidden_states, policy: Policy = FP32):
        pol = policy
        B, C, T, H, W = hidden_states.shape
        pt, ph, pw = self.patch_size
        grid = (T // pt, H // ph, W // pw)
        rope = wan_rope_table(grid, self.head_dim)
        x = pol.r(self.patch_embedding(hidden_states).flatten(2).transpose(1, 2))
        temb, tproj, ctx = self.condition_embedder(timestep, encoder_hidden_states, pol)
        temb6 = tproj.unflatten(1, (6, -1))
        for blk in self.blocks:
            x = blk(x, ctx, temb6, rope, pol)
        shift, scale = (self.scale_shift_table + temb.unsqueeze(1)).chunk(2, dim=1)
        x = pol.r(self.norm_out(x) * (1 + scale) + shift)
        x = pol.r(self.proj_out(x))
        x = x.reshape(B, grid[0], grid[1], grid[2], pt, ph, pw, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return x.flatten(6, 7).flatten(4, 5).flatten(2, 3)
