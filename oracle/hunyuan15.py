"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the HunyuanVideo-1.5 MM-DiT forward, following
/root/reference/apps/api/src/transformer/hunyuanvideo15/base/model.py:
  HunyuanVideo15Transformer3DModel.forward      :954-1165   (token reorder :1058-1108, un-patchify :1145-1156)
  HunyuanVideo15TransformerBlock.forward        :617-694    (AdaLN-Zero on both streams, joint attention, gated MLPs)
  HunyuanVideo15AttnProcessor2_0.__call__       :96-176     (latent tokens FIRST in the joint sequence, RoPE on them only)
  HunyuanVideo15TokenRefiner / ...RefinerBlock  :273-458    (masked mean pooling, key-padding mask, gates without shift/scale)
  HunyuanVideo15ByT5TextProjection / ImageProjection :506-540 (erf GELU)
  HunyuanVideo15RotaryPosEmbed                  :461-503    (theta 256, real cos/sin repeated per pair)
  HunyuanVideo15TimeEmbedding                   :222-270    (meanflow branch not restated: off for T2V/I2V 480p/720p)
  apply_cos_sin_rope_inplace                    transformer/efficiency/ops.py:163-233 (pinned in efficiency_ops.pt)
Leaves (Attention container + default processor, FeedForward incl. "linear-silu", AdaLayerNormZero/Continuous,
CombinedTimestepTextProjEmbeddings, Timesteps, TimestepEmbedding, get_1d_rotary_pos_embed) come from oracle.layers
(diffusers is absent: "parity unpinned" for those leaves).  The wiring is pinned by tests/golden/hunyuan15_hybrid.pt,
which runs the reference's own classes with those leaves.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .layers import Policy, FP32


def rope_table(grid_thw: Tuple[int, int, int], axes_dim=(16, 56, 56), theta: float = 256.0):
    """cos, sin [T*H*W, sum(axes)] (each angle repeated for its pair), model.py:476-503."""
    axes = [torch.arange(0, n, dtype=torch.float32) for n in grid_thw]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=0)
    cs = [L.get_1d_rotary_pos_embed(axes_dim[i], grid[i].reshape(-1), theta, use_real=True) for i in range(3)]
    return torch.cat([c[0] for c in cs], dim=1), torch.cat([c[1] for c in cs], dim=1)


def apply_cos_sin_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B, S, H, D]; full-layout tables are sub-sampled [..., ::2]; pairs (2i, 2i+1) rotate (ops.py:203-232)."""
    c, s = cos[None, :, None, ::2], sin[None, :, None, ::2]
    xr, xi = x.float().reshape(*x.shape[:-1], -1, 2).unbind(-1)
    return torch.stack([xr * c - xi * s, xr * s + xi * c], dim=-1).flatten(3).to(x.dtype)


class RefinerBlock(nn.Module):
    def __init__(self, heads: int, head_dim: int, mlp_ratio: float = 4.0):
        super().__init__()
        dim = heads * head_dim
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=True, eps=1e-6)
        self.attn = L.DiffusersAttention(query_dim=dim, cross_attention_dim=None, heads=heads, dim_head=head_dim, bias=True)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=True, eps=1e-6)
        self.ff = L.FeedForward(dim, mult=mlp_ratio, activation_fn="linear-silu")
        self.norm_out = nn.Module()
        self.norm_out.linear = nn.Linear(dim, 2 * dim)

    def forward(self, x, temb, mask, pol: Policy):
        # the projection output is not a storage point: the HIP GEMM adds gate * (o W^T + b) to the residual in f32 and rounds once
        a = L.attn_processor_2_0_policy(self.attn, pol.r(self.norm1(x)), mask, pol)
        g = self.norm_out.linear(F.silu(temb))
        gate_msa, gate_mlp = g.chunk(2, dim=1)
        x = pol.r(x + a * gate_msa.unsqueeze(1))
        h = pol.r(self.norm2(x))
        h = pol.r(F.silu(self.ff.net[0].proj(h)))
        return pol.r(x + self.ff.net[2](h) * gate_mlp.unsqueeze(1))


class TokenRefiner(nn.Module):
    def __init__(self, in_channels: int, heads: int, head_dim: int, num_layers: int):
        super().__init__()
        dim = heads * head_dim
        self.time_text_embed = L.CombinedTimestepTextProjEmbeddings(embedding_dim=dim, pooled_projection_dim=in_channels)
        self.proj_in = nn.Linear(in_channels, dim)
        self.token_refiner = nn.Module()
        self.token_refiner.refiner_blocks = nn.ModuleList([RefinerBlock(heads, head_dim) for _ in range(num_layers)])

    def forward(self, x, timestep, mask, pol: Policy):
        m = mask.float().unsqueeze(-1)
        pooled = (x * m).sum(dim=1) / m.sum(dim=1)
        temb = self.time_text_embed(timestep, pooled)
        h = pol.r(self.proj_in(x))
        add = None
        if not mask.bool().all():
            add = torch.zeros(x.shape[0], 1, 1, x.shape[1]).masked_fill(~mask.bool().view(x.shape[0], 1, 1, -1), float("-inf"))
        for blk in self.token_refiner.refiner_blocks:
            h = blk(h, temb, add, pol)
        return h


class ByT5Projection(nn.Module):
    def __init__(self, in_features: int, hidden: int, out_features: int):
        super().__init__()
        self.norm = nn.LayerNorm(in_features)
        self.linear_1 = nn.Linear(in_features, hidden)
        self.linear_2 = nn.Linear(hidden, hidden)
        self.linear_3 = nn.Linear(hidden, out_features)

    def forward(self, x, pol: Policy):
        h = pol.r(self.norm(x))
        h = pol.r(F.gelu(self.linear_1(h)))
        h = pol.r(F.gelu(self.linear_2(h)))
        return pol.r(self.linear_3(h))


class ImageProjection(nn.Module):
    def __init__(self, in_channels: int, hidden: int):
        super().__init__()
        self.norm_in = nn.LayerNorm(in_channels)
        self.linear_1 = nn.Linear(in_channels, in_channels)
        self.linear_2 = nn.Linear(in_channels, hidden)
        self.norm_out = nn.LayerNorm(hidden)

    def forward(self, x, pol: Policy):
        h = pol.r(self.norm_in(x))
        h = pol.r(F.gelu(self.linear_1(h)))
        return pol.r(self.norm_out(pol.r(self.linear_2(h))))


class TransformerBlock(nn.Module):
    def __init__(self, heads: int, head_dim: int, mlp_ratio: float = 4.0):
        super().__init__()
        dim = heads * head_dim
        self.heads = heads
        self.norm1 = L.AdaLayerNormZero(dim)
        self.norm1_context = L.AdaLayerNormZero(dim)
        self.attn = L.DiffusersAttention(query_dim=dim, cross_attention_dim=None, added_kv_proj_dim=dim, dim_head=head_dim,
                                         heads=heads, out_dim=dim, context_pre_only=False, bias=True, qk_norm="rms_norm",
                                         eps=1e-6)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = L.FeedForward(dim, mult=mlp_ratio, activation_fn="gelu-approximate")
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = L.FeedForward(dim, mult=mlp_ratio, activation_fn="gelu-approximate")

    def _attention(self, x, c, rope, pol: Policy):
        a, h = self.attn, self.heads
        B, S, _ = x.shape
        T = c.shape[1]
        q = pol.r(a.to_q(x)).view(B, S, h, -1)
        k = pol.r(a.to_k(x)).view(B, S, h, -1)
        v = pol.r(a.to_v(x)).view(B, S, h, -1)
        q, k = a.norm_q(q), a.norm_k(k)
        q, k = apply_cos_sin_rope(q, *rope), apply_cos_sin_rope(k, *rope)
        cq = a.norm_added_q(pol.r(a.add_q_proj(c)).view(B, T, h, -1))
        ck = a.norm_added_k(pol.r(a.add_k_proj(c)).view(B, T, h, -1))
        cv = pol.r(a.add_v_proj(c)).view(B, T, h, -1)
        q, k, v = (pol.r(torch.cat(p, dim=1)) for p in ((q, cq), (k, ck), (v, cv)))
        o = pol.r(L.sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), policy=pol).transpose(1, 2).flatten(2, 3))
        return a.to_out[0](o[:, :S]), a.to_add_out(o[:, S:])

    def forward(self, x, c, temb, rope, pol: Policy):
        nx, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(x, emb=temb)
        nc, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(c, emb=temb)
        ax, ac = self._attention(pol.r(nx), pol.r(nc), rope, pol)
        x = pol.r(x + ax * gate_msa.unsqueeze(1))
        c = pol.r(c + ac * c_gate_msa.unsqueeze(1))
        nx = pol.r(self.norm2(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None])
        nc = pol.r(self.norm2_context(c) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None])
        fx = self.ff.net[2](pol.r(self.ff.net[0](nx)))
        fc = self.ff_context.net[2](pol.r(self.ff_context.net[0](nc)))
        return pol.r(x + gate_mlp.unsqueeze(1) * fx), pol.r(c + c_gate_mlp.unsqueeze(1) * fc)


class HunyuanVideo15Transformer3DModel(nn.Module):
    def __init__(self, in_channels: int = 65, out_channels: int = 32, num_attention_heads: int = 16,
                 attention_head_dim: int = 128, num_layers: int = 54, num_refiner_layers: int = 2, mlp_ratio: float = 4.0,
                 patch_size: int = 1, patch_size_t: int = 1, qk_norm: str = "rms_norm", text_embed_dim: int = 3584,
                 text_embed_2_dim: int = 1472, image_embed_dim: int = 1152, rope_theta: float = 256.0,
                 rope_axes_dim=(16, 56, 56), use_meanflow: bool = False, **_unused):
        super().__init__()
        dim = num_attention_heads * attention_head_dim
        self.use_meanflow = use_meanflow
        self.p, self.pt, self.out_channels = patch_size, patch_size_t, out_channels or in_channels
        self.rope_axes_dim, self.rope_theta = tuple(rope_axes_dim), rope_theta
        self.x_embedder = nn.Module()
        self.x_embedder.proj = nn.Conv3d(in_channels, dim, kernel_size=(patch_size_t, patch_size, patch_size),
                                         stride=(patch_size_t, patch_size, patch_size))
        self.image_embedder = ImageProjection(image_embed_dim, dim)
        self.context_embedder = TokenRefiner(text_embed_dim, num_attention_heads, attention_head_dim, num_refiner_layers)
        self.context_embedder_2 = ByT5Projection(text_embed_2_dim, 2048, dim)
        self.time_embed = nn.Module()
        self.time_embed.time_proj = L.Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.time_embed.timestep_embedder = L.TimestepEmbedding(in_channels=256, time_embed_dim=dim)
        if use_meanflow:      # HunyuanVideo15TimeEmbedding, model.py:234-268 (time_proj_r has no parameters)
            self.time_embed.timestep_embedder_r = L.TimestepEmbedding(in_channels=256, time_embed_dim=dim)
        self.cond_type_embed = nn.Embedding(3, dim)
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlock(num_attention_heads, attention_head_dim, mlp_ratio) for _ in range(num_layers)])
        self.norm_out = L.AdaLayerNormContinuous(dim, dim, elementwise_affine=False, eps=1e-6)
        self.proj_out = nn.Linear(dim, patch_size_t * patch_size * patch_size * self.out_channels)

    @torch.no_grad()
    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, encoder_hidden_states_2,
                encoder_attention_mask_2, image_embeds, policy: Policy = FP32, timestep_r=None):
        pol = policy
        B, _, F_, H, W = hidden_states.shape
        grid = (F_ // self.pt, H // self.p, W // self.p)
        rope = rope_table(grid, self.rope_axes_dim, self.rope_theta)
        # the engine passes `t.expand(B).to(latents.dtype)` (engine/hunyuanvideo15/t2v.py:243-245): bf16 with bf16 latents
        t = timestep.to(torch.bfloat16).float() if pol.emulate_bf16 else timestep.float()
        temb = self.time_embed.timestep_embedder(self.time_embed.time_proj(t))
        if timestep_r is not None:
            tr = timestep_r.to(torch.bfloat16).float() if pol.emulate_bf16 else timestep_r.float()
            temb = temb + self.time_embed.timestep_embedder_r(self.time_embed.time_proj(tr))
        x = pol.r(self.x_embedder.proj(hidden_states).flatten(2).transpose(1, 2))
        c1 = self.context_embedder(encoder_hidden_states, t, encoder_attention_mask, pol)
        c1 = pol.r(c1 + self.cond_type_embed.weight[0])
        c2 = pol.r(self.context_embedder_2(encoder_hidden_states_2, pol) + self.cond_type_embed.weight[1])
        c3 = self.image_embedder(image_embeds, pol)
        is_t2v = bool(torch.all(image_embeds == 0))
        if is_t2v:
            c3 = c3 * 0.0
        m3 = torch.zeros(B, c3.shape[1], dtype=torch.bool) if is_t2v else torch.ones(B, c3.shape[1], dtype=torch.bool)
        c3 = pol.r(c3 + self.cond_type_embed.weight[2])
        m1, m2 = encoder_attention_mask.bool(), encoder_attention_mask_2.bool()
        rows = []
        for b in range(B):      # [valid image, valid byt5, valid mllm, invalid image, zeros(byt5), zeros(mllm)]
            rows.append(torch.cat([c3[b][m3[b]], c2[b][m2[b]], c1[b][m1[b]], c3[b][~m3[b]],
                                   torch.zeros_like(c2[b][~m2[b]]), torch.zeros_like(c1[b][~m1[b]])], dim=0))
        c = torch.stack(rows)
        for blk in self.transformer_blocks:
            x, c = blk(x, c, temb, rope, pol)
        x = pol.r(self.proj_out(pol.r(self.norm_out(x, temb))))
        x = x.reshape(B, grid[0], grid[1], grid[2], -1, self.pt, self.p, self.p)
        x = x.permute(0, 4, 1, 5, 2, 6, 3, 7)
        return x.flatten(6, 7).flatten(4, 5).flatten(2, 3)
