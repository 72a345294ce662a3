"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the QwenImage MM-DiT forward (QwenImage-Edit-2509 uses the same class):
  QwenImageTransformer2DModel.forward      reference transformer/qwenimage/base/model.py:851-993
  QwenImageTransformerBlock.forward        :679-750   (_modulate :640-677)
  QwenDoubleStreamAttnProcessor2_0         :495-578   (text tokens first in the joint sequence)
  QwenEmbedRope (scale_rope=True)          :187-314   (centred h/w positions, text offset max(h,w)//2)
  QwenTimestepProjEmbeddings               :154-184   (Timesteps scale=1000 on the engine's t/1000)
  apply_rotary_emb_qwen(use_real=False)    :100-151
Leaves (Attention container, RMSNorm, FeedForward, AdaLayerNormContinuous, Timesteps, TimestepEmbedding)
come from oracle.layers.  `zero_cond_t` (a second conditioning row at t = 0 that modulates the CONDITION images' tokens:
:692-702, :912-923, :980-981, `_modulate(index)` :640-677) and `use_additional_t_cond` (:164-182) are restated (round 6);
the layer3d rope variant is not.  Pinned by tests/golden/qwen_hybrid.pt and qwen_variants.pt.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import layers as L
from .layers import Policy, FP32


def qwen_rope_positions(img_shapes: Sequence[Tuple[int, int, int]], txt_len: int) -> torch.Tensor:
    """Integer (frame, height, width) positions for the joint sequence [text | image_0 | image_1 ...]."""
    vids, max_idx = [], 0
    for idx, (f, h, w) in enumerate(img_shapes):
        fr = torch.arange(idx, idx + f)
        hh = torch.cat([torch.arange(-(h - h // 2), 0), torch.arange(0, h // 2)])
        ww = torch.cat([torch.arange(-(w - w // 2), 0), torch.arange(0, w // 2)])
        g = torch.stack(torch.meshgrid(fr, hh, ww, indexing="ij"), dim=-1).reshape(-1, 3)
        vids.append(g)
        max_idx = max(h // 2, w // 2, max_idx)
    t = torch.arange(max_idx, max_idx + txt_len)
    return torch.cat([torch.stack([t, t, t], dim=-1)] + vids, dim=0)


def qwen_rope_table(positions: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0):
    """cos/sin [S, sum(axes)/2]; the reference builds the angles in float32 (rope_params :213-224)."""
    cols = []
    for a, d in enumerate(axes_dim):
        freqs = 1.0 / torch.pow(torch.tensor(float(theta)), torch.arange(0, d, 2).to(torch.float32).div(d))
        cols.append(torch.outer(positions[:, a].to(torch.float32), freqs))
    ang = torch.cat(cols, dim=1)
    return ang.cos(), ang.sin()


def apply_rope_complex(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B, S, H, D]; complex multiply of (2i, 2i+1) pairs by e^{i angle[s, i]}."""
    xr, xi = x.float().reshape(*x.shape[:-1], -1, 2).unbind(-1)
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    return torch.stack([xr * c - xi * s, xr * s + xi * c], dim=-1).flatten(3).to(x.dtype)


class QwenImageTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, eps: float = 1e-6):
        super().__init__()
        self.img_mod = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim))
        self.img_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.attn = L.DiffusersAttention(query_dim=dim, added_kv_proj_dim=dim, dim_head=head_dim, heads=heads,
                                         out_dim=dim, bias=True, qk_norm="rms_norm", eps=eps)
        self.img_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.img_mlp = L.FeedForward(dim, dim)
        self.txt_mod = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim))
        self.txt_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.txt_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.txt_mlp = L.FeedForward(dim, dim)

    @staticmethod
    def _mod(x, p, index=None):
        """`_modulate` (model.py:640-677).  index [B, L] (zero_cond_t): p holds 2 B rows — row b for tokens with index 0 (the
        target image, conditioned on t), row B + b for tokens with index 1 (the condition images, conditioned on t = 0)."""
        shift, scale, gate = p.chunk(3, dim=-1)
        if index is None:
            return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)
        B = shift.shape[0] // 2
        sel = (index == 0).unsqueeze(-1)
        pick = lambda v: torch.where(sel, v[:B].unsqueeze(1), v[B:].unsqueeze(1))      # noqa: E731
        return x * (1 + pick(scale)) + pick(shift), pick(gate)

    def _attention(self, img, txt, rope, pol: Policy):
        a, H = self.attn, self.attn.heads
        n_txt = txt.shape[1]
        q = torch.cat([a.norm_added_q(pol.r(a.add_q_proj(txt)).unflatten(-1, (H, -1))),
                       a.norm_q(pol.r(a.to_q(img)).unflatten(-1, (H, -1)))], dim=1)
        k = torch.cat([a.norm_added_k(pol.r(a.add_k_proj(txt)).unflatten(-1, (H, -1))),
                       a.norm_k(pol.r(a.to_k(img)).unflatten(-1, (H, -1)))], dim=1)
        v = torch.cat([pol.r(a.add_v_proj(txt)).unflatten(-1, (H, -1)),
                       pol.r(a.to_v(img)).unflatten(-1, (H, -1))], dim=1)
        q, k = pol.r(apply_rope_complex(q, *rope)), pol.r(apply_rope_complex(k, *rope))
        o = L.sdpa(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), policy=pol)
        o = pol.r(o.permute(0, 2, 1, 3).flatten(2, 3))
        return a.to_out[0](o[:, n_txt:]), a.to_add_out(o[:, :n_txt])

    @staticmethod
    def _ff(ff, x, pol):
        return ff.net[2](pol.r(ff.net[0](x)))

    def forward(self, img, txt, temb, rope, pol: Policy, index=None):
        im1, im2 = self.img_mod(temb).chunk(2, dim=-1)
        if index is not None:                  # zero_cond_t: the text stream sees the t rows only (model.py:692-693)
            temb = temb.chunk(2, dim=0)[0]
        tm1, tm2 = self.txt_mod(temb).chunk(2, dim=-1)
        im, ig1 = self._mod(self.img_norm1(img), im1, index)
        tm, tg1 = self._mod(self.txt_norm1(txt), tm1)
        ia, ta = self._attention(pol.r(im), pol.r(tm), rope, pol)
        img = pol.r(img + ig1 * ia)
        txt = pol.r(txt + tg1 * ta)
        im, ig2 = self._mod(self.img_norm2(img), im2, index)
        img = pol.r(img + ig2 * self._ff(self.img_mlp, pol.r(im), pol))
        tm, tg2 = self._mod(self.txt_norm2(txt), tm2)
        txt = pol.r(txt + tg2 * self._ff(self.txt_mlp, pol.r(tm), pol))
        return txt, img


class QwenTimestepProjEmbeddings(nn.Module):
    def __init__(self, dim: int, use_additional_t_cond: bool = False):
        super().__init__()
        self.time_proj = L.Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000)
        self.timestep_embedder = L.TimestepEmbedding(256, dim)
        self.use_additional_t_cond = use_additional_t_cond
        if use_additional_t_cond:
            self.addition_t_embedding = nn.Embedding(2, dim)

    def forward(self, timestep, addition_t_cond=None):
        emb = self.timestep_embedder(self.time_proj(timestep))
        if self.use_additional_t_cond:         # model.py:175-182
            if addition_t_cond is None:
                raise ValueError("When additional_t_cond is True, addition_t_cond must be provided.")
            emb = emb + self.addition_t_embedding(addition_t_cond).to(emb.dtype)
        return emb


class QwenImageTransformer2DModel(nn.Module):
    def __init__(self, patch_size: int = 2, in_channels: int = 64, out_channels: Optional[int] = 16,
                 num_layers: int = 60, attention_head_dim: int = 128, num_attention_heads: int = 24,
                 joint_attention_dim: int = 3584, guidance_embeds: bool = False,
                 axes_dims_rope: Tuple[int, int, int] = (16, 56, 56), zero_cond_t: bool = False,
                 use_additional_t_cond: bool = False):
        super().__init__()
        self.out_channels = out_channels or in_channels
        self.inner_dim = dim = num_attention_heads * attention_head_dim
        self.axes_dims_rope = tuple(axes_dims_rope)
        self.zero_cond_t = zero_cond_t
        self.time_text_embed = QwenTimestepProjEmbeddings(dim, use_additional_t_cond)
        self.txt_norm = L.RMSNorm(joint_attention_dim, eps=1e-6)
        self.img_in = nn.Linear(in_channels, dim)
        self.txt_in = nn.Linear(joint_attention_dim, dim)
        self.transformer_blocks = nn.ModuleList(
            [QwenImageTransformerBlock(dim, num_attention_heads, attention_head_dim) for _ in range(num_layers)])
        self.norm_out = L.AdaLayerNormContinuous(dim, dim)
        self.proj_out = nn.Linear(dim, patch_size * patch_size * self.out_channels)

    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, timestep, img_shapes, txt_seq_lens=None,
                policy: Policy = FP32, additional_t_cond=None):
        pol = policy
        shapes = img_shapes[0] if isinstance(img_shapes[0], (list, tuple)) and \
            isinstance(img_shapes[0][0], (list, tuple)) else img_shapes
        img = pol.r(self.img_in(hidden_states))
        txt = pol.r(self.txt_in(pol.r(self.txt_norm(encoder_hidden_states))))
        # reference: `timestep = timestep.to(hidden_states.dtype)` (model.py:905) — bf16 with bf16 latents
        tdt = torch.bfloat16 if pol.emulate_bf16 else hidden_states.dtype
        timestep = timestep.to(tdt).to(hidden_states.dtype)
        index = None
        if self.zero_cond_t:                   # model.py:912-923: a second row at t = 0; tokens of images 1.. are marked 1
            timestep = torch.cat([timestep, timestep * 0], dim=0)
            per = [shapes] * hidden_states.shape[0] if not isinstance(img_shapes[0][0], (list, tuple)) else img_shapes
            index = torch.tensor([[0] * (s[0][0] * s[0][1] * s[0][2]) + [1] * sum(a * b * c for a, b, c in s[1:]) for s in per],
                                 dtype=torch.int)
            if additional_t_cond is not None:
                additional_t_cond = torch.cat([additional_t_cond, additional_t_cond], dim=0)
        temb = self.time_text_embed(timestep, additional_t_cond)
        n_txt = encoder_hidden_states.shape[1]
        rope = qwen_rope_table(qwen_rope_positions(shapes, n_txt), self.axes_dims_rope)
        for blk in self.transformer_blocks:
            txt, img = blk(img, txt, temb, rope, pol, index)
        if self.zero_cond_t:
            temb = temb.chunk(2, dim=0)[0]
        img = pol.r(self.norm_out(img, temb))
        return pol.r(self.proj_out(img))
