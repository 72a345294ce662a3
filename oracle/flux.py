"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the Flux MM-DiT forward the reference runs per denoise step:
  FluxTransformer2DModel.forward        reference transformer/flux/base/model.py:473-657
  FluxTransformerBlock.forward          :265-328   (double stream)
  FluxSingleTransformerBlock.forward    :195-227   (single stream)
  FluxAttnProcessor.__call__            transformer/flux/base/attention.py:54-112
  FluxPosEmbed.forward                  model.py:338-359
plus packing / shift helpers of the engine (engine/flux/shared.py:29-68).
Leaf layers come from oracle.layers (diffusers restatements).  Parameter names equal the
reference's state-dict keys, so a state dict moves between this oracle, the reference classes and the
HIP model unchanged.  Pinned by tests/golden/flux_hybrid_*.pt (the reference's own wiring run here).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .layers import Policy, FP32


class FluxIPAdapterProcessor(nn.Module):
    """The parameters of `FluxIPAdapterAttnProcessor` (reference transformer/flux/base/attention.py:115-173): one key and one value
    projection of the image-prompt tokens per adapter, and a scale each.  State-dict keys as diffusers names them:
    `transformer_blocks.N.attn.processor.to_k_ip.M.weight`."""

    def __init__(self, hidden_size: int, cross_attention_dim: int, num_tokens=(4,), scale=1.0):
        super().__init__()
        num_tokens = list(num_tokens) if isinstance(num_tokens, (tuple, list)) else [num_tokens]
        self.scale = list(scale) if isinstance(scale, (list, tuple)) else [scale] * len(num_tokens)
        assert len(self.scale) == len(num_tokens)
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size) for _ in num_tokens])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size) for _ in num_tokens])


class FluxAttention(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, joint: bool, pre_only: bool, eps: float = 1e-6):
        super().__init__()
        self.processor = None          # a FluxIPAdapterProcessor on the double blocks of a model with IP adapters
        self.heads, self.head_dim, self.joint = heads, head_dim, joint
        inner = heads * head_dim
        self.norm_q = nn.RMSNorm(head_dim, eps=eps)
        self.norm_k = nn.RMSNorm(head_dim, eps=eps)
        self.to_q = nn.Linear(dim, inner)
        self.to_k = nn.Linear(dim, inner)
        self.to_v = nn.Linear(dim, inner)
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Dropout(0.0)])
        if joint:
            self.norm_added_q = nn.RMSNorm(head_dim, eps=eps)
            self.norm_added_k = nn.RMSNorm(head_dim, eps=eps)
            self.add_q_proj = nn.Linear(dim, inner)
            self.add_k_proj = nn.Linear(dim, inner)
            self.add_v_proj = nn.Linear(dim, inner)
            self.to_add_out = nn.Linear(inner, dim)

    def forward(self, x, ctx, rope, pol: Policy, ip_hidden_states=None):
        H = self.heads
        q = pol.r(self.to_q(x)).unflatten(-1, (H, -1))
        k = pol.r(self.to_k(x)).unflatten(-1, (H, -1))
        v = pol.r(self.to_v(x)).unflatten(-1, (H, -1))
        q, k = self.norm_q(q), self.norm_k(k)
        use_ip = ip_hidden_states is not None and self.processor is not None and ctx is not None
        # the image stream's normalised query BEFORE the rotary embedding (attention.py:199) — a storage point of the IP path only
        ip_q = pol.r(q) if use_ip else None
        if ctx is not None:
            cq = pol.r(self.add_q_proj(ctx)).unflatten(-1, (H, -1))
            ck = pol.r(self.add_k_proj(ctx)).unflatten(-1, (H, -1))
            cv = pol.r(self.add_v_proj(ctx)).unflatten(-1, (H, -1))
            cq, ck = self.norm_added_q(cq), self.norm_added_k(ck)
            q = torch.cat([cq, q], dim=1)  # text tokens first
            k = torch.cat([ck, k], dim=1)
            v = torch.cat([cv, v], dim=1)
        if rope is not None:
            q = L.apply_rotary_emb(q, rope, sequence_dim=1)
            k = L.apply_rotary_emb(k, rope, sequence_dim=1)
        q, k = pol.r(q), pol.r(k)
        o = L.sdpa(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), policy=pol)
        o = pol.r(o.permute(0, 2, 1, 3).flatten(2, 3))
        if ctx is not None:
            n_txt = ctx.shape[1]
            co, o = o[:, :n_txt], o[:, n_txt:]
            if use_ip:
                # IP-adapter (attention.py:232-262): per adapter, attention of the image queries over the projected image-prompt
                # tokens (no norm, no rotary embedding on those keys); returned per adapter with its scale — the block adds
                # scale x output to the image stream AFTER its feed-forward (model.py:308-309)
                B = x.shape[0]
                ips = []
                for h_ip, sc, wk, wv in zip(ip_hidden_states, self.processor.scale, self.processor.to_k_ip, self.processor.to_v_ip):
                    ik = pol.r(wk(h_ip)).view(B, -1, H, self.head_dim)
                    iv = pol.r(wv(h_ip)).view(B, -1, H, self.head_dim)
                    io = L.sdpa(ip_q.permute(0, 2, 1, 3), ik.permute(0, 2, 1, 3), iv.permute(0, 2, 1, 3), policy=pol)
                    ips.append((sc, pol.r(io.permute(0, 2, 1, 3).reshape(B, -1, H * self.head_dim))))
                return self.to_out[0](o), self.to_add_out(co), ips
            return self.to_out[0](o), self.to_add_out(co)
        return o


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int):
        super().__init__()
        self.norm1 = L.AdaLayerNormZero(dim)
        self.norm1_context = L.AdaLayerNormZero(dim)
        self.attn = FluxAttention(dim, heads, head_dim, joint=True, pre_only=False)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = L.FeedForward(dim, dim)
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = L.FeedForward(dim, dim)

    @staticmethod
    def _ff(ff, x, pol):
        h = pol.r(ff.net[0](x))
        return ff.net[2](h)

    def forward(self, x, ctx, temb, rope, pol: Policy, ip_hidden_states=None):
        nx, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(x, temb)
        nc, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(ctx, temb)
        outs = self.attn(pol.r(nx), pol.r(nc), rope, pol, ip_hidden_states)
        a, ca = outs[0], outs[1]
        x = pol.r(x + gate_msa.unsqueeze(1) * a)
        n2 = pol.r(self.norm2(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None])
        x = pol.r(x + gate_mlp.unsqueeze(1) * self._ff(self.ff, n2, pol))
        if len(outs) == 3:
            # `hidden_states = hidden_states + ip_attn_output` (model.py:308-309), ip_attn_output = sum of scale x adapter output.
            # Storage policy: the HIP path adds adapter after adapter into the stream (f32 multiply-add, one rounding each)
            if pol.emulate_bf16:
                for sc, o in outs[2]:
                    x = pol.r(x + sc * o)
            else:
                x = x + sum(sc * o for sc, o in outs[2])
        ctx = pol.r(ctx + c_gate_msa.unsqueeze(1) * ca)
        c2 = pol.r(self.norm2_context(ctx) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None])
        ctx = pol.r(ctx + c_gate_mlp.unsqueeze(1) * self._ff(self.ff_context, c2, pol))
        return ctx, x


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm = L.AdaLayerNormZeroSingle(dim)
        self.proj_mlp = nn.Linear(dim, self.mlp_hidden_dim)
        self.proj_out = nn.Linear(dim + self.mlp_hidden_dim, dim)
        self.attn = FluxAttention(dim, heads, head_dim, joint=False, pre_only=True)

    def forward(self, x, ctx, temb, rope, pol: Policy):
        n_txt = ctx.shape[1]
        h = torch.cat([ctx, x], dim=1)
        nh, gate = self.norm(h, temb)
        nh = pol.r(nh)
        mlp = pol.r(F.gelu(self.proj_mlp(nh), approximate="tanh"))
        a = self.attn(nh, None, rope, pol)
        h = pol.r(h + gate.unsqueeze(1) * self.proj_out(torch.cat([a, mlp], dim=2)))
        return h[:, :n_txt], h[:, n_txt:]


def flux_pos_embed(ids: torch.Tensor, axes_dim, theta: float = 10000.0):
    cos, sin = [], []
    pos = ids.float()
    for i, d in enumerate(axes_dim):
        c, s = L.get_1d_rotary_pos_embed(d, pos[:, i], theta=theta)
        cos.append(c)
        sin.append(s)
    return torch.cat(cos, dim=-1), torch.cat(sin, dim=-1)


class FluxTransformer2DModel(nn.Module):
    def __init__(self, patch_size: int = 1, in_channels: int = 64, out_channels: Optional[int] = None,
                 num_layers: int = 19, num_single_layers: int = 38, attention_head_dim: int = 128,
                 num_attention_heads: int = 24, joint_attention_dim: int = 4096,
                 pooled_projection_dim: int = 768, guidance_embeds: bool = False,
                 axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)):
        super().__init__()
        self.out_channels = out_channels or in_channels
        self.inner_dim = num_attention_heads * attention_head_dim
        self.axes_dims_rope = tuple(axes_dims_rope)
        self.guidance_embeds = guidance_embeds
        cls = (L.CombinedTimestepGuidanceTextProjEmbeddings if guidance_embeds
               else L.CombinedTimestepTextProjEmbeddings)
        self.time_text_embed = cls(self.inner_dim, pooled_projection_dim)
        self.context_embedder = nn.Linear(joint_attention_dim, self.inner_dim)
        self.x_embedder = nn.Linear(in_channels, self.inner_dim)
        self.transformer_blocks = nn.ModuleList(
            [FluxTransformerBlock(self.inner_dim, num_attention_heads, attention_head_dim)
             for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(self.inner_dim, num_attention_heads, attention_head_dim)
             for _ in range(num_single_layers)])
        self.norm_out = L.AdaLayerNormContinuous(self.inner_dim, self.inner_dim)
        self.proj_out = nn.Linear(self.inner_dim, patch_size * patch_size * self.out_channels)

    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids,
                txt_ids, guidance=None, policy: Policy = FP32, controlnet_block_samples=None,
                controlnet_single_block_samples=None, controlnet_blocks_repeat: bool = False, ip_hidden_states=None):
        """`ip_hidden_states`: the IP-adapter image-prompt tokens, one [B, tokens, joint_attention_dim] tensor per adapter — what
        `encoder_hid_proj(ip_adapter_image_embeds)` yields in the reference (model.py:562-571); the double blocks must carry a
        FluxIPAdapterProcessor (`attn.processor`)."""
        pol = policy
        if ip_hidden_states is not None:
            ip_hidden_states = [pol.r(h) for h in ip_hidden_states]
        x = pol.r(self.x_embedder(hidden_states))
        # reference: `timestep.to(hidden_states.dtype) * 1000` (model.py:535-537) — with bf16 hidden
        # states that product is rounded to bf16; the bf16 policy reproduces it.
        tdt = torch.bfloat16 if pol.emulate_bf16 else hidden_states.dtype
        timestep = (timestep.to(tdt) * 1000).to(hidden_states.dtype)
        if guidance is not None:
            guidance = (guidance.to(tdt) * 1000).to(hidden_states.dtype)
            temb = self.time_text_embed(timestep, guidance, pooled_projections)
        else:
            temb = self.time_text_embed(timestep, pooled_projections)
        ctx = pol.r(self.context_embedder(encoder_hidden_states))
        rope = flux_pos_embed(torch.cat((txt_ids, img_ids), dim=0), self.axes_dims_rope)
        # ControlNet residuals on the image stream after every block (reference model.py:594-612, :631-640): sample index =
        # block // ceil(blocks / samples), or block % samples with `controlnet_blocks_repeat` (double blocks only)
        import math
        for i, blk in enumerate(self.transformer_blocks):
            ctx, x = blk(x, ctx, temb, rope, pol, ip_hidden_states)
            if controlnet_block_samples is not None:
                n = len(controlnet_block_samples)
                j = i % n if controlnet_blocks_repeat else i // int(math.ceil(len(self.transformer_blocks) / n))
                x = pol.r(x + controlnet_block_samples[j])
        for i, blk in enumerate(self.single_transformer_blocks):
            ctx, x = blk(x, ctx, temb, rope, pol)
            if controlnet_single_block_samples is not None:
                n = len(controlnet_single_block_samples)
                x = pol.r(x + controlnet_single_block_samples[i // int(math.ceil(len(self.single_transformer_blocks) / n))])
        x = pol.r(self.norm_out(x, temb))
        return pol.r(self.proj_out(x))


# ---- engine-side helpers (reference engine/flux/shared.py:29-68, :197-215) ----

def pack_latents(latents: torch.Tensor) -> torch.Tensor:
    b, c, h, w = latents.shape
    x = latents.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, (h // 2) * (w // 2), c * 4)


def unpack_latents(latents: torch.Tensor, height: int, width: int, vae_scale_factor: int = 8):
    b, n, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.view(b, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(b, ch // 4, h, w)


def latent_image_ids(h2: int, w2: int) -> torch.Tensor:
    ids = torch.zeros(h2, w2, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(h2)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(w2)[None, :]
    return ids.reshape(h2 * w2, 3)


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.15) -> float:
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)
