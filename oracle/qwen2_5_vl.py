"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of `transformers.Qwen2_5_VLForConditionalGeneration` as the reference uses it for QwenImage /
QwenImage-Edit prompts (manifest `base: Qwen2_5_VLForConditionalGeneration`,
manifest/image/qwenimage-edit-2509-1.0.0.v1.yml:62-76; called with input_ids / attention_mask / pixel_values /
image_grid_thw / output_hidden_states=True and read as `hidden_states[-1]`, engine/qwenimage/shared.py:183-226): the
vision tower (patch embedding, 2-D rotary, windowed / full block attention, SwiGLU, patch merger), the scatter of the
image embeddings over the `<|image_pad|>` tokens, multimodal 3-D RoPE position ids, and the causal GQA decoder.

The algorithm lives in the third-party package (pinned transformers==4.57.1, requirements.txt:79), not in
/root/reference; it is restated from the published modeling code (models/qwen2_5_vl/modeling_qwen2_5_vl.py), with the
state-dict keys of the 4.57 module layout the reference's converter targets (`model.visual.*`, `model.language_model.*`,
`lm_head.weight`; converters/text_encoder_converters.py:30-45).  Parity IS pinned: tests/golden/make_golden.py runs
the `transformers` build installed in this container on a small config (text-only, and text + two images) and saves
inputs / outputs (tests/golden/qwen2_5_vl.pt).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import FP32, Policy


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        var = x.float().pow(2).mean(-1, keepdim=True)
        return self.weight * (x * torch.rsqrt(var + self.eps))


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def sdpa_masked(q, k, v, allowed, scale, pol: Policy):
    """q, k, v [H, S, D]; allowed bool [S, S] (or [1, S, S])."""
    sc = (q @ k.transpose(-1, -2)) * scale
    p = pol.r(torch.softmax(sc.masked_fill(~allowed, float("-inf")).float(), dim=-1))
    return p @ v


# ---- vision tower ------------------------------------------------------------------------------------------------

def vision_position_ids(grid_thw: List[List[int]], merge: int) -> torch.Tensor:
    """(h, w) index of every patch, laid out block-major over merge x merge blocks (vision_utils.get_vision_position_ids)."""
    out = []
    for t, h, w in grid_thw:
        hp, wp = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        shape = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shape).transpose(1, 2).flatten()
        wp = wp.reshape(shape).transpose(1, 2).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_window_index(grid_thw: List[List[int]], merge: int, window_size: int, patch_size: int):
    """Reorder index (in merged-token units) that makes attention windows contiguous, and the cumulative window
    boundaries in patch units (vision_utils.get_vision_window_index)."""
    window_index, cu = [], [0]
    base = 0
    vw = window_size // merge // patch_size
    unit = merge * merge
    for t, h, w in grid_thw:
        gh, gw = h // merge, w // merge
        index = torch.arange(t * gh * gw).reshape(t, gh, gw)
        pad_h, pad_w = vw - gh % vw, vw - gw % vw
        nh, nw = (gh + pad_h) // vw, (gw + pad_w) // vw
        ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        ip = ip.reshape(t, nh, vw, nw, vw).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, vw, vw)
        seqlens = (ip != -100).sum([2, 3]).reshape(-1)
        ip = ip.reshape(-1)
        window_index.append(ip[ip != -100] + base)
        cu.extend((seqlens.cumsum(0) * unit + cu[-1]).tolist())
        base += t * gh * gw
    cu = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int64))
    return torch.cat(window_index), cu


def segment_mask(cu: torch.Tensor, S: int) -> torch.Tensor:
    seg = torch.bucketize(torch.arange(S), cu[1:], right=True)
    return seg[:, None] == seg[None, :]


class VisionBlock(nn.Module):
    def __init__(self, d, heads, inter):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = RMSNorm(d), RMSNorm(d)
        self.attn = nn.Module()
        self.attn.qkv, self.attn.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.mlp = nn.Module()
        self.mlp.gate_proj, self.mlp.up_proj, self.mlp.down_proj = nn.Linear(d, inter), nn.Linear(d, inter), nn.Linear(inter, d)

    def forward(self, x, cos, sin, allowed, pol: Policy):
        S, d = x.shape
        hd = d // self.heads
        qkv = pol.r(self.attn.qkv(pol.r(self.norm1(x)))).view(S, 3, self.heads, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]                                  # [S, H, hd]
        q = pol.r(q * cos[:, None] + rotate_half(q) * sin[:, None])
        k = pol.r(k * cos[:, None] + rotate_half(k) * sin[:, None])
        o = sdpa_masked(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), allowed, hd ** -0.5, pol)
        x = pol.r(x + self.attn.proj(pol.r(o.transpose(0, 1).reshape(S, d))))
        h = pol.r(self.norm2(x))
        h = pol.r(pol.r(F.silu(self.mlp.gate_proj(h))) * pol.r(self.mlp.up_proj(h)))
        return pol.r(x + self.mlp.down_proj(h))


class VisionTransformer(nn.Module):
    def __init__(self, depth=32, hidden_size=1280, intermediate_size=3420, num_heads=16, in_channels=3, patch_size=14,
                 spatial_merge_size=2, temporal_patch_size=2, window_size=112, out_hidden_size=3584,
                 fullatt_block_indexes=(7, 15, 23, 31), **_):
        super().__init__()
        self.cfg = SimpleNamespace(merge=spatial_merge_size, patch=patch_size, window=window_size, heads=num_heads,
                                   hidden=hidden_size, full=tuple(fullatt_block_indexes), tps=temporal_patch_size,
                                   cin=in_channels)
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv3d(in_channels, hidden_size, (temporal_patch_size, patch_size, patch_size),
                                          stride=(temporal_patch_size, patch_size, patch_size), bias=False)
        self.blocks = nn.ModuleList([VisionBlock(hidden_size, num_heads, intermediate_size) for _ in range(depth)])
        self.merger = nn.Module()
        self.merger.ln_q = RMSNorm(hidden_size)
        m = hidden_size * spatial_merge_size ** 2
        self.merger.mlp = nn.Sequential(nn.Linear(m, m), nn.GELU(), nn.Linear(m, out_hidden_size))

    def forward(self, pixel_values, grid_thw, pol: Policy = FP32):
        c = self.cfg
        grid = [[int(v) for v in g] for g in grid_thw.tolist()]
        x = pol.r(pixel_values.float()) @ self.patch_embed.proj.weight.reshape(c.hidden, -1).t()
        x = pol.r(x)
        S = x.shape[0]
        unit = c.merge ** 2
        widx, cu_win = vision_window_index(grid, c.merge, c.window, c.patch)
        x = x.reshape(S // unit, unit, -1)[widx].reshape(S, -1)
        hd = c.hidden // c.heads
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
        rot = (vision_position_ids(grid, c.merge).unsqueeze(-1) * inv).flatten(1)       # [S, hd/2]
        rot = rot.reshape(S // unit, unit, -1)[widx].reshape(S, -1)
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos(), emb.sin()
        cu_full = torch.tensor([0] + [h * w for t, h, w in grid for _ in range(t)]).cumsum(0)
        m_full, m_win = segment_mask(cu_full, S), segment_mask(cu_win, S)
        for i, blk in enumerate(self.blocks):
            x = blk(x, cos, sin, m_full if i in c.full else m_win, pol)
        h = pol.r(self.merger.ln_q(x)).view(S // unit, -1)
        h = pol.r(F.gelu(self.merger.mlp[0](h)))
        h = pol.r(self.merger.mlp[2](h))
        return h[torch.argsort(widx)]


# ---- decoder -----------------------------------------------------------------------------------------------------

def rope_index(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], grid_thw, image_token_id: int,
               merge: int) -> torch.Tensor:
    """Qwen2_5_VLModel.get_rope_index for image inputs: [3, B, S] (t, h, w) positions.  Text runs count up in all three
    components; an image block starting at position p gets t = p, h = p + row, w = p + col, and the following text resumes
    at p + max(rows, cols).  Text-only input: cumsum(mask) - 1, masked positions 1 (4.57 behaviour)."""
    B, S = input_ids.shape
    if grid_thw is None:
        if attention_mask is None:
            return torch.arange(S).view(1, 1, -1).expand(3, B, -1).clone()
        pos = attention_mask.long().cumsum(-1) - 1
        pos = pos.masked_fill(attention_mask == 0, 1)
        return pos.unsqueeze(0).expand(3, -1, -1).clone()
    grids = iter([[int(v) for v in g] for g in grid_thw.tolist()])
    out = torch.ones(3, B, S, dtype=torch.long)
    for b in range(B):
        keep = attention_mask[b].bool() if attention_mask is not None else torch.ones(S, dtype=torch.bool)
        ids = input_ids[b][keep].tolist()
        pos, cur, i = [], 0, 0
        while i < len(ids):
            if ids[i] == image_token_id:
                t, h, w = next(grids)
                gh, gw = h // merge, w // merge
                n = t * gh * gw
                tt = torch.arange(t).view(-1, 1, 1).expand(t, gh, gw).flatten()
                hh = torch.arange(gh).view(1, -1, 1).expand(t, gh, gw).flatten()
                ww = torch.arange(gw).view(1, 1, -1).expand(t, gh, gw).flatten()
                pos.append(torch.stack([tt, hh, ww]) + cur)
                cur += max(gh, gw)
                i += n
            else:
                j = i
                while j < len(ids) and ids[j] != image_token_id:
                    j += 1
                pos.append(torch.arange(j - i).view(1, -1).expand(3, -1) + cur)
                cur += j - i
                i = j
        out[:, b, keep] = torch.cat(pos, dim=1)
    return out


class DecoderLayer(nn.Module):
    def __init__(self, d, heads, kv_heads, inter, eps):
        super().__init__()
        self.heads, self.kv = heads, kv_heads
        hd = d // heads
        a = self.self_attn = nn.Module()
        a.q_proj, a.k_proj, a.v_proj = nn.Linear(d, heads * hd), nn.Linear(d, kv_heads * hd), nn.Linear(d, kv_heads * hd)
        a.o_proj = nn.Linear(heads * hd, d, bias=False)
        m = self.mlp = nn.Module()
        m.gate_proj, m.up_proj, m.down_proj = (nn.Linear(d, inter, bias=False), nn.Linear(d, inter, bias=False),
                                               nn.Linear(inter, d, bias=False))
        self.input_layernorm, self.post_attention_layernorm = RMSNorm(d, eps), RMSNorm(d, eps)

    def forward(self, x, cos, sin, allowed, pol: Policy):
        S, d = x.shape
        hd = d // self.heads
        a = self.self_attn
        h = pol.r(self.input_layernorm(x))
        q = pol.r(a.q_proj(h)).view(S, self.heads, hd).transpose(0, 1)
        k = pol.r(a.k_proj(h)).view(S, self.kv, hd).transpose(0, 1)
        v = pol.r(a.v_proj(h)).view(S, self.kv, hd).transpose(0, 1)
        q = pol.r(q * cos + rotate_half(q) * sin)
        k = pol.r(k * cos + rotate_half(k) * sin)
        rep = self.heads // self.kv
        k, v = k.repeat_interleave(rep, dim=0), v.repeat_interleave(rep, dim=0)
        o = pol.r(sdpa_masked(q, k, v, allowed, hd ** -0.5, pol).transpose(0, 1).reshape(S, d))
        x = pol.r(x + a.o_proj(o))
        h = pol.r(self.post_attention_layernorm(x))
        h = pol.r(pol.r(F.silu(self.mlp.gate_proj(h))) * pol.r(self.mlp.up_proj(h)))
        return pol.r(x + self.mlp.down_proj(h))


class Qwen2_5_VLForConditionalGeneration(nn.Module):
    def __init__(self, vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
                 num_attention_heads=28, num_key_value_heads=4, rms_norm_eps=1e-6, rope_theta=1000000.0,
                 mrope_section=(16, 24, 24), image_token_id=151655, vision_config=None, **_):
        super().__init__()
        self.image_token_id, self.mrope_section, self.rope_theta = image_token_id, tuple(mrope_section), rope_theta
        self.heads = num_attention_heads
        self.model = nn.Module()
        self.model.visual = VisionTransformer(**{**(vision_config or {}), "out_hidden_size": hidden_size})
        lm = self.model.language_model = nn.Module()
        lm.embed_tokens = nn.Embedding(vocab_size, hidden_size)
        lm.layers = nn.ModuleList([DecoderLayer(hidden_size, num_attention_heads, num_key_value_heads, intermediate_size,
                                                rms_norm_eps) for _ in range(num_hidden_layers)])
        lm.norm = RMSNorm(hidden_size, rms_norm_eps)
        self.lm_head = nn.Linear(hidden_size, vocab_size, bias=False)

    def mrope_cos_sin(self, position_ids: torch.Tensor, hd: int):
        """[3, S] positions -> cos, sin [S, hd]: frequency band i of the half-dim takes the t / h / w position according
        to mrope_section (apply_multimodal_rotary_pos_emb)."""
        inv = 1.0 / (self.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
        freqs = position_ids[:, :, None].float() * inv[None, None, :]                   # [3, S, hd/2]
        emb = torch.cat((freqs, freqs), dim=-1)                                          # [3, S, hd]
        sec = list(self.mrope_section) * 2
        cos = torch.cat([m[i % 3] for i, m in enumerate(emb.cos().split(sec, dim=-1))], dim=-1)
        sin = torch.cat([m[i % 3] for i, m in enumerate(emb.sin().split(sec, dim=-1))], dim=-1)
        return cos, sin

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, pixel_values=None, image_grid_thw=None, policy: Policy = FP32):
        pol, lm = policy, self.model.language_model
        B, S = input_ids.shape
        x = pol.r(lm.embed_tokens(input_ids))
        if pixel_values is not None:
            img = self.model.visual(pixel_values, image_grid_thw, pol)
            x = x.masked_scatter((input_ids == self.image_token_id).unsqueeze(-1), img)
        pos = rope_index(input_ids, attention_mask, image_grid_thw if pixel_values is not None else None,
                         self.image_token_id, self.model.visual.cfg.merge)
        hd = x.shape[-1] // self.heads
        causal = torch.ones(S, S, dtype=torch.bool).tril()
        hidden, outs = [x], []
        for b in range(B):
            allowed = causal if attention_mask is None else causal & attention_mask[b].bool()[None, :]
            cos, sin = self.mrope_cos_sin(pos[:, b], hd)
            h = x[b]
            per = []
            for layer in lm.layers:
                h = layer(h, cos, sin, allowed, pol)
                per.append(h)
            outs.append(per)
        for i in range(len(lm.layers) - 1):
            hidden.append(torch.stack([o[i] for o in outs]))
        last = pol.r(lm.norm(torch.stack([o[-1] for o in outs])))
        hidden.append(last)
        return SimpleNamespace(last_hidden_state=last, hidden_states=tuple(hidden))
