"""Qwen2.5-VL prompt encoder on the MI355X HIP ops: drop-in for `transformers.Qwen2_5_VLForConditionalGeneration` as the
QwenImage / QwenImage-Edit (and HunyuanVideo-1.5) engines use it (manifest
`base: Qwen2_5_VLForConditionalGeneration`, manifest/image/qwenimage-edit-2509-1.0.0.v1.yml:62-76;
engine/qwenimage/shared.py:183-226 calls `model(input_ids=, attention_mask=, pixel_values=, image_grid_thw=,
output_hidden_states=True)` and reads `hidden_states[-1]`).

State-dict keys are the 4.57 module layout the reference's converter produces (`model.visual.*`,
`model.language_model.*`, `lm_head.weight`; converters/text_encoder_converters.py:30-45).  Only the encoder use is
covered: one forward over the whole prompt, no KV cache, no generation, images only (no video inputs).

Vision tower: patch embedding as one GEMM over flattened patches, window reorder, RMS norm, fused QKV GEMM with the
80-wide heads laid out in 128-wide slots (zero rows in the packed weight), in-place rotate-half RoPE, block-diagonal
attention (`apexmi_attn_fwd_bias` with per-token segment ids: windows, or whole images in the full-attention blocks),
SwiGLU with the activation in the GEMM epilogue, patch merger.  Decoder: token gather, image embeddings scattered over
the `<|image_pad|>` rows, 3-D RoPE tables from the position index, per layer RMS norm -> fused QKV GEMM (+bias) ->
RoPE -> grouped-query causal attention (shared key heads addressed with stride 0, never materialised) -> output
projection with the residual in the epilogue -> SwiGLU.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .flux import _Config
from .text_encoders import _Base, _Emb, _N, _W, _cfg_dict


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


# ---- index arithmetic (host; integers only) ------------------------------------------------------------------------

def vision_position_ids(grid: List[List[int]], merge: int) -> torch.Tensor:
    out = []
    for t, h, w in grid:
        hp, wp = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        shape = (h // merge, merge, w // merge, merge)
        out.append(torch.stack([hp.reshape(shape).transpose(1, 2).flatten(), wp.reshape(shape).transpose(1, 2).flatten()],
                               dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_window_index(grid: List[List[int]], merge: int, window_size: int, patch_size: int):
    """Window reorder index (merged-token units) and per-patch window id, after the reorder."""
    window_index, seqlens_all = [], []
    base = 0
    vw = window_size // merge // patch_size
    for t, h, w in grid:
        gh, gw = h // merge, w // merge
        index = torch.arange(t * gh * gw).reshape(t, gh, gw)
        pad_h, pad_w = vw - gh % vw, vw - gw % vw
        nh, nw = (gh + pad_h) // vw, (gw + pad_w) // vw
        ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        ip = ip.reshape(t, nh, vw, nw, vw).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, vw, vw)
        seqlens_all.append((ip != -100).sum([2, 3]).reshape(-1))
        ip = ip.reshape(-1)
        window_index.append(ip[ip != -100] + base)
        base += t * gh * gw
    seqlens = torch.cat(seqlens_all) * merge * merge
    seg = torch.repeat_interleave(torch.arange(seqlens.numel()), seqlens)        # empty windows contribute nothing
    return torch.cat(window_index), seg.to(torch.int32)


def rope_index(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], grid, image_token_id: int,
               merge: int) -> torch.Tensor:
    """Qwen2_5_VLModel.get_rope_index for image inputs (see the oracle's docstring): [3, B, S] long, on the CPU."""
    B, S = input_ids.shape
    if grid is None:
        if attention_mask is None:
            return torch.arange(S).view(1, 1, -1).expand(3, B, -1).clone()
        pos = attention_mask.long().cumsum(-1) - 1
        return pos.masked_fill(attention_mask == 0, 1).unsqueeze(0).expand(3, -1, -1).clone()
    grids = iter(grid)
    out = torch.ones(3, B, S, dtype=torch.long)
    for b in range(B):
        keep = attention_mask[b].bool() if attention_mask is not None else torch.ones(S, dtype=torch.bool)
        ids = input_ids[b][keep].tolist()
        pos, cur, i = [], 0, 0
        while i < len(ids):
            if ids[i] == image_token_id:
                t, h, w = next(grids)
                gh, gw = h // merge, w // merge
                tt = torch.arange(t).view(-1, 1, 1).expand(t, gh, gw).flatten()
                hh = torch.arange(gh).view(1, -1, 1).expand(t, gh, gw).flatten()
                ww = torch.arange(gw).view(1, 1, -1).expand(t, gh, gw).flatten()
                pos.append(torch.stack([tt, hh, ww]) + cur)
                cur += max(gh, gw)
                i += t * gh * gw
            else:
                j = i
                while j < len(ids) and ids[j] != image_token_id:
                    j += 1
                pos.append(torch.arange(j - i).view(1, -1).expand(3, -1) + cur)
                cur += j - i
                i = j
        out[:, b, keep] = torch.cat(pos, dim=1)
    return out


# ---- modules (parameter holders with the reference's names) --------------------------------------------------------

class _VisionBlock(nn.Module):
    def __init__(self, d, inter, **kw):
        super().__init__()
        self.norm1, self.norm2 = _N(d, False, **kw), _N(d, False, **kw)
        self.attn = nn.Module()
        self.attn.qkv, self.attn.proj = _W(3 * d, d, True, **kw), _W(d, d, True, **kw)
        self.mlp = nn.Module()
        self.mlp.gate_proj, self.mlp.up_proj = _W(inter, d, True, **kw), _W(inter, d, True, **kw)
        self.mlp.down_proj = _W(d, inter, True, **kw)


class _Conv3dW(nn.Module):
    def __init__(self, cout, cin, kt, kh, kw_, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, kt, kh, kw_, **kw), requires_grad=False)


class _Vision(nn.Module):
    def __init__(self, c, out_hidden, **kw):
        super().__init__()
        self.patch_embed = nn.Module()
        self.patch_embed.proj = _Conv3dW(c.hidden_size, c.in_channels, c.temporal_patch_size, c.patch_size, c.patch_size, **kw)
        self.blocks = nn.ModuleList([_VisionBlock(c.hidden_size, c.intermediate_size, **kw) for _ in range(c.depth)])
        self.merger = nn.Module()
        self.merger.ln_q = _N(c.hidden_size, False, **kw)
        m = c.hidden_size * c.spatial_merge_size ** 2
        self.merger.mlp = nn.ModuleList([_W(m, m, True, **kw), nn.Identity(), _W(out_hidden, m, True, **kw)])


class _DecoderLayer(nn.Module):
    def __init__(self, d, heads, kv, inter, **kw):
        super().__init__()
        hd = d // heads
        a = self.self_attn = nn.Module()
        a.q_proj, a.k_proj, a.v_proj = _W(heads * hd, d, True, **kw), _W(kv * hd, d, True, **kw), _W(kv * hd, d, True, **kw)
        a.o_proj = _W(d, heads * hd, False, **kw)
        m = self.mlp = nn.Module()
        m.gate_proj, m.up_proj, m.down_proj = _W(inter, d, False, **kw), _W(inter, d, False, **kw), _W(d, inter, False, **kw)
        self.input_layernorm, self.post_attention_layernorm = _N(d, False, **kw), _N(d, False, **kw)


_VISION_DEFAULTS = dict(depth=32, hidden_size=1280, intermediate_size=3420, num_heads=16, in_channels=3, patch_size=14,
                        spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=(7, 15, 23, 31))


class Qwen2_5_VLForConditionalGeneration(_Base):
    def __init__(self, config=None, device=None, dtype=torch.bfloat16, **kwargs):
        super().__init__()
        cfg = _cfg_dict(config, kwargs)
        text = dict(cfg.get("text_config") or {})
        for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                  "num_key_value_heads", "rms_norm_eps", "rope_theta", "rope_scaling", "rope_parameters"):
            if k in cfg and k not in text:          # 4.x configs keep the decoder fields at the top level
                text[k] = cfg[k]
        rope = dict(text.get("rope_parameters") or text.get("rope_scaling") or {})
        vis = {**_VISION_DEFAULTS, **{k: v for k, v in dict(cfg.get("vision_config") or {}).items() if k in _VISION_DEFAULTS}}
        c = self.config = _Config(
            vocab_size=text.get("vocab_size", 152064), hidden_size=text.get("hidden_size", 3584),
            intermediate_size=text.get("intermediate_size", 18944), num_hidden_layers=text.get("num_hidden_layers", 28),
            num_attention_heads=text.get("num_attention_heads", 28), num_key_value_heads=text.get("num_key_value_heads", 4),
            rms_norm_eps=text.get("rms_norm_eps", 1e-6), rope_theta=float(rope.get("rope_theta", text.get("rope_theta", 1e6))),
            mrope_section=tuple(rope.get("mrope_section", cfg.get("mrope_section", (16, 24, 24)))),
            image_token_id=cfg.get("image_token_id", 151655), vision_config=_Config(**vis))
        hd = c.hidden_size // c.num_attention_heads
        if hd % 64 or (c.num_key_value_heads * hd) % 128 or sum(c.mrope_section) * 2 != hd:
            raise NotImplementedError(f"qwen2.5-vl (mi355): head dim {hd} / kv heads {c.num_key_value_heads} unsupported")
        kw = dict(device=device, dtype=dtype)
        self.model = nn.Module()
        self.model.visual = _Vision(c.vision_config, c.hidden_size, **kw)
        lm = self.model.language_model = nn.Module()
        lm.embed_tokens = _Emb(c.vocab_size, c.hidden_size, **kw)
        lm.layers = nn.ModuleList([_DecoderLayer(c.hidden_size, c.num_attention_heads, c.num_key_value_heads,
                                                 c.intermediate_size, **kw) for _ in range(c.num_hidden_layers)])
        lm.norm = _N(c.hidden_size, False, **kw)
        self.lm_head = _W(c.vocab_size, c.hidden_size, False, **kw)     # present in the checkpoints; unused by the encoder
        self._fused: Dict = {}

    # ---- packed weights (built once per device placement) ----------------------------------------------------------
    def _vis_pack(self, i: int, blk: _VisionBlock):
        """80-wide heads in 128-wide slots, intermediate size padded to a multiple of 64 (zeros), so every GEMM sees
        K % 64 == 0 and the attention kernels a head dim of 128."""
        key = ("vis", i)
        f = self._fused.get(key)
        if f is None:
            v = self.config.vision_config
            d, H = v.hidden_size, v.num_heads
            hd, slot = d // H, _pad_to(d // H, 64)
            w = blk.attn.qkv.weight.data.view(3, H, hd, d)
            wq = torch.zeros(3, H, slot, d, dtype=w.dtype, device=w.device)
            wq[:, :, :hd] = w
            bq = torch.zeros(3, H, slot, dtype=w.dtype, device=w.device)
            bq[:, :, :hd] = blk.attn.qkv.bias.data.view(3, H, hd)
            wp = torch.zeros(d, H, slot, dtype=w.dtype, device=w.device)
            wp[:, :, :hd] = blk.attn.proj.weight.data.view(d, H, hd)
            inter = v.intermediate_size
            ip = _pad_to(inter, 64)
            def padrows(m):
                o = torch.zeros(ip, d, dtype=w.dtype, device=w.device)
                o[:inter] = m.weight.data
                b = torch.zeros(ip, dtype=w.dtype, device=w.device)
                b[:inter] = m.bias.data
                return o, b
            wg, bg = padrows(blk.mlp.gate_proj)
            wu, bu = padrows(blk.mlp.up_proj)
            wd = torch.zeros(d, ip, dtype=w.dtype, device=w.device)
            wd[:, :inter] = blk.mlp.down_proj.weight.data
            f = dict(wqkv=wq.view(3 * H * slot, d), bqkv=bq.view(-1), wproj=wp.view(d, H * slot).contiguous(), slot=slot,
                     wg=wg, bg=bg, wu=wu, bu=bu, wd=wd)
            self._fused[key] = f
        return f

    def _patch_weight(self):
        f = self._fused.get("patch")
        if f is None:
            w = self.model.visual.patch_embed.proj.weight.data
            k = w[0].numel()
            kp = _pad_to(k, 64)
            wp = torch.zeros(w.shape[0], kp, dtype=w.dtype, device=w.device)
            wp[:, :k] = w.reshape(w.shape[0], k)
            f = (wp, k, kp)
            self._fused["patch"] = f
        return f

    # ---- vision tower ----------------------------------------------------------------------------------------------
    @ops.on_model_device
    def get_image_features(self, pixel_values: torch.Tensor, image_grid_thw) -> torch.Tensor:
        """pixel_values [patches, C * T_p * P * P] (the processor's flattened patches) -> merged image tokens
        [patches / merge^2, hidden_size], in input order."""
        v = self.config.vision_config
        dev = self.device
        grid = [[int(x) for x in g] for g in (image_grid_thw.tolist() if torch.is_tensor(image_grid_thw) else image_grid_thw)]
        wp, k, kp = self._patch_weight()
        S = pixel_values.shape[0]
        px = torch.zeros(S, kp, dtype=self.storage_dtype, device=dev)
        px[:, :k] = pixel_values.to(dev, self.storage_dtype)
        unit = v.spatial_merge_size ** 2
        widx, seg_win = vision_window_index(grid, v.spatial_merge_size, v.window_size, v.patch_size)
        widx_d = widx.to(dev)
        x = ops.gemm(px, wp).view(S // unit, unit, -1)[widx_d].reshape(S, -1).contiguous()
        d, H = v.hidden_size, v.num_heads
        hd = d // H
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
        rot = (vision_position_ids(grid, v.spatial_merge_size).unsqueeze(-1) * inv).flatten(1)
        rot = rot.reshape(S // unit, unit, -1)[widx].reshape(S, -1)
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos().to(dev).contiguous(), emb.sin().to(dev).contiguous()
        seg_full = torch.repeat_interleave(torch.arange(sum(t for t, _, _ in grid)),
                                           torch.tensor([h * w for t, h, w in grid for _ in range(t)])).to(torch.int32)
        seg_win, seg_full = seg_win.to(dev), seg_full.to(dev)
        ones = self._ones(d)
        for i, blk in enumerate(self.model.visual.blocks):
            p = self._vis_pack(i, blk)
            slot, inner = p["slot"], H * p["slot"]
            qkv = ops.gemm(ops.ln_modulate(x, gamma=blk.norm1.weight.data, rms=True, eps=1e-6), p["wqkv"], p["bqkv"])
            ops.rope_half_(qkv[:, :2 * inner], 2 * H, slot, cos, sin)                    # q and k heads, first 80 columns each
            a = ops.attention_bias(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], H, hd ** -0.5,
                                   seg=seg_full if i in v.fullatt_block_indexes else seg_win)
            x = ops.gemm(a, p["wproj"], blk.attn.proj.bias.data, epilogue="gate_res", gate=ones, residual=x)
            h = ops.ln_modulate(x, gamma=blk.norm2.weight.data, rms=True, eps=1e-6)
            h = ops.mul(ops.gemm(h, p["wg"], p["bg"], epilogue="silu"), ops.gemm(h, p["wu"], p["bu"]))
            x = ops.gemm(h, p["wd"], blk.mlp.down_proj.bias.data, epilogue="gate_res", gate=ones, residual=x)
        mg = self.model.visual.merger
        h = ops.ln_modulate(x, gamma=mg.ln_q.weight.data, rms=True, eps=1e-6).view(S // unit, unit * d)
        h = ops.gemm(h, mg.mlp[0].weight.data, mg.mlp[0].bias.data, epilogue="gelu_erf")
        h = ops.gemm(h, mg.mlp[2].weight.data, mg.mlp[2].bias.data)
        return h[torch.argsort(widx).to(dev)]

    # ---- decoder ---------------------------------------------------------------------------------------------------
    def _mrope(self, pos: torch.Tensor, hd: int):
        c = self.config
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
        freqs = pos[:, :, None].float() * inv[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        sec = list(c.mrope_section) * 2
        cos = torch.cat([m[i % 3] for i, m in enumerate(emb.cos().split(sec, dim=-1))], dim=-1)
        sin = torch.cat([m[i % 3] for i, m in enumerate(emb.sin().split(sec, dim=-1))], dim=-1)
        return cos.to(self.device).contiguous(), sin.to(self.device).contiguous()

    @ops.on_model_device
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, image_grid_thw=None,
                output_hidden_states=False, return_dict=True, pixel_values_videos=None, video_grid_thw=None, **_):
        self._check(input_ids)
        if pixel_values_videos is not None or video_grid_thw is not None:
            raise NotImplementedError("qwen2.5-vl (mi355): video inputs are not built (no engine of the reference passes them)")
        c, lm, dev = self.config, self.model.language_model, self.device
        B, S = input_ids.shape
        ids_cpu = input_ids.detach().cpu()
        mask_cpu = attention_mask.detach().cpu() if attention_mask is not None else None
        ids = input_ids.to(dev, torch.int64).reshape(-1).contiguous()
        if int(ids.min()) < 0 or int(ids.max()) >= c.vocab_size:
            raise IndexError("input_ids out of range for the embedding table")
        x = ops.gather_rows(lm.embed_tokens.weight.data, ids, out_dtype=self.storage_dtype)
        grid = None
        if pixel_values is not None:
            grid = [[int(v) for v in g] for g in image_grid_thw.tolist()]
            img = self.get_image_features(pixel_values, grid)
            rows = (ids == c.image_token_id).nonzero(as_tuple=True)[0]
            if rows.numel() != img.shape[0]:
                raise ValueError(f"Image features and image tokens do not match: tokens: {rows.numel()}, features {img.shape[0]}")
            x[rows] = img
        pos = rope_index(ids_cpu, mask_cpu, grid, c.image_token_id, c.vision_config.spatial_merge_size)
        H, Hkv, d = c.num_attention_heads, c.num_key_value_heads, c.hidden_size
        hd = d // H
        nq, nkv = H * hd, Hkv * hd
        tables = [self._mrope(pos[:, b], hd) for b in range(B)]
        keep = None if attention_mask is None else (attention_mask.to(dev) != 0).to(torch.uint8).contiguous()
        ones = self._ones(d)
        hidden = [x.view(B, S, d)] if output_hidden_states else []
        n = len(lm.layers)
        for li, layer in enumerate(lm.layers):
            at = layer.self_attn
            wqkv, bqkv = self._qkv(id(at), (at.q_proj, at.k_proj, at.v_proj))
            qkv = ops.gemm(ops.ln_modulate(x, gamma=layer.input_layernorm.weight.data, rms=True, eps=c.rms_norm_eps), wqkv, bqkv)
            a = torch.empty((B * S, nq), dtype=x.dtype, device=dev)
            for b in range(B):
                r = slice(b * S, (b + 1) * S)
                ops.rope_half_(qkv[r, :nq + nkv], H + Hkv, hd, *tables[b])
                ops.attention_bias(qkv[r, :nq], qkv[r, nq:nq + nkv], qkv[r, nq + nkv:], H, hd ** -0.5,
                                   keep=None if keep is None else keep[b], causal=True, kv_heads=Hkv, out=a[r])
            x = ops.gemm(a, at.o_proj.weight.data, epilogue="gate_res", gate=ones, residual=x)
            h = ops.ln_modulate(x, gamma=layer.post_attention_layernorm.weight.data, rms=True, eps=c.rms_norm_eps)
            h = ops.mul(ops.gemm(h, layer.mlp.gate_proj.weight.data, epilogue="silu"), ops.gemm(h, layer.mlp.up_proj.weight.data))
            x = ops.gemm(h, layer.mlp.down_proj.weight.data, epilogue="gate_res", gate=ones, residual=x)
            if output_hidden_states and li < n - 1:
                hidden.append(x.view(B, S, d))
        last = ops.ln_modulate(x, gamma=lm.norm.weight.data, rms=True, eps=c.rms_norm_eps).view(B, S, d)
        if output_hidden_states:
            hidden.append(last)
        out = SimpleNamespace(last_hidden_state=last, hidden_states=tuple(hidden) if output_hidden_states else None)
        return out if return_dict else (last,) + ((out.hidden_states,) if output_hidden_states else ())
