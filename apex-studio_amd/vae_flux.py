"""AutoencoderKL (Flux 2-D VAE) decode on the MI355X HIP ops — drop-in for VAE registry key "auto".

Mirrors what the Flux engine uses of the reference wrapper (apps/api/src/vae/auto/model.py:44-549):
`from_config`, diffusers state-dict keys `decoder.*`, `.config` (scaling_factor, shift_factor,
latent_channels, block_out_channels), `denormalize_latents`, `decode(z, return_dict=False)[0]`.
The decoder arithmetic is diffusers' `Decoder` (conv_in -> mid block with one 512-channel attention ->
4 up blocks -> GroupNorm+SiLU -> conv_out).  Runs channels-last: 3x3 convs are the
implicit-GEMM MFMA conv kernel (kT = 1), GroupNorm32+SiLU a three-pass deterministic kernel, the mid
attention goes through the attention operator.  At 1024^2 the latent is 128x128 = the VAE's
tile_latent_min_size, so the reference does not tile (model.py:229-236) and neither does this.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .flux import _Config, _Linear
from .vae_wan import _Conv


class _GN(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c, **kw), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c, **kw), requires_grad=False)


class _Res2D(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.norm1 = _GN(cin, **kw)
        self.conv1 = _Conv(cin, cout, (3, 3), **kw)
        self.norm2 = _GN(cout, **kw)
        self.conv2 = _Conv(cout, cout, (3, 3), **kw)
        if cin != cout:
            self.conv_shortcut = _Conv(cin, cout, (1, 1), **kw)
        else:
            self.conv_shortcut = None


class _Attn2D(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.group_norm = _GN(c, **kw)
        self.to_q, self.to_k, self.to_v = _Linear(c, c, **kw), _Linear(c, c, **kw), _Linear(c, c, **kw)
        self.to_out = nn.ModuleList([_Linear(c, c, **kw), nn.Identity()])


class _Mid2D(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Res2D(c, c, **kw), _Res2D(c, c, **kw)])
        self.attentions = nn.ModuleList([_Attn2D(c, **kw)])


class _Upsample(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.conv = _Conv(c, c, (3, 3), **kw)


class _UpBlock2D(nn.Module):
    def __init__(self, cin, cout, n, up, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Res2D(cin if i == 0 else cout, cout, **kw) for i in range(n)])
        self.upsamplers = nn.ModuleList([_Upsample(cout, **kw)]) if up else None


class _Decoder2D(nn.Module):
    def __init__(self, latent_channels, out_channels, block_out_channels, layers_per_block, **kw):
        super().__init__()
        rev = list(block_out_channels)[::-1]
        self.conv_in = _Conv(latent_channels, rev[0], (3, 3), **kw)
        self.mid_block = _Mid2D(rev[0], **kw)
        ups, prev = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(_UpBlock2D(prev, c, layers_per_block + 1, i != len(rev) - 1, **kw))
            prev = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = _GN(rev[-1], **kw)
        self.conv_out = _Conv(rev[-1], out_channels, (3, 3), **kw)


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, latent_channels: int = 16,
                 block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2, norm_num_groups: int = 32,
                 scaling_factor: float = 0.3611, shift_factor: float = 0.1159, sample_size: int = 1024,
                 use_quant_conv: bool = False, use_post_quant_conv: bool = False, device=None,
                 dtype=torch.bfloat16, **_ignored):
        super().__init__()
        if use_post_quant_conv or norm_num_groups != 32:
            raise NotImplementedError("flux VAE: post_quant_conv / non-32 group norm are not configured for FLUX.1")
        self.config = _Config(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              norm_num_groups=norm_num_groups, scaling_factor=scaling_factor,
                              shift_factor=shift_factor, sample_size=sample_size)
        kw = dict(device=device, dtype=dtype)
        self.decoder = _Decoder2D(latent_channels, out_channels, block_out_channels, layers_per_block, **kw)
        self._packed: Dict[int, tuple] = {}
        self.storage_dtype = torch.bfloat16

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config) if isinstance(config, dict) else dict(vars(config))
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    _from_config = from_config

    # ---- activation storage ------------------------------------------------------------------------------------------
    def set_storage_dtype(self, dtype: torch.dtype):
        """torch.bfloat16 (production) or torch.float32: the f32-STORAGE VERIFICATION MODE — the same
        kernel sequence with every activation buffer float and the library's `_f32` entry points, which is what
        north_star's "within 1e-3 of the CPU fp32 reference" is tested with.  Weights stay bf16."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError(f"activation storage must be bfloat16 or float32, got {dtype}")
        self.storage_dtype = dtype
        return self

    @property
    def dtype(self):
        return self.decoder.conv_in.weight.dtype

    @property
    def device(self):
        return self.decoder.conv_in.weight.device

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        return super().load_state_dict(*a, **k)

    def _weights_changed(self):
        """Parameters were written in place (`weights.load_checkpoint_into`): drop the packed conv-weight cache."""
        self._packed = {}

    def enable_tiling(self, *a, **k):   # 1024^2 latents equal tile_latent_min_size: untiled (model.py:229-236)
        return None

    def enable_slicing(self):
        return None

    @torch.no_grad()
    def denormalize_latents(self, latents):
        return latents / self.config.scaling_factor + self.config.shift_factor

    @torch.no_grad()
    def normalize_latents(self, latents):
        return (latents - self.config.shift_factor) * self.config.scaling_factor

    def _conv(self, conv: _Conv, x, residual=None, upsample2x=False):
        key = id(conv)
        p = self._packed.get(key)
        if p is None:
            w = ops.pack_conv_weight(conv.weight.data)
            b = torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
            b[:conv.bias.numel()] = conv.bias.data
            p = (w, b)
            self._packed[key] = p
        return ops.conv3d_cl(x, p[0], p[1], (1,) + conv.ksize, residual=residual, upsample2x=upsample2x)

    def _res(self, blk: _Res2D, x):
        s = x if blk.conv_shortcut is None else self._conv(blk.conv_shortcut, x)
        h = self._conv(blk.conv1, ops.groupnorm_cl(x, blk.norm1.weight.data, blk.norm1.bias.data, silu=True))
        return self._conv(blk.conv2, ops.groupnorm_cl(h, blk.norm2.weight.data, blk.norm2.bias.data, silu=True),
                          residual=s)

    def _attn(self, a: _Attn2D, x):
        _, H, W, Cc = x.shape
        n = ops.groupnorm_cl(x, a.group_norm.weight.data, a.group_norm.bias.data).view(H * W, Cc)
        q = ops.gemm(n, a.to_q.weight.data, a.to_q.bias.data)
        k = ops.gemm(n, a.to_k.weight.data, a.to_k.bias.data)
        v = ops.gemm(n, a.to_v.weight.data, a.to_v.bias.data)
        o = ops.attention(q.view(1, 1, H * W, Cc), k.view(1, 1, H * W, Cc), v.view(1, 1, H * W, Cc))
        o = o.permute(0, 2, 1, 3).reshape(H * W, Cc)
        ones = torch.ones(Cc, dtype=torch.float32, device=x.device)
        out = ops.gemm(o, a.to_out[0].weight.data, a.to_out[0].bias.data, epilogue="gate_res", gate=ones,
                       residual=x.view(H * W, Cc))
        return out.view(1, H, W, Cc)

    @torch.no_grad()
    def _decode_one(self, z):
        if z.device.type != "cuda" or self.dtype != torch.bfloat16:
            raise _l.ApexMIError("flux VAE needs bf16 weights and latents on a ROCm device (no CPU fallback)")
        d = self.decoder
        x = z.to(self.storage_dtype).permute(1, 2, 0).unsqueeze(0).contiguous()      # [1, H, W, C]
        x = self._conv(d.conv_in, x)
        x = self._res(d.mid_block.resnets[0], x)
        x = self._attn(d.mid_block.attentions[0], x)
        x = self._res(d.mid_block.resnets[1], x)
        for up in d.up_blocks:
            for r in up.resnets:
                x = self._res(r, x)
            if up.upsamplers is not None:
                # diffusers Upsample2D (nearest 2x + conv) with the upsample folded into the convolution's gather
                x = self._conv(up.upsamplers[0].conv, x, upsample2x=True)
        x = ops.groupnorm_cl(x, d.conv_norm_out.weight.data, d.conv_norm_out.bias.data, silu=True)
        x = self._conv(d.conv_out, x)
        return x[0, :, :, :self.config.out_channels].permute(2, 0, 1).contiguous()

    @ops.on_model_device
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        dec = torch.stack([self._decode_one(z[b]) for b in range(z.shape[0])], dim=0).to(z.dtype)
        if not return_dict:
            return (dec,)
        return SimpleNamespace(sample=dec)
