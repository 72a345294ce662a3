"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the Wan / QwenImage 3-D causal VAE DECODE path:
  AutoencoderKLWan._decode / tiled_decode     reference vae/wan/model.py:1333-1376, :1516-1623
  WanDecoder3d.forward                        :972-1021
  WanMidBlock / WanUpBlock / WanResidualBlock :523-533, :853-878, :389-441
  WanResample.forward ("upsample2d/3d")       :291-353   (the "Rep" first-chunk rule)
  WanCausalConv3d.forward                     :178-185
  WanRMS_norm.forward                         :216-222
  WanAttentionBlock.forward                   :461-490
  blend_v / blend_h                           :1404-1422
  denormalize_latents                         :1649-1660

and of the ENCODE path the engines call for condition images / first frames (`vae_encode`,
engine/base_engine.py:2061-2165): AutoencoderKLWan._encode / tiled_encode (:1273-1304, :1424-1514),
WanEncoder3d.forward (:665-713), WanResample "downsample2d/3d" (:276-286, :336-365) — class AutoencoderKLWanEncoder.

Restated as ONE causal pass over a tile's whole frame sequence instead of the reference's per-frame
streaming with `feat_cache`: a kernel-3 causal convolution fed the last two cached input frames is the
same sum as the convolution over the full zero-left-padded sequence, and the temporal upsampler's
sentinel logic reduces to "frame 0 passes through, time_conv runs causally over frames 1.. (never seeing
frame 0)".  tests/golden/vae_wan.pt holds outputs of the REFERENCE class (streaming, tiled and untiled)
run in this container; the CPU test requires this restatement to match them to 1e-5, which is what pins
the reformulation the HIP path uses.  Parameter names equal the reference's `decoder.*` /
`post_quant_conv.*` state-dict keys.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .layers import Policy, FP32


class CausalConv3d(nn.Conv3d):
    """time padding 2*p on the left only; spatial padding symmetric."""

    def __init__(self, cin, cout, kernel_size, padding=0):
        super().__init__(cin, cout, kernel_size, padding=0)
        p = (padding,) * 3 if isinstance(padding, int) else tuple(padding)
        self._pad = (p[2], p[2], p[1], p[1], 2 * p[0], 0)

    def forward(self, x):
        return super().forward(F.pad(x, self._pad))


class RMSNorm(nn.Module):
    def __init__(self, dim: int, images: bool = True):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones((dim, 1, 1) if images else (dim, 1, 1, 1)))

    def forward(self, x):
        return F.normalize(x, dim=1) * self.scale * self.gamma


class ResidualBlock(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.norm1 = RMSNorm(cin, images=False)
        self.conv1 = CausalConv3d(cin, cout, 3, padding=1)
        self.norm2 = RMSNorm(cout, images=False)
        self.conv2 = CausalConv3d(cout, cout, 3, padding=1)
        self.conv_shortcut = CausalConv3d(cin, cout, 1) if cin != cout else nn.Identity()

    def forward(self, x, pol: Policy):
        # x is a stored tensor already: an Identity shortcut is not a new storage point
        h = x if isinstance(self.conv_shortcut, nn.Identity) else pol.r(self.conv_shortcut(x))
        y = pol.r(self.conv1(pol.r(F.silu(self.norm1(x)))))
        y = self.conv2(pol.r(F.silu(self.norm2(y))))
        return pol.r(y + h)


class AttentionBlock(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.norm = RMSNorm(dim)
        self.to_qkv = nn.Conv2d(dim, dim * 3, 1)
        self.proj = nn.Conv2d(dim, dim, 1)

    def forward(self, x, pol: Policy):
        b, c, t, h, w = x.shape
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        qkv = pol.r(self.to_qkv(pol.r(self.norm(y))))
        qkv = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2)
        q, k, v = qkv.chunk(3, dim=-1)
        o = pol.r(L.sdpa_dispatch(q, k, v, policy=pol))
        o = o.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
        o = self.proj(o).view(b, t, c, h, w).permute(0, 2, 1, 3, 4)
        return pol.r(o + x)


class Resample(nn.Module):
    def __init__(self, dim: int, mode: str):
        super().__init__()
        self.mode = mode
        self.resample = nn.Sequential(nn.Upsample(scale_factor=(2.0, 2.0), mode="nearest-exact"),
                                      nn.Conv2d(dim, dim // 2, 3, padding=1))
        if mode == "upsample3d":
            self.time_conv = CausalConv3d(dim, dim * 2, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x, pol: Policy):
        b, c, t, h, w = x.shape
        if self.mode == "upsample3d" and t > 1:
            y = pol.r(self.time_conv(x[:, :, 1:]))                  # never sees frame 0 (the "Rep" rule)
            y = y.reshape(b, 2, c, t - 1, h, w)
            y = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, 2 * (t - 1), h, w)
            x = torch.cat([x[:, :, :1], y], dim=2)
        t = x.shape[2]
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        y = pol.r(self.resample(y))
        return y.view(b, t, y.size(1), y.size(2), y.size(3)).permute(0, 2, 1, 3, 4)


class MidBlock(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.resnets = nn.ModuleList([ResidualBlock(dim, dim), ResidualBlock(dim, dim)])
        self.attentions = nn.ModuleList([AttentionBlock(dim)])

    def forward(self, x, pol):
        x = self.resnets[0](x, pol)
        x = self.attentions[0](x, pol)
        return self.resnets[1](x, pol)


class UpBlock(nn.Module):
    def __init__(self, cin: int, cout: int, num_res_blocks: int, mode: Optional[str]):
        super().__init__()
        res, cur = [], cin
        for _ in range(num_res_blocks + 1):
            res.append(ResidualBlock(cur, cout))
            cur = cout
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Resample(cout, mode)]) if mode else None

    def forward(self, x, pol):
        for r in self.resnets:
            x = r(x, pol)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, pol)
        return x


class Decoder3d(nn.Module):
    def __init__(self, dim: int, z_dim: int, dim_mult: List[int], num_res_blocks: int,
                 temperal_upsample: List[bool], out_channels: int = 3):
        super().__init__()
        dims = [dim * u for u in [dim_mult[-1]] + dim_mult[::-1]]
        self.conv_in = CausalConv3d(z_dim, dims[0], 3, padding=1)
        self.mid_block = MidBlock(dims[0])
        ups = []
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            if i > 0:
                cin = cin // 2
            up = i != len(dim_mult) - 1
            mode = ("upsample3d" if temperal_upsample[i] else "upsample2d") if up else None
            ups.append(UpBlock(cin, cout, num_res_blocks, mode))
        self.up_blocks = nn.ModuleList(ups)
        self.norm_out = RMSNorm(dims[-1], images=False)
        self.conv_out = CausalConv3d(dims[-1], out_channels, 3, padding=1)

    def forward(self, x, pol: Policy):
        x = pol.r(self.conv_in(x))
        x = self.mid_block(x, pol)
        for u in self.up_blocks:
            x = u(x, pol)
        return pol.r(self.conv_out(pol.r(F.silu(self.norm_out(x)))))


class AutoencoderKLWanDecoder(nn.Module):
    """decode half of AutoencoderKLWan (the encoder is not on the hot path)."""

    def __init__(self, base_dim: int = 96, z_dim: int = 16, dim_mult=(1, 2, 4, 4), num_res_blocks: int = 2,
                 temperal_downsample=(False, True, True), out_channels: int = 3,
                 latents_mean=None, latents_std=None, scale_factor_spatial: int = 8):
        super().__init__()
        self.z_dim = z_dim
        self.post_quant_conv = CausalConv3d(z_dim, z_dim, 1)
        self.decoder = Decoder3d(base_dim, z_dim, list(dim_mult), num_res_blocks,
                                 list(temperal_downsample)[::-1], out_channels)
        self.ratio = scale_factor_spatial
        self.latents_mean, self.latents_std = latents_mean, latents_std
        self.tile_min, self.tile_stride, self.use_tiling = (256, 256), (192, 192), False

    def enable_tiling(self, min_h=None, min_w=None, stride_h=None, stride_w=None):
        self.use_tiling = True
        self.tile_min = (min_h or self.tile_min[0], min_w or self.tile_min[1])
        self.tile_stride = (stride_h or self.tile_stride[0], stride_w or self.tile_stride[1])

    def denormalize_latents(self, z):
        mean = torch.tensor(self.latents_mean).view(1, self.z_dim, 1, 1, 1).to(z)
        inv_std = 1.0 / torch.tensor(self.latents_std).view(1, self.z_dim, 1, 1, 1).to(z)
        return z / inv_std + mean

    def _tile(self, z, pol):
        return self.decoder(pol.r(self.post_quant_conv(z)), pol)

    @staticmethod
    def _blend(a, b, extent, dim, pol: Policy = FP32):
        """Linear cross-fade of an overlap, written into `b` in place like the reference (model.py:1404-1422); the blended
        values are a storage point (the HIP crossfade kernel writes bf16 tiles)."""
        extent = min(a.shape[dim], b.shape[dim], extent)
        w = (torch.arange(extent, dtype=a.dtype) / extent).view([-1 if d == dim % 5 else 1 for d in range(5)])
        sa = [slice(None)] * 5
        sb = [slice(None)] * 5
        sa[dim], sb[dim] = slice(a.shape[dim] - extent, None), slice(0, extent)
        b[tuple(sb)] = pol.r(a[tuple(sa)] * (1 - w) + b[tuple(sb)] * w)
        return b

    @torch.no_grad()
    def decode(self, z, policy: Policy = FP32):
        pol = policy
        _, _, _, H, W = z.shape
        lat_min = (self.tile_min[0] // self.ratio, self.tile_min[1] // self.ratio)
        if not (self.use_tiling and (W > lat_min[1] or H > lat_min[0])):
            return torch.clamp(self._tile(z, pol), -1.0, 1.0)
        lat_stride = (self.tile_stride[0] // self.ratio, self.tile_stride[1] // self.ratio)
        blend = (self.tile_min[0] - self.tile_stride[0], self.tile_min[1] - self.tile_stride[1])
        rows = [[self._tile(z[:, :, :, i:i + lat_min[0], j:j + lat_min[1]], pol)
                 for j in range(0, W, lat_stride[1])] for i in range(0, H, lat_stride[0])]
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(rows[i - 1][j], tile, blend[0], 3, pol)   # in place, like the reference
                if j > 0:
                    tile = self._blend(row[j - 1], tile, blend[1], 4, pol)
                out.append(tile[:, :, :, :self.tile_stride[0], :self.tile_stride[1]])
            out_rows.append(torch.cat(out, dim=-1))
        dec = torch.cat(out_rows, dim=3)[:, :, :, :H * self.ratio, :W * self.ratio]
        return torch.clamp(dec, -1.0, 1.0)


# ---- encode half ------------------------------------------------------------------------------------------------

class Downsample(nn.Module):
    """WanResample "downsample2d" / "downsample3d".  Full-sequence form of the streaming temporal downsample: the first
    frame passes through and output j >= 1 is the kernel-3 convolution of frames (2j-2, 2j-1, 2j) — what the
    reference computes chunk by chunk from the cached last frame (model.py:340-365) for 1 + 4k input frames."""

    def __init__(self, dim: int, mode: str):
        super().__init__()
        self.mode = mode
        self.resample = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
        if mode == "downsample3d":
            self.time_conv = nn.Conv3d(dim, dim, (3, 1, 1), stride=(2, 1, 1))

    def forward(self, x, pol: Policy):
        b, c, t, h, w = x.shape
        y = pol.r(self.resample(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)))
        x = y.view(b, t, c, y.size(2), y.size(3)).permute(0, 2, 1, 3, 4)
        if self.mode == "downsample3d" and t > 1:
            x = torch.cat([x[:, :, :1], pol.r(self.time_conv(x))], dim=2)
        return x


class Encoder3d(nn.Module):
    def __init__(self, in_channels: int, dim: int, z_dim: int, dim_mult: List[int], num_res_blocks: int,
                 temperal_downsample: List[bool]):
        super().__init__()
        dims = [dim * u for u in [1] + list(dim_mult)]
        self.conv_in = CausalConv3d(in_channels, dims[0], 3, padding=1)
        blocks = []
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                blocks.append(ResidualBlock(cin, cout))
                cin = cout
            if i != len(dim_mult) - 1:
                blocks.append(Downsample(cout, "downsample3d" if temperal_downsample[i] else "downsample2d"))
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock(dims[-1])
        self.norm_out = RMSNorm(dims[-1], images=False)
        self.conv_out = CausalConv3d(dims[-1], z_dim, 3, padding=1)

    def forward(self, x, pol: Policy):
        x = pol.r(self.conv_in(x))
        for blk in self.down_blocks:
            x = blk(x, pol)
        x = self.mid_block(x, pol)
        return pol.r(self.conv_out(pol.r(F.silu(self.norm_out(x)))))


class AutoencoderKLWanEncoder(nn.Module):
    """encode half of AutoencoderKLWan: `encode(x)` returns the posterior parameters [B, 2 z, T', H/8, W/8] (mean | logvar);
    `.mode()` of the reference's DiagonalGaussianDistribution is the first z channels."""

    def __init__(self, base_dim: int = 96, z_dim: int = 16, dim_mult=(1, 2, 4, 4), num_res_blocks: int = 2,
                 temperal_downsample=(False, True, True), in_channels: int = 3, latents_mean=None, latents_std=None,
                 scale_factor_spatial: int = 8):
        super().__init__()
        self.z_dim = z_dim
        self.encoder = Encoder3d(in_channels, base_dim, z_dim * 2, list(dim_mult), num_res_blocks, list(temperal_downsample))
        self.quant_conv = CausalConv3d(z_dim * 2, z_dim * 2, 1)
        self.ratio = scale_factor_spatial
        self.latents_mean, self.latents_std = latents_mean, latents_std
        self.tile_min, self.tile_stride, self.use_tiling = (256, 256), (192, 192), False

    enable_tiling = AutoencoderKLWanDecoder.enable_tiling
    _blend = staticmethod(AutoencoderKLWanDecoder._blend)

    def normalize_latents(self, z):
        mean = torch.tensor(self.latents_mean).view(1, self.z_dim, 1, 1, 1).to(z)
        inv_std = 1.0 / torch.tensor(self.latents_std).view(1, self.z_dim, 1, 1, 1).to(z)
        return (z - mean) * inv_std

    def _tile(self, x, pol):
        return pol.r(self.quant_conv(self.encoder(x, pol)))

    @torch.no_grad()
    def encode(self, x, policy: Policy = FP32):
        pol = policy
        _, _, T, H, W = x.shape
        assert (T - 1) % 4 == 0, "the reference encodes 1 + 4k frames"
        if not (self.use_tiling and (W > self.tile_min[1] or H > self.tile_min[0])):
            return self._tile(x, pol)
        lat_stride = (self.tile_stride[0] // self.ratio, self.tile_stride[1] // self.ratio)
        blend = (self.tile_min[0] // self.ratio - lat_stride[0], self.tile_min[1] // self.ratio - lat_stride[1])
        rows = [[self._tile(x[:, :, :, i:i + self.tile_min[0], j:j + self.tile_min[1]], pol)
                 for j in range(0, W, self.tile_stride[1])] for i in range(0, H, self.tile_stride[0])]
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(rows[i - 1][j], tile, blend[0], 3, pol)
                if j > 0:
                    tile = self._blend(row[j - 1], tile, blend[1], 4, pol)
                out.append(tile[:, :, :, :lat_stride[0], :lat_stride[1]])
            out_rows.append(torch.cat(out, dim=-1))
        return torch.cat(out_rows, dim=3)[:, :, :, :H // self.ratio, :W // self.ratio]
