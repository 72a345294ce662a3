"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the HunyuanVideo-1.5 3-D causal VAE DECODE path, following
/root/reference/apps/api/src/vae/hunyuanvideo15/model.py:
  AutoencoderKLHunyuanVideo15._decode / tiled_decode   :929-940, :1060-1119   (8x8-latent tiles, stride 6, blend 32 px)
  HunyuanVideo15Decoder3D.forward                      :708-732   (conv_in + channel-repeat shortcut)
  HunyuanVideo15MidBlock / UpBlock3D / ResnetBlock     :383-428, :480-532, :338-380
  HunyuanVideo15Upsample.forward (DCAE pixel shuffle)  :217-274   (first frame: spatial only, half the channels)
  HunyuanVideo15AttnBlock.forward                      :130-214   (frame-causal mask: a token sees frames <= its own)
  HunyuanVideo15CausalConv3d.forward                   :52-90     (REPLICATE padding, 2 frames in front)
  HunyuanVideo15RMS_norm.forward                       :93-127
  blend_v / blend_h, denormalize_latents               :974-992, :1145-1150
and of its ENCODE path (image-to-video conditioning, engine/hunyuanvideo15/shared `_get_image_latents`):
  AutoencoderKLHunyuanVideo15._encode / tiled_encode   :890-897, :994-1058   (256-px tiles, stride 192, 2-latent blends)
  HunyuanVideo15Encoder3D.forward                      :535-636   (grouped-mean shortcut around norm_out/conv_out)
  HunyuanVideo15DownBlock3D / Downsample.forward       :440-478, :277-333  (DCAE pixel un-shuffle; first frame: spatial
                                                        only, its channels duplicated; shortcut = grouped channel mean)
tests/golden/vae_hunyuan15.pt and vae_hunyuan15_encode.pt hold outputs of the REFERENCE class run in this container (untiled
and tiled); the CPU tests require this restatement to match them.  Parameter names equal the reference's `decoder.*` /
`encoder.*` state-dict keys.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .layers import Policy, FP32


class CausalConv3d(nn.Module):
    def __init__(self, cin: int, cout: int, kernel_size: int = 3):
        super().__init__()
        k = kernel_size
        self._pad = (k // 2, k // 2, k // 2, k // 2, k - 1, 0)
        self.conv = nn.Conv3d(cin, cout, k)

    def forward(self, x):
        return self.conv(F.pad(x, self._pad, mode="replicate"))


class RMSNorm(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(dim, 1, 1, 1))

    def forward(self, x):
        return F.normalize(x, dim=1) * self.scale * self.gamma


def dcae_upsample_rearrange(t: torch.Tensor, r1: int = 1, r2: int = 2, r3: int = 2) -> torch.Tensor:
    """(b, r1*r2*r3*c, f, h, w) -> (b, c, r1*f, r2*h, r3*w), model.py:231-247."""
    b, pc, f, h, w = t.shape
    c = pc // (r1 * r2 * r3)
    return t.view(b, r1, r2, r3, c, f, h, w).permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(b, c, f * r1, h * r2, w * r3)


class ResnetBlock(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.norm1, self.conv1 = RMSNorm(cin), CausalConv3d(cin, cout)
        self.norm2, self.conv2 = RMSNorm(cout), CausalConv3d(cout, cout)
        self.conv_shortcut = nn.Conv3d(cin, cout, 1) if cin != cout else None

    def forward(self, x, pol: Policy):
        h = pol.r(self.conv1(pol.r(F.silu(self.norm1(x)))))
        h = self.conv2(pol.r(F.silu(self.norm2(h))))
        r = x if self.conv_shortcut is None else pol.r(self.conv_shortcut(x))
        return pol.r(h + r)


class AttnBlock(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.norm = RMSNorm(c)
        self.to_q, self.to_k, self.to_v, self.proj_out = (nn.Conv3d(c, c, 1) for _ in range(4))

    def forward(self, x, pol: Policy):
        b, c, f, h, w = x.shape
        n = pol.r(self.norm(x))
        q, k, v = (pol.r(m(n)).reshape(b, c, f * h * w).permute(0, 2, 1).unsqueeze(1) for m in (self.to_q, self.to_k, self.to_v))
        frame = torch.arange(f * h * w) // (h * w)
        keep = frame[None, :] <= frame[:, None]        # token i sees the keys of frames <= its own
        o = pol.r(L.sdpa_dispatch(q, k, v, policy=pol, mask=keep))
        o = o.squeeze(1).reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)
        return pol.r(self.proj_out(o) + x)


class Upsample(nn.Module):
    def __init__(self, cin: int, cout: int, temporal: bool):
        super().__init__()
        factor = 8 if temporal else 4
        self.conv = CausalConv3d(cin, cout * factor)
        self.temporal = temporal
        self.repeats = factor * cout // cin

    def forward(self, x, pol: Policy):
        h = pol.r(self.conv(x))
        if self.temporal:
            hf = dcae_upsample_rearrange(h[:, :, :1], 1, 2, 2)
            hf = hf[:, : hf.shape[1] // 2]
            hn = dcae_upsample_rearrange(h[:, :, 1:], 2, 2, 2)
            h = torch.cat([hf, hn], dim=2)
            xf = dcae_upsample_rearrange(x[:, :, :1], 1, 2, 2).repeat_interleave(self.repeats // 2, dim=1)
            xn = dcae_upsample_rearrange(x[:, :, 1:], 2, 2, 2).repeat_interleave(self.repeats, dim=1)
            sc = torch.cat([xf, xn], dim=2)
        else:
            h = dcae_upsample_rearrange(h, 1, 2, 2)
            sc = dcae_upsample_rearrange(x.repeat_interleave(self.repeats, dim=1), 1, 2, 2)
        return pol.r(h + sc)


def dcae_downsample_rearrange(t: torch.Tensor, r1: int = 1, r2: int = 2, r3: int = 2) -> torch.Tensor:
    """(b, c, r1*f, r2*h, r3*w) -> (b, r1*r2*r3*c, f, h, w), model.py:289-302."""
    b, c, pf, ph, pw = t.shape
    f, h, w = pf // r1, ph // r2, pw // r3
    return t.view(b, c, f, r1, h, r2, w, r3).permute(0, 3, 5, 7, 1, 2, 4, 6).reshape(b, r1 * r2 * r3 * c, f, h, w)


def group_mean(t: torch.Tensor, out_channels: int) -> torch.Tensor:
    """Channel groups averaged: (b, C, ...) -> (b, out_channels, ...), group g = channels g*gs .. (g+1)*gs - 1."""
    b, c = t.shape[:2]
    return t.view(b, out_channels, c // out_channels, *t.shape[2:]).mean(dim=2)


class Downsample(nn.Module):
    def __init__(self, cin: int, cout: int, temporal: bool):
        super().__init__()
        factor = 8 if temporal else 4
        self.conv = CausalConv3d(cin, cout // factor)
        self.temporal = temporal
        self.cout = cout

    def forward(self, x, pol: Policy):
        h = pol.r(self.conv(x))
        if self.temporal:
            hf = dcae_downsample_rearrange(h[:, :, :1], 1, 2, 2)
            hf = torch.cat([hf, hf], dim=1)
            h = torch.cat([hf, dcae_downsample_rearrange(h[:, :, 1:], 2, 2, 2)], dim=2)
            xf = pol.r(group_mean(dcae_downsample_rearrange(x[:, :, :1], 1, 2, 2), self.cout))   # one storage point each:
            if x.shape[2] > 1:                                                                      # the HIP path writes two
                xn = pol.r(group_mean(dcae_downsample_rearrange(x[:, :, 1:], 2, 2, 2), self.cout))
                sc = torch.cat([xf, xn], dim=2)
            else:
                sc = xf
        else:
            h = dcae_downsample_rearrange(h, 1, 2, 2)
            sc = pol.r(group_mean(dcae_downsample_rearrange(x, 1, 2, 2), self.cout))
        return pol.r(h + sc)


class DownBlock(nn.Module):
    def __init__(self, cin: int, cout: int, n: int, down_out, temporal: bool):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout) for i in range(n)])
        self.downsamplers = None if down_out is None else nn.ModuleList([Downsample(cout, down_out, temporal)])

    def forward(self, x, pol: Policy):
        for r in self.resnets:
            x = r(x, pol)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x, pol)
        return x


class MidBlock(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(c, c), ResnetBlock(c, c)])
        self.attentions = nn.ModuleList([AttnBlock(c)])

    def forward(self, x, pol: Policy):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, pol), pol), pol)


class UpBlock(nn.Module):
    def __init__(self, cin: int, cout: int, n: int, up_out, temporal: bool):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout) for i in range(n)])
        self.upsamplers = None if up_out is None else nn.ModuleList([Upsample(cout, up_out, temporal)])

    def forward(self, x, pol: Policy):
        for r in self.resnets:
            x = r(x, pol)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, pol)
        return x


class Decoder3D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, block_out_channels: Tuple[int, ...], layers_per_block: int,
                 spatial_compression_ratio: int, temporal_compression_ratio: int):
        super().__init__()
        import math
        ch = list(block_out_channels)
        self.repeat = ch[0] // in_channels
        self.conv_in = CausalConv3d(in_channels, ch[0])
        self.mid_block = MidBlock(ch[0])
        self.up_blocks = nn.ModuleList()
        cin = ch[0]
        for i, cout in enumerate(ch):
            sp, tp = i < math.log2(spatial_compression_ratio), i < math.log2(temporal_compression_ratio)
            if sp or tp:
                self.up_blocks.append(UpBlock(cin, cout, layers_per_block + 1, ch[i + 1], tp))
                cin = ch[i + 1]
            else:
                self.up_blocks.append(UpBlock(cin, cout, layers_per_block + 1, None, False))
                cin = cout
        self.norm_out = RMSNorm(ch[-1])
        self.conv_out = CausalConv3d(ch[-1], out_channels)

    def forward(self, z, pol: Policy = FP32):
        x = pol.r(self.conv_in(z) + z.repeat_interleave(self.repeat, dim=1))
        x = self.mid_block(x, pol)
        for ub in self.up_blocks:
            x = ub(x, pol)
        return pol.r(self.conv_out(pol.r(F.silu(self.norm_out(x)))))


class Encoder3D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, block_out_channels: Tuple[int, ...], layers_per_block: int,
                 spatial_compression_ratio: int, temporal_compression_ratio: int):
        super().__init__()
        import math
        ch = list(block_out_channels)
        self.out_channels = out_channels
        self.conv_in = CausalConv3d(in_channels, ch[0])
        self.down_blocks = nn.ModuleList()
        cin = ch[0]
        for i, cout in enumerate(ch):
            if i < math.log2(spatial_compression_ratio):
                tp = i >= math.log2(spatial_compression_ratio // temporal_compression_ratio)
                self.down_blocks.append(DownBlock(cin, cout, layers_per_block, ch[i + 1], tp))
                cin = ch[i + 1]
            else:
                self.down_blocks.append(DownBlock(cin, cout, layers_per_block, None, False))
                cin = cout
        self.mid_block = MidBlock(ch[-1])
        self.norm_out = RMSNorm(ch[-1])
        self.conv_out = CausalConv3d(ch[-1], out_channels)

    def forward(self, x, pol: Policy = FP32):
        x = pol.r(self.conv_in(x))
        for db in self.down_blocks:
            x = db(x, pol)
        x = self.mid_block(x, pol)
        sc = pol.r(group_mean(x, self.out_channels))
        return pol.r(self.conv_out(pol.r(F.silu(self.norm_out(x)))) + sc)


class AutoencoderKLHunyuanVideo15(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, latent_channels: int = 32,
                 block_out_channels=(128, 256, 512, 1024, 1024), layers_per_block: int = 2,
                 spatial_compression_ratio: int = 16, temporal_compression_ratio: int = 4, scaling_factor: float = 1.03682,
                 **_unused):
        super().__init__()
        self.encoder = Encoder3D(in_channels, latent_channels * 2, tuple(block_out_channels), layers_per_block,
                                 spatial_compression_ratio, temporal_compression_ratio)
        self.decoder = Decoder3D(latent_channels, out_channels, tuple(reversed(block_out_channels)), layers_per_block,
                                 spatial_compression_ratio, temporal_compression_ratio)
        self.scaling_factor = scaling_factor
        self.spatial_compression_ratio = spatial_compression_ratio
        self.tile_sample_min = 128
        self.tile_latent_min = 128 // spatial_compression_ratio
        self.tile_overlap_factor = 0.25
        self.use_tiling = False

    def enable_tiling(self):
        self.use_tiling = True

    def denormalize_latents(self, latents):
        return latents / self.scaling_factor

    def normalize_latents(self, latents):
        return latents * self.scaling_factor

    @torch.no_grad()
    def encode(self, x, policy: Policy = FP32, tile_sample_min: int = 128):
        """Posterior parameters [B, 2 * latent_channels, T', H/16, W/16] (mean | logvar); `.mode()` = the first half."""
        _, _, _, H, W = x.shape
        ts = tile_sample_min
        if not (self.use_tiling and (W > ts or H > ts)):
            return self.encoder(x, policy)
        tl = ts // self.spatial_compression_ratio
        ov = int(ts * (1 - self.tile_overlap_factor))
        blend = int(tl * self.tile_overlap_factor)
        limit = tl - blend
        rows = [[self.encoder(x[:, :, :, i:i + ts, j:j + ts], policy) for j in range(0, W, ov)] for i in range(0, H, ov)]
        out_rows = []
        for i, row in enumerate(rows):
            res = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = policy.r(self._blend(rows[i - 1][j], tile, blend, 3))
                if j > 0:
                    tile = policy.r(self._blend(row[j - 1], tile, blend, 4))
                res.append(tile[:, :, :, :limit, :limit])
            out_rows.append(torch.cat(res, dim=-1))
        return torch.cat(out_rows, dim=-2)

    @staticmethod
    def _blend(a, b, extent: int, dim: int):
        extent = min(a.shape[dim], b.shape[dim], extent)
        for i in range(extent):
            ia = [slice(None)] * 5
            ib = [slice(None)] * 5
            ia[dim], ib[dim] = a.shape[dim] - extent + i, i
            b[tuple(ib)] = a[tuple(ia)] * (1 - i / extent) + b[tuple(ib)] * (i / extent)
        return b

    @torch.no_grad()
    def decode(self, z, policy: Policy = FP32):
        _, _, _, H, W = z.shape
        tl = self.tile_latent_min
        if not (self.use_tiling and (W > tl or H > tl)):
            return self.decoder(z, policy)
        ov = int(tl * (1 - self.tile_overlap_factor))
        blend = int(self.tile_sample_min * self.tile_overlap_factor)
        limit = self.tile_sample_min - blend
        rows = [[self.decoder(z[:, :, :, i:i + tl, j:j + tl], policy) for j in range(0, W, ov)] for i in range(0, H, ov)]
        out_rows = []
        for i, row in enumerate(rows):
            res = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(rows[i - 1][j], tile, blend, 3)
                if j > 0:
                    tile = self._blend(row[j - 1], tile, blend, 4)
                res.append(tile[:, :, :, :limit, :limit])
            out_rows.append(torch.cat(res, dim=-1))
        return torch.cat(out_rows, dim=-2)
