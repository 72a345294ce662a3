"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of the 2-D VAE DECODER the Flux engine uses: the reference's AutoencoderKL wrapper
(apps/api/src/vae/auto/model.py:229-275 `decode/_decode`, :537-549 de/normalize) builds its `Decoder`
from the un-vendored diffusers package (`diffusers.models.autoencoders.vae.Decoder`, imported at
vae/auto/model.py:35-41), so the arithmetic below is restated from the published diffusers semantics
summarised in SURVEY.md App. A:
  conv_in(z -> C3, 3x3) -> UNetMidBlock2D [ResnetBlock2D, Attention(1 head, GroupNorm32, residual), ResnetBlock2D]
  -> UpDecoderBlock2D x len(block_out_channels) (layers_per_block+1 ResnetBlock2D each; nearest 2x Upsample2D +
  conv3x3 on all but the last) -> GroupNorm(32, eps 1e-6) -> SiLU -> conv_out(C0 -> 3).
  ResnetBlock2D: GN32 -> SiLU -> conv3x3 -> GN32 -> SiLU -> conv3x3 (+ 1x1 conv_shortcut when C changes).
Parity: the BLOCKS are pinned against the reference's own in-tree copies run in the build container
(tests/golden/leaf_pins.pt, tests/test_oracle_leaf_pins.py): ResnetBlock2D against vae/seedvr/modules/__model.py:73-142
and against the LDM-style ResnetBlock of vae/hunyuanimage3/model.py:202-240 on a one-frame clip, the single-head
GroupNorm attention block against :169-199, the nearest-2x + 3x3 upsampler against :297-308.  Only the decoder TOPOLOGY
(which blocks in which order, `layers_per_block + 1` resnets per up block, no upsampler on the last) is still "parity
unpinned": diffusers is absent and no in-tree class assembles these blocks the way diffusers' Decoder does; it is
restated from the published class and from the state-dict keys of the FLUX.1 VAE checkpoint the manifest names
(decoder.mid_block.attentions.0.to_q, decoder.up_blocks.0.resnets.2..., decoder.up_blocks.2.upsamplers.0.conv).
FLUX.1-dev VAE config: latent 16, block_out_channels (128, 256, 512, 512), layers_per_block 2,
scaling_factor 0.3611, shift_factor 0.1159, no post_quant_conv.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .layers import Policy, FP32


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, pol: Policy):
        s = x if self.conv_shortcut is None else pol.r(self.conv_shortcut(x))      # storage points in launch order
        h = pol.r(self.conv1(pol.r(F.silu(self.norm1(x)))))
        h = self.conv2(pol.r(F.silu(self.norm2(h))))
        return pol.r(h + s)


class AttnBlock(nn.Module):
    """diffusers Attention(heads=1, dim_head=C, norm_num_groups=32, residual_connection=True, bias=True)."""

    def __init__(self, c, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x, pol: Policy):
        b, c, h, w = x.shape
        y = pol.r(self.group_norm(x)).view(b, c, h * w).transpose(1, 2)
        q, k, v = pol.r(self.to_q(y)), pol.r(self.to_k(y)), pol.r(self.to_v(y))
        o = pol.r(L.sdpa_dispatch(q, k, v, policy=pol))
        o = self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)
        return pol.r(o + x)


class MidBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])
        self.attentions = nn.ModuleList([AttnBlock(c)])

    def forward(self, x, pol):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, pol), pol), pol)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x, pol):
        return pol.r(self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest")))


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, x, pol):
        for r in self.resnets:
            x = r(x, pol)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, pol)
        return x


class Decoder(nn.Module):
    def __init__(self, latent_channels=16, out_channels=3, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2):
        super().__init__()
        rev = list(block_out_channels)[::-1]
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0])
        ups, prev = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(UpDecoderBlock2D(prev, c, layers_per_block + 1, add_upsample=i != len(rev) - 1))
            prev = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z, pol: Policy):
        x = self.mid_block(pol.r(self.conv_in(z)), pol)
        for u in self.up_blocks:
            x = u(x, pol)
        return pol.r(self.conv_out(pol.r(F.silu(self.conv_norm_out(x)))))


class AutoencoderKLDecoder(nn.Module):
    def __init__(self, latent_channels=16, out_channels=3, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, scaling_factor=0.3611, shift_factor=0.1159):
        super().__init__()
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self.scaling_factor, self.shift_factor = scaling_factor, shift_factor

    def denormalize_latents(self, z):
        return z / self.scaling_factor + self.shift_factor      # vae/auto/model.py:544-549

    @torch.no_grad()
    def decode(self, z, policy: Policy = FP32):
        return self.decoder(z, policy)
