"""ORACLE — test infrastructure, not product code.

fp32 CPU restatement of TAEHV — the DECODER behind HunyuanVideo-1.5's `use_light_vae` switch and the encoder
half of the same class (`encode_video`, tae/model.py:214-236, :299-316; pinned by tests/golden/vae_taehv_encode.pt) —, following
  /root/reference/apps/api/src/vae/tae/model.py
      conv / Clamp / MemBlock / TGrow                   :20-66    (3x3 "same" Conv2d; 3 tanh(x/3); act(conv(cat[x, past]) +
                                                                   skip(x)); 1x1 conv whose channel blocks become frames)
      apply_model_with_memblocks                        :69-176   (parallel: past = the sequence shifted by one frame, zeros
                                                                   first; sequential: a graph walk with per-block memory —
                                                                   the same function of the input)
      TAEHV.__init__ (decoder stack), decode_video      :180-264, :318-333 (clamp to [-1, 1] for "hy15", pixel_shuffle, trim)
  /root/reference/apps/api/src/vae/hunyuanvideo15/model.py
      AutoencoderKLHunyuanVideo15Light.decode           :1225-1234 (latents / scaling_factor, NCTHW <-> NTCHW, unsqueeze(0))
      AutoencoderKLHunyuanVideo15.decode (light branch) :958-962   (parallel=False)
tests/golden/vae_taehv.pt holds outputs of the REFERENCE classes run in this container on seeded weights (sequential AND
parallel mode); tests/test_oracle_golden.py requires this restatement to match them.  Parameter names equal the reference's
`decoder.*` state-dict keys (the Sequential's indices).

Where the bf16 STORAGE policy rounds: after every convolution's epilogue (bias, residual and activation are applied in f32
before the one rounding — the HIP conv kernel's epilogue), after the input clamp, after each 1x1 TGrow.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import FP32, Policy


class MemBlock(nn.Module):
    """tae/model.py:29-45 with n_in == n_out (every decoder block): skip is the identity."""

    def __init__(self, n: int):
        super().__init__()
        self.conv = nn.ModuleList([nn.Conv2d(2 * n, n, 3, padding=1), nn.Identity(), nn.Conv2d(n, n, 3, padding=1),
                                   nn.Identity(), nn.Conv2d(n, n, 3, padding=1)])

    def forward(self, x, past, act, pol: Policy):
        h = pol.r(act(self.conv[0](torch.cat([x, past], 1))))
        h = pol.r(act(self.conv[2](h)))
        return pol.r(act(self.conv[4](h) + x))


class TGrow(nn.Module):
    def __init__(self, n: int, stride: int):
        super().__init__()
        self.stride = stride
        self.conv = nn.Conv2d(n, n * stride, 1, bias=False)

    def forward(self, x, pol: Policy):
        _, c, h, w = x.shape
        return pol.r(self.conv(x)).reshape(-1, c, h, w)


class TAEHVDecoder(nn.Module):
    """The `decoder` Sequential of TAEHV (tae/model.py:239-264) as an indexable list with the same indices."""

    def __init__(self, latent_channels: int = 32, patch_size: int = 2, image_channels: int = 3,
                 n_f: Sequence[int] = (256, 128, 64, 64), decoder_time_upscale=(True, True),
                 decoder_space_upscale=(True, True, True), model_type: str = "hy15"):
        super().__init__()
        self.patch_size, self.latent_channels, self.image_channels, self.model_type = patch_size, latent_channels, image_channels, model_type
        self.slope = 0.2 if model_type == "hy15" else 0.0
        self.frames_to_trim = 2 ** sum(decoder_time_upscale) - 1
        self.space = [2 if s else 1 for s in decoder_space_upscale]
        tg = [1, 2 if decoder_time_upscale[0] else 1, 2 if decoder_time_upscale[1] else 1]
        n = list(n_f)
        mods: List[nn.Module] = [nn.Identity(), nn.Conv2d(latent_channels, n[0], 3, padding=1), nn.Identity()]
        for s in range(3):
            mods += [MemBlock(n[s]), MemBlock(n[s]), MemBlock(n[s]), nn.Identity(), TGrow(n[s], tg[s]),
                     nn.Conv2d(n[s], n[s + 1], 3, padding=1, bias=False)]
        mods += [nn.Identity(), nn.Conv2d(n[3], image_channels * patch_size ** 2, 3, padding=1)]
        self.decoder = nn.ModuleList(mods)

    def act(self, x):
        return F.leaky_relu(x, self.slope)

    def _mem(self, x, n_batch):
        """past of every frame = the previous frame of ITS clip, zeros before the first (tae/model.py:92-96)."""
        nt, c, h, w = x.shape
        xx = x.reshape(n_batch, nt // n_batch, c, h, w)
        return F.pad(xx, (0, 0, 0, 0, 0, 0, 1, 0))[:, :nt // n_batch].reshape(x.shape)

    def decode_video(self, x: torch.Tensor, policy: Policy = FP32) -> torch.Tensor:
        """x [N, T, C, H, W] latents -> [N, 4T - 3, 3, H r 8, W r 8] (defaults), tae/model.py:318-333."""
        pol, d = policy, self.decoder
        N, T, C, H, W = x.shape
        x = pol.r(torch.tanh(x.reshape(N * T, C, H, W) / 3) * 3)
        x = pol.r(self.act(d[1](x)))
        i = 3
        for s in range(3):
            for b in range(3):
                x = d[i + b](x, self._mem(x, N), self.act, pol)
            # reference order: nn.Upsample (nearest) -> TGrow (1x1 conv, channel blocks -> frames) -> 3x3 conv (:246-248).  A 1x1
            # convolution and a nearest upsample commute EXACTLY (every output pixel is the same function of one input
            # pixel), so TGrow's storage point is taken at the low resolution, where the HIP path computes it
            x = d[i + 4](x, pol)
            if self.space[s] > 1:
                x = F.interpolate(x, scale_factor=self.space[s])     # nn.Upsample default: nearest
            x = d[i + 5](x)
            x = pol.r(self.act(x)) if s == 2 else pol.r(x)            # only the last of these convs is followed by act (:261)
            i += 6
        x = pol.r(d[22](x))
        lo = -1.0 if self.model_type == "hy15" else 0.0
        x = x.clamp(lo, 1.0)                                          # exact on stored values (the bounds are representable)
        if self.patch_size > 1:
            x = F.pixel_shuffle(x, self.patch_size)
        nt, c, h, w = x.shape
        return x.view(N, nt // N, c, h, w)[:, self.frames_to_trim:]


class TPool(nn.Module):
    """tae/model.py:48-56: `stride` consecutive frames stacked along the channels, then a 1x1 convolution."""

    def __init__(self, n: int, stride: int):
        super().__init__()
        self.stride = stride
        self.conv = nn.Conv2d(n * stride, n, 1, bias=False)

    def forward(self, x, pol: Policy):
        _, c, h, w = x.shape
        return pol.r(self.conv(x.reshape(-1, self.stride * c, h, w)))


class TAEHVEncoder(nn.Module):
    """The `encoder` Sequential of TAEHV (tae/model.py:214-236) with the same indices, and `encode_video` (:299-316): pixel
    un-shuffle by the patch size, the clip padded to a multiple of 4 frames by repeating the last one, NTCHW latents."""

    def __init__(self, latent_channels: int = 32, patch_size: int = 2, image_channels: int = 3, model_type: str = "hy15"):
        super().__init__()
        self.patch_size, self.slope = patch_size, (0.2 if model_type == "hy15" else 0.0)
        mods: List[nn.Module] = [nn.Conv2d(image_channels * patch_size ** 2, 64, 3, padding=1), nn.Identity()]
        for stride in (2, 2, 1):
            mods += [TPool(64, stride), nn.Conv2d(64, 64, 3, padding=1, stride=2, bias=False), MemBlock(64), MemBlock(64), MemBlock(64)]
        mods += [nn.Conv2d(64, latent_channels, 3, padding=1)]
        self.encoder = nn.ModuleList(mods)

    def act(self, x):
        return F.leaky_relu(x, self.slope)

    def encode_video(self, x: torch.Tensor, policy: Policy = FP32) -> torch.Tensor:
        """x [N, T, 3, H, W] in [0, 1] -> latents [N, T' / 4, C, H / (8 p), W / (8 p)]."""
        pol, e = policy, self.encoder
        if self.patch_size > 1:
            x = F.pixel_unshuffle(x, self.patch_size)
        if x.shape[1] % 4 != 0:
            x = torch.cat([x, x[:, -1:].repeat_interleave(4 - x.shape[1] % 4, dim=1)], 1)
        N, T, C, H, W = x.shape
        x = pol.r(self.act(e[0](x.reshape(N * T, C, H, W))))
        i = 2
        for _ in range(3):
            x = e[i](x, pol)                                   # TPool
            x = pol.r(e[i + 1](x))                             # stride-2 conv, no bias, no activation
            for b in range(3):
                xx = x.reshape(N, -1, *x.shape[1:])
                past = F.pad(xx, (0, 0, 0, 0, 0, 0, 1, 0))[:, :xx.shape[1]].reshape(x.shape)
                x = e[i + 2 + b](x, past, self.act, pol)
            i += 5
        x = pol.r(e[17](x))
        return x.view(N, -1, *x.shape[1:])


class AutoencoderKLHunyuanVideo15Light(nn.Module):
    """vae/hunyuanvideo15/model.py:1163-1234; `taehv.decoder.*` keys.  decode returns what the reference's caller indexes
    with [0] (base_engine.vae_decode): [N, 3, T', H', W']."""

    def __init__(self, scaling_factor: float = 1.03682, latent_channels: int = 32, patch_size: int = 2, **kw):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.taehv = TAEHVDecoder(latent_channels=latent_channels, patch_size=patch_size, model_type="hy15", **kw)

    def decode(self, latents: torch.Tensor, policy: Policy = FP32) -> torch.Tensor:
        x = latents.float() / self.scaling_factor      # under the bf16 policy the caller passes bf16-representable latents
        return self.taehv.decode_video(x.transpose(1, 2), policy).transpose(1, 2)
