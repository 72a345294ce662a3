"""AutoencoderKLHunyuanVideo15 decode on the MI355X HIP ops — drop-in for VAE registry key "hunyuanvideo15"
(decode half).

Mirrors what the engines use of the reference class (apps/api/src/vae/hunyuanvideo15/model.py:735-1158):
`from_config`, state-dict keys `decoder.*` (encoder keys in a checkpoint are ignored by `strict=False`), `.config`
(scaling_factor, spatial / temporal compression), `enable_tiling(...)`, `denormalize_latents`,
`decode(z, return_dict=False)[0]`, `.dtype`.

Decode runs channels-last [T, H, W, C].  Per tile (the reference decodes 8x8-latent tiles with stride 6 over the WHOLE
frame sequence and blends 32-pixel overlaps, `tiled_decode` :1060-1119 — tiling is part of the numerical contract):
implicit-GEMM causal conv3d on MFMA with REPLICATE padding (clamped tap coordinates) and fused bias + residual,
RMS-norm(channel)+SiLU, GEMMs for the 1x1x1 convs, the frame-causal single-head attention of the mid block
(materialised through the GEMM kernel with the mask applied in the row softmax), DCAE pixel-shuffle upsampling as
pure data movement plus one add, crossfades for the blends.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .flux import _Config


class _CConv(nn.Module):
    """HunyuanVideo15CausalConv3d: parameters live under `.conv` (model.py:82-84)."""

    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Module()
        self.conv.weight = nn.Parameter(torch.empty(cout, cin, 3, 3, 3, **kw), requires_grad=False)
        self.conv.bias = nn.Parameter(torch.empty(cout, **kw), requires_grad=False)


class _Conv1(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1, 1, **kw), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout, **kw), requires_grad=False)


class _Gamma(nn.Module):
    def __init__(self, dim, **kw):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim, 1, 1, 1, **kw), requires_grad=False)


class _Res(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.norm1, self.conv1 = _Gamma(cin, **kw), _CConv(cin, cout, **kw)
        self.norm2, self.conv2 = _Gamma(cout, **kw), _CConv(cout, cout, **kw)
        self.conv_shortcut = _Conv1(cin, cout, **kw) if cin != cout else None


class _Attn(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.norm = _Gamma(c, **kw)
        self.to_q, self.to_k, self.to_v, self.proj_out = (_Conv1(c, c, **kw) for _ in range(4))


class _Upsample(nn.Module):
    def __init__(self, cin, cout, temporal, **kw):
        super().__init__()
        self.factor = 8 if temporal else 4
        self.conv = _CConv(cin, cout * self.factor, **kw)
        self.temporal = temporal
        self.repeats = self.factor * cout // cin


class _Mid(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(c, c, **kw), _Res(c, c, **kw)])
        self.attentions = nn.ModuleList([_Attn(c, **kw)])


class _Up(nn.Module):
    def __init__(self, cin, cout, n, up_out, temporal, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout, **kw) for i in range(n)])
        self.upsamplers = None if up_out is None else nn.ModuleList([_Upsample(cout, up_out, temporal, **kw)])


class _Downsample(nn.Module):
    def __init__(self, cin, cout, temporal, **kw):
        super().__init__()
        self.factor = 8 if temporal else 4
        self.conv = _CConv(cin, cout // self.factor, **kw)
        self.temporal = temporal
        self.cout = cout


class _Down(nn.Module):
    def __init__(self, cin, cout, n, down_out, temporal, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout, **kw) for i in range(n)])
        self.downsamplers = None if down_out is None else nn.ModuleList([_Downsample(cout, down_out, temporal, **kw)])


class _Encoder(nn.Module):
    """HunyuanVideo15Encoder3D (model.py:535-636): same parameter names (`encoder.*`)."""

    def __init__(self, in_channels, out_channels, ch, layers_per_block, spatial_ratio, temporal_ratio, **kw):
        super().__init__()
        self.out_channels = out_channels
        self.conv_in = _CConv(in_channels, ch[0], **kw)
        downs, cin = [], ch[0]
        for i, cout in enumerate(ch):
            if i < math.log2(spatial_ratio):
                tp = i >= math.log2(spatial_ratio // temporal_ratio)
                downs.append(_Down(cin, cout, layers_per_block, ch[i + 1], tp, **kw))
                cin = ch[i + 1]
            else:
                downs.append(_Down(cin, cout, layers_per_block, None, False, **kw))
                cin = cout
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _Mid(ch[-1], **kw)
        self.norm_out = _Gamma(ch[-1], **kw)
        self.conv_out = _CConv(ch[-1], out_channels, **kw)


class _Posterior:
    """DiagonalGaussianDistribution as the engines use it (`.mode()`, `.sample(generator)`; engine/base_engine.py:2144-2149):
    parameters [B, 2 C, T, H, W] = mean | logvar, logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        # diffusers' DiagonalGaussianDistribution draws with randn_tensor(dtype=parameters.dtype): same stream, same values
        dev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise


class _Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, ch, layers_per_block, spatial_ratio, temporal_ratio, **kw):
        super().__init__()
        self.repeat = ch[0] // in_channels
        self.conv_in = _CConv(in_channels, ch[0], **kw)
        self.mid_block = _Mid(ch[0], **kw)
        ups, cin = [], ch[0]
        for i, cout in enumerate(ch):
            sp, tp = i < math.log2(spatial_ratio), i < math.log2(temporal_ratio)
            if sp or tp:
                ups.append(_Up(cin, cout, layers_per_block + 1, ch[i + 1], tp, **kw))
                cin = ch[i + 1]
            else:
                ups.append(_Up(cin, cout, layers_per_block + 1, None, False, **kw))
                cin = cout
        self.up_blocks = nn.ModuleList(ups)
        self.norm_out = _Gamma(ch[-1], **kw)
        self.conv_out = _CConv(ch[-1], out_channels, **kw)


def _pack_cl(x: torch.Tensor, r1: int, r2: int, r3: int) -> torch.Tensor:
    """channels-last form of `_dcae_downsample_rearrange` (model.py:289-302): [r1 T, r2 H, r3 W, c] -> [T, H, W, r1*r2*r3*c]."""
    PT, PH, PW, c = x.shape
    T, H, W = PT // r1, PH // r2, PW // r3
    return x.view(T, r1, H, r2, W, r3, c).permute(0, 2, 4, 1, 3, 5, 6).reshape(T, H, W, r1 * r2 * r3 * c).contiguous()


def _rearrange_cl(x: torch.Tensor, r1: int, r2: int, r3: int) -> torch.Tensor:
    """channels-last form of `_dcae_upsample_rearrange` (model.py:231-247): [T, H, W, r1*r2*r3*c] -> [r1 T, r2 H, r3 W, c]."""
    T, H, W, pc = x.shape
    c = pc // (r1 * r2 * r3)
    return x.view(T, H, W, r1, r2, r3, c).permute(0, 3, 1, 4, 2, 5, 6).reshape(T * r1, H * r2, W * r3, c)


class AutoencoderKLHunyuanVideo15(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, latent_channels: int = 32,
                 block_out_channels: Tuple[int, ...] = (128, 256, 512, 1024, 1024), layers_per_block: int = 2,
                 spatial_compression_ratio: int = 16, temporal_compression_ratio: int = 4,
                 downsample_match_channel: bool = True, upsample_match_channel: bool = True,
                 scaling_factor: float = 1.03682, light_vae_path: Optional[str] = None, device=None,
                 dtype=torch.bfloat16):
        super().__init__()
        if not upsample_match_channel:
            raise NotImplementedError("hunyuanvideo15_mi355 VAE: upsample_match_channel=False is not a shipped configuration")
        self.config = _Config(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              spatial_compression_ratio=spatial_compression_ratio,
                              temporal_compression_ratio=temporal_compression_ratio, scaling_factor=scaling_factor,
                              shift_factor=None)
        kw = dict(device=device, dtype=dtype)
        self.encoder = _Encoder(in_channels, 2 * latent_channels, list(block_out_channels), layers_per_block,
                                spatial_compression_ratio, temporal_compression_ratio, **kw)
        self.decoder = _Decoder(latent_channels, out_channels, list(reversed(block_out_channels)), layers_per_block,
                                spatial_compression_ratio, temporal_compression_ratio, **kw)
        self.spatial_compression_ratio = spatial_compression_ratio
        self.temporal_compression_ratio = temporal_compression_ratio
        self.use_tiling = False
        # the reference class keeps ONE pair of tile sizes for both directions (model.py:866-888): the decode reads the
        # latent pair, the encode the sample pair
        self.tile_sample_min_height = self.tile_sample_min_width = 128
        self.tile_latent_min_height = self.tile_latent_min_width = 128 // spatial_compression_ratio
        self.tile_overlap_factor = 0.25
        self._packed: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.storage_dtype = torch.bfloat16
        # up blocks (after conv_in + mid) whose launches cover all equally shaped tiles at once; measured on the 480p x 121-frame
        # decode (tools/hunyuan_vae_streams_ab.py): 0 / 1 / 2 = 2382 / 2366 / 2351 ms at 7 / 10 / 26 GiB peak — 1 by default; stages
        # whose single-tile launch already takes the conv-shaped tiles must stay outside (another summation order)
        self.batch_head_blocks = 1
        self.decode_streams = 2      # spatial tiles decoded side by side on their own HIP streams (see _decode_tiles)
        self._streams: list = []
        # the TAEHV "light" decoder (model.py:794-846): built lazily when `enable_tiling(use_light_vae=True)` asks for it, or
        # at once when a path is configured and this module is not being built on the meta device
        self.use_light_vae = False
        self.light_vae_path = light_vae_path
        self.light_vae = None
        if light_vae_path is not None and not self.decoder.conv_in.conv.weight.is_meta:
            self._ensure_light_vae_loaded()

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config) if isinstance(config, dict) else dict(vars(config))
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_") and k != "shift_factor"}
        cfg.update(kwargs)
        return cls(**cfg)

    _from_config = from_config

    @property
    def dtype(self):
        return self.decoder.conv_in.conv.weight.dtype

    @property
    def device(self):
        return self.decoder.conv_in.conv.weight.device

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        return super().load_state_dict(*a, **k)

    def _weights_changed(self):
        """Parameters were written in place (`weights.load_checkpoint_into`): drop the packed conv-weight cache."""
        self._packed = {}

    def _ensure_light_vae_loaded(self) -> None:
        """model.py:821-846."""
        if self.light_vae is not None:
            return
        if not self.light_vae_path:
            raise ValueError("Light VAE requested but `light_vae_path` is not set in the VAE config.")
        from .vae_taehv import AutoencoderKLHunyuanVideo15Light
        # on the main VAE's device (model.py:838-846); a VAE still on the meta device gets a CPU light VAE that follows the
        # later `.to(device)` as a submodule
        dev = None if self.device.type == "meta" else self.device
        self.light_vae = AutoencoderKLHunyuanVideo15Light(taehv_checkpoint_path=self.light_vae_path,
                                                          scaling_factor=self.config.scaling_factor, device=dev)

    def set_light_vae(self, light_vae) -> None:
        """Attach an already built `AutoencoderKLHunyuanVideo15Light` (weights streamed by the host's own loader)."""
        self.light_vae = light_vae

    def set_storage_dtype(self, dtype: torch.dtype):
        """torch.bfloat16 (production) or torch.float32: the f32-STORAGE VERIFICATION MODE of the decoder AND the encoder
        (DESIGN.md §1.2): every activation float, the library's `_f32` entry points (the convolutions on the exact three-way bf16
        split), and the frame-causal mid-block attention as one f32 attention call per frame over the keys of the frames up to
        it — the same softmax, without the bf16 probabilities of the materialised production path.  Weights stay bf16; the
        encoder's moments come back float."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError(f"activation storage must be bfloat16 or float32, got {dtype}")
        self.storage_dtype = dtype
        return self

    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None, tile_latent_min_height=None,
                      tile_latent_min_width=None, tile_overlap_factor=None, use_light_vae: Optional[bool] = None):
        # model.py:848-888.  A call without `use_light_vae` (the engine's vae_decode issues one, base_engine.py:2051) leaves
        # the switch as it is — what the reference's `if self.light_vae is None:` guard amounts to for such calls.  An
        # explicit True / False is honoured at any time (the reference ignores it once its light VAE has been built, so a
        # run could never go back to the full decoder; that defect is not mirrored).
        if use_light_vae is not None:
            if use_light_vae:
                self._ensure_light_vae_loaded()
            self.use_light_vae = bool(use_light_vae)
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_latent_min_height = tile_latent_min_height or self.tile_latent_min_height
        self.tile_latent_min_width = tile_latent_min_width or self.tile_latent_min_width
        self.tile_overlap_factor = tile_overlap_factor or self.tile_overlap_factor

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        return None

    def denormalize_latents(self, latents):
        return latents / self.config.scaling_factor

    def normalize_latents(self, latents):
        return latents * self.config.scaling_factor

    # ---- kernels per layer ----------------------------------------------------------------------
    def _w(self, mod, weight, bias):
        key = id(mod)
        p = self._packed.get(key)
        if p is None:
            wt = weight.data
            if wt.shape[1] % 8:      # the RGB input convolution: channels zero-padded to 8, as the activations are
                pad = torch.zeros(wt.shape[0], 8 - wt.shape[1] % 8, *wt.shape[2:], dtype=wt.dtype, device=wt.device)
                wt = torch.cat([wt, pad], dim=1)
            w = ops.pack_conv_weight(wt)
            b = torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
            b[:bias.numel()] = bias.data
            p = (w, b)
            self._packed[key] = p
        return p

    def _cconv(self, c: _CConv, x, residual=None, clip=0):
        """`clip`: frames per clip when x stacks several tiles' clips along T (0 = one clip)."""
        w, b = self._w(c, c.conv.weight, c.conv.bias)
        return ops.conv3d_cl(x, w, b, (3, 3, 3), residual=residual, replicate=True, clip_frames=clip)

    def _conv1(self, c: _Conv1, x2d, **kw):
        return ops.gemm(x2d, c.weight.data.reshape(c.weight.shape[0], c.weight.shape[1]), c.bias.data, **kw)

    @staticmethod
    def _g(n: _Gamma):
        return n.gamma.data.reshape(-1).contiguous()

    def _res(self, blk: _Res, x, clip=0):
        T, H, W, Cc = x.shape
        h = self._cconv(blk.conv1, ops.rmsnorm_cl(x, self._g(blk.norm1), silu=True), clip=clip)
        sc = x if blk.conv_shortcut is None else self._conv1(blk.conv_shortcut, x.view(T * H * W, Cc)).view(T, H, W, -1)
        return self._cconv(blk.conv2, ops.rmsnorm_cl(h, self._g(blk.norm2), silu=True), residual=sc, clip=clip)

    def _attn(self, blk: _Attn, x, clip=0):
        if clip and clip != x.shape[0]:      # stacked clips: the frame-causal attention is per clip
            return torch.cat([self._attn(blk, x[i:i + clip]) for i in range(0, x.shape[0], clip)], dim=0)
        T, H, W, Cc = x.shape
        S = T * H * W
        n = ops.rmsnorm_cl(x, self._g(blk.norm)).view(S, Cc)
        q, k, v = (self._conv1(m, n).view(1, 1, S, Cc) for m in (blk.to_q, blk.to_k, blk.to_v))
        if x.dtype == torch.float32:     # verification mode: frame f's queries over the keys of frames <= f, f32 softmax
            hw = H * W
            o = torch.cat([ops.attention(q[:, :, f * hw:(f + 1) * hw], k[:, :, :(f + 1) * hw], v[:, :, :(f + 1) * hw])
                           for f in range(T)], dim=2).permute(0, 2, 1, 3).reshape(S, Cc)
        else:
            o = ops.attention_framecausal(q, k, v, H * W).permute(0, 2, 1, 3).reshape(S, Cc)
        ones = torch.ones(Cc, dtype=torch.float32, device=x.device)
        return self._conv1(blk.proj_out, o, epilogue="gate_res", gate=ones, residual=x.view(S, Cc)).view(T, H, W, Cc)

    def _upsample(self, up: _Upsample, x, clip=0):
        h = self._cconv(up.conv, x, clip=clip)
        if up.temporal and clip and clip != x.shape[0]:
            # stacked clips: the first-frame rule applies to every clip's first frame.  Both rearrangements treat frames
            # independently, so all first frames go through the (1, 2, 2) form at once and all others through (2, 2, 2)
            B = x.shape[0] // clip

            def split(t):
                t5 = t.view(B, clip, *t.shape[1:])
                return t5[:, :1].reshape(B, *t.shape[1:]), t5[:, 1:].reshape(B * (clip - 1), *t.shape[1:])

            def join(first, rest):           # [B, ...], [B * 2 (clip - 1), ...] -> [B * (2 clip - 1), ...]
                rest = rest.view(B, 2 * (clip - 1), *rest.shape[1:])
                return torch.cat([first.unsqueeze(1), rest], dim=1).reshape(B * (2 * clip - 1), *first.shape[1:])
            h0, h1 = split(h)
            hf = _rearrange_cl(h0, 1, 2, 2)
            h = join(hf[..., : hf.shape[-1] // 2], _rearrange_cl(h1, 2, 2, 2))
            x0, x1 = split(x)
            sc = join(_rearrange_cl(x0, 1, 2, 2).repeat_interleave(up.repeats // 2, dim=-1),
                      _rearrange_cl(x1, 2, 2, 2).repeat_interleave(up.repeats, dim=-1))
        elif up.temporal:
            hf = _rearrange_cl(h[:1], 1, 2, 2)
            hf = hf[..., : hf.shape[-1] // 2]
            h = torch.cat([hf, _rearrange_cl(h[1:], 2, 2, 2)], dim=0)
            xf = _rearrange_cl(x[:1], 1, 2, 2).repeat_interleave(up.repeats // 2, dim=-1)
            xn = _rearrange_cl(x[1:], 2, 2, 2).repeat_interleave(up.repeats, dim=-1)
            sc = torch.cat([xf, xn], dim=0)
        else:
            h = _rearrange_cl(h, 1, 2, 2)
            sc = _rearrange_cl(x.repeat_interleave(up.repeats, dim=-1), 1, 2, 2)
        return ops.add(h.contiguous(), sc.contiguous())

    def _downsample(self, dn: _Downsample, x):
        """HunyuanVideo15Downsample.forward (model.py:304-333) on a channels-last tile: conv -> pixel un-shuffle into the
        channels; the shortcut is the grouped channel mean of the un-shuffled input.  With temporal downsampling the
        first frame is un-shuffled in space only and its channels duplicated."""
        h = self._cconv(dn.conv, x)[..., :dn.conv.conv.weight.shape[0]]
        if dn.temporal:
            hf = _pack_cl(h[:1], 1, 2, 2)
            hf = torch.cat([hf, hf], dim=-1)
            h = torch.cat([hf, _pack_cl(h[1:], 2, 2, 2)], dim=0) if h.shape[0] > 1 else hf
            sc = ops.group_mean(_pack_cl(x[:1], 1, 2, 2), dn.cout)
            if x.shape[0] > 1:
                sc = torch.cat([sc, ops.group_mean(_pack_cl(x[1:], 2, 2, 2), dn.cout)], dim=0)
        else:
            h = _pack_cl(h, 1, 2, 2)
            sc = ops.group_mean(_pack_cl(x, 1, 2, 2), dn.cout)
        return ops.add(h.contiguous(), sc.contiguous())

    def _encode_tile(self, x):
        """x [T, H, W, 8] channels-last pixels (3 channels + zero pad) -> moments [T', H/16, W/16, 2 * latent_channels]."""
        e = self.encoder
        x = self._cconv(e.conv_in, x)
        for db in e.down_blocks:
            for r in db.resnets:
                x = self._res(r, x)
            if db.downsamplers is not None:
                x = self._downsample(db.downsamplers[0], x)
        x = self._res(e.mid_block.resnets[0], x)
        x = self._attn(e.mid_block.attentions[0], x)
        x = self._res(e.mid_block.resnets[1], x)
        sc = ops.group_mean(x, e.out_channels)
        return self._cconv(e.conv_out, ops.rmsnorm_cl(x, self._g(e.norm_out), silu=True), residual=sc)

    @torch.no_grad()
    def _encode_one(self, x):
        """x [3, T, H, W] in [-1, 1] -> moments [2 C, T', H/16, W/16] bf16 (`_encode` / `tiled_encode`, model.py:890-897,
        :994-1058: 128-px tiles at stride 96, blended over 25 % of a latent tile)."""
        if x.device.type != "cuda" or self.dtype != torch.bfloat16:
            raise _l.ApexMIError("hunyuanvideo15_mi355 VAE needs bf16 weights and pixels on a ROCm device (no CPU fallback)")
        Cc, T, H, W = x.shape
        if (T - 1) % self.temporal_compression_ratio:
            raise ValueError(f"hunyuanvideo15 VAE encodes 1 + {self.temporal_compression_ratio} k frames, got {T}")
        xc = torch.zeros(T, H, W, 8, dtype=self.storage_dtype, device=x.device)
        xc[..., :Cc] = x.to(self.storage_dtype).permute(1, 2, 3, 0)
        tsh, tsw = self.tile_sample_min_height, self.tile_sample_min_width
        if not (self.use_tiling and (W > tsw or H > tsh)):
            out = self._encode_tile(xc)
        else:
            ovh, ovw = int(tsh * (1 - self.tile_overlap_factor)), int(tsw * (1 - self.tile_overlap_factor))
            bh = int(self.tile_latent_min_height * self.tile_overlap_factor)
            bw = int(self.tile_latent_min_width * self.tile_overlap_factor)
            lh, lw = self.tile_latent_min_height - bh, self.tile_latent_min_width - bw
            rows = [[self._encode_tile(xc[:, i:i + tsh, j:j + tsw].contiguous()) for j in range(0, W, ovw)]
                    for i in range(0, H, ovh)]
            out_rows = []
            for i, row in enumerate(rows):
                parts = []
                for j, tile in enumerate(row):
                    if i > 0 and bh > 0:
                        a = rows[i - 1][j]
                        e = min(a.shape[1], tile.shape[1], bh)
                        ops.crossfade_(a[:, a.shape[1] - e:, :tile.shape[2]], tile[:, :e], dim=1)
                    if j > 0 and bw > 0:
                        a = row[j - 1]
                        e = min(a.shape[2], tile.shape[2], bw)
                        ops.crossfade_(a[:, :tile.shape[1], a.shape[2] - e:], tile[:, :, :e], dim=2)
                    parts.append(tile[:, :lh, :lw])
                out_rows.append(torch.cat(parts, dim=2))
            out = torch.cat(out_rows, dim=1)
        return out[..., :self.encoder.out_channels].permute(3, 0, 1, 2).contiguous()

    @ops.on_model_device
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """`vae.encode(x, return_dict=False)[0].mode()` (engine/base_engine.py:2061-2165): x [B, 3, T, H, W]."""
        if x.dim() == 4:
            x = x.unsqueeze(2)
        post = _Posterior(torch.stack([self._encode_one(x[b]) for b in range(x.shape[0])], dim=0))
        if not return_dict:
            return (post,)
        return SimpleNamespace(latent_dist=post)

    def _decode_head(self, z, clip, nblocks):
        """conv_in, the mid block and the first `nblocks` up blocks over z [B * clip, h, w, latent_channels] = B tiles' clips
        stacked along T (B = 1: an ordinary tile).  Returns (x, frames per clip now)."""
        d = self.decoder
        x = self._cconv(d.conv_in, z, residual=z.repeat_interleave(d.repeat, dim=-1).contiguous(), clip=clip)
        x = self._res(d.mid_block.resnets[0], x, clip)
        x = self._attn(d.mid_block.attentions[0], x, clip)
        x = self._res(d.mid_block.resnets[1], x, clip)
        for ub in list(d.up_blocks)[:nblocks]:
            for r in ub.resnets:
                x = self._res(r, x, clip)
            if ub.upsamplers is not None:
                x = self._upsample(ub.upsamplers[0], x, clip)
                if ub.upsamplers[0].temporal:
                    clip = 2 * clip - 1
        return x, clip

    def _decode_tail(self, x, nblocks):
        d = self.decoder
        for ub in list(d.up_blocks)[nblocks:]:
            for r in ub.resnets:
                x = self._res(r, x)
            if ub.upsamplers is not None:
                x = self._upsample(ub.upsamplers[0], x)
        return self._cconv(d.conv_out, ops.rmsnorm_cl(x, self._g(d.norm_out), silu=True))

    def _decode_tile(self, z):
        """z [T, h, w, latent_channels] channels-last -> [4 (T - 1) + 1, 16 h, 16 w, 4] (3 channels + 1 pad)."""
        x, _ = self._decode_head(z, z.shape[0], len(self.decoder.up_blocks))
        return self._decode_tail(x, len(self.decoder.up_blocks))

    def _decode_tiles(self, ztiles):
        """Every spatial tile through the decoder.  The tiles are independent until the cross-fades, and the decoder's
        low-resolution stages launch far fewer workgroups than the chip has slots (a 31-frame 8 x 8-latent tile is 1984
        positions = 128 workgroups of the 128 x 128 convolution tile in the 1024-channel stages): `decode_streams` tiles run side
        by side on their own HIP streams, so those launches overlap, while the full-size launches of the late stages simply queue.
        Same kernels on the same data: bit-identical to the sequential walk (`decode_streams = 1`)."""
        flat = [(i, j, zt) for i, row in enumerate(ztiles) for j, zt in enumerate(row)]
        nb = max(0, min(int(self.batch_head_blocks), len(self.decoder.up_blocks)))
        if self.storage_dtype != torch.bfloat16:
            nb = 0                       # the verification mode walks tile by tile
        if nb > 0 and len(flat) > 1:
            # The lowest-resolution stages (conv_in, mid block, the first `batch_head_blocks` up blocks) of ALL equally shaped
            # tiles run as ONE launch per layer over the tiles' clips stacked along T (`apexmi_conv3d_cl_clips`: a 31-frame
            # 8 x 8-latent tile is 1984 positions = 128 workgroups in the 1024-channel stages; 32 of them fill the chip);
            # bit-identical to the tile-by-tile walk.  The rest of each tile follows, side by side on the streams below.
            groups: Dict[tuple, list] = {}
            for n, (_, _, zt) in enumerate(flat):
                groups.setdefault(tuple(zt.shape), []).append(n)
            heads = [None] * len(flat)
            for idxs in groups.values():
                T0 = flat[idxs[0]][2].shape[0]
                x, clip = self._decode_head(torch.cat([flat[n][2] for n in idxs], dim=0), T0, nb)
                for b, n in enumerate(idxs):
                    heads[n] = x[b * clip:(b + 1) * clip]
            flat = [(i, j, h) for (i, j, _), h in zip(flat, heads)]
            tile_fn = lambda h: self._decode_tail(h, nb)   # noqa: E731
        else:
            tile_fn = self._decode_tile
        ns = max(1, min(int(self.decode_streams), len(flat)))
        if ns == 1:
            out = [[None] * len(row) for row in ztiles]
            for i, j, zt in flat:
                out[i][j] = tile_fn(zt)
            return out
        main = torch.cuda.current_stream()
        if len(self._streams) < ns:
            self._streams += [torch.cuda.Stream(device=self.device) for _ in range(ns - len(self._streams))]
        out = [[None] * len(row) for row in ztiles]
        i0, j0, z0 = flat[0]
        out[i0][j0] = tile_fn(z0)                # on the main stream: fills the packed-weight caches every other tile reads
        flat = flat[1:]
        for s_ in self._streams[:ns]:
            s_.wait_stream(main)                 # the latents and the packed weights are ready on the main stream
        for n, (i, j, zt) in enumerate(flat):
            st = self._streams[n % ns]
            with torch.cuda.stream(st):
                zt.record_stream(st)
                t = tile_fn(zt)
                t.record_stream(main)            # consumed (cross-faded, concatenated) on the main stream below
                out[i][j] = t
        for s_ in self._streams[:ns]:
            main.wait_stream(s_)
        return out

    @torch.no_grad()
    def _decode_one(self, z):
        """z [C, T, H, W] -> [3, T', 16 H, 16 W] bf16."""
        if z.device.type != "cuda" or self.dtype != torch.bfloat16:
            raise _l.ApexMIError("hunyuanvideo15_mi355 VAE needs bf16 weights and latents on a ROCm device (no CPU fallback)")
        _, T, H, W = z.shape
        zc = z.to(self.storage_dtype).permute(1, 2, 3, 0).contiguous()
        tlh, tlw = self.tile_latent_min_height, self.tile_latent_min_width
        if not (self.use_tiling and (W > tlw or H > tlh)):
            out = self._decode_tile(zc)
        else:
            ovh, ovw = int(tlh * (1 - self.tile_overlap_factor)), int(tlw * (1 - self.tile_overlap_factor))
            bh = int(self.tile_sample_min_height * self.tile_overlap_factor)
            bw = int(self.tile_sample_min_width * self.tile_overlap_factor)
            lh, lw = self.tile_sample_min_height - bh, self.tile_sample_min_width - bw
            rows = self._decode_tiles([[zc[:, i:i + tlh, j:j + tlw].contiguous() for j in range(0, W, ovw)]
                                       for i in range(0, H, ovh)])
            out_rows = []
            for i, row in enumerate(rows):
                parts = []
                for j, tile in enumerate(row):
                    if i > 0:
                        a = rows[i - 1][j]
                        e = min(a.shape[1], tile.shape[1], bh)
                        ops.crossfade_(a[:, a.shape[1] - e:, :tile.shape[2]], tile[:, :e], dim=1)
                    if j > 0:
                        a = row[j - 1]
                        e = min(a.shape[2], tile.shape[2], bw)
                        ops.crossfade_(a[:, :tile.shape[1], a.shape[2] - e:], tile[:, :, :e], dim=2)
                    parts.append(tile[:, :lh, :lw])
                out_rows.append(torch.cat(parts, dim=2))
            out = torch.cat(out_rows, dim=1)
        return out[..., :self.config.out_channels].permute(3, 0, 1, 2).contiguous()

    @ops.on_model_device
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        if self.use_light_vae:
            # model.py:958-962: the light decoder's [1, N, 3, T', H', W'] result is returned AS IS (no tuple, no
            # DecoderOutput), which the caller's `[0]` turns into the video
            self._ensure_light_vae_loaded()
            return self.light_vae.decode(z, parallel=False, show_progress_bar=True, skip_trim=False)
        dec = torch.stack([self._decode_one(z[b]) for b in range(z.shape[0])], dim=0).to(z.dtype)
        if not return_dict:
            return (dec,)
        return SimpleNamespace(sample=dec)
