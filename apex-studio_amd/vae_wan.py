"""AutoencoderKLWan on the MI355X HIP ops — drop-in for VAE registry keys "wan" / "qwenimage".

Mirrors what the engines use of the reference class (apps/api/src/vae/wan/model.py:1083-1672, identical
architecture at vae/qwenimage/model.py:774): `from_config`, the reference's state-dict keys (`encoder.*`, `quant_conv.*`,
`decoder.*`, `post_quant_conv.*`), `.config` (z_dim, latents_mean, latents_std), `enable_tiling(...)`,
`denormalize_latents` / `normalize_latents`, `decode(z, return_dict=False)[0]`,
`encode(x, return_dict=False)[0].mode() / .sample(generator)` (BaseEngine.vae_encode, engine/base_engine.py:2061-2165:
the condition image of QwenImage-Edit, first frames of image-to-video), `.dtype`.

Encode mirrors decode: every spatial tile's whole 1 + 4k frame sequence in one causal pass (the reference streams the
first frame and then chunks of four with `feat_cache`, model.py:1273-1304; oracle/vae_wan.py proves both give the same
numbers), the stride-2 spatial downsampling as a strided implicit-GEMM convolution with the zero pad on the right / bottom
only, the temporal downsampling as "frame 0 passes through, output j is the 3-tap convolution ending at frame 2j", tiles
of 256 px with stride 192 blended in latent space (`tiled_encode`, :1424-1514).

Decode runs channels-last and processes every spatial tile's WHOLE frame sequence in one causal pass
(the reference streams frame by frame with `feat_cache`; oracle/vae_wan.py proves both give the same
numbers, including the first-frame "Rep" rule).  Tiles and their in-place linear blends follow
`tiled_decode` exactly (model.py:1516-1623): tiling is part of the numerical contract because each tile
sees zero padding at its borders.  Per tile the kernels are: implicit-GEMM causal conv3d on MFMA with
fused bias + residual, RMS-norm(channel)+SiLU, nearest 2x upsample, frame interleave, GEMMs for the
1x1 convs, the attention operator for the 384-channel single-head mid block, and a crossfade for the blends.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .flux import _Config

WAN_LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715,
                    0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
WAN_LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652,
                   1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class _Conv(nn.Module):
    def __init__(self, cin, cout, ksize, **kw):
        super().__init__()
        self.ksize = tuple(ksize)
        self.weight = nn.Parameter(torch.empty(cout, cin, *ksize, **kw), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout, **kw), requires_grad=False)


class _Gamma(nn.Module):
    def __init__(self, dim, images, **kw):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones((dim, 1, 1) if images else (dim, 1, 1, 1), **kw), requires_grad=False)


class _Res(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.norm1 = _Gamma(cin, False, **kw)
        self.conv1 = _Conv(cin, cout, (3, 3, 3), **kw)
        self.norm2 = _Gamma(cout, False, **kw)
        self.conv2 = _Conv(cout, cout, (3, 3, 3), **kw)
        self.conv_shortcut = _Conv(cin, cout, (1, 1, 1), **kw) if cin != cout else nn.Identity()


class _Attn(nn.Module):
    def __init__(self, dim, **kw):
        super().__init__()
        self.norm = _Gamma(dim, True, **kw)
        self.to_qkv = _Conv(dim, 3 * dim, (1, 1), **kw)
        self.proj = _Conv(dim, dim, (1, 1), **kw)


class _Resample(nn.Module):
    def __init__(self, dim, mode, **kw):
        super().__init__()
        self.mode = mode
        self.resample = nn.ModuleList([nn.Identity(), _Conv(dim, dim // 2, (3, 3), **kw)])
        if mode == "upsample3d":
            self.time_conv = _Conv(dim, 2 * dim, (3, 1, 1), **kw)


class _Down(nn.Module):
    def __init__(self, dim, mode, **kw):
        super().__init__()
        self.mode = mode
        self.resample = nn.ModuleList([nn.Identity(), _Conv(dim, dim, (3, 3), **kw)])
        if mode == "downsample3d":
            self.time_conv = _Conv(dim, dim, (3, 1, 1), **kw)


class _Mid(nn.Module):
    def __init__(self, dim, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(dim, dim, **kw), _Res(dim, dim, **kw)])
        self.attentions = nn.ModuleList([_Attn(dim, **kw)])


class _Up(nn.Module):
    def __init__(self, cin, cout, n, mode, **kw):
        super().__init__()
        res, cur = [], cin
        for _ in range(n + 1):
            res.append(_Res(cur, cout, **kw))
            cur = cout
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([_Resample(cout, mode, **kw)]) if mode else None


class _Decoder(nn.Module):
    def __init__(self, dim, z_dim, dim_mult, num_res_blocks, temperal_upsample, out_channels, **kw):
        super().__init__()
        dims = [dim * u for u in [dim_mult[-1]] + dim_mult[::-1]]
        self.conv_in = _Conv(z_dim, dims[0], (3, 3, 3), **kw)
        self.mid_block = _Mid(dims[0], **kw)
        ups = []
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            if i > 0:
                cin = cin // 2
            up = i != len(dim_mult) - 1
            mode = ("upsample3d" if temperal_upsample[i] else "upsample2d") if up else None
            ups.append(_Up(cin, cout, num_res_blocks, mode, **kw))
        self.up_blocks = nn.ModuleList(ups)
        self.norm_out = _Gamma(dims[-1], False, **kw)
        self.conv_out = _Conv(dims[-1], out_channels, (3, 3, 3), **kw)


class _Encoder(nn.Module):
    def __init__(self, in_channels, dim, z_dim, dim_mult, num_res_blocks, temperal_downsample, **kw):
        super().__init__()
        dims = [dim * u for u in [1] + dim_mult]
        self.conv_in = _Conv(in_channels, dims[0], (3, 3, 3), **kw)
        blocks = []
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                blocks.append(_Res(cin, cout, **kw))
                cin = cout
            if i != len(dim_mult) - 1:
                blocks.append(_Down(cout, "downsample3d" if temperal_downsample[i] else "downsample2d", **kw))
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid(dims[-1], **kw)
        self.norm_out = _Gamma(dims[-1], False, **kw)
        self.conv_out = _Conv(dims[-1], z_dim, (3, 3, 3), **kw)


class DiagonalGaussianDistribution:
    """The posterior object `encode(...)[0]` returns (diffusers DiagonalGaussianDistribution as the reference uses it,
    model.py:1327): parameters [B, 2 z, T, H, W] = mean | logvar, logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self) -> torch.Tensor:
        return self.mean

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        dev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise


class AutoencoderKLWan(nn.Module):
    def __init__(self, base_dim: int = 96, decoder_base_dim: Optional[int] = None, z_dim: int = 16,
                 dim_mult: List[int] = (1, 2, 4, 4), num_res_blocks: int = 2, attn_scales=(),
                 temperal_downsample=(False, True, True), dropout: float = 0.0,
                 latents_mean=WAN_LATENTS_MEAN, latents_std=WAN_LATENTS_STD, is_residual: bool = False,
                 in_channels: int = 3, out_channels: int = 3, patch_size: Optional[int] = None,
                 scale_factor_temporal: int = 4, scale_factor_spatial: int = 8, device=None,
                 dtype=torch.bfloat16):
        super().__init__()
        if is_residual or patch_size is not None or list(attn_scales):
            raise NotImplementedError("wan_mi355 VAE: residual (Wan 2.2 TI2V-5B) / patchified / attn_scales "
                                      "variants are outside the configured hot path")
        self.config = _Config(base_dim=base_dim, decoder_base_dim=decoder_base_dim, z_dim=z_dim,
                              dim_mult=list(dim_mult), num_res_blocks=num_res_blocks,
                              temperal_downsample=list(temperal_downsample), latents_mean=list(latents_mean),
                              latents_std=list(latents_std), in_channels=in_channels, out_channels=out_channels,
                              patch_size=patch_size,
                              scale_factor_temporal=scale_factor_temporal,
                              scale_factor_spatial=scale_factor_spatial)
        kw = dict(device=device, dtype=dtype)
        self.z_dim = z_dim
        self.temperal_downsample = list(temperal_downsample)
        self.encoder = _Encoder(in_channels, base_dim, z_dim * 2, list(dim_mult), num_res_blocks,
                                list(temperal_downsample), **kw)
        self.quant_conv = _Conv(z_dim * 2, z_dim * 2, (1, 1, 1), **kw)
        self.post_quant_conv = _Conv(z_dim, z_dim, (1, 1, 1), **kw)
        self.decoder = _Decoder(decoder_base_dim or base_dim, z_dim, list(dim_mult), num_res_blocks,
                                list(temperal_downsample)[::-1], out_channels, **kw)
        self.spatial_compression_ratio = scale_factor_spatial
        self.use_tiling = False
        self.tile_sample_min_height = self.tile_sample_min_width = 256
        self.tile_sample_stride_height = self.tile_sample_stride_width = 192
        self._packed: Dict[int, torch.Tensor] = {}
        self._indep = False      # inside a batched pass over independent single-frame tiles
        self.tile_streams = 2            # video tiles decoded / encoded side by side on HIP streams (see _run_tiles_on_streams)
        self._streams: list = []
        self.batch_single_frame_tiles = True
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
        return self.post_quant_conv.weight.dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        return super().load_state_dict(*a, **k)

    def _weights_changed(self):
        """Parameters were written in place (`weights.load_checkpoint_into`): drop the packed conv-weight cache."""
        self._packed = {}

    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None,
                      tile_sample_stride_height=None, tile_sample_stride_width=None):
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_sample_stride_height = tile_sample_stride_height or self.tile_sample_stride_height
        self.tile_sample_stride_width = tile_sample_stride_width or self.tile_sample_stride_width

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        return None

    @torch.no_grad()
    def denormalize_latents(self, latents):
        mean = torch.tensor(self.config.latents_mean).view(1, self.z_dim, 1, 1, 1).to(latents.device, latents.dtype)
        inv_std = 1.0 / torch.tensor(self.config.latents_std).view(1, self.z_dim, 1, 1, 1).to(latents.device,
                                                                                               latents.dtype)
        return latents / inv_std + mean

    @torch.no_grad()
    def normalize_latents(self, latents):
        mean = torch.tensor(self.config.latents_mean).view(1, self.z_dim, 1, 1, 1).to(latents.device, latents.dtype)
        inv_std = 1.0 / torch.tensor(self.config.latents_std).view(1, self.z_dim, 1, 1, 1).to(latents.device,
                                                                                               latents.dtype)
        return (latents - mean) * inv_std

    # ---- kernels per layer ----------------------------------------------------------------------
    def _w(self, conv: _Conv):
        key = id(conv)
        p = self._packed.get(key)
        if p is None:
            wt = conv.weight.data
            if wt.shape[1] % 8:      # the RGB input convolution: channels zero-padded to 8, as the activations are
                pad = torch.zeros(wt.shape[0], 8 - wt.shape[1] % 8, *wt.shape[2:], dtype=wt.dtype, device=wt.device)
                wt = torch.cat([wt, pad], dim=1)
            w = ops.pack_conv_weight(wt)
            b = torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
            b[:conv.bias.numel()] = conv.bias.data
            p = (w, b)
            self._packed[key] = p
        return p

    def _conv(self, conv: _Conv, x, residual=None, upsample2x=False):
        w, b = self._w(conv)
        k = conv.ksize if len(conv.ksize) == 3 else (1,) + conv.ksize
        return ops.conv3d_cl(x, w, b, k, residual=residual, independent_frames=self._indep and k[0] > 1,
                             upsample2x=upsample2x)

    def _run_tiles(self, fn, tiles):
        """fn over every tile.  Single-frame tiles (QwenImage's image VAE, first-frame encodes) of equal shape run as ONE
        pass over a stacked [n, h, w, C] tensor whose frames are treated as independent one-frame clips
        (`apexmi_conv3d_cl_frames`; the temporal resamplers are identities for one frame): a 1024x1024 image is 4 shape
        groups instead of 36 launch-bound tile passes, with bit-identical results."""
        if len(tiles) == 1 or tiles[0].shape[0] != 1 or not self.batch_single_frame_tiles:
            return self._run_tiles_on_streams(fn, tiles)
        groups: Dict[tuple, List[int]] = {}
        for i, t in enumerate(tiles):
            groups.setdefault(tuple(t.shape), []).append(i)
        outs = [None] * len(tiles)
        self._indep = True
        try:
            for idxs in groups.values():
                y = fn(torch.cat([tiles[i] for i in idxs], dim=0))
                for n, i in enumerate(idxs):
                    outs[i] = y[n:n + 1]
        finally:
            self._indep = False
        return outs

    def _run_tiles_on_streams(self, fn, tiles):
        """Video tiles are independent until the cross-fades, and the first (lowest-resolution) stages of a tile launch fewer
        workgroups than the chip has slots: `tile_streams` tiles run side by side on their own HIP streams so those launches
        overlap (the full-size launches of the late stages simply queue).  Same kernels on the same data: bit-identical to the
        sequential walk (`tile_streams = 1`).  The first tile runs on the calling stream and fills the packed-weight caches."""
        ns = max(1, min(int(self.tile_streams), len(tiles) - 1))
        if ns <= 1 or len(tiles) < 3 or not tiles[0].is_cuda:
            return [fn(t) for t in tiles]
        main = torch.cuda.current_stream()
        if len(self._streams) < ns:
            self._streams += [torch.cuda.Stream(device=tiles[0].device) for _ in range(ns - len(self._streams))]
        outs = [fn(tiles[0])]
        for s_ in self._streams[:ns]:
            s_.wait_stream(main)
        for n, t in enumerate(tiles[1:]):
            st = self._streams[n % ns]
            with torch.cuda.stream(st):
                t.record_stream(st)
                y = fn(t)
                y.record_stream(main)
                outs.append(y)
        for s_ in self._streams[:ns]:
            main.wait_stream(s_)
        return outs

    @staticmethod
    def _gamma(n):
        return n.gamma.data.reshape(-1).contiguous()

    def _conv_norm(self, conv: _Conv, x, gamma, silu=True, residual=None, want_raw=True, upsample2x=False):
        """A convolution whose output's RMS norm (+ SiLU) — the consumer's `norm1` / `norm2` / `norm_out` — is produced
        in the conv's own epilogue (`apexmi_conv3d_cl_norm`) instead of a separate read-modify-write pass over the
        tensor; shapes the fused tiles do not cover fall back to the two launches inside `ops.conv3d_cl_norm`."""
        w, b = self._w(conv)
        k = conv.ksize if len(conv.ksize) == 3 else (1,) + conv.ksize
        return ops.conv3d_cl_norm(x, w, b, k, gamma, silu=silu, residual=residual, want_raw=want_raw, upsample2x=upsample2x,
                                  independent_frames=self._indep and k[0] > 1)

    def _res(self, blk: _Res, x, xn=None, next_norm=None, want_raw=True):
        """WanResidualBlock (reference vae/wan/model.py:389-441).  `xn` = silu(norm1(x)) when the producer of x already
        made it; `next_norm` = (gamma, silu) of the norm that will read this block's output.  conv1's output is read by
        norm2 only, so it is never stored: conv1 writes silu(norm2(.)) directly.  Returns (out, normed out or None)."""
        h = x if isinstance(blk.conv_shortcut, nn.Identity) else self._conv(blk.conv_shortcut, x)
        if xn is None:
            xn = ops.rmsnorm_cl(x, self._gamma(blk.norm1), silu=True)
        _, n2 = self._conv_norm(blk.conv1, xn, self._gamma(blk.norm2), silu=True, want_raw=False)
        if next_norm is None:
            return self._conv(blk.conv2, n2, residual=h), None
        # want_raw=False (the decoder's LAST block): only norm_out reads this output, so the raw tensor — 1 GB per full-resolution
        # tile — is never stored
        return self._conv_norm(blk.conv2, n2, next_norm[0], silu=next_norm[1], residual=h, want_raw=want_raw)

    def _attn(self, blk: _Attn, x, n=None):
        T, H, W, Cc = x.shape
        if n is None:
            n = ops.rmsnorm_cl(x, blk.norm.gamma.data.reshape(-1).contiguous())
        qkv = ops.gemm(n.view(T * H * W, Cc), blk.to_qkv.weight.data.reshape(3 * Cc, Cc), blk.to_qkv.bias.data)
        qkv = qkv.view(T, 1, H * W, 3 * Cc)
        o = ops.attention(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:])     # [T,1,HW,C] view
        o = o.permute(0, 2, 1, 3).reshape(T * H * W, Cc)
        ones = torch.ones(Cc, dtype=torch.float32, device=x.device)
        out = ops.gemm(o, blk.proj.weight.data.reshape(Cc, Cc), blk.proj.bias.data, epilogue="gate_res",
                       gate=ones, residual=x.view(T * H * W, Cc))
        return out.view(T, H, W, Cc)

    def _resample(self, up: _Resample, x, next_norm=None):
        T = x.shape[0]
        if up.mode == "upsample3d" and T > 1 and not self._indep:
            y = self._conv(up.time_conv, x[1:].contiguous())        # never sees frame 0 (the "Rep" rule)
            x = torch.cat([x[:1], ops.time_interleave_cl(y)], dim=0)
        # WanUpsample (nearest-exact 2x) is folded into the convolution's gather: no 4x larger intermediate
        if next_norm is None:
            return self._conv(up.resample[1], x, upsample2x=True), None
        return self._conv_norm(up.resample[1], x, next_norm[0], silu=next_norm[1], upsample2x=True)

    def _decode_tile(self, z):
        """z [T, h, w, z_dim] channels-last -> [T', 8h, 8w, 4] (3 channels + 1 pad).  Every RMS norm whose input comes
        out of a convolution is produced by that convolution (`_conv_norm`); `xn` carries it to its consumer."""
        d = self.decoder
        g = self._gamma
        mid = d.mid_block
        x = self._conv(self.post_quant_conv, z)
        x, xn = self._conv_norm(d.conv_in, x, g(mid.resnets[0].norm1))
        x, xn = self._res(mid.resnets[0], x, xn, next_norm=(g(mid.attentions[0].norm), False))
        x = self._attn(mid.attentions[0], x, xn)
        # the chain of residual blocks / resamplers that follows; each one is told which norm reads its output
        chain = [mid.resnets[1]]
        for up in d.up_blocks:
            chain += list(up.resnets)
            if up.upsamplers is not None:
                chain.append(up.upsamplers[0])
        xn = None
        for i, m in enumerate(chain):
            nxt = chain[i + 1] if i + 1 < len(chain) else None
            if nxt is None:
                nn_ = (g(d.norm_out), True)
            elif isinstance(nxt, _Res):
                nn_ = (g(nxt.norm1), True)
            else:
                nn_ = None                          # a resampler reads the raw tensor
            if isinstance(m, _Res):
                x, xn = self._res(m, x, xn, next_norm=nn_, want_raw=nxt is not None)
            else:
                x, xn = self._resample(m, x, next_norm=nn_)
        return self._conv(d.conv_out, xn)

    @torch.no_grad()
    def _decode_one(self, z):
        """z [C, T, H, W] -> [3, T', 8H, 8W] bf16."""
        if z.device.type != "cuda" or self.dtype != torch.bfloat16:
            raise _l.ApexMIError("wan_mi355 VAE needs bf16 weights and latents on a ROCm device (no CPU fallback)")
        Cz, T, H, W = z.shape
        ratio = self.spatial_compression_ratio
        zc = z.to(self.storage_dtype).permute(1, 2, 3, 0).contiguous()           # [T, H, W, C]
        lat_min_h, lat_min_w = self.tile_sample_min_height // ratio, self.tile_sample_min_width // ratio
        if not (self.use_tiling and (W > lat_min_w or H > lat_min_h)):
            out = self._decode_tile(zc)
        else:
            sh, sw = self.tile_sample_stride_height, self.tile_sample_stride_width
            lsh, lsw = sh // ratio, sw // ratio
            bh, bw = self.tile_sample_min_height - sh, self.tile_sample_min_width - sw
            cols = list(range(0, W, lsw))
            flat = self._run_tiles(self._decode_tile, [zc[:, i:i + lat_min_h, j:j + lat_min_w].contiguous()
                                                       for i in range(0, H, lsh) for j in cols])
            rows = [flat[r * len(cols):(r + 1) * len(cols)] for r in range(len(flat) // len(cols))]
            out_rows = []
            for i, row in enumerate(rows):
                parts = []
                for j, tile in enumerate(row):
                    if i > 0:      # blend_v: top rows of this tile with the bottom rows of the tile above
                        a = rows[i - 1][j]
                        e = min(a.shape[1], tile.shape[1], bh)
                        ops.crossfade_(a[:, a.shape[1] - e:, :tile.shape[2]], tile[:, :e], dim=1)
                    if j > 0:      # blend_h: left columns with the right columns of the (already blended) left tile
                        a = row[j - 1]
                        e = min(a.shape[2], tile.shape[2], bw)
                        ops.crossfade_(a[:, :tile.shape[1], a.shape[2] - e:], tile[:, :, :e], dim=2)
                    parts.append(tile[:, :sh, :sw])
                out_rows.append(torch.cat(parts, dim=2))
            out = torch.cat(out_rows, dim=1)[:, :H * ratio, :W * ratio]
        out = out[..., :self.config.out_channels].clamp(-1.0, 1.0)
        return out.permute(3, 0, 1, 2).contiguous()

    # ---- encode ----------------------------------------------------------------------------------
    def _down(self, dn: _Down, x):
        w, b = self._w(dn.resample[1])
        x = ops.conv2d_cl_down2(x, w, b)
        if dn.mode == "downsample3d" and x.shape[0] > 1 and not self._indep:
            # frame 0 passes through; output j >= 1 is the causal 3-tap convolution ending at frame 2j (temporal stride in
            # the gather: only those frames are computed)
            w, b = self._w(dn.time_conv)
            y = ops.conv3d_cl_tstrided(x, w, b, dn.time_conv.ksize, 2, 2, (x.shape[0] - 1) // 2)
            x = torch.cat([x[:1], y], dim=0)
        return x

    def _encode_tile(self, x):
        """x [T, H, W, 8] channels-last (RGB + zero pad) -> posterior parameters [T', H/8, W/8, 2 z]."""
        e = self.encoder
        x = self._conv(e.conv_in, x)
        for blk in e.down_blocks:
            x = self._down(blk, x) if isinstance(blk, _Down) else self._res(blk, x)[0]
        x = self._res(e.mid_block.resnets[0], x)[0]
        x = self._attn(e.mid_block.attentions[0], x)
        x = self._res(e.mid_block.resnets[1], x)[0]
        x = self._conv(e.conv_out, ops.rmsnorm_cl(x, e.norm_out.gamma.data.reshape(-1).contiguous(), silu=True))
        return self._conv(self.quant_conv, x)

    @torch.no_grad()
    def _encode_one(self, x):
        """x [3, T, H, W] in [-1, 1] -> [2 z, T', H/8, W/8] bf16."""
        if x.device.type != "cuda" or self.dtype != torch.bfloat16:
            raise _l.ApexMIError("wan_mi355 VAE needs bf16 weights and inputs on a ROCm device (no CPU fallback)")
        Cin, T, H, W = x.shape
        if (T - 1) % 4:
            raise ValueError(f"encode expects 1 + 4k frames (the reference's chunking), got {T}")
        ratio = self.spatial_compression_ratio
        if H % ratio or W % ratio:
            raise ValueError(f"encode expects height and width divisible by {ratio}, got {H}x{W}")
        xc = torch.zeros((T, H, W, 8), dtype=self.storage_dtype, device=x.device)
        xc[..., :Cin] = x.to(self.storage_dtype).permute(1, 2, 3, 0)
        mh, mw = self.tile_sample_min_height, self.tile_sample_min_width
        if not (self.use_tiling and (W > mw or H > mh)):
            out = self._encode_tile(xc)
        else:
            sh, sw = self.tile_sample_stride_height, self.tile_sample_stride_width
            lsh, lsw = sh // ratio, sw // ratio
            bh, bw = mh // ratio - lsh, mw // ratio - lsw
            cols = list(range(0, W, sw))
            flat = self._run_tiles(self._encode_tile, [xc[:, i:i + mh, j:j + mw].contiguous()
                                                       for i in range(0, H, sh) for j in cols])
            rows = [flat[r * len(cols):(r + 1) * len(cols)] for r in range(len(flat) // len(cols))]
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
                    parts.append(tile[:, :lsh, :lsw])
                out_rows.append(torch.cat(parts, dim=2))
            out = torch.cat(out_rows, dim=1)[:, :H // ratio, :W // ratio]
        return out[..., :2 * self.z_dim].permute(3, 0, 1, 2).contiguous()

    @ops.on_model_device
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        h = torch.stack([self._encode_one(x[b]) for b in range(x.shape[0])], dim=0).to(x.dtype)
        posterior = DiagonalGaussianDistribution(h)
        if not return_dict:
            return (posterior,)
        return SimpleNamespace(latent_dist=posterior)

    @ops.on_model_device
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        dec = torch.stack([self._decode_one(z[b]) for b in range(z.shape[0])], dim=0).to(z.dtype)
        if not return_dict:
            return (dec,)
        return SimpleNamespace(sample=dec)
