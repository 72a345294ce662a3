"""TAEHV decoder — the "light VAE" behind HunyuanVideo-1.5's `use_light_vae` switch — on the MI355X HIP ops.

Mirrors what the engines use of the reference classes:
  * `TAEHV` (apps/api/src/vae/tae/model.py:179-333): constructor arguments, `decoder.*` state-dict keys (the Sequential's
    indices), `patch_tgrow_layers`, `decode_video(x [N, T, C, H, W], parallel, show_progress_bar)`, `frames_to_trim`.
    `encode_video` too (`encoder.*` keys); the light-VAE wrapper below builds it decode-only and drops `encoder.*` keys on load
    (no engine of SURVEY.md §8 encodes with TAEHV).
  * `AutoencoderKLHunyuanVideo15Light` (apps/api/src/vae/hunyuanvideo15/model.py:1163-1234): `taehv.*` keys,
    `decode(latents, parallel, show_progress_bar, skip_trim)` = TAEHV over latents / scaling_factor, returned with the
    reference's extra leading axis ([1, N, 3, T', H', W'] — its caller indexes [0], base_engine.py:2055-2057).

How it runs (channels-last [T, H, W, C] bf16, one clip at a time):
  * every 3x3 convolution is the implicit-GEMM MFMA kernel with bias, residual and leaky-ReLU applied in f32 in its epilogue
    (`apexmi_conv3d_cl_act`);
  * MemBlock's `conv(torch.cat([x, past], 1))` (:44; past = the previous frame, zeros before the first) is ONE causal kT = 2
    convolution over the clip — no concatenated copy, no shifted copy — and `act(conv(...) + skip(x))` is that kernel's
    residual + activation epilogue.  The reference's sequential graph walk (:102-175, `parallel=False`) and its parallel mode
    are the same function of the input; both map to this full-sequence form;
  * `nn.Upsample(2)` -> TGrow (1x1) -> 3x3 conv (:246-248) runs as: TGrow at the LOW resolution (a 1x1 convolution commutes
    with a nearest upsample exactly) as one GEMM, its channel blocks re-read as frames, then the 3x3 convolution reading
    through the 2x upsample in its gather — the 4x larger images are never written;
  * input `Clamp` and the output clamp + pixel-shuffle + frame trim are one small kernel each.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, bias=True, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k, **kw), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(cout, **kw), requires_grad=False)
        else:
            self.bias = None


class _None(nn.Module):
    """Parameter-less entry of the reference Sequential (Clamp, activation, nn.Upsample): keeps the indices."""


class _MemBlock(nn.Module):
    def __init__(self, n, **kw):
        super().__init__()
        self.conv = nn.ModuleList([_Conv(2 * n, n, 3, **kw), _None(), _Conv(n, n, 3, **kw), _None(), _Conv(n, n, 3, **kw)])


class _TPool(nn.Module):
    def __init__(self, n, stride, **kw):
        super().__init__()
        self.stride = stride
        self.conv = _Conv(n * stride, n, 1, bias=False, **kw)


class _TGrow(nn.Module):
    def __init__(self, n, stride, **kw):
        super().__init__()
        self.stride = stride
        self.conv = _Conv(n, n * stride, 1, bias=False, **kw)


class TAEHV(nn.Module):
    N_F = (256, 128, 64, 64)

    def __init__(self, checkpoint_path: Optional[str] = None, decoder_time_upscale: Sequence[bool] = (True, True),
                 decoder_space_upscale: Sequence[bool] = (True, True, True), patch_size: int = 1, latent_channels: int = 32,
                 model_type: str = "wan21", with_encoder: bool = True, device=None, dtype=torch.bfloat16):
        super().__init__()
        if dtype != torch.bfloat16:
            raise ValueError("taehv_mi355 computes in bf16")
        self.with_encoder = with_encoder
        self.patch_size, self.latent_channels, self.image_channels, self.model_type = patch_size, latent_channels, 3, model_type
        self.is_cogvideox = checkpoint_path is not None and "taecvx" in checkpoint_path
        if model_type == "wan22":
            self.patch_size, self.latent_channels = 2, 48
        self.slope = 0.2 if model_type == "hy15" else 0.0            # LeakyReLU(0.2) for "hy15", ReLU otherwise (:208-211)
        self.frames_to_trim = 2 ** sum(bool(t) for t in decoder_time_upscale) - 1
        self.space = [bool(s) for s in decoder_space_upscale]
        kw = dict(device=device, dtype=dtype)
        n = self.N_F
        tg = (1, 2 if decoder_time_upscale[0] else 1, 2 if decoder_time_upscale[1] else 1)
        mods = [_None(), _Conv(self.latent_channels, n[0], 3, **kw), _None()]
        for s in range(3):
            mods += [_MemBlock(n[s], **kw), _MemBlock(n[s], **kw), _MemBlock(n[s], **kw), _None(), _TGrow(n[s], tg[s], **kw),
                     _Conv(n[s], n[s + 1], 3, bias=False, **kw)]
        mods += [_None(), _Conv(n[3], self.image_channels * self.patch_size ** 2, 3, **kw)]
        self.decoder = nn.ModuleList(mods)
        if with_encoder:      # tae/model.py:214-236 (the light VAE of HunyuanVideo-1.5 only decodes and is built without it)
            enc = [_Conv(self.image_channels * self.patch_size ** 2, 64, 3, **kw), _None()]
            for stride in (2, 2, 1):
                enc += [_TPool(64, stride, **kw), _Conv(64, 64, 3, bias=False, **kw), _MemBlock(64, **kw), _MemBlock(64, **kw),
                        _MemBlock(64, **kw)]
            enc += [_Conv(64, self.latent_channels, 3, **kw)]
            self.encoder = nn.ModuleList(enc)
        self._packed: Dict[int, Tuple[torch.Tensor, Optional[torch.Tensor]]] = {}
        if checkpoint_path is not None:
            self.load_state_dict(self.patch_tgrow_layers(_read_checkpoint(checkpoint_path)))

    # ---- state ------------------------------------------------------------------------------------------------------
    storage_dtype = torch.bfloat16

    def set_storage_dtype(self, dtype: torch.dtype):
        """torch.bfloat16 (production) or torch.float32: the f32-storage verification mode — activations float,
        the `_f32` entry points (convolutions on the exact bf16 split).  Weights stay bf16."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError(f"activation storage must be bfloat16 or float32, got {dtype}")
        self.storage_dtype = dtype
        return self

    @property
    def dtype(self):
        return self.decoder[1].weight.dtype

    @property
    def device(self):
        return self.decoder[1].weight.device

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def _weights_changed(self):
        self._packed = {}

    def patch_tgrow_layers(self, sd):
        """tae/model.py:283-297: a checkpoint trained with more temporal upscaling keeps the LAST-timestep output channels."""
        new_sd = self.state_dict()
        for i, layer in enumerate(self.decoder):
            if isinstance(layer, _TGrow):
                key = f"decoder.{i}.conv.weight"
                if key in sd and sd[key].shape[0] > new_sd[key].shape[0]:
                    sd[key] = sd[key][-new_sd[key].shape[0]:]
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._packed = {}
        sd = state_dict if self.with_encoder else {k: v for k, v in state_dict.items() if not k.startswith("encoder.")}
        return super().load_state_dict(sd, strict=strict, assign=assign)

    # ---- kernels per layer ------------------------------------------------------------------------------------------
    def _w(self, c: _Conv, mem: bool = False, pool: int = 0):
        p = self._packed.get(id(c))
        if p is None:
            wt = c.weight.data
            if wt.shape[1] % 8:      # the encoder's RGB (x patch^2) input: channels zero-padded to a multiple of 8, as the activations are
                wt = torch.cat([wt, wt.new_zeros(wt.shape[0], 8 - wt.shape[1] % 8, *wt.shape[2:])], dim=1)
            if pool == 2:    # TPool: [n, 2n, 1, 1] over two stacked frames -> causal [n, n, kT=2, 1, 1], tap 0 = the earlier frame
                n = wt.shape[0]
                wt = torch.stack([wt[:, :n], wt[:, n:]], dim=2)
            if mem:      # [n, 2n, 3, 3] over cat([x, past]) -> causal [n, n, kT=2, 3, 3]: tap 1 = this frame, tap 0 = the previous
                n = wt.shape[0]
                wt = torch.stack([wt[:, n:], wt[:, :n]], dim=2)
            w = ops.pack_conv_weight(wt.contiguous())
            b = None
            if c.bias is not None:
                b = torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
                b[:c.bias.numel()] = c.bias.data
            p = (w, b)
            self._packed[id(c)] = p
        return p

    def _conv(self, c: _Conv, x, act: bool, residual=None, mem: bool = False, up: bool = False):
        w, b = self._w(c, mem)
        return ops.conv3d_cl_act(x, w, b, (2 if mem else 1, 3, 3), residual=residual, slope=self.slope if act else None,
                                 upsample2x=up)

    def _memblock(self, blk: _MemBlock, x):
        h = self._conv(blk.conv[0], x, True, mem=True)
        h = self._conv(blk.conv[2], h, True)
        return self._conv(blk.conv[4], h, True, residual=x)

    def _tgrow(self, g: _TGrow, x):
        T, H, W, C = x.shape
        y = ops.gemm(x.view(T * H * W, C), g.conv.weight.data.view(g.stride * C, C)).view(T, H, W, g.stride * C)
        if g.stride == 1:
            return y
        if g.stride != 2:
            raise NotImplementedError("taehv_mi355: TGrow stride must be 1 or 2")
        return ops.time_interleave_cl(y)

    @ops.on_model_device
    def _decode_clip(self, z: torch.Tensor, inv_scale: float, trim: int) -> torch.Tensor:
        """z [C, T, H, W] -> [3, T', H', W'] bf16."""
        d = self.decoder
        x = z.to(self.device, self.storage_dtype).permute(1, 2, 3, 0).contiguous()
        x = ops.tanh_clamp(x, inv_scale)
        x = self._conv(d[1], x, True)
        i = 3
        for s in range(3):
            for b in range(3):
                x = self._memblock(d[i + b], x)
            x = self._tgrow(d[i + 4], x)
            x = self._conv(d[i + 5], x, act=(s == 2), up=self.space[s])
            i += 6
        x = self._conv(d[22], x, False)
        lo = -1.0 if self.model_type == "hy15" else 0.0
        return ops.pixel_shuffle_clamp(x, self.image_channels, self.patch_size, trim=trim, lo=lo, hi=1.0)

    def decode_video(self, x: torch.Tensor, parallel: bool = True, show_progress_bar: bool = True, _inv_scale: float = 1.0):
        """x [N, T, C, H, W] latents -> [N, T', 3, H', W'] (tae/model.py:318-333).  `parallel` selects nothing here: both of
        the reference's modes compute this function."""
        if x.dim() != 5 or x.shape[2] != self.latent_channels:
            raise ValueError(f"TAEHV operates on NTCHW tensors with C={self.latent_channels}, got {tuple(x.shape)}")
        skip_trim = self.is_cogvideox and x.shape[1] % 2 == 0
        trim = 0 if skip_trim else self.frames_to_trim
        if x.shape[1] * (self.frames_to_trim + 1) <= trim:
            raise ValueError("TAEHV: no frame left after trimming")
        outs = [self._decode_clip(x[n].transpose(0, 1), _inv_scale, trim) for n in range(x.shape[0])]
        return torch.stack(outs, 0).transpose(1, 2)

    @ops.on_model_device
    def _encode_clip(self, x: torch.Tensor) -> torch.Tensor:
        """x [T, 3 p^2, h, w] (pixel-unshuffled, T a multiple of 4) -> [T / 4, latent, h / 8, w / 8] bf16."""
        e = self.encoder
        T, C, H, W = x.shape
        xc = torch.zeros((T, H, W, (C + 7) // 8 * 8), dtype=self.storage_dtype, device=self.device)
        xc[..., :C] = x.to(self.device, self.storage_dtype).permute(0, 2, 3, 1)
        x = self._conv(e[0], xc, True)
        i = 2
        for _ in range(3):
            pool = e[i]
            if pool.stride == 2:     # frames (2j, 2j + 1) stacked along the channels + 1x1 conv == a kT = 2 conv at temporal stride 2
                w, _b = self._w(pool.conv, pool=2)
                x = ops.conv3d_cl_tstrided(x, w, None, (2, 1, 1), 2, 1, x.shape[0] // 2)
            else:
                t_, h_, w_, c_ = x.shape
                x = ops.gemm(x.view(t_ * h_ * w_, c_), pool.conv.weight.data.view(c_, c_)).view(t_, h_, w_, c_)
            w, _b = self._w(e[i + 1])
            x = ops.conv2d_cl_strided(x, w, None, stride=2, pad=1)
            for b in range(3):
                x = self._memblock(e[i + 2 + b], x)
            i += 5
        x = self._conv(e[17], x, False)
        return x[..., :self.latent_channels].permute(0, 3, 1, 2)

    def encode_video(self, x: torch.Tensor, parallel: bool = True, show_progress_bar: bool = True):
        """x [N, T, 3, H, W] RGB in [0, 1] -> latents [N, T' / 4, C, H / (8 p), W / (8 p)] (tae/model.py:299-316); the clip is
        padded at the end to a multiple of 4 frames by repeating the last one.  `parallel` selects nothing here."""
        if not self.with_encoder:
            raise RuntimeError("this TAEHV was built without its encoder (with_encoder=False: the decode-only light VAE)")
        if x.dim() != 5 or x.shape[2] != self.image_channels:
            raise ValueError(f"TAEHV operates on NTCHW RGB tensors, got {tuple(x.shape)}")
        if self.patch_size > 1:
            x = torch.nn.functional.pixel_unshuffle(x, self.patch_size)
        if x.shape[1] % 4 != 0:
            x = torch.cat([x, x[:, -1:].repeat_interleave(4 - x.shape[1] % 4, dim=1)], 1)
        return torch.stack([self._encode_clip(x[n]) for n in range(x.shape[0])], 0)


def _read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """Local .pth / .safetensors file (the reference resolves `light_vae_path` through its DownloadMixin first,
    vae/hunyuanvideo15/model.py:1203-1223; here the path must already be a file)."""
    if not os.path.isfile(path):
        raise FileNotFoundError(f"light VAE checkpoint {path!r} is not a local file")
    low = path.lower()
    if low.endswith(".pth"):
        return torch.load(path, map_location="cpu", weights_only=True)
    if low.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    raise ValueError(f"Unsupported checkpoint format for light VAE: {path}. Supported formats: .pth, .safetensors")


class AutoencoderKLHunyuanVideo15Light(nn.Module):
    def __init__(self, scaling_factor: float = 1.03682, taehv_checkpoint_path: Optional[str] = None, taehv_model_type: str = "hy15",
                 taehv_latent_channels: int = 32, taehv_patch_size: int = 2, load_on_init: bool = True, device=None,
                 dtype=torch.bfloat16):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.taehv_checkpoint_path = taehv_checkpoint_path
        self.taehv = TAEHV(checkpoint_path=None, model_type=taehv_model_type, latent_channels=taehv_latent_channels,
                           patch_size=taehv_patch_size, with_encoder=False, device=device, dtype=dtype)
        on_meta = self.taehv.decoder[1].weight.is_meta
        if load_on_init and taehv_checkpoint_path is not None and not on_meta:
            self.load_taehv_weights(taehv_checkpoint_path)

    @property
    def dtype(self):
        return self.taehv.dtype

    @property
    def device(self):
        return self.taehv.device

    def load_taehv_weights(self, taehv_checkpoint_path: str) -> None:
        sd = self.taehv.patch_tgrow_layers(_read_checkpoint(taehv_checkpoint_path))
        self.taehv.load_state_dict({k: v.to(self.taehv.dtype) for k, v in sd.items()}, strict=True)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {k: v for k, v in state_dict.items() if not k.startswith("taehv.encoder.")}
        self.taehv._packed = {}
        return super().load_state_dict(sd, strict=strict, assign=assign)

    def _weights_changed(self):
        self.taehv._weights_changed()

    def decode(self, latents: torch.Tensor, parallel: bool = False, show_progress_bar: bool = True, skip_trim: bool = False):
        """vae/hunyuanvideo15/model.py:1225-1234 (`skip_trim` is accepted and unused there too)."""
        out = self.taehv.decode_video(latents.transpose(1, 2), parallel, show_progress_bar, _inv_scale=1.0 / self.scaling_factor)
        return out.transpose(1, 2).unsqueeze(0)
