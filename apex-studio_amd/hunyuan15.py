"""HunyuanVideo15Transformer3DModel on the MI355X HIP ops — drop-in for registry key "hunyuanvideo15.base"
(the "Hunyuan passes" of the editor's queue).

Mirrors the reference class (apps/api/src/transformer/hunyuanvideo15/base/model.py:697-1165): same config, same
state-dict keys (x_embedder.proj.weight, context_embedder.token_refiner.refiner_blocks.N..., transformer_blocks.N.
norm1.linear / norm1_context.linear / attn.add_q_proj / ff_context.net.0.proj, cond_type_embed.weight ...) and
`forward(hidden_states [B,C,F,H,W], timestep [B] (0-1000 scale), encoder_hidden_states [B,T1,3584] +
encoder_attention_mask, encoder_hidden_states_2 [B,T2,1472] + encoder_attention_mask_2, image_embeds [B,N,1152],
return_dict=False)`.

The 54 blocks are MM-DiT double-stream blocks (model.py:543-694): AdaLN-Zero on both streams, per-head RMSNorm,
RoPE on the latent tokens only (theta 256, cos/sin repeated per pair), joint attention WITHOUT a mask (the
reference passes None, :1110-1116: padded condition tokens are zeroed and attended), gated residuals, GELU-tanh
MLPs.  The step reuses the Flux/Qwen kernel sequence over one joint buffer; internally the condition tokens come
first (the reference concatenates latent tokens first — attention does not depend on the order of keys).
Around it: the token refiner (two masked self-attention blocks over the MLLM tokens with gates from the pooled
prompt, "linear-silu" MLP), the ByT5 and image projections (erf GELU), the `[valid image | valid byt5 | valid mllm |
invalid image | zeros | zeros]` token reorder (:1058-1108), AdaLayerNormContinuous + un-patchify.
The MeanFlow branch (`use_meanflow`: a second timestep embedder for `timestep_r`, model.py:234-268) adds its embedding to
temb with one more GEMV pair.
"""
from __future__ import annotations

import contextlib
import os
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .flux import _AdaNorm, _Config, _FF, _Linear, _TimestepEmbedding, _repoint
from .lora import LoraAdapterMixin
from .qwenimage import _QwenAttn


class _LN(nn.Module):
    def __init__(self, dim: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim, device=device, dtype=dtype), requires_grad=False)


class _SelfAttn(nn.Module):
    def __init__(self, dim: int, **kw):
        super().__init__()
        self.to_q, self.to_k, self.to_v = _Linear(dim, dim, **kw), _Linear(dim, dim, **kw), _Linear(dim, dim, **kw)
        self.to_out = nn.ModuleList([_Linear(dim, dim, **kw), nn.Identity()])


class _RefinerBlock(nn.Module):
    def __init__(self, dim: int, **kw):
        super().__init__()
        self.norm1 = _LN(dim, **kw)
        self.attn = _SelfAttn(dim, **kw)
        self.norm2 = _LN(dim, **kw)
        self.ff = _FF(dim, 4 * dim, **kw)
        self.norm_out = _AdaNorm(dim, 2, **kw)


class _TextTimeEmbed(nn.Module):
    def __init__(self, dim: int, pooled_dim: int, **kw):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedding(256, dim, **kw)
        self.text_embedder = _TimestepEmbedding(pooled_dim, dim, **kw)   # PixArtAlphaTextProjection: linear_1/linear_2


class _TokenRefiner(nn.Module):
    def __init__(self, in_channels: int, dim: int, num_layers: int, **kw):
        super().__init__()
        self.time_text_embed = _TextTimeEmbed(dim, in_channels, **kw)
        self.proj_in = _Linear(in_channels, dim, **kw)
        self.token_refiner = nn.Module()
        self.token_refiner.refiner_blocks = nn.ModuleList([_RefinerBlock(dim, **kw) for _ in range(num_layers)])


class _ByT5(nn.Module):
    def __init__(self, in_features: int, hidden: int, out_features: int, **kw):
        super().__init__()
        self.norm = _LN(in_features, **kw)
        self.linear_1 = _Linear(in_features, hidden, **kw)
        self.linear_2 = _Linear(hidden, hidden, **kw)
        self.linear_3 = _Linear(hidden, out_features, **kw)


class _ImageProj(nn.Module):
    def __init__(self, in_channels: int, hidden: int, **kw):
        super().__init__()
        self.norm_in = _LN(in_channels, **kw)
        self.linear_1 = _Linear(in_channels, in_channels, **kw)
        self.linear_2 = _Linear(in_channels, hidden, **kw)
        self.norm_out = _LN(hidden, **kw)


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, inner: int, **kw):
        super().__init__()
        self.norm1 = _AdaNorm(dim, 6, **kw)
        self.norm1_context = _AdaNorm(dim, 6, **kw)
        self.attn = _QwenAttn(dim, heads, head_dim, **kw)
        self.ff = _FF(dim, inner, **kw)
        self.ff_context = _FF(dim, inner, **kw)


class _PatchEmbed(nn.Module):
    def __init__(self, patch: Tuple[int, int, int], in_chans: int, dim: int, device=None, dtype=None):
        super().__init__()
        self.proj = nn.Module()
        self.proj.weight = nn.Parameter(torch.empty(dim, in_chans, *patch, device=device, dtype=dtype), requires_grad=False)
        self.proj.bias = nn.Parameter(torch.empty(dim, device=device, dtype=dtype), requires_grad=False)


class _Embedding(nn.Module):
    def __init__(self, n: int, dim: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim, device=device, dtype=dtype), requires_grad=False)


class HunyuanVideo15Transformer3DModel(LoraAdapterMixin, nn.Module):
    _converter_base = "hunyuanvideo15.base"      # which key-converter table original-format weight files / LoRAs go through (converters.py)
    _no_split_modules = ["_Block", "_RefinerBlock"]

    def __init__(self, in_channels: int = 65, out_channels: int = 32, num_attention_heads: int = 16,
                 attention_head_dim: int = 128, num_layers: int = 54, num_refiner_layers: int = 2, mlp_ratio: float = 4.0,
                 patch_size: int = 1, patch_size_t: int = 1, qk_norm: str = "rms_norm", text_embed_dim: int = 3584,
                 text_embed_2_dim: int = 1472, image_embed_dim: int = 1152, rope_theta: float = 256.0,
                 rope_axes_dim: Tuple[int, ...] = (16, 56, 56), target_size: int = 640, task_type: str = "i2v",
                 use_meanflow: bool = False, chunking_profile: str = "none", ffn_chunk_size=None, ffn_chunk_dim: int = 1,
                 rope_chunk_size=None, device=None, dtype=torch.bfloat16):
        super().__init__()
        if attention_head_dim != 128 or qk_norm != "rms_norm":
            raise _l.ApexMIError("hunyuanvideo15.mi355: attention_head_dim must be 128 and qk_norm 'rms_norm'")
        out_channels = out_channels or in_channels
        self.config = _Config(in_channels=in_channels, out_channels=out_channels, num_attention_heads=num_attention_heads,
                              attention_head_dim=attention_head_dim, num_layers=num_layers,
                              num_refiner_layers=num_refiner_layers, mlp_ratio=mlp_ratio, patch_size=patch_size,
                              patch_size_t=patch_size_t, qk_norm=qk_norm, text_embed_dim=text_embed_dim,
                              text_embed_2_dim=text_embed_2_dim, image_embed_dim=image_embed_dim, rope_theta=rope_theta,
                              rope_axes_dim=tuple(rope_axes_dim), target_size=target_size, task_type=task_type,
                              use_meanflow=use_meanflow)
        kw = dict(device=device, dtype=dtype)
        self.inner_dim = dim = num_attention_heads * attention_head_dim
        self.x_embedder = _PatchEmbed((patch_size_t, patch_size, patch_size), in_channels, dim, **kw)
        self.image_embedder = _ImageProj(image_embed_dim, dim, **kw)
        self.context_embedder = _TokenRefiner(text_embed_dim, dim, num_refiner_layers, **kw)
        self.context_embedder_2 = _ByT5(text_embed_2_dim, 2048, dim, **kw)
        self.time_embed = nn.Module()
        self.time_embed.timestep_embedder = _TimestepEmbedding(256, dim, **kw)
        if use_meanflow:      # HunyuanVideo15TimeEmbedding (model.py:234-268): temb = embed(t) + embed_r(r)
            self.time_embed.timestep_embedder_r = _TimestepEmbedding(256, dim, **kw)
        self.cond_type_embed = _Embedding(3, dim, **kw)
        self.transformer_blocks = nn.ModuleList(
            [_Block(dim, num_attention_heads, attention_head_dim, int(dim * mlp_ratio), **kw) for _ in range(num_layers)])
        self.norm_out = _AdaNorm(dim, 2, **kw)
        self.proj_out = _Linear(dim, patch_size_t * patch_size * patch_size * out_channels, **kw)
        self._packed = False
        self._ws: Dict[Any, Any] = {}
        self._rope: Dict[Any, torch.Tensor] = {}
        self._side = None
        self.storage_dtype = torch.bfloat16
        # q/k/v preparation in the QKV GEMM's epilogue where the launch allows it (APEX_FUSE_QKV=0: A/B)
        self.fuse_qkv = os.environ.get("APEX_FUSE_QKV", "1") != "0"

    # ---- the duck-typed surface LoaderMixin / the engines use ----
    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config) if isinstance(config, dict) else dict(vars(config))
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    _from_config = from_config

    @property
    def dtype(self):
        return self.proj_out.weight.dtype

    @property
    def device(self):
        return self.proj_out.weight.device

    @contextlib.contextmanager
    def cache_context(self, name: str):
        yield

    def set_chunking_profile(self, profile_name: str) -> None:
        """Chunking exists in the reference to fit 24 GB cards (model.py:904-924); nothing to do with 288 GB."""

    def _apply(self, fn, *a, **k):
        self._packed = False
        self._ws, self._rope = {}, {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = False
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def init_synthetic(self, seed: int = 0, std: float = 0.02):
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith(".bias"):
                p.data.copy_((torch.randn(p.shape, generator=g, device=p.device) * 0.01).to(p.dtype))
            elif p.dim() == 1:
                p.data.fill_(1.0)
            else:
                p.data.copy_((torch.randn(p.shape, generator=g, device=p.device) * std).to(p.dtype))
        self._packed = False
        return self

    @torch.no_grad()
    def pack(self):
        if self._packed:
            return
        dev, dt = self.device, self.dtype
        if dev.type != "cuda" or dt != torch.bfloat16:
            raise _l.ApexMIError(f"hunyuanvideo15.mi355 needs bf16 weights on a ROCm device (got {dt} on {dev}); "
                                 "there is no CPU fallback")
        dim = self.inner_dim
        mods_w, mods_b = [], []
        for blk in self.transformer_blocks:
            a = blk.attn
            blk._wqkv = torch.empty(3 * dim, dim, device=dev, dtype=dt)
            blk._bqkv = torch.empty(3 * dim, device=dev, dtype=dt)
            _repoint([a.to_q.weight, a.to_k.weight, a.to_v.weight], blk._wqkv)
            _repoint([a.to_q.bias, a.to_k.bias, a.to_v.bias], blk._bqkv)
            blk._wqkv_c = torch.empty(3 * dim, dim, device=dev, dtype=dt)
            blk._bqkv_c = torch.empty(3 * dim, device=dev, dtype=dt)
            _repoint([a.add_q_proj.weight, a.add_k_proj.weight, a.add_v_proj.weight], blk._wqkv_c)
            _repoint([a.add_q_proj.bias, a.add_k_proj.bias, a.add_v_proj.bias], blk._bqkv_c)
            mods_w += [blk.norm1.linear.weight, blk.norm1_context.linear.weight]
            mods_b += [blk.norm1.linear.bias, blk.norm1_context.linear.bias]
        mods_w.append(self.norm_out.linear.weight)
        mods_b.append(self.norm_out.linear.bias)
        total = sum(w.shape[0] for w in mods_w)
        self._mod_w = torch.empty(total, dim, device=dev, dtype=dt)
        self._mod_b = torch.empty(total, device=dev, dtype=dt)
        _repoint(mods_w, self._mod_w)
        _repoint(mods_b, self._mod_b)
        self._mod_total = total
        self._mod_first = min(12 * dim, total)
        for rb in self.context_embedder.token_refiner.refiner_blocks:
            a = rb.attn
            rb._wqkv = torch.empty(3 * dim, dim, device=dev, dtype=dt)
            rb._bqkv = torch.empty(3 * dim, device=dev, dtype=dt)
            _repoint([a.to_q.weight, a.to_k.weight, a.to_v.weight], rb._wqkv)
            _repoint([a.to_q.bias, a.to_k.bias, a.to_v.bias], rb._bqkv)
        self._packed = True
        self._weights_changed()

    @torch.no_grad()
    def _weights_changed(self):
        """Rebuild what is DERIVED from parameter values: the patch embedding as a GEMM operand over
        [S, C*pt*p*p] rows, K padded to a multiple of 64 (zeros).  `weights.load_checkpoint_into` writes parameters in
        place after `pack()` and calls this."""
        if not self._packed:
            return
        dim = self.inner_dim
        w = self.x_embedder.proj.weight.data.reshape(dim, -1)
        kp = (w.shape[1] + 63) // 64 * 64
        self._w_patch = torch.zeros(dim, kp, device=w.device, dtype=w.dtype)
        self._w_patch[:, :w.shape[1]] = w

    def _workspace(self, s_txt: int, s_img: int):
        key = (s_txt, s_img)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev, dim, H = self.device, self.inner_dim, self.config.num_attention_heads
        inner = self.transformer_blocks[0].ff.net[0].proj.weight.shape[0] if len(self.transformer_blocks) else 4 * dim
        S = s_txt + s_img
        skp = (S + 63) // 64 * 64
        bf = dict(device=dev, dtype=self.storage_dtype)     # activations: bf16, or float in the verification mode
        f32 = dict(device=dev, dtype=torch.float32)
        ws = SimpleNamespace(
            X=torch.empty(S, dim, **bf), XN=torch.empty(S, dim, **bf), QKV=torch.empty(S, 3 * dim, **bf),
            Q=torch.empty(1, H, S, 128, **bf), K=torch.empty(1, H, S, 128, **bf), VT=torch.zeros(1, H, 128, skp, **bf),
            ATT=torch.empty(S, dim, **bf), FFH=torch.empty(S, inner, **bf), MOD=torch.empty(1, self._mod_total, **f32),
            TEMB=torch.empty(1, dim, **f32))
        self._ws = {key: ws}
        return ws

    def set_storage_dtype(self, dtype: torch.dtype):
        """torch.bfloat16 (production) or torch.float32: the f32-STORAGE VERIFICATION MODE — the same kernel
        sequence with every activation buffer float and the library's `_f32` entry points.  Weights stay bf16."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError(f"activation storage must be bfloat16 or float32, got {dtype}")
        self.storage_dtype = dtype
        self._ws = {}
        return self

    def _rope_table(self, grid: Tuple[int, int, int], s_txt: int):
        key = (grid, s_txt)
        t = self._rope.get(key)
        if t is None:
            axes = [torch.arange(0, n, dtype=torch.float32) for n in grid]
            ids = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3)
            ids = torch.cat([torch.zeros(s_txt, 3), ids], dim=0).to(self.device)      # condition rows: angle 0
            t = ops.rope_table_axes(ids.contiguous(), self.config.rope_axes_dim, float(self.config.rope_theta))
            ops.rope_pairs(t, trusted=True)     # the compact copy the fused q/k/v epilogue reads, made with the table
            self._rope = {key: t}
        return t

    def _temb(self, te: _TimestepEmbedding, t: torch.Tensor, out=None):
        tp = ops.timestep_embedding(t, 256, scale=1.0)
        h = ops.gemv(te.linear_1.weight, tp, te.linear_1.bias, post="silu")
        return ops.gemv(te.linear_2.weight, h, te.linear_2.bias, out=out)

    def _refine_text(self, text: torch.Tensor, mask: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """HunyuanVideo15TokenRefiner.forward (model.py:434-454) for one sample: text [T1, C] bf16, mask [T1] bool."""
        ce = self.context_embedder
        dim, H = self.inner_dim, self.config.num_attention_heads
        T1, Cc = text.shape
        valid = torch.nonzero(mask, as_tuple=False).flatten()
        n_valid = int(valid.numel())
        # masked mean over tokens = (mask / n) @ text, as a GEMV with text^T as the weight operand (K padded to 8)
        kp = (T1 + 7) // 8 * 8
        xt = torch.zeros(Cc, kp, device=text.device, dtype=torch.bfloat16)
        xt[:, :T1] = text.t().to(torch.bfloat16)    # (an input: bf16-representable in either storage mode)
        wv = torch.zeros(1, kp, device=text.device, dtype=torch.float32)
        wv[0, :T1] = mask.float() / float(n_valid)
        pooled = ops.gemv(xt, wv)
        tte = ce.time_text_embed
        temb = self._temb(tte.timestep_embedder, t)
        hp = ops.gemv(tte.text_embedder.linear_1.weight, pooled, tte.text_embedder.linear_1.bias, post="silu")
        ops.gemv(tte.text_embedder.linear_2.weight, hp, tte.text_embedder.linear_2.bias, out=temb, accum=True)
        x = ops.gemm(text, ce.proj_in.weight, ce.proj_in.bias)
        all_valid = n_valid == T1
        for rb in ce.token_refiner.refiner_blocks:
            gates = ops.gemv(rb.norm_out.linear.weight, temb, rb.norm_out.linear.bias, pre_silu=True)[0]
            n = ops.ln_modulate(x, gamma=rb.norm1.weight, beta=rb.norm1.bias, eps=1e-6)
            qkv = ops.gemm(n, rb._wqkv, rb._bqkv)
            q = qkv[:, :dim].unflatten(-1, (H, 128)).permute(1, 0, 2).unsqueeze(0)          # [1, H, T1, 128] view
            kv = qkv if all_valid else qkv.index_select(0, valid)                           # key-padding mask
            k = kv[:, dim:2 * dim].unflatten(-1, (H, 128)).permute(1, 0, 2).unsqueeze(0)
            v = kv[:, 2 * dim:].unflatten(-1, (H, 128)).permute(1, 0, 2).unsqueeze(0)
            o = ops.attention(q, k, v).permute(0, 2, 1, 3).reshape(T1, dim)
            ops.gemm(o, rb.attn.to_out[0].weight, rb.attn.to_out[0].bias, out=x, epilogue="gate_res",
                     gate=gates[:dim].contiguous(), residual=x)
            n = ops.ln_modulate(x, gamma=rb.norm2.weight, beta=rb.norm2.bias, eps=1e-6)
            h = ops.gemm(n, rb.ff.net[0].proj.weight, rb.ff.net[0].proj.bias, epilogue="silu")
            ops.gemm(h, rb.ff.net[2].weight, rb.ff.net[2].bias, out=x, epilogue="gate_res",
                     gate=gates[dim:].contiguous(), residual=x)
        return x

    @torch.no_grad()
    def _forward_one(self, latent, timestep, text, mask, text2, mask2, image, timestep_r=None):
        cfg = self.config
        dim, H = self.inner_dim, cfg.num_attention_heads
        pt, p = cfg.patch_size_t, cfg.patch_size
        Cin, F_, Hh, Ww = latent.shape
        grid = (F_ // pt, Hh // p, Ww // p)
        s_img = grid[0] * grid[1] * grid[2]
        t = timestep.to(self.storage_dtype).float().reshape(1)   # `t.expand(B).to(latents.dtype)`, engine t2v.py:243-245
        E = self.cond_type_embed.weight

        # ---- condition streams ----
        c1 = ops.add_rowvec(self._refine_text(text, mask, t), E[0].contiguous())
        b2 = self.context_embedder_2
        h = ops.ln_modulate(text2, gamma=b2.norm.weight, beta=b2.norm.bias, eps=1e-5)
        h = ops.gemm(h, b2.linear_1.weight, b2.linear_1.bias, epilogue="gelu_erf")
        h = ops.gemm(h, b2.linear_2.weight, b2.linear_2.bias, epilogue="gelu_erf")
        c2 = ops.add_rowvec(ops.gemm(h, b2.linear_3.weight, b2.linear_3.bias), E[1].contiguous())
        is_t2v = bool((image == 0).all())
        n3 = image.shape[0]
        if is_t2v:
            c3 = E[2].reshape(1, dim).expand(n3, dim).to(self.storage_dtype)   # `projection * 0.0 + cond_type_embed(2)`, :1031-1056
        else:
            ie = self.image_embedder
            h = ops.ln_modulate(image, gamma=ie.norm_in.weight, beta=ie.norm_in.bias, eps=1e-5)
            h = ops.gemm(h, ie.linear_1.weight, ie.linear_1.bias, epilogue="gelu_erf")
            h = ops.gemm(h, ie.linear_2.weight, ie.linear_2.bias)
            c3 = ops.add_rowvec(ops.ln_modulate(h, gamma=ie.norm_out.weight, beta=ie.norm_out.bias, eps=1e-5),
                                E[2].contiguous())
        m1, m2 = mask.bool(), mask2.bool()
        z = torch.zeros
        parts = [c2[m2], c1[m1]] if is_t2v else [c3, c2[m2], c1[m1]]
        tail = ([c3] if is_t2v else []) + [z(int((~m2).sum()), dim, device=c1.device, dtype=c1.dtype),
                                           z(int((~m1).sum()), dim, device=c1.device, dtype=c1.dtype)]
        cond = torch.cat(parts + tail, dim=0)
        s_txt = cond.shape[0]

        # ---- joint buffer [condition | latent] ----
        self.pack()
        S = s_txt + s_img
        ws = self._workspace(s_txt, s_img)
        X, XN, QKV, ATT, FFH = ws.X, ws.XN, ws.QKV, ws.ATT, ws.FFH
        Xt, Xi, XNt, XNi = X[:s_txt], X[s_txt:], XN[:s_txt], XN[s_txt:]
        Xt.copy_(cond)
        # patchify: [C, F, H, W] -> [S, C*pt*p*p] (Conv3d with kernel = stride = patch is a GEMM over patches)
        pat = latent.reshape(Cin, grid[0], pt, grid[1], p, grid[2], p).permute(1, 3, 5, 0, 2, 4, 6).reshape(s_img, -1)
        A = torch.zeros(s_img, self._w_patch.shape[1], device=latent.device, dtype=self.storage_dtype)
        A[:, :pat.shape[1]] = pat
        ops.gemm(A, self._w_patch, self.x_embedder.proj.bias, out=Xi)

        self._temb(self.time_embed.timestep_embedder, t, out=ws.TEMB)
        if timestep_r is not None:
            ter = self.time_embed.timestep_embedder_r
            tr = timestep_r.to(self.storage_dtype).float().reshape(1)   # `timestep_r.expand(B).to(latents.dtype)`, i2v.py:281-286
            hr = ops.gemv(ter.linear_1.weight, ops.timestep_embedding(tr, 256, scale=1.0), ter.linear_1.bias, post="silu")
            ops.gemv(ter.linear_2.weight, hr, ter.linear_2.bias, out=ws.TEMB, accum=True)
        n_first = self._mod_first
        ops.gemv(self._mod_w[:n_first], ws.TEMB, self._mod_b[:n_first], out=ws.MOD[:, :n_first], pre_silu=True)
        mod_ready = None
        if n_first < self._mod_total:   # the other blocks' modulation streams on a side stream under block 0
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                ops.gemv(self._mod_w[n_first:], ws.TEMB, self._mod_b[n_first:], out=ws.MOD[:, n_first:], pre_silu=True)
                mod_ready = torch.cuda.Event()
                mod_ready.record(self._side)
        rope = self._rope_table(grid, s_txt)

        q_in, k_in, v_in = QKV[:, :dim], QKV[:, dim:2 * dim], QKV[:, 2 * dim:]
        att_v = ATT.unflatten(-1, (H, 128)).unsqueeze(0)
        # fused q/k/v preparation (apexmi_gemm_bf16_grouped_qkv) where the launch allows it: bf16 storage, 8-aligned streams,
        # >= 1024 rows; `fuse_qkv = False` keeps the [S, 3 dim] projection as a storage point (tests/stage_parity.py)
        fuse = (getattr(self, "fuse_qkv", True) and getattr(self, "storage_dtype", torch.bfloat16) == torch.bfloat16
                and len(self.transformer_blocks) > 0 and tuple(rope.shape) == (2, S, 128)
                and ops.qkv_fusable([XNi, XNt], [self.transformer_blocks[0]._wqkv, self.transformer_blocks[0]._wqkv_c], [s_txt, 0], H))
        for i, blk in enumerate(self.transformer_blocks):
            if i == 1 and mod_ready is not None:
                torch.cuda.current_stream().wait_event(mod_ready)
                mod_ready = None
            a = blk.attn
            base = i * 12 * dim
            mi = lambda j: ws.MOD[0, base + j * dim: base + (j + 1) * dim]              # noqa: E731  latent stream
            mt = lambda j: ws.MOD[0, base + (6 + j) * dim: base + (7 + j) * dim]        # noqa: E731  condition stream
            # AdaLayerNormZero chunk order: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
            ops.ln_modulate(X, mi(1), mi(0), out=XN, split=s_txt, scale2=mt(1), shift2=mt(0))
            if fuse:
                # q/k norm + RoPE + [H, S, D] layout and V^T leave the QKV GEMM's epilogue (bit-identical to the two passes)
                ops.gemm_grouped_qkv([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c], [None, None], "bias",
                                     [1, 1], [a.norm_q.weight, a.norm_added_q.weight], [a.norm_k.weight, a.norm_added_k.weight],
                                     [s_txt, 0], H, 1e-6, rope, ws.Q[0], ws.K[0], ws.VT[0])
            else:
                ops.gemm_grouped([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c],
                                 [QKV[s_txt:], QKV[:s_txt]])
                ops.qkv_prepare(q_in, k_in, v_in, H, ws.Q[0], ws.K[0], ws.VT[0], wq=a.norm_q.weight,
                                wk=a.norm_k.weight, wq2=a.norm_added_q.weight, wk2=a.norm_added_k.weight,
                                split=s_txt, eps=1e-6, rope=rope, rope_mode=_l.ROPE_INTERLEAVED)
            ops.attention_prepared(ws.Q, ws.K, ws.VT, att_v, S)
            ops.gemm_grouped([ATT[s_txt:], ATT[:s_txt]], [a.to_out[0].weight, a.to_add_out.weight],
                             [a.to_out[0].bias, a.to_add_out.bias], [Xi, Xt], epilogue="gate_res",
                             gate_list=[mi(2), mt(2)], residual_list=[Xi, Xt])
            ops.ln_modulate(X, mi(4), mi(3), out=XN, split=s_txt, scale2=mt(4), shift2=mt(3))
            fi, ft = blk.ff.net, blk.ff_context.net
            ops.gemm_grouped([XNi, XNt], [fi[0].proj.weight, ft[0].proj.weight], [fi[0].proj.bias, ft[0].proj.bias],
                             [FFH[s_txt:], FFH[:s_txt]], epilogue="gelu")
            ops.gemm_grouped([FFH[s_txt:], FFH[:s_txt]], [fi[2].weight, ft[2].weight], [fi[2].bias, ft[2].bias],
                             [Xi, Xt], epilogue="gate_res", gate_list=[mi(5), mt(5)], residual_list=[Xi, Xt])
        if mod_ready is not None:
            torch.cuda.current_stream().wait_event(mod_ready)
        o = len(self.transformer_blocks) * 12 * dim
        ops.ln_modulate(Xi, ws.MOD[0, o:o + dim], ws.MOD[0, o + dim:o + 2 * dim], out=XNi)   # scale first, then shift
        y = ops.gemm(XNi, self.proj_out.weight, self.proj_out.bias)
        y = y.reshape(grid[0], grid[1], grid[2], -1, pt, p, p).permute(3, 0, 4, 1, 5, 2, 6)
        return y.reshape(-1, grid[0] * pt, grid[1] * p, grid[2] * p)

    @ops.on_model_device
    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                encoder_attention_mask: torch.Tensor, timestep_r: Optional[torch.Tensor] = None,
                encoder_hidden_states_2: Optional[torch.Tensor] = None,
                encoder_attention_mask_2: Optional[torch.Tensor] = None, image_embeds: Optional[torch.Tensor] = None,
                attention_kwargs=None, rope_on_cpu=None, return_dict: bool = True):
        if timestep_r is not None and not self.config.use_meanflow:
            raise ValueError("hunyuanvideo15.mi355: timestep_r given but the model was built with use_meanflow=False")
        self.pack()
        bf = self.storage_dtype
        outs = []
        for b in range(hidden_states.shape[0]):
            outs.append(self._forward_one(
                hidden_states[b].to(bf).contiguous(), timestep[b:b + 1], encoder_hidden_states[b].to(bf).contiguous(),
                encoder_attention_mask[b], encoder_hidden_states_2[b].to(bf).contiguous(), encoder_attention_mask_2[b],
                image_embeds[b].to(bf).contiguous(), None if timestep_r is None else timestep_r[b:b + 1]))
        out = torch.stack(outs, dim=0).to(hidden_states.dtype)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
