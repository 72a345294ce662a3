"""WanTransformer3DModel on the MI355X HIP ops — drop-in for registry key "wan.base" (text-to-video).

Mirrors the reference class (apps/api/src/transformer/wan/base/model.py:1336-1891): same config, same
state-dict keys (blocks.N.attn1.to_q.weight, blocks.N.scale_shift_table, condition_embedder.*), same
`forward(hidden_states [B,16,F,H,W], timestep [B], encoder_hidden_states [B,512,4096], return_dict=False)`.
Memory knobs of the reference that exist for 8-24 GB GPUs (`set_chunking_profile`, `rope_on_cpu`,
`enable_easy_cache`) are accepted and ignored: one 75 600-token activation is 0.77 GB of 288 GB.

Per block (reference model.py:1101-1333 + attention.py:305-413), as libapex_mi355.so launches:
  ln_modulate -> fused QKV gemm -> RMSNorm over all 5120 channels on q and k (in place)
  -> qkv_prepare (RoPE + attention layout + V^T) -> attention -> out-proj gemm (gate * y + residual)
  -> affine LayerNorm -> q gemm + RMSNorm | text k,v gemm + RMSNorm -> prepare x2 -> cross attention
  -> out-proj gemm (+ residual) -> ln_modulate -> FFN-up gemm (+GELU) -> FFN-down gemm (gate + residual)
Image-conditioning branches (`added_kv_proj_dim`, IP adapter) are outside the text-to-video scope and
raise NotImplementedError.
"""
from __future__ import annotations

import os

import contextlib
import math
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn


from .lora import LoraAdapterMixin  # noqa: E402
from . import lib as _l
from . import ops
from .flux import _Config, _Linear, _Norm, _FF, _repoint


class _WanAttn(nn.Module):
    def __init__(self, dim: int, heads: int, **kw):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = _Linear(dim, dim, **kw), _Linear(dim, dim, **kw), _Linear(dim, dim, **kw)
        self.to_out = nn.ModuleList([_Linear(dim, dim, **kw), nn.Identity()])
        self.norm_q, self.norm_k = _Norm(dim, **kw), _Norm(dim, **kw)


class _AffineNorm(nn.Module):
    def __init__(self, dim: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim, device=device, dtype=dtype), requires_grad=False)


class _WanBlock(nn.Module):
    def __init__(self, dim: int, ffn_dim: int, heads: int, cross_attn_norm: bool, **kw):
        super().__init__()
        self.attn1 = _WanAttn(dim, heads, **kw)
        self.attn2 = _WanAttn(dim, heads, **kw)
        self.norm2 = _AffineNorm(dim, **kw) if cross_attn_norm else nn.Identity()
        self.ffn = _FF(dim, ffn_dim, **kw)
        self.scale_shift_table = nn.Parameter(torch.empty(1, 6, dim, **kw), requires_grad=False)


class _TimeEmb(nn.Module):
    def __init__(self, a: int, b: int, **kw):
        super().__init__()
        self.linear_1, self.linear_2 = _Linear(a, b, **kw), _Linear(b, b, **kw)


class _Cond(nn.Module):
    def __init__(self, dim: int, freq_dim: int, proj_dim: int, text_dim: int, **kw):
        super().__init__()
        self.time_embedder = _TimeEmb(freq_dim, dim, **kw)
        self.time_proj = _Linear(dim, proj_dim, **kw)
        self.text_embedder = _TimeEmb(text_dim, dim, **kw)


class _Conv3dParams(nn.Module):
    def __init__(self, cin: int, cout: int, k, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, *k, **kw), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout, **kw), requires_grad=False)


class WanTransformer3DModel(LoraAdapterMixin, nn.Module):
    _converter_base = "wan.base"      # which key-converter table original-format weight files / LoRAs go through (converters.py)
    _no_split_modules = ["_WanBlock"]

    def __init__(self, patch_size: Tuple[int, int, int] = (1, 2, 2), num_attention_heads: int = 40,
                 attention_head_dim: int = 128, in_channels: int = 16, out_channels: int = 16,
                 text_dim: int = 4096, freq_dim: int = 256, ffn_dim: int = 13824, num_layers: int = 40,
                 cross_attn_norm: bool = True, qk_norm: Optional[str] = "rms_norm_across_heads",
                 eps: float = 1e-6, image_dim: Optional[int] = None, added_kv_proj_dim: Optional[int] = None,
                 rope_max_seq_len: int = 1024, pos_embed_seq_len: Optional[int] = None, ip_adapter: bool = False,
                 use_enhance: bool = False, ffn_chunk_size: Optional[int] = None, ffn_chunk_dim: int = 1,
                 device=None, dtype=torch.bfloat16):
        super().__init__()
        if attention_head_dim != 128:
            raise _l.ApexMIError("wan.mi355: attention_head_dim must be 128 (MFMA attention tile)")
        if image_dim is not None or added_kv_proj_dim is not None or ip_adapter or use_enhance:
            raise NotImplementedError("wan.mi355: image conditioning / IP adapter / enhance are outside the "
                                      "text-to-video hot-path scope")
        if qk_norm != "rms_norm_across_heads":
            raise NotImplementedError(f"wan.mi355: qk_norm={qk_norm!r}")
        self.config = _Config(patch_size=tuple(patch_size), num_attention_heads=num_attention_heads,
                              attention_head_dim=attention_head_dim, in_channels=in_channels,
                              out_channels=out_channels, text_dim=text_dim, freq_dim=freq_dim, ffn_dim=ffn_dim,
                              num_layers=num_layers, cross_attn_norm=cross_attn_norm, qk_norm=qk_norm, eps=eps,
                              image_dim=image_dim, added_kv_proj_dim=added_kv_proj_dim,
                              rope_max_seq_len=rope_max_seq_len)
        kw = dict(device=device, dtype=dtype)
        self.inner_dim = dim = num_attention_heads * attention_head_dim
        self.patch_embedding = _Conv3dParams(in_channels, dim, tuple(patch_size), **kw)
        self.condition_embedder = _Cond(dim, freq_dim, dim * 6, text_dim, **kw)
        self.blocks = nn.ModuleList([_WanBlock(dim, ffn_dim, num_attention_heads, cross_attn_norm, **kw)
                                     for _ in range(num_layers)])
        self.proj_out = _Linear(dim, out_channels * math.prod(patch_size), **kw)
        self.scale_shift_table = nn.Parameter(torch.empty(1, 2, dim, **kw), requires_grad=False)
        self._packed = False
        self._ws: Dict[Any, Any] = {}
        self.storage_dtype = torch.bfloat16
        self.fuse_qkv = os.environ.get("APEX_FUSE_QKV", "1") != "0"     # see _forward_one
        self._rope: Dict[Any, torch.Tensor] = {}

    # ---- reference-compatible plumbing -------------------------------------------------------
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
        self._ws = {}
        return self

    @property
    def dtype(self):
        return self.proj_out.weight.dtype

    @property
    def device(self):
        return self.proj_out.weight.device

    @contextlib.contextmanager
    def cache_context(self, name: str):
        yield

    def set_chunking_profile(self, *a, **k):  # memory knobs of the reference: no-ops on 288 GB
        return None

    def set_chunk_feed_forward(self, *a, **k):
        return None

    def enable_easy_cache(self, num_steps: int, thresh: float, ret_steps: int = 10, should_reset_global_cache: bool = True):
        """The reference's EasyCache switch (R/src/transformer/wan/base/model.py:1645-1672): from now on `forward` serves
        conditional / unconditional call pairs from the cache while the accumulated predicted change stays under `thresh`
        (easycache.py).  `should_reset_global_cache=False` keeps the running state (the reference's state is global)."""
        from .easycache import EasyCache
        if should_reset_global_cache or getattr(self, "_easy_cache", None) is None:
            self._easy_cache = EasyCache(num_steps, thresh, ret_steps)
        else:
            ec = self._easy_cache
            ec.num_steps, ec.thresh, ec.ret_steps = int(num_steps) * 2, float(thresh), int(ret_steps) * 2
        return self

    def share_easy_cache_state(self, other):
        """Continue `other`'s EasyCache state on this model (the reference's state is module-global, so the low-noise expert
        picks up where the high-noise one stopped: R/src/engine/wan/shared/__init__.py:435-444 enables it with
        `should_reset_global_cache=False`).  Follow with `enable_easy_cache(..., should_reset_global_cache=False)`."""
        self._easy_cache = getattr(other, "_easy_cache", None)
        return self

    def disable_easy_cache(self):
        self._easy_cache = None
        return self

    def _apply(self, fn, *a, **k):
        self._packed = False
        self._ws = {}
        self._rope = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = False
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def init_synthetic(self, seed: int = 0, std: float = 0.02):
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith(("norm_q.weight", "norm_k.weight", "norm2.weight")):
                p.data.fill_(1.0)
            elif name.endswith("scale_shift_table"):
                p.data.copy_((torch.randn(p.shape, generator=g, device=p.device) / p.shape[-1] ** 0.5).to(p.dtype))
            elif name.endswith(".bias"):
                p.data.copy_((torch.randn(p.shape, generator=g, device=p.device) * 0.01).to(p.dtype))
            else:
                flat = p.data.view(-1)
                step = 1 << 26
                for i in range(0, flat.numel(), step):
                    n = min(step, flat.numel() - i)
                    flat[i:i + n] = (torch.randn(n, generator=g, device=p.device) * std).to(p.dtype)
        self._packed = False
        return self

    @torch.no_grad()
    def pack(self):
        if self._packed:
            return
        if getattr(self, "_fp8_bytes", 0):
            raise _l.ApexMIError("wan.mi355: the block weights are resident fp8 (keep_fp8 load) and their bf16 storage is gone; "
                                 "moving / re-packing such a model is not supported — construct and load again")
        dev, dt = self.device, self.dtype
        if dev.type != "cuda" or dt != torch.bfloat16:
            raise _l.ApexMIError(f"wan.mi355 needs bf16 weights on a ROCm device (got {dt} on {dev}); "
                                 "there is no CPU fallback")
        dim = self.inner_dim
        L = len(self.blocks)
        for blk in self.blocks:
            a1, a2 = blk.attn1, blk.attn2
            blk._wqkv = torch.empty(3 * dim, dim, device=dev, dtype=dt)
            blk._bqkv = torch.empty(3 * dim, device=dev, dtype=dt)
            _repoint([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], blk._wqkv)
            _repoint([a1.to_q.bias, a1.to_k.bias, a1.to_v.bias], blk._bqkv)
            blk._wkv2 = torch.empty(2 * dim, dim, device=dev, dtype=dt)
            blk._bkv2 = torch.empty(2 * dim, device=dev, dtype=dt)
            _repoint([a2.to_k.weight, a2.to_v.weight], blk._wkv2)
            _repoint([a2.to_k.bias, a2.to_v.bias], blk._bkv2)
        self._ones = torch.ones(dim, device=dev, dtype=torch.float32)
        self._packed = True
        self._weights_changed()

    # ---- fp8-scaled expert weights resident in HBM -------------------------------------------------------------
    @staticmethod
    def _fp8_resident_key(key: str) -> bool:
        """Which fp8-scaled checkpoint tensors `weights.load_checkpoint_into(keep_fp8=True)` keeps as float8 + scale: the
        Linear weights of the blocks (attention projections and FFN: 99 % of an expert's bytes).  The embedders feed GEMVs and the
        modulation tables are f32 copies: those stay dequantised."""
        return key.startswith("blocks.") and key.endswith(".weight") and ".norm" not in key

    @torch.no_grad()
    def _fp8_adopt(self):
        """After a keep_fp8 load: fused projections become fused `ops.Fp8Weight`s (per-row scales), every adopted parameter's bf16
        storage is released.  The blocks' GEMMs then dequantise per call (`ops._bf16_weight`)."""
        self.pack()
        dev, dt = self.device, self.dtype

        names = {id(p): k[:-len(".weight")] for k, p in self.named_parameters() if k.endswith(".weight")}
        for p in self.parameters():            # every record knows which Linear its rows are (run-time LoRA addresses them by name)
            f = getattr(p, "_fp8", None)
            if f is not None:
                f.parts = [(names[id(p)], 0, int(f.shape[0]))]
                p._fp8_shape = tuple(f.shape)

        def fused(parts):
            f8 = [getattr(p, "_fp8", None) for p in parts]
            if all(f is None for f in f8):
                return None
            if any(f is None for f in f8):
                raise _l.ApexMIError("wan.mi355 keep_fp8: a fused projection mixes fp8-scaled and plain weights")
            return ops.Fp8Weight.cat(f8)
        n = 0
        self._fp8_records = {}                 # module path -> the record holding its rows
        for blk in self.blocks:
            a1, a2 = blk.attn1, blk.attn2
            for name, parts in (("_wqkv", [a1.to_q.weight, a1.to_k.weight, a1.to_v.weight]), ("_wkv2", [a2.to_k.weight, a2.to_v.weight])):
                f = fused(parts)
                if f is not None:
                    setattr(blk, name, f)
                    for p in parts:
                        del p._fp8           # the fused record is the one that is read; the parts' views of the bf16 buffer go
                        p.data = torch.empty(0, device=dev, dtype=dt)
                    n += f.nbytes()
                    self._fp8_records.update({m: f for m, _, _ in f.parts})
            for p in (a1.to_out[0].weight, a2.to_q.weight, a2.to_out[0].weight, blk.ffn.net[0].proj.weight, blk.ffn.net[2].weight):
                if getattr(p, "_fp8", None) is not None:
                    p.data = torch.empty(0, device=dev, dtype=dt)
                    n += p._fp8.nbytes()
                    self._fp8_records[p._fp8.parts[0][0]] = p._fp8
        self._fp8_bytes = n
        torch.cuda.empty_cache()
        return self

    # ---- LoRA on resident-fp8 weights: applied at RUN TIME, as the reference does (R/src/lora/manager.py:454-606 around
    # FPScaledLinear) — there is no bf16 weight to merge into.  Modules whose weights are ordinary bf16 parameters still merge.
    @torch.no_grad()
    def _lora_remerge(self, modules):
        recs = getattr(self, "_fp8_records", None) or {}
        modules = set(modules)
        resident = {m for m in modules if m in recs}
        if resident:
            for m in resident:
                if any("bias" in d[m] for d in self._lora_adapters.values() if m in d):
                    raise NotImplementedError(f"wan.mi355: a lora_B.bias on the resident-fp8 Linear '{m}' is not supported")
            touched = {id(recs[m]): recs[m] for m in resident}
            pad = 0
            for rec in touched.values():
                act = [(m, d[m]["A"], d[m]["B"], self._lora_scales[n]) for m, _, _ in rec.parts
                       for n, d in self._lora_adapters.items()
                       if m in d and self._lora_scales[n] != 0.0 and self._lora_enabled]
                rec.set_lora(act)
            for rec in {id(r): r for r in recs.values()}.values():
                pad = max(pad, 0 if rec.lora_A is None else int(rec.lora_A.shape[0]))
            if pad != getattr(self, "_lora_pad", 0):
                self._lora_pad = pad         # the activation buffers grow by this many columns (see _workspace)
                self._ws = {}
        super()._lora_remerge(modules - resident)

    def state_dict(self, *a, **k):
        if getattr(self, "_fp8_bytes", 0):
            raise _l.ApexMIError("wan.mi355: this model was loaded with keep_fp8=True — its block weights live as float8 + scale "
                                 "records (inference only); it has no bf16 state dict to save.  Load without keep_fp8 to serialise.")
        return super().state_dict(*a, **k)

    @torch.no_grad()
    def _weights_changed(self):
        """Rebuild what is DERIVED from parameter values (f32 copies of the modulation tables, [L, 6*dim] and
        [2*dim]).  `weights.load_checkpoint_into` writes parameters in place after `pack()` and calls this."""
        if not self._packed:
            return
        dim, dev = self.inner_dim, self.device
        self._sst = torch.stack([b.scale_shift_table.data.float().reshape(-1) for b in self.blocks]) \
            if len(self.blocks) else torch.empty(0, 6 * dim, device=dev)
        self._sst_out = self.scale_shift_table.data.float().reshape(-1).contiguous()
        # patch_embedding as a GEMM needs K = C pt ph pw in multiples of 64: 64 for the 16-channel text-to-video experts; the
        # image-to-video experts take 36 channels (K = 144) -> the weight gets zero columns up to 192 (exact zeros in the f32 sums)
        w = self.patch_embedding.weight.data.reshape(dim, -1)
        kp = (w.shape[1] + 63) // 64 * 64
        self._pe_w = w if kp == w.shape[1] else torch.cat([w, w.new_zeros(dim, kp - w.shape[1])], dim=1).contiguous()

    def _workspace(self, S: int, s_txt: int):
        key = (S, s_txt)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev, dim, H = self.device, self.inner_dim, self.config.num_attention_heads
        ffn = self.config.ffn_dim
        skp = (S + 63) // 64 * 64
        tkp = (s_txt + 63) // 64 * 64
        bf = dict(device=dev, dtype=self.storage_dtype)     # activation buffers
        f32 = dict(device=dev, dtype=torch.float32)
        # run-time LoRA on resident-fp8 weights (ops._gemm_fp8_lora): the buffers a Linear READS carry `pad` spare columns behind
        # their rows for the adapters' rank-space activations; pad = 0 (no such adapter) is the plain layout
        pad = int(getattr(self, "_lora_pad", 0))
        XNf, ATTf = torch.empty(S, dim + pad, **bf), torch.empty(S, dim + pad, **bf)
        FFHf, CTXf = torch.empty(S, ffn + pad, **bf), torch.empty(s_txt, dim + pad, **bf)
        ws = SimpleNamespace(
            X=torch.empty(S, dim, **bf), XN=XNf[:, :dim], QKV=torch.empty(S, 3 * dim, **bf),
            Q=torch.empty(1, H, S, 128, **bf), K=torch.empty(1, H, S, 128, **bf),
            VT=torch.zeros(1, H, 128, skp, **bf), ATT=ATTf[:, :dim], FFH=FFHf[:, :ffn],
            CTX=CTXf[:, :dim], CTXH=torch.empty(s_txt, dim, **bf), XNf=XNf, ATTf=ATTf, FFHf=FFHf, CTXf=CTXf,
            KV2=torch.empty(s_txt, 2 * dim, **bf), K2=torch.empty(1, H, s_txt, 128, **bf),
            VT2=torch.zeros(1, H, 128, tkp, **bf),
            MOD=torch.empty(max(len(self.blocks), 1), 6 * dim, **f32), MOD2=torch.empty(1, 2 * dim, **f32),
            TEMB=torch.empty(1, dim, **f32), TPROJ=torch.empty(1, 6 * dim, **f32))
        self._ws = {key: ws}
        return ws

    def _rope_table(self, grid):
        t = self._rope.get(grid)
        if t is None:
            f, h, w = grid
            dev = self.device
            ids = torch.stack(torch.meshgrid(torch.arange(f, device=dev), torch.arange(h, device=dev),
                                             torch.arange(w, device=dev), indexing="ij"), dim=-1)
            ids = ids.reshape(-1, 3).float().contiguous()
            hd = self.config.attention_head_dim
            hw_dim = 2 * (hd // 6)
            t = ops.rope_table_axes(ids, (hd - 2 * hw_dim, hw_dim, hw_dim), 10000.0)
            self._rope = {grid: t}
        return t

    @torch.no_grad()
    def _forward_one(self, hidden_states, timestep, text):
        cfg = self.config
        dim, H = self.inner_dim, cfg.num_attention_heads
        C, T, Hh, Ww = hidden_states.shape
        pt, ph, pw = cfg.patch_size
        grid = (T // pt, Hh // ph, Ww // pw)
        S = grid[0] * grid[1] * grid[2]
        s_txt = text.shape[0]
        ws = self._workspace(S, s_txt)
        X, XN, QKV, ATT, FFH = ws.X, ws.XN, ws.QKV, ws.ATT, ws.FFH
        eps = cfg.eps

        # patchify (layout only) + patch_embedding as a K = C*pt*ph*pw GEMM
        tok = hidden_states.to(self.storage_dtype).reshape(C, grid[0], pt, grid[1], ph, grid[2], pw) \
            .permute(1, 3, 5, 0, 2, 4, 6).reshape(S, C * pt * ph * pw).contiguous()
        pe = self.patch_embedding
        if self._pe_w.shape[1] != tok.shape[1]:
            tok = torch.cat([tok, tok.new_zeros(S, self._pe_w.shape[1] - tok.shape[1])], dim=1)
        ops.gemm(tok, self._pe_w, pe.bias, out=X)

        ce = self.condition_embedder
        tp = ops.timestep_embedding(timestep.float().reshape(1), cfg.freq_dim)
        h = ops.gemv(ce.time_embedder.linear_1.weight, tp, ce.time_embedder.linear_1.bias, post="silu")
        ops.gemv(ce.time_embedder.linear_2.weight, h, ce.time_embedder.linear_2.bias, out=ws.TEMB)
        ops.gemv(ce.time_proj.weight, ws.TEMB, ce.time_proj.bias, out=ws.TPROJ, pre_silu=True)
        ops.gemm(text, ce.text_embedder.linear_1.weight, ce.text_embedder.linear_1.bias, out=ws.CTXH,
                 epilogue="gelu")
        ops.gemm(ws.CTXH, ce.text_embedder.linear_2.weight, ce.text_embedder.linear_2.bias, out=ws.CTX)
        if len(self.blocks):
            ops.add_bcast(self._sst, ws.TPROJ[0], out=ws.MOD)    # scale_shift_table + temb.float()
        ops.add_bcast(self._sst_out.reshape(1, -1), torch.cat([ws.TEMB[0], ws.TEMB[0]]), out=ws.MOD2)
        rope = self._rope_table(grid)

        q_in, k_in, v_in = QKV[:, :dim], QKV[:, dim:2 * dim], QKV[:, 2 * dim:]
        att_v = ATT.unflatten(-1, (H, 128)).unsqueeze(0)
        # q / k across-heads RMSNorm + RoPE + layout (+ V^T) as one pass (apexmi_qk_rms_rope_rows, bit-identical to the three it
        # replaces); `fuse_qkv = False` keeps the normalised [S, dim] q / k as storage points (tests/stage_parity.py)
        fuse = self.fuse_qkv and dim in (3072, 5120)
        for i, blk in enumerate(self.blocks):
            a1, a2 = blk.attn1, blk.attn2
            m = lambda j: ws.MOD[i, j * dim:(j + 1) * dim]  # noqa: E731 shift, scale, gate, c_shift, c_scale, c_gate
            # 1. self attention
            ops.ln_modulate(X, m(1), m(0), out=XN, eps=eps)
            ops.gemm(XN, blk._wqkv, blk._bqkv, out=QKV, lora_buf=ws.XNf)
            if fuse:       # across-heads RMSNorm of q and k + RoPE + layout + V^T in one read of the projection
                ops.qk_rms_rope_rows(q_in, k_in, v_in, H, ws.Q[0], ws.K[0], ws.VT[0], wq=a1.norm_q.weight, wk=a1.norm_k.weight,
                                     eps=eps, rope=rope, rope_mode=_l.ROPE_INTERLEAVED)
            else:
                ops.ln_modulate(q_in, gamma=a1.norm_q.weight, out=q_in, eps=eps, rms=True)
                ops.ln_modulate(k_in, gamma=a1.norm_k.weight, out=k_in, eps=eps, rms=True)
                ops.qkv_prepare(q_in, k_in, v_in, H, ws.Q[0], ws.K[0], ws.VT[0], rope=rope,
                                rope_mode=_l.ROPE_INTERLEAVED)
            ops.attention_prepared(ws.Q, ws.K, ws.VT, att_v, S)
            ops.gemm(ATT, a1.to_out[0].weight, a1.to_out[0].bias, out=X, epilogue="gate_res", gate=m(2),
                     residual=X, lora_buf=ws.ATTf)
            # 2. cross attention over the text tokens (no RoPE, ungated residual)
            if isinstance(blk.norm2, _AffineNorm):
                ops.ln_modulate(X, gamma=blk.norm2.weight, beta=blk.norm2.bias, out=XN, eps=eps)
                src = XN
            else:
                src = X
            ops.gemm(src, a2.to_q.weight, a2.to_q.bias, out=q_in, lora_buf=ws.XNf if src is XN else None)
            if fuse:
                ops.qk_rms_rope_rows(q_in, None, None, H, ws.Q[0], None, None, wq=a2.norm_q.weight, eps=eps)
            else:
                ops.ln_modulate(q_in, gamma=a2.norm_q.weight, out=q_in, eps=eps, rms=True)
            ops.gemm(ws.CTX, blk._wkv2, blk._bkv2, out=ws.KV2, lora_buf=ws.CTXf)
            ops.ln_modulate(ws.KV2[:, :dim], gamma=a2.norm_k.weight, out=ws.KV2[:, :dim], eps=eps, rms=True)
            if not fuse:
                ops.qkv_prepare(q_in, None, None, H, ws.Q[0], None, None)
            ops.qkv_prepare(ws.KV2[:, :dim], None, ws.KV2[:, dim:], H, ws.K2[0], None, ws.VT2[0])
            ops.attention_prepared(ws.Q, ws.K2, ws.VT2, att_v, s_txt)
            ops.gemm(ATT, a2.to_out[0].weight, a2.to_out[0].bias, out=X, epilogue="gate_res", gate=self._ones,
                     residual=X, lora_buf=ws.ATTf)
            # 3. feed-forward
            ops.ln_modulate(X, m(4), m(3), out=XN, eps=eps)
            ops.gemm(XN, blk.ffn.net[0].proj.weight, blk.ffn.net[0].proj.bias, out=FFH, epilogue="gelu", lora_buf=ws.XNf)
            ops.gemm(FFH, blk.ffn.net[2].weight, blk.ffn.net[2].bias, out=X, epilogue="gate_res", gate=m(5),
                     residual=X, lora_buf=ws.FFHf)

        # (scale_shift_table + temb).chunk(2): shift first, then scale (model.py:1849-1856)
        ops.ln_modulate(X, ws.MOD2[0, dim:], ws.MOD2[0, :dim], out=XN, eps=eps)
        out = ops.gemm(XN, self.proj_out.weight, self.proj_out.bias)
        out = out.reshape(grid[0], grid[1], grid[2], pt, ph, pw, -1).permute(6, 0, 3, 1, 4, 2, 5)
        return out.reshape(-1, T, Hh, Ww)

    @ops.on_model_device
    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, timestep: torch.Tensor = None,
                encoder_hidden_states: torch.Tensor = None, encoder_hidden_states_image=None,
                ip_image_hidden_states=None, return_dict: bool = True, attention_kwargs=None,
                enhance_kwargs=None, rope_on_cpu=None):
        if encoder_hidden_states_image is not None or ip_image_hidden_states is not None:
            raise NotImplementedError("wan.mi355: image conditioning is outside the text-to-video scope")
        if timestep.ndim != 1:
            raise NotImplementedError("wan.mi355: per-token timesteps are not supported")
        self.pack()
        enc = encoder_hidden_states.to(self.storage_dtype)

        def run():
            return torch.stack([self._forward_one(hidden_states[b], timestep[b:b + 1], enc[b].contiguous())
                                for b in range(hidden_states.shape[0])], dim=0)
        ec = getattr(self, "_easy_cache", None)
        if ec is not None:       # EasyCache: this call may be served from the cache; float32 out, as the reference returns
            out = ec(hidden_states, self.config.out_channels, run)
        else:
            out = run().to(hidden_states.dtype)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
