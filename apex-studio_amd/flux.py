"""FluxTransformer2DModel on the MI355X HIP ops — drop-in for registry key "flux.base".

Mirrors the interface of the reference class (apps/api/src/transformer/flux/base/model.py:362-657):
same constructor config, same state-dict keys (transformer_blocks.N.attn.to_q.weight, ...), same
keyword forward returning a 1-tuple when return_dict=False, `.config` with attribute/`.get`
access, `.dtype`/`.device`, `cache_context(name)`, `from_config`.  Underneath, a denoise step is
a fixed sequence of libapex_mi355.so kernels over one joint [S_txt + S_img, dim] residual buffer
(text rows first), so none of the reference's torch.cat / split / permute copies exist:

  per step      : timestep/guidance/pooled embeds (gemv) -> ONE batched gemv for every block's AdaLN
                  modulation vectors -> rope table
  double block  : ln_modulate x2 -> QKV gemm x2 (img / txt weights, one joint output)
                  -> qkv_prepare (per-head RMSNorm + RoPE + V^T) -> attention
                  -> out-proj gemm x2 with fused gate*y + residual -> ln_modulate x2
                  -> MLP-up gemm (+GELU) x2 -> MLP-down gemm x2 with fused gate + residual
  single block  : ln_modulate -> QKV gemm + MLP gemm(+GELU) into the concat buffer -> qkv_prepare
                  -> attention (writes into the concat buffer) -> 15360->3072 gemm with fused
                  gate + residual

Weights are packed once after loading (fused QKV, all modulation projections in one matrix) and the
original nn.Parameters are re-pointed at views of the packed storage, so state_dict()/load_state_dict
keep working and memory is not doubled.
"""
from __future__ import annotations

import contextlib
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn


from .lora import LoraAdapterMixin  # noqa: E402
from . import lib as _l
from .schedule import ModulationSchedule, ScheduleRegistry
from . import ops


class _Config(SimpleNamespace):
    def get(self, key, default=None):
        return getattr(self, key, default)

    def __getitem__(self, key):
        return getattr(self, key)

    def __contains__(self, key):
        return hasattr(self, key)


class _Linear(nn.Module):
    """Parameter holder with nn.Linear's names/shapes (weight [out,in], bias [out])."""

    def __init__(self, in_features: int, out_features: int, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype),
                                   requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype), requires_grad=False)


class _Norm(nn.Module):
    def __init__(self, dim: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype), requires_grad=False)


class _AdaNorm(nn.Module):
    def __init__(self, dim: int, mult: int, cond_dim: Optional[int] = None, **kw):
        super().__init__()
        self.linear = _Linear(cond_dim or dim, mult * dim, **kw)


class _FF(nn.Module):
    def __init__(self, dim: int, inner: int, **kw):
        super().__init__()
        proj = nn.Module()
        proj.proj = _Linear(dim, inner, **kw)
        self.net = nn.ModuleList([proj, nn.Identity(), _Linear(inner, dim, **kw)])


class _Attn(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, joint: bool, pre_only: bool, **kw):
        super().__init__()
        inner = heads * head_dim
        self.heads, self.head_dim = heads, head_dim
        self.processor = None          # an _IPAdapterProcessor on the double blocks once IP adapters are loaded
        self.norm_q, self.norm_k = _Norm(head_dim, **kw), _Norm(head_dim, **kw)
        self.to_q, self.to_k, self.to_v = (_Linear(dim, inner, **kw), _Linear(dim, inner, **kw),
                                           _Linear(dim, inner, **kw))
        if not pre_only:
            self.to_out = nn.ModuleList([_Linear(inner, dim, **kw), nn.Identity()])
        if joint:
            self.norm_added_q, self.norm_added_k = _Norm(head_dim, **kw), _Norm(head_dim, **kw)
            self.add_q_proj, self.add_k_proj, self.add_v_proj = (
                _Linear(dim, inner, **kw), _Linear(dim, inner, **kw), _Linear(dim, inner, **kw))
            self.to_add_out = _Linear(inner, dim, **kw)


class _IPAdapterProcessor(nn.Module):
    """Parameters of `FluxIPAdapterAttnProcessor` (R/src/transformer/flux/base/attention.py:115-173) on a double block's attention:
    per adapter one key and one value projection of the image-prompt tokens (`...attn.processor.to_k_ip.M.weight`) and a scale."""

    def __init__(self, hidden_size: int, cross_attention_dim: int, num_tokens=(4,), scale=1.0, **kw):
        super().__init__()
        num_tokens = list(num_tokens) if isinstance(num_tokens, (tuple, list)) else [num_tokens]
        self.scale = list(scale) if isinstance(scale, (list, tuple)) else [float(scale)] * len(num_tokens)
        if len(self.scale) != len(num_tokens):
            raise ValueError("`scale` should be a list with the same length as `num_tokens`.")
        self.to_k_ip = nn.ModuleList([_Linear(cross_attention_dim, hidden_size, **kw) for _ in num_tokens])
        self.to_v_ip = nn.ModuleList([_Linear(cross_attention_dim, hidden_size, **kw) for _ in num_tokens])


class _ImageProjection(nn.Module):
    """diffusers `ImageProjection` (what `_load_ip_adapter_weights` builds for the Flux IP-adapter; the class is diffusers' and
    absent from this image — restated from its definition, parity unpinned): Linear(image_embed_dim -> tokens x cross_attention_dim),
    reshape to [B, tokens, cross_attention_dim], LayerNorm(cross_attention_dim, eps 1e-5)."""

    def __init__(self, image_embed_dim: int, cross_attention_dim: int, num_image_text_embeds: int, **kw):
        super().__init__()
        self.num_image_text_embeds, self.cross_attention_dim = num_image_text_embeds, cross_attention_dim
        self.image_embeds = _Linear(image_embed_dim, num_image_text_embeds * cross_attention_dim, **kw)
        self.norm = nn.LayerNorm(cross_attention_dim, **kw)

    def forward(self, image_embeds: torch.Tensor) -> torch.Tensor:
        B = image_embeds.shape[0]
        x = image_embeds.reshape(B, -1).to(self.image_embeds.weight.dtype).contiguous()
        y = ops.gemm(x, self.image_embeds.weight, self.image_embeds.bias).reshape(B * self.num_image_text_embeds, -1)
        y = ops.ln_modulate(y, gamma=self.norm.weight, beta=self.norm.bias, eps=self.norm.eps)
        return y.reshape(B, self.num_image_text_embeds, self.cross_attention_dim)


class _MultiIPAdapterImageProjection(nn.Module):
    """diffusers `MultiIPAdapterImageProjection` (the model's `encoder_hid_proj` once IP adapters are loaded; engines read its
    `num_ip_adapters`, R/src/engine/flux/shared.py:84-96): one `ImageProjection` per adapter; each entry of the input list is
    [B, num_images, image_embed_dim] (or [B, image_embed_dim]) -> [B, num_images x tokens, cross_attention_dim]."""

    def __init__(self, layers):
        super().__init__()
        self.image_projection_layers = nn.ModuleList(layers)

    @property
    def num_ip_adapters(self) -> int:
        return len(self.image_projection_layers)

    def forward(self, image_embeds):
        if not isinstance(image_embeds, (list, tuple)):
            image_embeds = [image_embeds.unsqueeze(1)]
        if len(image_embeds) != len(self.image_projection_layers):
            raise ValueError(f"image_embeds must have the same length as image_projection_layers, got {len(image_embeds)} and "
                             f"{len(self.image_projection_layers)}")
        out = []
        for e, layer in zip(image_embeds, self.image_projection_layers):
            e = e if e.dim() == 3 else e.unsqueeze(1)
            B, n = e.shape[0], e.shape[1]
            out.append(layer(e.reshape(B * n, -1)).reshape(B, n * layer.num_image_text_embeds, -1))
        return out


class _DoubleBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, **kw):
        super().__init__()
        self.norm1 = _AdaNorm(dim, 6, **kw)
        self.norm1_context = _AdaNorm(dim, 6, **kw)
        self.attn = _Attn(dim, heads, head_dim, joint=True, pre_only=False, **kw)
        self.ff = _FF(dim, 4 * dim, **kw)
        self.ff_context = _FF(dim, 4 * dim, **kw)


class _SingleBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, mlp_ratio: float = 4.0, **kw):
        super().__init__()
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm = _AdaNorm(dim, 3, **kw)
        self.proj_mlp = _Linear(dim, self.mlp_hidden_dim, **kw)
        self.proj_out = _Linear(dim + self.mlp_hidden_dim, dim, **kw)
        self.attn = _Attn(dim, heads, head_dim, joint=False, pre_only=True, **kw)


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int, **kw):
        super().__init__()
        self.linear_1 = _Linear(in_dim, dim, **kw)
        self.linear_2 = _Linear(dim, dim, **kw)


class _TimeTextEmbed(nn.Module):
    def __init__(self, dim: int, pooled_dim: int, guidance: bool, **kw):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedding(256, dim, **kw)
        if guidance:
            self.guidance_embedder = _TimestepEmbedding(256, dim, **kw)
        self.text_embedder = _TimestepEmbedding(pooled_dim, dim, **kw)


def _repoint(params, packed_rows):
    """Copy each parameter into its slice of `packed_rows` and make the parameter a view of it."""
    r = 0
    for p in params:
        n = p.shape[0]
        dst = packed_rows[r:r + n]
        dst.copy_(p.data)
        p.data = dst
        r += n
    assert r == packed_rows.shape[0]


class FluxTransformer2DModel(LoraAdapterMixin, nn.Module):
    _converter_base = "flux.base"      # which key-converter table original-format weight files / LoRAs go through (converters.py)
    _supports_gradient_checkpointing = False
    _no_split_modules = ["_DoubleBlock", "_SingleBlock"]

    def __init__(self, patch_size: int = 1, in_channels: int = 64, out_channels: Optional[int] = None,
                 num_layers: int = 19, num_single_layers: int = 38, attention_head_dim: int = 128,
                 num_attention_heads: int = 24, joint_attention_dim: int = 4096,
                 pooled_projection_dim: int = 768, guidance_embeds: bool = False,
                 axes_dims_rope: Tuple[int, int, int] = (16, 56, 56), device=None,
                 dtype=torch.bfloat16):
        super().__init__()
        if attention_head_dim != 128:
            raise _l.ApexMIError("flux.mi355: attention_head_dim must be 128 (MFMA attention tile)")
        self.config = _Config(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels,
                              num_layers=num_layers, num_single_layers=num_single_layers,
                              attention_head_dim=attention_head_dim,
                              num_attention_heads=num_attention_heads,
                              joint_attention_dim=joint_attention_dim,
                              pooled_projection_dim=pooled_projection_dim,
                              guidance_embeds=guidance_embeds, axes_dims_rope=tuple(axes_dims_rope))
        kw = dict(device=device, dtype=dtype)
        self.out_channels = out_channels or in_channels
        self.inner_dim = dim = num_attention_heads * attention_head_dim
        self.time_text_embed = _TimeTextEmbed(dim, pooled_projection_dim, guidance_embeds, **kw)
        self.context_embedder = _Linear(joint_attention_dim, dim, **kw)
        self.x_embedder = _Linear(in_channels, dim, **kw)
        self.transformer_blocks = nn.ModuleList(
            [_DoubleBlock(dim, num_attention_heads, attention_head_dim, **kw) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [_SingleBlock(dim, num_attention_heads, attention_head_dim, **kw)
             for _ in range(num_single_layers)])
        self.norm_out = _AdaNorm(dim, 2, **kw)
        self.proj_out = _Linear(dim, patch_size * patch_size * self.out_channels, **kw)
        self._packed = False
        self._ws: Dict[Any, Any] = {}
        self._side = None
        self._rope_cache = None
        self._scheds = ScheduleRegistry()  # modulation schedules of the clips in flight (begin_schedule), one handle per clip
        self.batch_streams = 2           # images of a batch run side by side on HIP streams (see forward); 1 = sequential
        self._bstreams: List[Any] = []
        self.storage_dtype = torch.bfloat16
        # q/k/v preparation in the QKV GEMM's epilogue where the launch allows it (see _forward_one; APEX_FLUX_FUSE_QKV=0: A/B)
        self.fuse_qkv = os.environ.get("APEX_FLUX_FUSE_QKV", "1") != "0"

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
        return self.x_embedder.weight.dtype

    @property
    def device(self):
        return self.x_embedder.weight.device

    @contextlib.contextmanager
    def cache_context(self, name: str):
        yield

    def _apply(self, fn, *a, **k):
        self._packed = False
        self._ws = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = False
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def init_synthetic(self, seed: int = 0, std: float = 0.02):
        """N(0, std^2) weights, unit norm weights, small biases (SURVEY.md §8d synthetic inputs)."""
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or \
                    name.endswith("norm_added_q.weight") or name.endswith("norm_added_k.weight"):
                p.data.fill_(1.0)
            elif name.endswith(".bias"):
                p.data.copy_((torch.randn(p.shape, generator=g, device=p.device) * 0.01).to(p.dtype))
            else:
                # chunked to keep the f32 temporary small for the 12B-parameter model
                flat = p.data.view(-1)
                step = 1 << 26
                for i in range(0, flat.numel(), step):
                    n = min(step, flat.numel() - i)
                    flat[i:i + n] = (torch.randn(n, generator=g, device=p.device) * std).to(p.dtype)
        self._packed = False
        return self

    # ---- weight packing ----------------------------------------------------------------------
    @torch.no_grad()
    def pack(self):
        if self._packed:
            return
        self._scheds.clear()
        dev, dt = self.device, self.dtype
        if dev.type != "cuda" or dt != torch.bfloat16:
            raise _l.ApexMIError(f"flux.mi355 needs bf16 weights on a ROCm device (got {dt} on {dev}); "
                                 "there is no CPU fallback")
        dim = self.inner_dim
        mods_w, mods_b = [], []
        self._mod_off = {}
        off = 0

        def reg(key, lin):
            nonlocal off
            mods_w.append(lin.weight)
            mods_b.append(lin.bias)
            self._mod_off[key] = off
            off += lin.weight.shape[0]

        for i, blk in enumerate(self.transformer_blocks):
            a = blk.attn
            blk._wqkv = torch.empty(3 * dim, dim, device=dev, dtype=dt)
            blk._bqkv = torch.empty(3 * dim, device=dev, dtype=dt)
            _repoint([a.to_q.weight, a.to_k.weight, a.to_v.weight], blk._wqkv)
            _repoint([a.to_q.bias, a.to_k.bias, a.to_v.bias], blk._bqkv)
            blk._wqkv_c = torch.empty(3 * dim, dim, device=dev, dtype=dt)
            blk._bqkv_c = torch.empty(3 * dim, device=dev, dtype=dt)
            _repoint([a.add_q_proj.weight, a.add_k_proj.weight, a.add_v_proj.weight], blk._wqkv_c)
            _repoint([a.add_q_proj.bias, a.add_k_proj.bias, a.add_v_proj.bias], blk._bqkv_c)
            reg(("d", i, "img"), blk.norm1.linear)
            reg(("d", i, "txt"), blk.norm1_context.linear)
        for i, blk in enumerate(self.single_transformer_blocks):
            a = blk.attn
            blk._wqkv = torch.empty(3 * dim, dim, device=dev, dtype=dt)
            blk._bqkv = torch.empty(3 * dim, device=dev, dtype=dt)
            _repoint([a.to_q.weight, a.to_k.weight, a.to_v.weight], blk._wqkv)
            _repoint([a.to_q.bias, a.to_k.bias, a.to_v.bias], blk._bqkv)
            reg(("s", i), blk.norm.linear)
        reg(("out",), self.norm_out.linear)
        self._mod_w = torch.empty(off, dim, device=dev, dtype=dt)
        self._mod_b = torch.empty(off, device=dev, dtype=dt)
        _repoint(mods_w, self._mod_w)
        _repoint(mods_b, self._mod_b)
        self._mod_total = off
        # rows of the first block that runs (its modulation is needed before the side stream is joined)
        offs = sorted(self._mod_off.values())
        self._mod_first = offs[1] if len(offs) > 1 else off
        if self.transformer_blocks and len(offs) > 2:
            self._mod_first = offs[2]  # img + txt projections of double block 0
        self._packed = True

    def _workspace(self, s_txt: int, s_img: int):
        # one workspace per (shape, HIP stream): two clips may run through one resident model on two streams
        key = (s_txt, s_img, torch.cuda.current_stream().cuda_stream)
        ws = self._ws.get(key)
        if ws is not None and ws.shipped == ops.shipped_verification():
            return ws
        dev, dim = self.device, self.inner_dim
        H = self.config.num_attention_heads
        S = s_txt + s_img
        skp = (S + 63) // 64 * 64
        bf = dict(device=dev, dtype=self.storage_dtype)     # activation buffers
        f32 = dict(device=dev, dtype=torch.float32)
        mlp = 4 * dim
        ws = SimpleNamespace(
            X=torch.empty(S, dim, **bf), XN=torch.empty(S, dim, **bf), QKV=torch.empty(S, 3 * dim, **bf),
            Q=torch.empty(1, H, S, 128, **bf), K=torch.empty(1, H, S, 128, **bf),
            VT=torch.zeros(1, H, 128, skp, **bf), CAT=torch.empty(S, dim + mlp, **bf),
            FFH=torch.empty(S, mlp, **bf), MOD=torch.empty(1, self._mod_total, **f32),
            TEMB=torch.empty(1, dim, **f32), OUT=torch.empty(s_img, self.proj_out.out_features, **bf),
            shipped=ops.shipped_verification(),
        )
        if ws.shipped and self.storage_dtype == torch.float32:
            # verification through the shipped kernels (ops.verify_through_shipped_kernels): the fused QKV epilogue and the flash
            # attention kernel exchange q / k / v^T in bf16, as in production
            b16 = dict(device=dev, dtype=torch.bfloat16)
            ws.Qb, ws.Kb, ws.VTb = torch.empty(1, H, S, 128, **b16), torch.empty(1, H, S, 128, **b16), torch.zeros(1, H, 128, skp, **b16)
        self._ws = {k: v for k, v in self._ws.items() if k[:2] == key[:2]}  # one shape resident at a time
        self._ws[key] = ws
        return ws

    # ---- the denoise step --------------------------------------------------------------------
    def _mod(self, mod, key, idx):
        off = self._mod_off[key] + idx * self.inner_dim
        return mod[0, off:off + self.inner_dim]

    # ---- the whole clip's modulation vectors in one pass over the AdaLN weights ------------------------------------------
    def _cond_rows(self, timestep, guidance, pooled):
        """Conditioning vectors `time_text_embed(timestep, guidance, pooled)` (model.py:535-545) for R rows at once, f32 [R, dim];
        row r is bit-identical to what `_forward_one` builds for (timestep[r], guidance[r], pooled[r]) — same kernels (the
        multi-row GEMV reproduces the single-row arithmetic), same accumulation order t, + guidance, + text."""
        cfg, tte, cdt = self.config, self.time_text_embed, self.storage_dtype
        t = (timestep.to(cdt) * 1000).float().reshape(-1)
        R = t.shape[0]

        def emb(e, proj):
            h = ops.gemv(e.linear_1.weight, proj, e.linear_1.bias, post="silu")
            return ops.gemv(e.linear_2.weight, h, e.linear_2.bias)
        out = emb(tte.timestep_embedder, ops.timestep_embedding(t, 256))
        if cfg.guidance_embeds:
            if guidance is None:
                raise ValueError("guidance_embeds=True model called without `guidance`")
            g = (guidance.to(cdt) * 1000).float().reshape(-1)
            out = emb(tte.guidance_embedder, ops.timestep_embedding(g, 256)) + out      # the ACCUM epilogue's `new + old`
        return emb(tte.text_embedder, pooled.float().reshape(R, -1).contiguous()) + out

    @torch.no_grad()
    def begin_schedule(self, timesteps, guidance, pooled_projections):
        """Every AdaLN shift / scale / gate vector of EVERY step of a clip, computed once when the sampler's timesteps are known
        (VERDICT r3 item 1b).  Per step the stacked `norm*.linear` projections are a 6.4 GB weight-streaming GEMV whose input
        depends only on (t, guidance, pooled) — all known before the loop — so the n steps become ONE pass over those weights
        (`apexmi_gemv` with M = n·B rows, each row bit-identical to the per-step launch) instead of n.

        `timesteps`: [n] or [n, B], the values `forward(timestep=…)` will receive (i.e. already / 1000); `guidance`: [B] or
        None; `pooled_projections`: one [B, P] tensor or a list of them (conditional / unconditional pass of true CFG).
        `begin_schedule` returns the clip's HANDLE (schedule.ModulationSchedule).  `forward(..., joint_attention_kwargs=
        {"modulation_step": i, "modulation_schedule": handle})` with one of these pooled tensors then reads row i of its table;
        any other call computes its vectors as before.  Without a handle the call is served only while exactly one schedule is
        live on the model (two clips through one resident model on two streams must pass theirs).  `end_schedule(handle)` frees
        the tables (n·B × 4.2 MB each) and raises if a scheduled step was called with another timestep / guidance."""
        self.pack()
        n = int(timesteps.shape[0])
        pls = pooled_projections if isinstance(pooled_projections, (list, tuple)) else [pooled_projections]
        sched = None
        for pooled in pls:
            if pooled is None:
                continue
            pooled = pooled.to(self.device)
            B = pooled.shape[0]
            ts = timesteps.to(self.device)
            ts = ts.reshape(n, 1).expand(n, B) if ts.dim() == 1 else ts
            if ts.shape != (n, B):
                raise ValueError(f"begin_schedule: timesteps {tuple(timesteps.shape)} do not match a batch of {B}")
            g = None if guidance is None else guidance.to(self.device).reshape(1, -1).expand(n, B)
            cond = self._cond_rows(ts.reshape(-1), None if g is None else g.reshape(-1),
                                   pooled.unsqueeze(0).expand(n, B, pooled.shape[-1]).reshape(n * B, -1))
            if sched is None:
                sched = ModulationSchedule(n, B, ts, None if g is None else g[0])
                sched.pooled = []
            elif B != sched.B:
                raise ValueError("begin_schedule: the pooled tensors of one clip must share a batch size")
            sched.tables[(pooled.data_ptr(), tuple(pooled.shape))] = ops.gemv(self._mod_w, cond, self._mod_b, pre_silu=True)  # [n B, mod_total] f32
            sched.pooled.append(pooled)           # keeps the key's storage alive for the clip
        return None if sched is None else self._scheds.add(sched)

    def end_schedule(self, handle=None):
        self._scheds.end(handle)
        return self

    def _sched_row(self, pooled_projections, jkw, b, timestep=None, guidance=None):
        """[1, mod_total] view of the table row for (step, image b), or None when this call is not part of a scheduled clip."""
        sc = self._scheds.find(jkw)
        if sc is None:
            return None
        return sc.row((pooled_projections.data_ptr(), tuple(pooled_projections.shape)), int(jkw["modulation_step"]), b,
                      timestep, guidance)

    def _embed_t(self, emb: _TimestepEmbedding, proj: torch.Tensor, out: torch.Tensor, accum: bool):
        h = ops.gemv(emb.linear_1.weight, proj, emb.linear_1.bias, post="silu")
        ops.gemv(emb.linear_2.weight, h, emb.linear_2.bias, out=out, accum=accum)

    def _rope(self, txt_ids, img_ids):
        """The rotary table depends on the position ids only, which a sampler loop passes unchanged every step: keep the last table
        while the SAME tensor objects come back unmodified (the cache holds references to them, so their storage cannot be
        recycled under another tensor, and an in-place edit bumps `_version`)."""
        rk = self._rope_cache
        ver = (ops.tensor_version(txt_ids), ops.tensor_version(img_ids))
        # inference tensors (ids built inside the host's `@torch.inference_mode()` run, R/src/engine/registry.py:196) carry no
        # version counter: an in-place edit would go unseen, so the table is rebuilt for them (one ~20 us launch per step)
        cacheable = ver[0] is not None and ver[1] is not None
        if (cacheable and rk is not None and rk[0] is txt_ids and rk[1] is img_ids and rk[2] == ver
                and rk[4] == self.storage_dtype):
            return rk[3]
        ids = torch.cat((txt_ids, img_ids), dim=0).float()
        rope = ops.rope_table_axes(ids, self.config.axes_dims_rope, 10000.0)
        ops.rope_pairs(rope, trusted=True)      # the compact copy the fused q/k/v epilogue reads, made HERE with the table (before any stream forks)
        self._rope_cache = (txt_ids, img_ids, ver, rope, self.storage_dtype) if cacheable else None
        return rope

    def _step_modulation(self, ws, pooled, timestep, guidance):
        """This step's modulation vectors into ws.MOD (a call outside a scheduled clip).  Returns the event to join before the
        second block, or None."""
        cfg = self.config
        # conditioning vector (f32): timestep.to(dtype) * 1000 as the reference does (model.py:535)
        tte = self.time_text_embed
        cdt = self.storage_dtype     # the reference's `hidden_states.dtype`: bf16 in production, f32 when verifying
        t = (timestep.to(cdt) * 1000).float().reshape(1)
        self._embed_t(tte.timestep_embedder, ops.timestep_embedding(t, 256), ws.TEMB, accum=False)
        if cfg.guidance_embeds:
            if guidance is None:
                raise ValueError("guidance_embeds=True model called without `guidance`")
            g = (guidance.to(cdt) * 1000).float().reshape(1)
            self._embed_t(tte.guidance_embedder, ops.timestep_embedding(g, 256), ws.TEMB, accum=True)
        self._embed_t(tte.text_embedder, pooled.float().reshape(1, -1), ws.TEMB, accum=True)
        # Every AdaLN projection of every block is one weight-streaming GEMV (6.4 GB for FLUX-dev).
        # Only the first block's slice is needed right away: the rest streams on a side HIP stream
        # underneath the first block's MFMA-bound GEMMs and is joined before the second block.
        n_first = self._mod_first
        ops.gemv(self._mod_w[:n_first], ws.TEMB, self._mod_b[:n_first], out=ws.MOD[:, :n_first], pre_silu=True)
        mod_ready = None
        if n_first < self._mod_total:
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                ops.gemv(self._mod_w[n_first:], ws.TEMB, self._mod_b[n_first:], out=ws.MOD[:, n_first:],
                         pre_silu=True)
                mod_ready = torch.cuda.Event()
                mod_ready.record(self._side)
        return mod_ready

    @torch.no_grad()
    def _forward_one(self, hidden_states, encoder_hidden_states, pooled, timestep, img_ids, txt_ids,
                     guidance, mod_row=None, cn_d=None, cn_s=None, ip=None):
        cfg = self.config
        dim, H = self.inner_dim, cfg.num_attention_heads
        s_img, s_txt = hidden_states.shape[0], encoder_hidden_states.shape[0]
        S = s_txt + s_img
        ws = self._workspace(s_txt, s_img)
        X, XN, QKV, CAT, FFH = ws.X, ws.XN, ws.QKV, ws.CAT, ws.FFH
        Xt, Xi = X[:s_txt], X[s_txt:]
        XNt, XNi = XN[:s_txt], XN[s_txt:]

        ops.gemm(hidden_states, self.x_embedder.weight, self.x_embedder.bias, out=Xi)
        ops.gemm(encoder_hidden_states, self.context_embedder.weight, self.context_embedder.bias, out=Xt)

        MOD = mod_row if mod_row is not None else ws.MOD
        mod_ready = None
        if mod_row is None:
            mod_ready = self._step_modulation(ws, pooled, timestep, guidance)

        rope = self._rope(txt_ids, img_ids)

        q_in, k_in, v_in = QKV[:, :dim], QKV[:, dim:2 * dim], QKV[:, 2 * dim:]
        Qp, Kp, VT = ws.Q, ws.K, ws.VT
        att = CAT[:, :dim]
        att_v = att.unflatten(-1, (H, 128)).unsqueeze(0)  # [1, S, H, 128] strided view

        nblk = 0
        overlap = os.environ.get("APEX_FLUX_OVERLAP") == "1"
        # fused q/k/v preparation (apexmi_gemm_bf16_grouped_qkv): bit-identical to the two-pass path; `fuse_qkv = False` keeps
        # the [S, 3 dim] projection as a storage point (tests/stage_parity.py reads it)
        mixed = self.storage_dtype == torch.float32 and ops.shipped_verification()
        fuse = self.fuse_qkv and (self.storage_dtype == torch.bfloat16 or mixed) and self.transformer_blocks is not None
        # IP-adapter: the image queries are needed normalised but NOT rotated (attention.py:199) -> the two-pass q/k/v preparation
        fuse_d = fuse and ip is None and len(self.transformer_blocks) > 0 and ops.qkv_fusable(
            [XNi, XNt], [self.transformer_blocks[0]._wqkv, self.transformer_blocks[0]._wqkv_c], [s_txt, 0], H)
        fuse_s = fuse and len(self.single_transformer_blocks) > 0 and ops.qkv_fusable(
            [XN, XN], [self.single_transformer_blocks[0]._wqkv, self.single_transformer_blocks[0].proj_mlp.weight], [0, 0], H)
        if overlap and self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        if mixed:        # the fused launches and the flash kernel behind them exchange bf16 q / k / v^T (ws.Qb ..), as in production
            qkv_d = (ws.Qb, ws.Kb, ws.VTb) if fuse_d else (ws.Q, ws.K, ws.VT)
            qkv_s = (ws.Qb, ws.Kb, ws.VTb) if fuse_s else (ws.Q, ws.K, ws.VT)
        else:
            qkv_d = qkv_s = (ws.Q, ws.K, ws.VT)

        def join_mod():
            nonlocal mod_ready
            if mod_ready is not None:
                torch.cuda.current_stream().wait_event(mod_ready)
                mod_ready = None

        for i, blk in enumerate(self.transformer_blocks):
            if nblk == 1:
                join_mod()
            nblk += 1
            a = blk.attn
            Qp, Kp, VT = qkv_d
            mi = lambda j: self._mod(MOD, ("d", i, "img"), j)  # noqa: E731
            mt = lambda j: self._mod(MOD, ("d", i, "txt"), j)  # noqa: E731
            # chunk order: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
            ops.ln_modulate(X, mi(1), mi(0), out=XN, split=s_txt, scale2=mt(1), shift2=mt(0))
            if fuse_d:
                # q/k norm + RoPE + [H, S, D] layout and V^T leave the QKV GEMM's epilogue: no [S, 3 dim] round trip
                ops.gemm_grouped_qkv([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c], [None, None], "bias",
                                     [1, 1], [a.norm_q.weight, a.norm_added_q.weight], [a.norm_k.weight, a.norm_added_k.weight],
                                     [s_txt, 0], H, 1e-6, rope, Qp[0], Kp[0], VT[0])
            else:
                ops.gemm_grouped([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c],
                                 [QKV[s_txt:], QKV[:s_txt]])
                ops.qkv_prepare(q_in, k_in, v_in, H, Qp[0], Kp[0], VT[0], wq=a.norm_q.weight,
                                wk=a.norm_k.weight, wq2=a.norm_added_q.weight, wk2=a.norm_added_k.weight,
                                split=s_txt, eps=1e-6, rope=rope, rope_mode=_l.ROPE_INTERLEAVED)
            if ip is not None:
                if getattr(ws, "IPQ", None) is None:
                    ws.IPQ = torch.empty(1, H, s_img, 128, device=X.device, dtype=self.storage_dtype)
                ops.qkv_prepare(q_in[s_txt:], None, None, H, ws.IPQ[0], None, None, wq=a.norm_q.weight, eps=1e-6)
            ops.attention_prepared(Qp, Kp, VT, att_v, S)
            ops.gemm_grouped([att[s_txt:], att[:s_txt]], [a.to_out[0].weight, a.to_add_out.weight],
                             [a.to_out[0].bias, a.to_add_out.bias], [Xi, Xt], epilogue="gate_res",
                             gate_list=[mi(2), mt(2)], residual_list=[Xi, Xt])
            ops.ln_modulate(X, mi(4), mi(3), out=XN, split=s_txt, scale2=mt(4), shift2=mt(3))
            ff, ffc = blk.ff.net, blk.ff_context.net
            ops.gemm_grouped([XNi, XNt], [ff[0].proj.weight, ffc[0].proj.weight],
                             [ff[0].proj.bias, ffc[0].proj.bias], [FFH[s_txt:], FFH[:s_txt]],
                             epilogue="gelu")
            ops.gemm_grouped([FFH[s_txt:], FFH[:s_txt]], [ff[2].weight, ffc[2].weight],
                             [ff[2].bias, ffc[2].bias], [Xi, Xt], epilogue="gate_res",
                             gate_list=[mi(5), mt(5)], residual_list=[Xi, Xt])
            if ip is not None:
                # `hidden_states + ip_attn_output` AFTER the feed-forward (model.py:308-309); ip_attn_output = sum over adapters of
                # scale x attention(image queries, to_k_ip(tokens), to_v_ip(tokens)) (attention.py:232-262): a few keys per adapter
                proc = a.processor
                for h_ip, sc, wk, wv in zip(ip, proc.scale, proc.to_k_ip, proc.to_v_ip):
                    n = h_ip.shape[0]
                    ik = ops.gemm(h_ip, wk.weight, wk.bias).view(1, n, H, 128).permute(0, 2, 1, 3)
                    iv = ops.gemm(h_ip, wv.weight, wv.bias).view(1, n, H, 128).permute(0, 2, 1, 3)
                    o = ops.attention(ws.IPQ, ik, iv).permute(0, 2, 1, 3).reshape(s_img, dim)
                    if self.storage_dtype == torch.bfloat16:
                        ops.euler_step(Xi, o, float(sc), out=Xi)      # x + scale x o in f32, one rounding
                    else:
                        Xi.add_(o, alpha=float(sc))                    # float-storage verification mode
            if cn_d is not None:
                ops.add(Xi, cn_d[i], out=Xi)

        for i, blk in enumerate(self.single_transformer_blocks):
            if nblk == 1:
                join_mod()
            nblk += 1
            a = blk.attn
            Qp, Kp, VT = qkv_s
            ms = lambda j: self._mod(MOD, ("s", i), j)  # noqa: E731  (shift, scale, gate)
            ops.ln_modulate(X, ms(1), ms(0), out=XN)
            if overlap:
                # experiment (APEX_FLUX_OVERLAP=1): the MLP-up GEMM on the side stream underneath q/k prepare +
                # attention, whose 1.69-round launch leaves 80 CUs idle in its second round
                main = torch.cuda.current_stream()
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(self._side):
                    self._side.wait_event(ev)
                    ops.gemm(XN, blk.proj_mlp.weight, blk.proj_mlp.bias, out=CAT[:, dim:], epilogue="gelu")
                    mlp_done = torch.cuda.Event()
                    mlp_done.record(self._side)
                ops.gemm(XN, blk._wqkv, blk._bqkv, out=QKV)
            elif fuse_s:
                ops.gemm_grouped_qkv([XN, XN], [blk._wqkv, blk.proj_mlp.weight], [blk._bqkv, blk.proj_mlp.bias],
                                     [None, CAT[:, dim:]], ["bias", "gelu"], [1, 0], [a.norm_q.weight, None],
                                     [a.norm_k.weight, None], [0, 0], H, 1e-6, rope, Qp[0], Kp[0], VT[0])
            else:
                # QKV and MLP-up read the same XN: one launch, 1512 tiles = 5.9 rounds of the 256 CUs
                ops.gemm_grouped([XN, XN], [blk._wqkv, blk.proj_mlp.weight], [blk._bqkv, blk.proj_mlp.bias],
                                 [QKV, CAT[:, dim:]], epilogue=["bias", "gelu"])
            if overlap or not fuse_s:
                ops.qkv_prepare(q_in, k_in, v_in, H, Qp[0], Kp[0], VT[0], wq=a.norm_q.weight,
                                wk=a.norm_k.weight, split=0, eps=1e-6, rope=rope,
                                rope_mode=_l.ROPE_INTERLEAVED)
            ops.attention_prepared(Qp, Kp, VT, att_v, S)
            if overlap:
                torch.cuda.current_stream().wait_event(mlp_done)
            ops.gemm(CAT, blk.proj_out.weight, blk.proj_out.bias, out=X, epilogue="gate_res", gate=ms(2),
                     residual=X)
            if cn_s is not None:
                ops.add(Xi, cn_s[i], out=Xi)

        join_mod()
        # AdaLayerNormContinuous: scale first, then shift
        ops.ln_modulate(Xi, self._mod(MOD, ("out",), 0), self._mod(MOD, ("out",), 1), out=XNi)
        out = torch.empty(s_img, self.proj_out.out_features, device=X.device, dtype=self.storage_dtype)
        ops.gemm(XNi, self.proj_out.weight, self.proj_out.bias, out=out)
        return out

    # ---- IP-adapter (optional; off in every BASELINE config) ---------------------------------------------------------------
    @torch.no_grad()
    def load_ip_adapter_weights(self, state_dicts, scale=1.0):
        """diffusers `FluxTransformer2DLoadersMixin._load_ip_adapter_weights(state_dicts)` for already-converted adapter files:
        each state dict = {"image_proj": {"proj.weight", "proj.bias", "norm.weight", "norm.bias"}, "ip_adapter": {"<block>.to_k_ip.
        weight", "<block>.to_k_ip.bias", "<block>.to_v_ip.weight", "<block>.to_v_ip.bias"}} (block = double-block index).  Builds
        `encoder_hid_proj` and one `attn.processor` per double block; several adapters stack (one projection + one k / v pair each)."""
        if not isinstance(state_dicts, (list, tuple)):
            state_dicts = [state_dicts]
        kw = dict(device=self.device, dtype=self.dtype)
        dim, ctx = self.inner_dim, self.config.joint_attention_dim
        layers, tokens = [], []
        for sd in state_dicts:
            pw = sd["image_proj"]["proj.weight"]
            n_tok = pw.shape[0] // ctx
            lay = _ImageProjection(pw.shape[1], ctx, n_tok, **kw)
            lay.image_embeds.weight.data.copy_(pw)
            lay.image_embeds.bias.data.copy_(sd["image_proj"]["proj.bias"])
            lay.norm.weight.data.copy_(sd["image_proj"]["norm.weight"])
            lay.norm.bias.data.copy_(sd["image_proj"]["norm.bias"])
            layers.append(lay)
            tokens.append(n_tok)
        self.encoder_hid_proj = _MultiIPAdapterImageProjection(layers)
        for i, blk in enumerate(self.transformer_blocks):
            proc = _IPAdapterProcessor(dim, ctx, tokens, scale, **kw)
            for j, sd in enumerate(state_dicts):
                for name in ("to_k_ip", "to_v_ip"):
                    lin = getattr(proc, name)[j]
                    lin.weight.data.copy_(sd["ip_adapter"][f"{i}.{name}.weight"])
                    lin.bias.data.copy_(sd["ip_adapter"][f"{i}.{name}.bias"])
            blk.attn.processor = proc
        return self

    def set_ip_adapter(self, num_tokens=(4,), scale=1.0):
        """Empty IP-adapter processors on every double block (their weights then arrive with `load_state_dict`: keys
        `transformer_blocks.N.attn.processor.to_k_ip.M.weight`, as a diffusers model with adapters saves them)."""
        kw = dict(device=self.device, dtype=self.dtype)
        for blk in self.transformer_blocks:
            blk.attn.processor = _IPAdapterProcessor(self.inner_dim, self.config.joint_attention_dim, num_tokens, scale, **kw)
        return self

    def set_ip_adapter_scale(self, scale):
        for blk in self.transformer_blocks:
            p = blk.attn.processor
            if p is not None:
                p.scale = list(scale) if isinstance(scale, (list, tuple)) else [float(scale)] * len(p.to_k_ip)
        return self

    def unload_ip_adapter(self):
        for blk in self.transformer_blocks:
            blk.attn.processor = None
        if hasattr(self, "encoder_hid_proj"):
            del self.encoder_hid_proj
        return self

    @ops.on_model_device
    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None,
                img_ids: torch.Tensor = None, txt_ids: torch.Tensor = None,
                guidance: torch.Tensor = None, joint_attention_kwargs: Optional[Dict[str, Any]] = None,
                controlnet_block_samples=None, controlnet_single_block_samples=None,
                return_dict: bool = True, controlnet_blocks_repeat: bool = False):
        # ControlNet residuals (reference model.py:594-612, :631-640): per-block tensors [B, S_img, dim] added to the image stream
        # after the block — sample index = block // ceil(blocks / samples), or block % samples with `controlnet_blocks_repeat`
        # (double blocks only).  The ControlNet that produces them is not part of this path.
        def _cn(samples, nblocks, repeat):
            if samples is None:
                return None
            n = len(samples)
            if n == 0:
                return None
            step = -(-nblocks // n)
            idx = [(i % n if repeat else i // step) for i in range(nblocks)]
            ss = [smp.to(self.device, self.storage_dtype).contiguous() for smp in samples]
            return [ss[j] for j in idx]
        cn_d = _cn(controlnet_block_samples, len(self.transformer_blocks), controlnet_blocks_repeat)
        cn_s = _cn(controlnet_single_block_samples, len(self.single_transformer_blocks), False)
        # IP-adapter inputs (model.py:562-571): image embeddings through `encoder_hid_proj`, or the projected tokens directly
        ip = None
        if joint_attention_kwargs and ("ip_adapter_image_embeds" in joint_attention_kwargs or "ip_hidden_states" in joint_attention_kwargs):
            if any(blk.attn.processor is None for blk in self.transformer_blocks) or not len(self.transformer_blocks):
                raise _l.ApexMIError("flux.mi355: IP-adapter inputs were passed but no adapter is loaded (load_ip_adapter_weights / "
                                     "set_ip_adapter)")
            if "ip_adapter_image_embeds" in joint_attention_kwargs:
                if not hasattr(self, "encoder_hid_proj"):
                    raise _l.ApexMIError("flux.mi355: `ip_adapter_image_embeds` needs `encoder_hid_proj` (load_ip_adapter_weights)")
                emb = joint_attention_kwargs["ip_adapter_image_embeds"]
                emb = [e.to(self.device) for e in emb] if isinstance(emb, (list, tuple)) else emb.to(self.device)
                ip = self.encoder_hid_proj(emb)
            else:
                ip = list(joint_attention_kwargs["ip_hidden_states"])
            n_ad = len(self.transformer_blocks[0].attn.processor.to_k_ip)
            if len(ip) != n_ad:
                raise ValueError(f"{len(ip)} image-prompt tensors for {n_ad} IP adapters")
            ip = [t.to(self.device, self.storage_dtype) for t in ip]
        self.pack()
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        B = hidden_states.shape[0]
        hs = hidden_states.to(self.storage_dtype)
        enc = encoder_hidden_states.to(self.storage_dtype)
        def one(b):
            return self._forward_one(
                hs[b].contiguous(), enc[b].contiguous(), pooled_projections[b], timestep[b:b + 1],
                img_ids, txt_ids, None if guidance is None else guidance[b:b + 1],
                mod_row=self._sched_row(pooled_projections, joint_attention_kwargs, b, timestep[b:b + 1],
                                        None if guidance is None else guidance[b:b + 1]),
                cn_d=None if cn_d is None else [t[b] for t in cn_d], cn_s=None if cn_s is None else [t[b] for t in cn_s],
                ip=None if ip is None else [t[b if t.shape[0] > 1 else 0].contiguous() for t in ip])

        ns = min(int(self.batch_streams), B)
        if ns <= 1 or not hs.is_cuda:
            outs = [one(b) for b in range(B)]
        else:
            # The images of a batch (`num_images`, reference engine/flux/t2i.py:88) are independent and one B=1 step leaves tile-
            # quantisation gaps (216 / 648 / 864 GEMM tiles and 432 attention workgroups on 256 CUs): `batch_streams` images run
            # side by side on their own HIP streams (own workspaces, `_workspace`) and fill them — +4 % images/s measured
            # (profiles/r03_two_clips_ab.json), same kernels on the same data, bit-identical to the sequential walk.
            self._rope(txt_ids, img_ids)        # table made on the calling stream, before the side streams fork from it
            outs = ops.run_on_streams(self._bstreams, ns, B, one, hs.device)
        out = torch.stack(outs, dim=0).to(hidden_states.dtype)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
