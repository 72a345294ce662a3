"""QwenImageTransformer2DModel on the MI355X HIP ops — drop-in for registry key "qwenimage.base"
(QwenImage / QwenImage-Edit-2509).

Mirrors the reference class (apps/api/src/transformer/qwenimage/base/model.py:753-993): same config and
state-dict keys (transformer_blocks.N.img_mod.1.weight, ...attn.add_q_proj..., txt_norm.weight, ...),
`forward(hidden_states [B,S_img,64], encoder_hidden_states [B,T,3584], encoder_hidden_states_mask,
timestep [B] (already /1000), img_shapes, txt_seq_lens, return_dict=False)`.

The 60 blocks are MM-DiT double-stream blocks (model.py:679-750): the step reuses the Flux kernel
sequence over one joint [text | image] buffer — joint-stream LN+modulate, grouped img/txt GEMMs,
per-head RMSNorm + RoPE + V^T, joint attention, gate+residual epilogues.  Differences handled here:
every block owns its img_mod / txt_mod projection (all 120 of them are one batched GEMV per step, chunk
order [shift1, scale1, gate1, shift2, scale2, gate2]); complex RoPE with centred image positions and
offset text positions (QwenEmbedRope, model.py:187-314); RMSNorm(3584) on the text embeddings.
`zero_cond_t` (the condition images' tokens are modulated by a second conditioning row at t = 0: model.py:640-677, :692-702,
:912-923, :980-981) and `use_additional_t_cond` (`addition_t_embedding`, :164-182) are served since round 6 (off in the
Edit-2509 configuration; pinned by tests/golden/qwen_variants.pt); layer-3D RoPE and ControlNet residuals raise
NotImplementedError.
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
from . import ops
from .schedule import ModulationSchedule, ScheduleRegistry
from .flux import _Config, _Linear, _Norm, _FF, _AdaNorm, _TimestepEmbedding, _repoint


class _QwenAttn(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, **kw):
        super().__init__()
        inner = heads * head_dim
        self.heads = heads
        self.to_q, self.to_k, self.to_v = _Linear(dim, inner, **kw), _Linear(dim, inner, **kw), _Linear(dim, inner, **kw)
        self.add_q_proj, self.add_k_proj, self.add_v_proj = (_Linear(dim, inner, **kw), _Linear(dim, inner, **kw),
                                                             _Linear(dim, inner, **kw))
        self.norm_q, self.norm_k = _Norm(head_dim, **kw), _Norm(head_dim, **kw)
        self.norm_added_q, self.norm_added_k = _Norm(head_dim, **kw), _Norm(head_dim, **kw)
        self.to_out = nn.ModuleList([_Linear(inner, dim, **kw), nn.Identity()])
        self.to_add_out = _Linear(inner, dim, **kw)


def _mod_seq(dim: int, **kw):
    return nn.ModuleList([nn.Identity(), _Linear(dim, 6 * dim, **kw)])  # keys: img_mod.1.weight / .bias


class _QwenBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, **kw):
        super().__init__()
        self.img_mod = _mod_seq(dim, **kw)
        self.attn = _QwenAttn(dim, heads, head_dim, **kw)
        self.img_mlp = _FF(dim, 4 * dim, **kw)
        self.txt_mod = _mod_seq(dim, **kw)
        self.txt_mlp = _FF(dim, 4 * dim, **kw)


class _TimeTextEmbed(nn.Module):
    def __init__(self, dim: int, use_additional_t_cond: bool = False, **kw):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedding(256, dim, **kw)
        if use_additional_t_cond:      # QwenTimestepProjEmbeddings, model.py:164-166: key time_text_embed.addition_t_embedding.weight
            self.addition_t_embedding = nn.Embedding(2, dim, **kw)


class QwenImageTransformer2DModel(LoraAdapterMixin, nn.Module):
    _converter_base = "qwenimage.base"      # which key-converter table original-format weight files / LoRAs go through (converters.py)
    _no_split_modules = ["_QwenBlock"]

    def __init__(self, patch_size: int = 2, in_channels: int = 64, out_channels: Optional[int] = 16,
                 num_layers: int = 60, attention_head_dim: int = 128, num_attention_heads: int = 24,
                 joint_attention_dim: int = 3584, guidance_embeds: bool = False,
                 axes_dims_rope: Tuple[int, int, int] = (16, 56, 56), zero_cond_t: bool = False,
                 use_additional_t_cond: bool = False, use_layer3d_rope: bool = False, device=None,
                 dtype=torch.bfloat16):
        super().__init__()
        if attention_head_dim != 128:
            raise _l.ApexMIError("qwenimage.mi355: attention_head_dim must be 128 (MFMA attention tile)")
        if use_layer3d_rope:
            raise NotImplementedError("qwenimage.mi355: the layer3d rope variant")
        self.config = _Config(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels,
                              num_layers=num_layers, attention_head_dim=attention_head_dim,
                              num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                              guidance_embeds=guidance_embeds, axes_dims_rope=tuple(axes_dims_rope),
                              zero_cond_t=zero_cond_t, use_additional_t_cond=use_additional_t_cond)
        kw = dict(device=device, dtype=dtype)
        self.out_channels = out_channels or in_channels
        self.inner_dim = dim = num_attention_heads * attention_head_dim
        self.time_text_embed = _TimeTextEmbed(dim, use_additional_t_cond, **kw)
        self.txt_norm = _Norm(joint_attention_dim, **kw)
        self.img_in = _Linear(in_channels, dim, **kw)
        self.txt_in = _Linear(joint_attention_dim, dim, **kw)
        self.transformer_blocks = nn.ModuleList(
            [_QwenBlock(dim, num_attention_heads, attention_head_dim, **kw) for _ in range(num_layers)])
        self.norm_out = _AdaNorm(dim, 2, **kw)
        self.proj_out = _Linear(dim, patch_size * patch_size * self.out_channels, **kw)
        self._packed = False
        self._ws: Dict[Any, Any] = {}
        self.storage_dtype = torch.bfloat16
        # q/k/v preparation in the QKV GEMM's epilogue where the launch allows it (APEX_FUSE_QKV=0: A/B)
        self.fuse_qkv = os.environ.get("APEX_FUSE_QKV", "1") != "0"
        self._rope: Dict[Any, torch.Tensor] = {}
        self._side = None
        self._scheds = ScheduleRegistry()  # modulation schedules of the clips in flight (begin_schedule), one handle per clip
        self.batch_streams = 2           # images of a batch on side-by-side HIP streams (forward); 1 = sequential
        self._bstreams: List[Any] = []

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
        return self.img_in.weight.dtype

    @property
    def device(self):
        return self.img_in.weight.device

    @contextlib.contextmanager
    def cache_context(self, name: str):
        yield

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
            if "norm_" in name or name == "txt_norm.weight":
                p.data.fill_(1.0)
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
        self._scheds.clear()
        dev, dt = self.device, self.dtype
        if dev.type != "cuda" or dt != torch.bfloat16:
            raise _l.ApexMIError(f"qwenimage.mi355 needs bf16 weights on a ROCm device (got {dt} on {dev}); "
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
            mods_w += [blk.img_mod[1].weight, blk.txt_mod[1].weight]
            mods_b += [blk.img_mod[1].bias, blk.txt_mod[1].bias]
        mods_w.append(self.norm_out.linear.weight)
        mods_b.append(self.norm_out.linear.bias)
        total = sum(w.shape[0] for w in mods_w)
        self._mod_w = torch.empty(total, dim, device=dev, dtype=dt)
        self._mod_b = torch.empty(total, device=dev, dtype=dt)
        _repoint(mods_w, self._mod_w)
        _repoint(mods_b, self._mod_b)
        self._mod_total = total
        self._mod_first = min(12 * dim, total)   # img_mod + txt_mod of block 0
        self._packed = True

    def _workspace(self, s_txt: int, s_img: int):
        key = (s_txt, s_img, torch.cuda.current_stream().cuda_stream)   # per stream: the images of a batch run side by side
        ws = self._ws.get(key)
        if ws is not None and ws.shipped == ops.shipped_verification():
            return ws
        dev, dim, H = self.device, self.inner_dim, self.config.num_attention_heads
        S = s_txt + s_img
        skp = (S + 63) // 64 * 64
        bf = dict(device=dev, dtype=self.storage_dtype)     # activation buffers
        f32 = dict(device=dev, dtype=torch.float32)
        ws = SimpleNamespace(
            X=torch.empty(S, dim, **bf), XN=torch.empty(S, dim, **bf), QKV=torch.empty(S, 3 * dim, **bf),
            Q=torch.empty(1, H, S, 128, **bf), K=torch.empty(1, H, S, 128, **bf),
            VT=torch.zeros(1, H, 128, skp, **bf), ATT=torch.empty(S, dim, **bf),
            FFH=torch.empty(S, 4 * dim, **bf), TXTN=torch.empty(s_txt, self.config.joint_attention_dim, **bf),
            # zero_cond_t: a second conditioning row (t = 0) for the condition images' tokens
            MOD=torch.empty(2 if self.config.zero_cond_t else 1, self._mod_total, **f32),
            TEMB=torch.empty(2 if self.config.zero_cond_t else 1, dim, **f32), shipped=ops.shipped_verification())
        if ws.shipped and self.storage_dtype == torch.float32:     # verification through the shipped kernels (flux.py `_workspace`)
            b16 = dict(device=dev, dtype=torch.bfloat16)
            ws.Qb, ws.Kb, ws.VTb = torch.empty(1, H, S, 128, **b16), torch.empty(1, H, S, 128, **b16), torch.zeros(1, H, 128, skp, **b16)
        self._ws.pop(key, None)
        # a few workspaces stay resident (the images of a batch on two streams; the cond / uncond passes of true CFG with their own
        # text lengths, engine_qwenimage.py): the oldest goes when a fifth shows up
        while len(self._ws) >= 4:
            self._ws.pop(next(iter(self._ws)))
        self._ws[key] = ws
        return ws

    def _rope_table(self, shapes, s_txt: int):
        # keyed on the stream too: a table is made and read on one stream only (no cross-stream ordering to get wrong)
        key = (tuple(tuple(int(v) for v in s) for s in shapes), s_txt, torch.cuda.current_stream().cuda_stream if self.device.type == "cuda" else 0)
        t = self._rope.get(key)
        if t is None:
            vids, max_idx = [], 0
            for idx, (f, h, w) in enumerate(key[0]):
                fr = torch.arange(idx, idx + f)
                hh = torch.cat([torch.arange(-(h - h // 2), 0), torch.arange(0, h // 2)])
                ww = torch.cat([torch.arange(-(w - w // 2), 0), torch.arange(0, w // 2)])
                vids.append(torch.stack(torch.meshgrid(fr, hh, ww, indexing="ij"), dim=-1).reshape(-1, 3))
                max_idx = max(h // 2, w // 2, max_idx)
            tt = torch.arange(max_idx, max_idx + s_txt)
            ids = torch.cat([torch.stack([tt, tt, tt], dim=-1)] + vids, dim=0).float().to(self.device)
            t = ops.rope_table_axes(ids.contiguous(), self.config.axes_dims_rope, 10000.0)
            ops.rope_pairs(t, trusted=True)     # the compact copy the fused q/k/v epilogue reads, made with the table on its stream
            while len(self._rope) >= 4:
                self._rope.pop(next(iter(self._rope)))
            self._rope[key] = t
        return t

    @torch.no_grad()
    def begin_schedule(self, timesteps):
        """Every block's img_mod / txt_mod vectors and norm_out's, for EVERY step of a clip, from one pass over the stacked
        projection weights (6.8 GB for the 60-block model) instead of one weight-streaming GEMV per step (flux.py
        `begin_schedule`; here the conditioning vector depends on the timestep alone, so the conditional and the unconditional
        pass of true CFG share a row).  `timesteps`: [n] or [n, B], the values `forward(timestep=…)` will receive (already
        / 1000).  Rows are bit-identical to the per-step launches.  Returns the clip's HANDLE (schedule.ModulationSchedule):
        `forward(attention_kwargs={"modulation_step": i, "modulation_schedule": handle})` reads row i of THAT clip's table; a call
        without a handle is served only while exactly one schedule is live on the model, any other call computes its own vectors
        as before.  `end_schedule(handle)` frees the table and raises if a scheduled step ran with another timestep."""
        self.pack()
        n = int(timesteps.shape[0])
        ts = timesteps.to(self.device).reshape(n, -1)
        B = ts.shape[1]
        te = self.time_text_embed.timestep_embedder
        t = ts.to(self.storage_dtype).float().reshape(-1)
        h = ops.gemv(te.linear_1.weight, ops.timestep_embedding(t, 256, scale=1000.0), te.linear_1.bias, post="silu")
        temb = ops.gemv(te.linear_2.weight, h, te.linear_2.bias)
        sched = ModulationSchedule(n, B, ts)
        sched.tables["mod"] = ops.gemv(self._mod_w, temb, self._mod_b, pre_silu=True)
        sched.table, sched.temb = sched.tables["mod"], temb
        return self._scheds.add(sched)

    def end_schedule(self, handle=None):
        self._scheds.end(handle)
        return self

    def _sched_row(self, kw, b, timestep=None):
        sc = self._scheds.find(kw)
        if sc is None:
            return None
        return sc.row("mod", int(kw["modulation_step"]), b, timestep, clamp_b=True)   # an [n] schedule serves every image of the batch

    @torch.no_grad()
    def _forward_one(self, hidden_states, text, timestep, shapes, mod_row=None, addition_t_cond=None):
        cfg = self.config
        zc = bool(cfg.zero_cond_t)
        dim, H = self.inner_dim, cfg.num_attention_heads
        s_img, s_txt = hidden_states.shape[0], text.shape[0]
        if sum(int(f) * int(h) * int(w) for f, h, w in shapes) != s_img:
            raise ValueError(f"img_shapes {shapes} do not cover {s_img} image tokens")
        S = s_txt + s_img
        ws = self._workspace(s_txt, s_img)
        X, XN, QKV, ATT, FFH = ws.X, ws.XN, ws.QKV, ws.ATT, ws.FFH
        Xt, Xi, XNt, XNi = X[:s_txt], X[s_txt:], XN[:s_txt], XN[s_txt:]

        ops.gemm(hidden_states, self.img_in.weight, self.img_in.bias, out=Xi)
        ops.ln_modulate(text, gamma=self.txt_norm.weight, out=ws.TXTN, eps=1e-6, rms=True)
        ops.gemm(ws.TXTN, self.txt_in.weight, self.txt_in.bias, out=Xt)

        mod_ready = None
        MOD = mod_row
        if MOD is None:
            MOD = ws.MOD
            te = self.time_text_embed.timestep_embedder
            # `timestep.to(hidden_states.dtype)`, model.py:905: bf16 in production, f32 in the verification mode
            t = timestep.to(self.storage_dtype).float().reshape(1)
            if zc:                         # `timestep = torch.cat([timestep, timestep * 0])`, model.py:912-913
                t = torch.cat([t, t * 0])
            tp = ops.timestep_embedding(t, 256, scale=1000.0)
            h = ops.gemv(te.linear_1.weight, tp, te.linear_1.bias, post="silu")
            ops.gemv(te.linear_2.weight, h, te.linear_2.bias, out=ws.TEMB)
            if cfg.use_additional_t_cond:  # conditioning + addition_t_embedding[addition_t_cond] (both rows), model.py:175-182
                if addition_t_cond is None:
                    raise ValueError("When additional_t_cond is True, addition_t_cond must be provided.")
                ws.TEMB.add_(self.time_text_embed.addition_t_embedding.weight[addition_t_cond.reshape(1).long()].float())

            n_first = self._mod_first
            ops.gemv(self._mod_w[:n_first], ws.TEMB, self._mod_b[:n_first], out=ws.MOD[:, :n_first], pre_silu=True)
            if n_first < self._mod_total:   # the other blocks' modulation streams on a side stream under block 0
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
        rope = self._rope_table(shapes, s_txt)

        q_in, k_in, v_in = QKV[:, :dim], QKV[:, dim:2 * dim], QKV[:, 2 * dim:]
        att_v = ATT.unflatten(-1, (H, 128)).unsqueeze(0)
        # fused q/k/v preparation (apexmi_gemm_bf16_grouped_qkv) where the launch allows it: bf16 storage, 8-aligned streams,
        # >= 1024 rows; `fuse_qkv = False` keeps the [S, 3 dim] projection as a storage point (tests/stage_parity.py)
        mixed = self.storage_dtype == torch.float32 and ops.shipped_verification()
        fuse = (getattr(self, "fuse_qkv", True) and (getattr(self, "storage_dtype", torch.bfloat16) == torch.bfloat16 or mixed)
                and len(self.transformer_blocks) > 0 and tuple(rope.shape) == (2, S, 128)
                and ops.qkv_fusable([XNi, XNt], [self.transformer_blocks[0]._wqkv, self.transformer_blocks[0]._wqkv_c], [s_txt, 0], H))
        Qp, Kp, VTp = (ws.Qb, ws.Kb, ws.VTb) if (mixed and fuse) else (ws.Q, ws.K, ws.VT)
        # zero_cond_t: the first image (the target) is conditioned on t, the images after it on t = 0 (`modulate_index`, :914-921)
        n0 = int(shapes[0][0]) * int(shapes[0][1]) * int(shapes[0][2]) if zc else s_img
        for i, blk in enumerate(self.transformer_blocks):
            if i == 1 and mod_ready is not None:
                torch.cuda.current_stream().wait_event(mod_ready)
                mod_ready = None
            a = blk.attn
            base = i * 12 * dim
            mi = lambda j: MOD[0, base + j * dim: base + (j + 1) * dim]              # noqa: E731
            mt = lambda j: MOD[0, base + (6 + j) * dim: base + (7 + j) * dim]        # noqa: E731
            # chunk order: shift1, scale1, gate1 | shift2, scale2, gate2
            if zc:
                self._block_zero_cond(blk, ws, MOD, base, s_txt, n0, rope, fuse, Qp, Kp, VTp, q_in, k_in, v_in, att_v, H)
                continue
            ops.ln_modulate(X, mi(1), mi(0), out=XN, split=s_txt, scale2=mt(1), shift2=mt(0))
            if fuse:
                # q/k norm + RoPE + [H, S, D] layout and V^T leave the QKV GEMM's epilogue (bit-identical to the two passes)
                ops.gemm_grouped_qkv([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c], [None, None], "bias",
                                     [1, 1], [a.norm_q.weight, a.norm_added_q.weight], [a.norm_k.weight, a.norm_added_k.weight],
                                     [s_txt, 0], H, 1e-6, rope, Qp[0], Kp[0], VTp[0])
            else:
                ops.gemm_grouped([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c],
                                 [QKV[s_txt:], QKV[:s_txt]])
                ops.qkv_prepare(q_in, k_in, v_in, H, Qp[0], Kp[0], VTp[0], wq=a.norm_q.weight,
                                wk=a.norm_k.weight, wq2=a.norm_added_q.weight, wk2=a.norm_added_k.weight,
                                split=s_txt, eps=1e-6, rope=rope, rope_mode=_l.ROPE_INTERLEAVED)
            ops.attention_prepared(Qp, Kp, VTp, att_v, S)
            ops.gemm_grouped([ATT[s_txt:], ATT[:s_txt]], [a.to_out[0].weight, a.to_add_out.weight],
                             [a.to_out[0].bias, a.to_add_out.bias], [Xi, Xt], epilogue="gate_res",
                             gate_list=[mi(2), mt(2)], residual_list=[Xi, Xt])
            ops.ln_modulate(X, mi(4), mi(3), out=XN, split=s_txt, scale2=mt(4), shift2=mt(3))
            fi, ft = blk.img_mlp.net, blk.txt_mlp.net
            ops.gemm_grouped([XNi, XNt], [fi[0].proj.weight, ft[0].proj.weight], [fi[0].proj.bias, ft[0].proj.bias],
                             [FFH[s_txt:], FFH[:s_txt]], epilogue="gelu")
            ops.gemm_grouped([FFH[s_txt:], FFH[:s_txt]], [fi[2].weight, ft[2].weight], [fi[2].bias, ft[2].bias],
                             [Xi, Xt], epilogue="gate_res", gate_list=[mi(5), mt(5)], residual_list=[Xi, Xt])
        if mod_ready is not None:
            torch.cuda.current_stream().wait_event(mod_ready)
        o = len(self.transformer_blocks) * 12 * dim
        # AdaLayerNormContinuous: scale first, then shift
        ops.ln_modulate(Xi, MOD[0, o:o + dim], MOD[0, o + dim:o + 2 * dim], out=XNi)
        return ops.gemm(XNi, self.proj_out.weight, self.proj_out.bias)

    def _block_zero_cond(self, blk, ws, MOD, base, s_txt, n0, rope, fuse, Qp, Kp, VTp, q_in, k_in, v_in, att_v, H):
        """One block with `zero_cond_t`: the same launches, with the image stream's modulation and gates taken per ROW RANGE —
        target tokens [s_txt, s_txt + n0) from conditioning row 0 (t), condition-image tokens behind them from row 1 (t = 0); the
        text stream from row 0 (`_modulate(index)`, model.py:640-677; `txt_mod(temb.chunk(2)[0])`, :692-693).  The LN + modulate
        pass runs once per range, the gate / residual GEMMs carry the two image ranges as two problems of the grouped launch."""
        dim = self.inner_dim
        a = blk.attn
        X, XN, QKV, ATT, FFH = ws.X, ws.XN, ws.QKV, ws.ATT, ws.FFH
        S = X.shape[0]
        c0 = s_txt + n0                                   # first condition-image row
        Xt, Xi, XNt, XNi = X[:s_txt], X[s_txt:], XN[:s_txt], XN[s_txt:]
        m0 = lambda j: MOD[0, base + j * dim: base + (j + 1) * dim]                # noqa: E731  image stream, row t
        m1 = lambda j: MOD[1, base + j * dim: base + (j + 1) * dim]                # noqa: E731  image stream, row t = 0
        mt = lambda j: MOD[0, base + (6 + j) * dim: base + (7 + j) * dim]          # noqa: E731  text stream, row t
        rng = [(s_txt, c0), (c0, S), (0, s_txt)] if c0 < S else [(s_txt, S), (0, s_txt)]
        gates = lambda j: ([m0(j), m1(j), mt(j)] if c0 < S else [m0(j), mt(j)])    # noqa: E731

        def ln(jscale, jshift):
            ops.ln_modulate(X[:c0], m0(jscale), m0(jshift), out=XN[:c0], split=s_txt, scale2=mt(jscale), shift2=mt(jshift))
            if c0 < S:
                ops.ln_modulate(X[c0:], m1(jscale), m1(jshift), out=XN[c0:])

        def gated(src, w_img, b_img, w_txt, b_txt, j):
            ws_ = [w_img] * (len(rng) - 1) + [w_txt]
            bs_ = [b_img] * (len(rng) - 1) + [b_txt]
            ops.gemm_grouped([src[lo:hi] for lo, hi in rng], ws_, bs_, [X[lo:hi] for lo, hi in rng], epilogue="gate_res",
                             gate_list=gates(j), residual_list=[X[lo:hi] for lo, hi in rng])

        ln(1, 0)
        if fuse:
            ops.gemm_grouped_qkv([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c], [None, None], "bias",
                                 [1, 1], [a.norm_q.weight, a.norm_added_q.weight], [a.norm_k.weight, a.norm_added_k.weight],
                                 [s_txt, 0], H, 1e-6, rope, Qp[0], Kp[0], VTp[0])
        else:
            ops.gemm_grouped([XNi, XNt], [blk._wqkv, blk._wqkv_c], [blk._bqkv, blk._bqkv_c], [QKV[s_txt:], QKV[:s_txt]])
            ops.qkv_prepare(q_in, k_in, v_in, H, Qp[0], Kp[0], VTp[0], wq=a.norm_q.weight, wk=a.norm_k.weight,
                            wq2=a.norm_added_q.weight, wk2=a.norm_added_k.weight, split=s_txt, eps=1e-6, rope=rope,
                            rope_mode=_l.ROPE_INTERLEAVED)
        ops.attention_prepared(Qp, Kp, VTp, att_v, S)
        gated(ATT, a.to_out[0].weight, a.to_out[0].bias, a.to_add_out.weight, a.to_add_out.bias, 2)
        ln(4, 3)
        fi, ft = blk.img_mlp.net, blk.txt_mlp.net
        ops.gemm_grouped([XNi, XNt], [fi[0].proj.weight, ft[0].proj.weight], [fi[0].proj.bias, ft[0].proj.bias],
                         [FFH[s_txt:], FFH[:s_txt]], epilogue="gelu")
        gated(FFH, fi[2].weight, fi[2].bias, ft[2].weight, ft[2].bias, 5)

    @ops.on_model_device
    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                encoder_hidden_states_mask: torch.Tensor = None, timestep: torch.Tensor = None,
                img_shapes=None, txt_seq_lens=None, guidance: torch.Tensor = None, attention_kwargs=None,
                controlnet_block_samples=None, additional_t_cond=None, return_dict: bool = True):
        if controlnet_block_samples is not None:
            raise NotImplementedError("qwenimage.mi355: controlnet residuals are out of scope")
        if self.config.use_additional_t_cond and additional_t_cond is None:
            raise ValueError("When additional_t_cond is True, addition_t_cond must be provided.")
        # the per-clip modulation table holds ONE row per step (conditioning on t alone): the variants compute theirs per step
        variant = bool(self.config.zero_cond_t or self.config.use_additional_t_cond)
        self.pack()
        B = hidden_states.shape[0]
        hs = hidden_states.to(self.storage_dtype)
        enc = encoder_hidden_states.to(self.storage_dtype)
        def shapes_of(b):
            return img_shapes[b] if isinstance(img_shapes[0], (list, tuple)) and \
                isinstance(img_shapes[0][0], (list, tuple)) else img_shapes

        def one(b):
            return self._forward_one(hs[b].contiguous(), enc[b].contiguous(), timestep[b:b + 1], shapes_of(b),
                                     mod_row=None if variant else self._sched_row(attention_kwargs, b, timestep[b:b + 1]),
                                     addition_t_cond=additional_t_cond[b:b + 1] if self.config.use_additional_t_cond else None)

        ns = min(int(self.batch_streams), B)
        if ns <= 1 or not hs.is_cuda or any(shapes_of(b) != shapes_of(0) for b in range(1, B)):
            outs = [one(b) for b in range(B)]
        else:
            # the images of a batch side by side on HIP streams (see flux.py forward; same mechanism, bit-identical results)
            outs = ops.run_on_streams(self._bstreams, ns, B, one, hs.device)
        out = torch.stack(outs, dim=0).to(hidden_states.dtype)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
