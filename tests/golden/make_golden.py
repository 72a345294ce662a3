"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

Run from the repo root:  python tests/golden/make_golden.py   (needs /root/reference; CPU only)

Nothing from the reference is copied: its modules are imported from /root/reference/apps/api and
executed; only inputs and outputs (tensors) are saved.  What each fixture pins:

  attention_sdpa.pt     reference attention_register "sdpa" (attention/functions.py:338-377) on
                        seeded q,k,v, f32 and bf16 — pins oracle.layers.sdpa and the HIP kernel.
  efficiency_ops.pt     reference transformer/efficiency/ops.py apply_gate_inplace /
                        apply_scale_shift_inplace / apply_cos_sin_rope_inplace and mod.py InplaceRMSNorm
                        (bf16 input; the fp32 path has the aliasing defect of SURVEY.md App. B-2).
  flux_hybrid.pt        the reference's OWN FluxTransformer2DModel / blocks / attention processor
                        (transformer/flux/base/model.py, attention.py) executed on a tiny config with
                        the un-vendored diffusers leaf layers supplied by oracle.layers ("hybrid
                        oracle"): pins block wiring, chunk orders, RoPE layout, concat
                        order and reshapes of oracle.flux — not the diffusers leaf arithmetic.
  hunyuan15_hybrid.pt   the reference's OWN HunyuanVideo15Transformer3DModel / block / attention processor / token refiner
                        (transformer/hunyuanvideo15/base/model.py) on a tiny config, t2v and i2v token orders, leaves
                        from oracle.layers: pins the wiring of oracle.hunyuan15.
  flux_controlnet.pt    the reference Flux model fed ControlNet residual samples (interval and repeat placement) — pins oracle.flux's
                        and the HIP model's `controlnet_block_samples` / `controlnet_single_block_samples` inputs.
  wan_easycache.pt      the reference's EasyCache forward (transformer/wan/base/model.py:202-520) on the tiny reference Wan model driven
                        like a CFG sampler: which calls run / are served from the cache, every output — pins oracle.easycache.
  flux_scheduler.pt     oracle FlowMatch-Euler trajectory (restatement only; diffusers absent).
  fp_scaled.pt          reference fp8_activation_dequant / FPScaledLinear._scale_and_cast_weight
                        (quantize/scaled_layer.py:154-167, :496-549) on seeded float8_e4m3fn / e5m2 weights with
                        scalar and per-row scales, every fp8 code point included — pins oracle.weights and the
                        HIP dequant kernel.
  vae_hunyuan15.pt      the reference's AutoencoderKLHunyuanVideo15 decode (replicate-padded causal convs, frame-causal
                        mid-block attention, DCAE pixel-shuffle upsampling, 8x8-latent tiling) — pins oracle.vae_hunyuan15.
  lora_convert.pt       reference LoraConverter().convert (lora/lora_converter.py:80-183) on seeded PEFT-with-alpha
                        and lora_down/lora_up state dicts — pins key normalisation and alpha folding of
                        apex_studio_amd.lora / oracle.lora (the PEFT runtime arithmetic itself is absent).

The diffusers stubs below carry NO arithmetic except the leaves re-exported from oracle.layers.
"""
from __future__ import annotations

import contextlib
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/apps/api"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import layers as OL  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    os.environ.setdefault("APEX_HOME_DIR", "/tmp/apexhome")

    class _Logger:
        def __getattr__(self, _):
            return lambda *a, **k: None

    _mod("loguru", logger=_Logger())

    class ConfigMixin:
        config_name = "config.json"

    def register_to_config(init):
        import functools
        import inspect

        @functools.wraps(init)
        def wrapper(self, *args, **kwargs):
            sig = inspect.signature(init)
            bound = sig.bind(self, *args, **kwargs)
            bound.apply_defaults()
            cfg = {k: v for k, v in bound.arguments.items() if k != "self"}

            class _Cfg(dict):
                __getattr__ = dict.__getitem__

            init(self, *args, **kwargs)
            self.config = _Cfg(cfg)

        return wrapper

    class ModelMixin(nn.Module):
        @property
        def dtype(self):
            return next(self.parameters()).dtype

        @property
        def device(self):
            return next(self.parameters()).device

    class _Empty:
        pass

    class AttentionModuleMixin:
        fused_projections = False

        def set_processor(self, processor):
            self.processor = processor

    class CacheMixin:
        @contextlib.contextmanager
        def cache_context(self, name):
            yield

    class _Out:
        def __init__(self, sample=None):
            self.sample = sample

    class _LoggingNS:
        @staticmethod
        def get_logger(name):
            return _Logger()

    d = _mod("diffusers")
    d.__path__ = []
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.loaders", FluxTransformer2DLoadersMixin=type("A", (), {}),
         FromOriginalModelMixin=type("B", (), {}), PeftAdapterMixin=type("Cc", (), {}))
    _mod("diffusers.utils", USE_PEFT_BACKEND=False, deprecate=lambda *a, **k: None, logging=_LoggingNS,
         scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None)
    _mod("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda c: c)
    _mod("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)
    m = _mod("diffusers.models")
    m.__path__ = []
    _mod("diffusers.models.attention", AttentionMixin=type("D", (), {}),
         AttentionModuleMixin=AttentionModuleMixin, FeedForward=OL.FeedForward, Attention=type("Attention", (), {}))
    _mod("diffusers.models.attention_processor", Attention=OL.DiffusersAttention)
    _mod("diffusers.models.cache_utils", CacheMixin=CacheMixin)
    _mod("diffusers.models.embeddings",
         CombinedTimestepGuidanceTextProjEmbeddings=OL.CombinedTimestepGuidanceTextProjEmbeddings,
         CombinedTimestepTextProjEmbeddings=OL.CombinedTimestepTextProjEmbeddings,
         get_1d_rotary_pos_embed=OL.get_1d_rotary_pos_embed, apply_rotary_emb=OL.apply_rotary_emb,
         Timesteps=OL.Timesteps, TimestepEmbedding=OL.TimestepEmbedding,
         PixArtAlphaTextProjection=OL.PixArtAlphaTextProjection)
    _mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=_Out)
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.normalization", AdaLayerNormContinuous=OL.AdaLayerNormContinuous,
         AdaLayerNormZero=OL.AdaLayerNormZero, AdaLayerNormZeroSingle=OL.AdaLayerNormZeroSingle,
         FP32LayerNorm=OL.FP32LayerNorm, RMSNorm=OL.RMSNorm)

    sys.path.insert(0, REF)
    # shell package for src.transformer: skip the directory auto-scan of its __init__
    import src  # noqa: F401  (reference top-level package)
    import src.register  # noqa: F401
    base_spec = importlib.util.spec_from_file_location(
        "src.transformer.base", os.path.join(REF, "src/transformer/base.py"))
    shell = types.ModuleType("src.transformer")
    shell.__path__ = [os.path.join(REF, "src/transformer")]
    sys.modules["src.transformer"] = shell
    base = importlib.util.module_from_spec(base_spec)
    sys.modules["src.transformer.base"] = base
    base_spec.loader.exec_module(base)
    shell.TRANSFORMERS_REGISTRY = base.TRANSFORMERS_REGISTRY


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


from tests.golden.seeded import (seeded, spec_tensors, synthetic_state_dict, tensor_digest, text_encoder_state_dict,  # noqa: E402
                                 vae_synthetic_state_dict)


def gen_attention():
    from src.attention.functions import attention_register
    cases = []
    specs = [((1, 2, 8, 64), (1, 2, 8, 64), 11), ((1, 4, 200, 128), (1, 4, 200, 128), 12),
             ((1, 2, 130, 128), (1, 2, 77, 128), 13), ((2, 3, 96, 128), (2, 3, 64, 128), 14)]
    for qs, ks, seed in specs:
        for dt in (torch.float32, torch.bfloat16):
            q, k, v = seeded(qs, seed, dt), seeded(ks, seed + 100, dt), seeded(ks, seed + 200, dt)
            # callers pass permuted [B,S,H,D] views (flux/base/attention.py:89-94)
            qv = q.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
            out = attention_register.call(qv, k, v, key="sdpa")
            # inputs are regenerated from (shape, seed) by tests.golden.seeded.seeded
            cases.append(dict(q_shape=qs, k_shape=ks, seed=seed, out=out.contiguous(), dtype=str(dt)))
    torch.save(cases, os.path.join(OUT, "attention_sdpa.pt"))
    print("attention_sdpa.pt", len(cases))


def gen_efficiency():
    ops = load_by_path("ref_eff_ops", "src/transformer/efficiency/ops.py")
    mod = load_by_path("ref_eff_mod", "src/transformer/efficiency/mod.py")
    out = {}
    x = seeded((1, 40, 256), 21, torch.bfloat16)
    gate = seeded((1, 1, 256), 22, torch.bfloat16)
    y = x.clone()
    ops.apply_gate_inplace(y, gate)
    out["gate"] = dict(x=x, gate=gate, out=y)
    scale, shift = seeded((1, 1, 256), 23, torch.bfloat16), seeded((1, 1, 256), 24, torch.bfloat16)
    y = x.clone()
    ops.apply_scale_shift_inplace(y, scale, shift)
    out["scale_shift"] = dict(x=x, scale=scale, shift=shift, out=y)
    norm = mod.InplaceRMSNorm(256, eps=1e-6)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.1 * seeded((256,), 25))
    y = norm(x.clone())
    out["rmsnorm_bf16"] = dict(x=x, weight=norm.weight.detach().clone(), eps=1e-6, out=y)
    torch.save(out, os.path.join(OUT, "efficiency_ops.pt"))
    print("efficiency_ops.pt", list(out))


TINY_FLUX = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2,
                 attention_head_dim=128, num_attention_heads=2, joint_attention_dim=128,
                 pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))


def tiny_flux_inputs(s_img_hw=(8, 8), s_txt=16, seed=31):
    from oracle.flux import latent_image_ids
    h2, w2 = s_img_hw
    cfg = TINY_FLUX
    return dict(
        hidden_states=seeded((1, h2 * w2, cfg["in_channels"]), seed),
        encoder_hidden_states=seeded((1, s_txt, cfg["joint_attention_dim"]), seed + 1),
        pooled_projections=seeded((1, cfg["pooled_projection_dim"]), seed + 2),
        timestep=torch.tensor([0.5]), guidance=torch.tensor([4.0]),
        img_ids=latent_image_ids(h2, w2), txt_ids=torch.zeros(s_txt, 3))


def gen_flux_hybrid():
    load_by_path("src.transformer.flux", "src/transformer/flux/__init__.py") \
        if os.path.exists(os.path.join(REF, "src/transformer/flux/__init__.py")) else None
    from src.transformer.flux.base.model import FluxTransformer2DModel as RefFlux
    from oracle.flux import FluxTransformer2DModel as OracleFlux
    torch.manual_seed(0)
    ref = RefFlux(**TINY_FLUX).eval()
    sd = synthetic_state_dict(ref)
    missing = ref.load_state_dict(sd, strict=True)
    orc = OracleFlux(**TINY_FLUX).eval()
    assert sorted(orc.state_dict().keys()) == sorted(sd.keys()), "state-dict keys differ from reference"
    inp = tiny_flux_inputs()
    with torch.no_grad():
        out = ref(return_dict=False, **inp)[0]
    torch.save(dict(config=TINY_FLUX, seed=7, inputs=inp, out=out, keys=sorted(sd.keys())),
               os.path.join(OUT, "flux_hybrid.pt"))
    print("flux_hybrid.pt", tuple(out.shape), float(out.abs().mean()), missing)


def gen_flux_controlnet():
    """The reference FluxTransformer2DModel fed ControlNet residuals (`controlnet_block_samples` / `controlnet_single_block_samples`,
    `controlnet_blocks_repeat`; transformer/flux/base/model.py:594-640): a 3 + 3-block model with 2 double samples (interval
    ceil(3 / 2) = 2, or index mod 2 with `controlnet_blocks_repeat`) and 2 single samples — pins the residual placement of
    oracle.flux and the HIP model."""
    from src.transformer.flux.base.model import FluxTransformer2DModel as RefFlux
    cfg = dict(TINY_FLUX, num_layers=3, num_single_layers=3)
    ref = RefFlux(**cfg).eval()
    sd = synthetic_state_dict(ref, 17)
    ref.load_state_dict(sd, strict=True)
    inp = tiny_flux_inputs()
    n_img, dim = inp["hidden_states"].shape[1], cfg["num_attention_heads"] * cfg["attention_head_dim"]
    cd = [seeded((1, n_img, dim), 70 + i) * 0.5 for i in range(2)]
    cs = [seeded((1, n_img, dim), 80 + i) * 0.5 for i in range(2)]
    outs = {}
    with torch.no_grad():
        for name, kw in (("interval", dict(controlnet_block_samples=cd, controlnet_single_block_samples=cs)),
                         ("repeat", dict(controlnet_block_samples=cd, controlnet_blocks_repeat=True)),
                         ("none", dict())):
            outs[name] = ref(return_dict=False, **dict(inp), **kw)[0]
    assert float((outs["interval"] - outs["none"]).abs().max()) > 1e-3 and float((outs["repeat"] - outs["interval"]).abs().max()) > 1e-3
    torch.save(dict(config=cfg, seed=17, inputs=inp, double_seeds=(70, 71), single_seeds=(80, 81), scale=0.5, out=outs,
                    keys=sorted(sd.keys())), os.path.join(OUT, "flux_controlnet.pt"))
    print("flux_controlnet.pt", {k: float(v.abs().mean()) for k, v in outs.items()})


def gen_flux_ip_adapter():
    """The reference FluxTransformer2DModel whose double blocks run `FluxIPAdapterAttnProcessor` (transformer/flux/base/
    attention.py:115-265; the block adds its third output to the image stream after the feed-forward, model.py:291-309), fed the
    image-prompt tokens directly as `joint_attention_kwargs={"ip_hidden_states": [...]}` (what `encoder_hid_proj` would hand it,
    model.py:562-571 — that projection class is diffusers' and absent here): one adapter, and two adapters with different scales."""
    from src.transformer.flux.base.model import FluxTransformer2DModel as RefFlux
    from src.transformer.flux.base.attention import FluxIPAdapterAttnProcessor
    from oracle.flux import FluxTransformer2DModel as OracleFlux, FluxIPAdapterProcessor
    cfg = dict(TINY_FLUX)
    dim, ctx_dim = cfg["num_attention_heads"] * cfg["attention_head_dim"], cfg["joint_attention_dim"]
    inp = tiny_flux_inputs()
    cases = {}
    for name, num_tokens, scale in (("one", (4,), 0.7), ("two", (4, 6), [0.7, 0.35])):
        orc = OracleFlux(**cfg).eval()
        for blk in orc.transformer_blocks:
            blk.attn.processor = FluxIPAdapterProcessor(dim, ctx_dim, num_tokens, scale)
        sd = synthetic_state_dict(orc, 21)
        ref = RefFlux(**cfg).eval()
        for blk in ref.transformer_blocks:
            blk.attn.processor = FluxIPAdapterAttnProcessor(hidden_size=dim, cross_attention_dim=ctx_dim, num_tokens=num_tokens, scale=scale)
        miss = ref.load_state_dict(sd, strict=False)
        ip_keys = [k for k in sd if ".processor." in k]
        assert not miss.unexpected_keys and len(ip_keys) == 4 * len(num_tokens) * cfg["num_layers"], (miss, ip_keys)
        # `processor` is a plain attribute of the reference attention module when set this way: load its parameters explicitly
        for i, blk in enumerate(ref.transformer_blocks):
            blk.attn.processor.load_state_dict({k.split(".processor.")[1]: v for k, v in sd.items()
                                                if k.startswith(f"transformer_blocks.{i}.attn.processor.")}, strict=True)
        ips = [seeded((1, n, ctx_dim), 90 + j) for j, n in enumerate(num_tokens)]
        with torch.no_grad():
            out = ref(return_dict=False, joint_attention_kwargs={"ip_hidden_states": ips}, **dict(inp))[0]
            plain = ref(return_dict=False, joint_attention_kwargs={"ip_hidden_states": [torch.zeros_like(t) for t in ips]}, **dict(inp))[0]
        assert float((out - plain).abs().max()) > 1e-3
        cases[name] = dict(num_tokens=num_tokens, scale=scale, ip_seeds=[90 + j for j in range(len(num_tokens))], out=out,
                           keys=sorted(sd.keys()))
    torch.save(dict(config=cfg, seed=21, inputs=inp, cases=cases), os.path.join(OUT, "flux_ip_adapter.pt"))
    print("flux_ip_adapter.pt", {k: float(v["out"].abs().mean()) for k, v in cases.items()})


TINY_WAN = dict(patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=128, in_channels=16,
                out_channels=16, text_dim=64, freq_dim=256, ffn_dim=512, num_layers=2, cross_attn_norm=True,
                eps=1e-6)


def gen_wan_hybrid():
    """Reference WanTransformer3DModel run in float64: `x.float()` then copies, so the fp32 aliasing
    defect of InplaceRMSNorm does not trigger and the result is the intended math."""
    from src.transformer.wan.base.model import WanTransformer3DModel as RefWan
    from oracle.wan import WanTransformer3DModel as OracleWan
    ref = RefWan(**TINY_WAN, rope_max_seq_len=64).eval()
    orc = OracleWan(**TINY_WAN).eval()
    sd = synthetic_state_dict(orc, 9)
    assert sorted(sd.keys()) == sorted(ref.state_dict().keys()), \
        set(sd.keys()) ^ set(ref.state_dict().keys())
    ref.load_state_dict(sd, strict=True)
    ref = ref.double()
    inp = dict(hidden_states=seeded((1, 16, 3, 8, 12), 41), timestep=torch.tensor([500.0]),
               encoder_hidden_states=seeded((1, 20, 64), 42))
    with torch.no_grad():
        out = ref(hidden_states=inp["hidden_states"].double(), timestep=inp["timestep"].double(),
                  encoder_hidden_states=inp["encoder_hidden_states"].double(), return_dict=False)[0]
    torch.save(dict(config=TINY_WAN, seed=9, inputs=inp, out=out.float(), keys=sorted(sd.keys())),
               os.path.join(OUT, "wan_hybrid.pt"))
    print("wan_hybrid.pt", tuple(out.shape), float(out.abs().mean()))


def gen_wan_easycache():
    """The reference's EasyCache forward (`easycache_forward_`, transformer/wan/base/model.py:202-520, enabled by
    `enable_easy_cache(num_steps, thresh, ret_steps)` :1645-1672) on the tiny Wan model, float64, driven like a CFG sampler:
    per step one conditional (even call) and one unconditional (odd call) forward on the same latent, then an Euler-like
    update small enough that the input-change predictor skips some pairs.  Saved: every call's output and whether the blocks
    ran — pins oracle.easycache and, through it, `wan.mi355`'s `enable_easy_cache`."""
    import src.transformer.wan.base.model as RM
    from oracle.wan import WanTransformer3DModel as OracleWan
    ref = RM.WanTransformer3DModel(**TINY_WAN, rope_max_seq_len=64).eval()
    sd = synthetic_state_dict(OracleWan(**TINY_WAN), 9)
    ref.load_state_dict(sd, strict=True)
    ref = ref.double()
    ran = []
    blk0 = ref.blocks[0]
    orig_fwd = blk0.forward
    blk0.forward = lambda *a, **k: (ran.append(1), orig_fwd(*a, **k))[1]
    n, ret_steps, thresh, dt, g = 10, 2, 1.0, 0.02, 3.0
    x = seeded((1, 16, 3, 8, 12), 41).double()
    txt_c, txt_u = seeded((1, 20, 64), 42).double(), seeded((1, 20, 64), 43).double()
    ts = [float(t) for t in torch.linspace(900.0, 100.0, n)]
    ref.enable_easy_cache(n, thresh, ret_steps, should_reset_global_cache=True)
    outs, computed = [], []
    with torch.no_grad():
        for i in range(n):
            pair = []
            for txt in (txt_c, txt_u):
                k0 = len(ran)
                o = ref(hidden_states=x, timestep=torch.tensor([ts[i]], dtype=torch.float64), encoder_hidden_states=txt, return_dict=False)[0]
                computed.append(len(ran) > k0)
                outs.append(o.float().clone())
                pair.append(o.double())
            x = x - dt * (pair[1] + g * (pair[0] - pair[1]))
    ref.disable_easy_cache()
    assert any(computed[2 * ret_steps:]) and not all(computed), computed
    torch.save(dict(config=TINY_WAN, seed=9, n=n, ret_steps=ret_steps, thresh=thresh, dt=dt, guidance=g, timesteps=ts,
                    x_seed=41, txt_seeds=(42, 43), outs=outs, computed=computed, x_final=x.float()),
               os.path.join(OUT, "wan_easycache.pt"))
    print("wan_easycache.pt", "computed:", "".join("C" if c else "-" for c in computed))



TINY_QWEN = dict(patch_size=2, in_channels=64, out_channels=16, num_layers=2, attention_head_dim=128,
                 num_attention_heads=2, joint_attention_dim=64, guidance_embeds=False,
                 axes_dims_rope=(16, 56, 56))


def gen_qwen_hybrid():
    """Reference QwenImageTransformer2DModel (edit layout: target image + one condition image)."""
    # the reference does `from src.attention import attention_register`
    import src.attention  # noqa: F401
    from src.transformer.qwenimage.base.model import QwenImageTransformer2DModel as RefQwen
    from oracle.qwenimage import QwenImageTransformer2DModel as OracleQwen
    ref = RefQwen(**TINY_QWEN).eval()
    orc = OracleQwen(**TINY_QWEN).eval()
    sd = synthetic_state_dict(orc, 11)
    assert sorted(sd.keys()) == sorted(ref.state_dict().keys()), set(sd) ^ set(ref.state_dict())
    ref.load_state_dict(sd, strict=True)
    shapes = [[(1, 6, 8), (1, 4, 6)]]
    n_img = 6 * 8 + 4 * 6
    inp = dict(hidden_states=seeded((1, n_img, 64), 51), encoder_hidden_states=seeded((1, 13, 64), 52),
               timestep=torch.tensor([0.5]), img_shapes=shapes, txt_seq_lens=[13])
    with torch.no_grad():
        out = ref(hidden_states=inp["hidden_states"], encoder_hidden_states=inp["encoder_hidden_states"],
                  encoder_hidden_states_mask=torch.ones(1, 13), timestep=inp["timestep"], img_shapes=shapes,
                  txt_seq_lens=[13], return_dict=False)[0]
    torch.save(dict(config=TINY_QWEN, seed=11, inputs=inp, out=out, keys=sorted(sd.keys())),
               os.path.join(OUT, "qwen_hybrid.pt"))
    print("qwen_hybrid.pt", tuple(out.shape), float(out.abs().mean()))


def gen_qwen_variants():
    """The reference QwenImageTransformer2DModel with `zero_cond_t` (a second conditioning row at t = 0 that modulates the tokens
    of the condition images: transformer/qwenimage/base/model.py:640-677, :692-702, :912-923, :980-981) and
    `use_additional_t_cond` (`addition_t_embedding`, :164-182), alone and together, on the edit layout (target + 2 condition
    images) — pins oracle.qwenimage's restatement of both switches and, through it, `qwenimage.mi355`."""
    import src.attention  # noqa: F401
    from src.transformer.qwenimage.base.model import QwenImageTransformer2DModel as RefQwen
    from oracle.qwenimage import QwenImageTransformer2DModel as OracleQwen
    shapes = [[(1, 6, 8), (1, 4, 6), (1, 2, 4)]]
    n_img = 6 * 8 + 4 * 6 + 2 * 4
    inp = dict(hidden_states=seeded((1, n_img, 64), 53), encoder_hidden_states=seeded((1, 13, 64), 54),
               timestep=torch.tensor([0.625]), img_shapes=shapes, txt_seq_lens=[13])
    out = {}
    for name, kw, atc in (("zero_cond_t", dict(zero_cond_t=True), None),
                          ("additional_t_cond", dict(use_additional_t_cond=True), torch.tensor([1])),
                          ("both", dict(zero_cond_t=True, use_additional_t_cond=True), torch.tensor([1]))):
        cfg = dict(TINY_QWEN, **kw)
        ref = RefQwen(**cfg).eval()
        sd = synthetic_state_dict(OracleQwen(**cfg), 12)
        assert sorted(sd.keys()) == sorted(ref.state_dict().keys()), set(sd) ^ set(ref.state_dict())
        ref.load_state_dict(sd, strict=True)
        with torch.no_grad():
            y = ref(hidden_states=inp["hidden_states"], encoder_hidden_states=inp["encoder_hidden_states"],
                    encoder_hidden_states_mask=torch.ones(1, 13), timestep=inp["timestep"], img_shapes=shapes, txt_seq_lens=[13],
                    additional_t_cond=atc, return_dict=False)[0]
        out[name] = dict(config=cfg, additional_t_cond=atc, out=y, keys=sorted(sd.keys()))
    assert float((out["both"]["out"] - out["zero_cond_t"]["out"]).abs().max()) > 1e-3
    torch.save(dict(seed=12, inputs=inp, cases=out), os.path.join(OUT, "qwen_variants.pt"))
    print("qwen_variants.pt", {k: float(v["out"].abs().mean()) for k, v in out.items()})


TINY_VAE = dict(base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=1,
                temperal_downsample=[False, True, True])


def install_vae_stubs():
    class _Out:
        def __init__(self, sample=None, latent_dist=None):
            self.sample, self.latent_dist = sample, latent_dist

    _mod("diffusers.models.activations", get_activation=lambda name: nn.SiLU())
    sys.modules["diffusers.models.modeling_outputs"].AutoencoderKLOutput = _Out
    ae = _mod("diffusers.models.autoencoders")
    ae.__path__ = []
    _mod("diffusers.models.autoencoders.vae", AutoencoderMixin=type("AutoencoderMixin", (), {}),
         DecoderOutput=_Out, DiagonalGaussianDistribution=type("DGD", (), {}))


def gen_vae_wan():
    """The REFERENCE AutoencoderKLWan (streaming decoder with feat_cache) run here, untiled and tiled."""
    install_vae_stubs()
    ref_mod = load_by_path("ref_vae_wan", "src/vae/wan/model.py")
    from oracle.vae_wan import AutoencoderKLWanDecoder
    ref = ref_mod.AutoencoderKLWan(**TINY_VAE).eval()
    orc = AutoencoderKLWanDecoder(**TINY_VAE)
    sd = vae_synthetic_state_dict(orc, 13)
    missing = ref.load_state_dict(sd, strict=False)      # encoder / quant_conv keep their init
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all(k.startswith(("encoder.", "quant_conv.")) for k in missing.missing_keys)
    z = seeded((1, 16, 3, 16, 20), 61)
    with torch.no_grad():
        untiled = ref.decode(z, return_dict=False)[0]
        ref.enable_tiling(tile_sample_min_height=96, tile_sample_min_width=96,
                          tile_sample_stride_height=64, tile_sample_stride_width=64)
        tiled = ref.decode(z, return_dict=False)[0]
        zn = ref.denormalize_latents(z)
    assert float((tiled - untiled).abs().max()) > 1e-3    # tiling IS part of the numerical contract
    torch.save(dict(config=TINY_VAE, seed=13, z_shape=(1, 16, 3, 16, 20), z_seed=61,
                    tile=(96, 96, 64, 64), untiled=untiled.to(torch.bfloat16), tiled=tiled.to(torch.bfloat16),
                    untiled_f32_sample=untiled[0, :, :, ::8, ::8].clone(), tiled_f32_sample=tiled[0, :, :, ::8, ::8].clone(),
                    denorm_sample=zn[0, :, 0, 0, 0].clone(), keys=sorted(sd.keys())),
               os.path.join(OUT, "vae_wan.pt"))
    print("vae_wan.pt", tuple(untiled.shape), float(untiled.abs().mean()), float((tiled - untiled).abs().max()))


def gen_vae_wan_encode():
    """The REFERENCE AutoencoderKLWan._encode (streaming encoder with feat_cache: first frame, then chunks of 4) on a 9-frame
    clip and on a single image, untiled and tiled; outputs are the posterior parameters (mean | logvar)."""
    install_vae_stubs()
    ref_mod = load_by_path("ref_vae_wan_enc", "src/vae/wan/model.py")
    from oracle.vae_wan import AutoencoderKLWanEncoder
    ref = ref_mod.AutoencoderKLWan(**TINY_VAE).eval()
    orc = AutoencoderKLWanEncoder(**TINY_VAE)
    sd = vae_synthetic_state_dict(orc, 15)
    missing = ref.load_state_dict(sd, strict=False)      # decoder / post_quant_conv keep their init
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all(k.startswith(("decoder.", "post_quant_conv.")) for k in missing.missing_keys)
    x = seeded((1, 3, 9, 64, 80), 63)
    tile = dict(tile_sample_min_height=48, tile_sample_min_width=48, tile_sample_stride_height=32,
                tile_sample_stride_width=32)
    with torch.no_grad():
        video = ref._encode(x)
        image = ref._encode(x[:, :, :1])
        ref.enable_tiling(**tile)
        video_tiled = ref._encode(x)
        image_tiled = ref._encode(x[:, :, :1])
        zn = ref.normalize_latents(video[:, :16])
    assert float((video_tiled - video).abs().max()) > 1e-3
    torch.save(dict(config=TINY_VAE, seed=15, x_shape=(1, 3, 9, 64, 80), x_seed=63, tile=(48, 48, 32, 32), video=video,
                    image=image, video_tiled=video_tiled, image_tiled=image_tiled, norm_sample=zn[0, :, 0, 0, 0].clone(),
                    keys=sorted(sd.keys())), os.path.join(OUT, "vae_wan_encode.pt"))
    print("vae_wan_encode.pt", tuple(video.shape), tuple(image.shape), float(video.abs().mean()),
          float((video_tiled - video).abs().max()))


TINY_VAE_HY15 = dict(in_channels=3, out_channels=3, latent_channels=32, block_out_channels=(32, 64, 64, 128, 128),
                     layers_per_block=1, spatial_compression_ratio=16, temporal_compression_ratio=4)


def gen_vae_hunyuan15():
    """The REFERENCE AutoencoderKLHunyuanVideo15 (vae/hunyuanvideo15/model.py) decode, untiled and tiled (8x8-latent tiles,
    stride 6, 32-px blends), on a small channel plan.  Stubs carry no arithmetic: the light-VAE class, the download mixin
    and the components-path helper are never used on the decode path."""
    install_vae_stubs()
    import src.attention  # noqa: F401
    d = _mod("src.utils.defaults", get_components_path=lambda *a, **k: "/tmp")
    t = _mod("src.vae.tae")
    t.__path__ = []
    _mod("src.vae.tae.model", TAEHV=type("TAEHV", (), {}))
    _mod("src.mixins.download_mixin", DownloadMixin=type("DownloadMixin", (), {}))
    ref_mod = load_by_path("ref_vae_hy15", "src/vae/hunyuanvideo15/model.py")
    from oracle.vae_hunyuan15 import AutoencoderKLHunyuanVideo15 as Orc
    ref = ref_mod.AutoencoderKLHunyuanVideo15(**TINY_VAE_HY15).eval()
    orc = Orc(**TINY_VAE_HY15)
    sd = vae_synthetic_state_dict(orc, 17)
    missing = ref.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all(k.startswith("encoder.") for k in missing.missing_keys), missing.missing_keys[:5]
    z = seeded((1, 32, 3, 10, 14), 71)
    with torch.no_grad():
        untiled = ref.decode(z, return_dict=False)[0]
        ref.enable_tiling()
        tiled = ref.decode(z, return_dict=False)[0]
    assert float((tiled - untiled).abs().max()) > 1e-3
    torch.save(dict(config=TINY_VAE_HY15, seed=17, z_shape=(1, 32, 3, 10, 14), z_seed=71,
                    untiled=untiled.to(torch.bfloat16), tiled=tiled.to(torch.bfloat16),
                    untiled_f32_sample=untiled[0, :, :, ::8, ::8].clone(), tiled_f32_sample=tiled[0, :, :, ::8, ::8].clone(),
                    keys=sorted(k for k in sd if k.startswith("decoder."))), os.path.join(OUT, "vae_hunyuan15.pt"))
    print("vae_hunyuan15.pt", tuple(untiled.shape), float(untiled.abs().mean()), float((tiled - untiled).abs().max()))


def gen_vae_hunyuan15_encode():
    """The REFERENCE AutoencoderKLHunyuanVideo15 ENCODE path (image-to-video conditioning): one frame and a 5-frame clip,
    untiled, and a tiled encode (tile 64 px, stride 48, 1-latent blends via the constructor-free attributes)."""
    install_vae_stubs()
    import src.attention  # noqa: F401
    _mod("src.utils.defaults", get_components_path=lambda *a, **k: "/tmp")
    t = _mod("src.vae.tae")
    t.__path__ = []
    _mod("src.vae.tae.model", TAEHV=type("TAEHV", (), {}))
    _mod("src.mixins.download_mixin", DownloadMixin=type("DownloadMixin", (), {}))
    ref_mod = load_by_path("ref_vae_hy15e", "src/vae/hunyuanvideo15/model.py")
    from oracle.vae_hunyuan15 import AutoencoderKLHunyuanVideo15 as Orc
    ref = ref_mod.AutoencoderKLHunyuanVideo15(**TINY_VAE_HY15).eval()
    orc = Orc(**TINY_VAE_HY15)
    sd = vae_synthetic_state_dict(orc, 19)
    res = ref.load_state_dict(sd, strict=True)
    out = dict(config=TINY_VAE_HY15, seed=19, keys=sorted(sd.keys()))
    with torch.no_grad():
        for name, shape, seed in (("image", (1, 3, 1, 64, 96), 81), ("clip", (1, 3, 5, 64, 64), 82)):
            x = seeded(shape, seed).clamp(-1, 1)
            h = ref._encode(x)
            out[name] = dict(shape=shape, seed=seed, moments=h)
        # tiled: 128 x 160 px with 64-px tiles (4 latents), stride 48 px, blend 1 latent
        ref.enable_tiling(tile_sample_min_height=64, tile_sample_min_width=64, tile_latent_min_height=4,
                          tile_latent_min_width=4)
        x = seeded((1, 3, 1, 128, 160), 83).clamp(-1, 1)
        out["tiled"] = dict(shape=(1, 3, 1, 128, 160), seed=83, tile=64, moments=ref._encode(x))
    torch.save(out, os.path.join(OUT, "vae_hunyuan15_encode.pt"))
    print("vae_hunyuan15_encode.pt", {k: tuple(v["moments"].shape) for k, v in out.items() if isinstance(v, dict) and "moments" in v}, res)


def gen_vae_taehv():
    """The REFERENCE TAEHV decoder behind `use_light_vae` (vae/tae/model.py, wrapped by AutoencoderKLHunyuanVideo15Light,
    vae/hunyuanvideo15/model.py:1163-1234) on seeded weights: sequential mode (what the engine runs, parallel=False) and
    parallel mode, a 3-latent-frame clip and a single latent frame.  Channel widths are fixed by the reference class
    (256/128/64/64); the spatial size is what keeps this small."""
    install_vae_stubs()
    import src.attention  # noqa: F401
    _mod("src.utils.defaults", get_components_path=lambda *a, **k: "/tmp")
    _mod("src.mixins.download_mixin", DownloadMixin=type("DownloadMixin", (), {}))
    for name in ("src.vae.tae", "src.vae.tae.model"):
        sys.modules.pop(name, None)
    t = _mod("src.vae.tae")
    t.__path__ = []
    tae = load_by_path("src.vae.tae.model", "src/vae/tae/model.py")
    ref_mod = load_by_path("ref_vae_hy15_light", "src/vae/hunyuanvideo15/model.py")
    from oracle.vae_taehv import AutoencoderKLHunyuanVideo15Light as Orc
    ref = ref_mod.AutoencoderKLHunyuanVideo15Light(taehv_checkpoint_path=None).eval()
    assert isinstance(ref.taehv, tae.TAEHV)
    orc = Orc().eval()
    sd = vae_synthetic_state_dict(orc, 29)                       # taehv.decoder.* (the product's keys)
    full = dict(ref.state_dict())
    enc = vae_synthetic_state_dict(ref, 30)
    full.update({k: v for k, v in enc.items() if k.startswith("taehv.encoder.")})
    full.update(sd)
    res = ref.load_state_dict(full, strict=True)
    assert sorted(k for k in ref.state_dict() if k.startswith("taehv.decoder.")) == sorted(sd.keys())
    out = dict(seed=29, keys=sorted(sd.keys()), all_keys=sorted(full.keys()), scaling_factor=ref.scaling_factor)
    with torch.no_grad():
        for name, shape, seed in (("clip", (1, 32, 3, 4, 6), 84), ("frame", (1, 32, 1, 6, 4), 85)):
            z = seeded(shape, seed) * 1.5
            seq = ref.decode(z.clone(), parallel=False, show_progress_bar=False)
            par = ref.decode(z.clone(), parallel=True, show_progress_bar=False)
            assert seq.shape[0] == 1 and seq.dim() == 6          # the reference's extra leading axis; callers take [0]
            out[name] = dict(shape=shape, seed=seed, scale=1.5, sequential=seq[0].clone(),
                             parallel_max_abs_diff=float((seq - par).abs().max()))
    torch.save(out, os.path.join(OUT, "vae_taehv.pt"))
    print("vae_taehv.pt", {k: tuple(v["sequential"].shape) for k, v in out.items() if isinstance(v, dict)}, res,
          {k: v["parallel_max_abs_diff"] for k, v in out.items() if isinstance(v, dict)})


def gen_vae_taehv_encode():
    """The REFERENCE TAEHV encoder (vae/tae/model.py:214-236, encode_video :299-316), model_type "hy15": a 9-frame clip
    (padded to 12 by repeating the last frame) and a 4-frame one, parallel and sequential mode."""
    for name in ("src.vae.tae", "src.vae.tae.model"):
        sys.modules.pop(name, None)
    t = _mod("src.vae.tae")
    t.__path__ = []
    tae = load_by_path("src.vae.tae.model", "src/vae/tae/model.py")
    from oracle.vae_taehv import TAEHVEncoder
    ref = tae.TAEHV(checkpoint_path=None, model_type="hy15", latent_channels=32, patch_size=2).eval()
    orc = TAEHVEncoder().eval()
    sd = vae_synthetic_state_dict(orc, 31)
    assert sorted(sd.keys()) == sorted(k for k in ref.state_dict() if k.startswith("encoder."))
    res = ref.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith("decoder.") for k in res.missing_keys)
    out = dict(seed=31, keys=sorted(sd.keys()))
    with torch.no_grad():
        for name, shape, seed in (("clip9", (1, 9, 3, 64, 96), 86), ("clip4", (1, 4, 3, 32, 32), 87)):
            x = (seeded(shape, seed) * 0.25 + 0.5).clamp(0, 1)
            par = ref.encode_video(x.clone(), parallel=True, show_progress_bar=False)
            seq = ref.encode_video(x.clone(), parallel=False, show_progress_bar=False)
            out[name] = dict(shape=shape, seed=seed, latents=par.clone(), sequential_max_abs_diff=float((par - seq).abs().max()))
    torch.save(out, os.path.join(OUT, "vae_taehv_encode.pt"))
    print("vae_taehv_encode.pt", {k: (tuple(v["latents"].shape), v["sequential_max_abs_diff"]) for k, v in out.items() if isinstance(v, dict)})


def gen_unipc():
    """In-tree UniPC (reference scheduler/unipc.py) trajectory: 6 steps, shift 3, fp32 latents."""
    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    sch = _mod("diffusers.schedulers")
    sch.__path__ = []
    _mod("diffusers.schedulers.scheduling_utils", KarrasDiffusionSchedulers=[], SchedulerMixin=type("SM", (), {}),
         SchedulerOutput=SchedulerOutput)
    mod = load_by_path("ref_unipc", "src/scheduler/unipc.py")
    s = mod.UniPCMultistepScheduler(shift=3.0)
    s.set_timesteps(6)
    x = seeded((1, 4, 3, 5, 5), 71)
    vs = [seeded((1, 4, 3, 5, 5), 72 + i) for i in range(6)]
    traj = []
    for i, t in enumerate(s.timesteps):
        x = s.step(vs[i], t, x, return_dict=False)[0]
        traj.append(x.clone())
    torch.save(dict(steps=6, shift=3.0, shape=(1, 4, 3, 5, 5), seed=71, timesteps=s.timesteps.clone(),
                    sigmas=s.sigmas.clone(), traj=traj), os.path.join(OUT, "unipc.pt"))
    print("unipc.pt", s.timesteps.tolist(), float(traj[-1].abs().mean()))


TINY_HY15 = dict(in_channels=9, out_channels=8, num_attention_heads=2, attention_head_dim=128, num_layers=2,
                 num_refiner_layers=2, mlp_ratio=4.0, patch_size=1, patch_size_t=1, qk_norm="rms_norm",
                 text_embed_dim=64, text_embed_2_dim=128, image_embed_dim=64, rope_theta=256.0,
                 rope_axes_dim=(16, 56, 56))


def hy15_inputs():
    m1 = torch.ones(1, 12)
    m1[0, 9:] = 0                                   # mllm prompt: 9 valid tokens, right padded
    m2 = torch.ones(1, 8)
    m2[0, 5:] = 0                                   # byt5 glyph tokens: 5 valid
    return dict(hidden_states=seeded((1, 9, 2, 4, 6), 61), timestep=torch.tensor([500.0]),
                encoder_hidden_states=seeded((1, 12, 64), 62), encoder_attention_mask=m1,
                encoder_hidden_states_2=seeded((1, 8, 128), 63), encoder_attention_mask_2=m2)


def gen_hunyuan15_hybrid():
    """Reference HunyuanVideo15Transformer3DModel (transformer/hunyuanvideo15/base/model.py) in float64 (InplaceRMSNorm,
    see gen_wan_hybrid), two cases: t2v (image_embeds all zero -> image tokens masked to the back) and i2v."""
    import src.attention  # noqa: F401
    from src.transformer.hunyuanvideo15.base.model import HunyuanVideo15Transformer3DModel as Ref
    from oracle.hunyuan15 import HunyuanVideo15Transformer3DModel as Orc
    ref = Ref(**TINY_HY15).eval()
    orc = Orc(**TINY_HY15).eval()
    sd = synthetic_state_dict(orc, 15)
    assert sorted(sd.keys()) == sorted(ref.state_dict().keys()), set(sd) ^ set(ref.state_dict())
    ref.load_state_dict(sd, strict=True)
    ref = ref.double()
    inp = hy15_inputs()
    outs = {}
    for name, img in (("t2v", torch.zeros(1, 3, 64)), ("i2v", seeded((1, 3, 64), 64))):
        with torch.no_grad():
            outs[name] = ref(hidden_states=inp["hidden_states"].double(), timestep=inp["timestep"].double(),
                             encoder_hidden_states=inp["encoder_hidden_states"].double(),
                             encoder_attention_mask=inp["encoder_attention_mask"],
                             encoder_hidden_states_2=inp["encoder_hidden_states_2"].double(),
                             encoder_attention_mask_2=inp["encoder_attention_mask_2"],
                             image_embeds=img.double(), return_dict=False)[0].float()
        print("hunyuan15_hybrid", name, tuple(outs[name].shape), float(outs[name].abs().mean()))
    torch.save(dict(config=TINY_HY15, seed=15, inputs=inp, image_embeds_i2v=seeded((1, 3, 64), 64), out=outs,
                    keys=sorted(sd.keys())), os.path.join(OUT, "hunyuan15_hybrid.pt"))


def gen_hunyuan15_meanflow():
    """The reference HunyuanVideo15Transformer3DModel built with use_meanflow=True (a second timestep embedder for
    `timestep_r`, model.py:234-268), float64, i2v inputs, timestep 500 with timestep_r 300 and with timestep_r None."""
    import src.attention  # noqa: F401
    from src.transformer.hunyuanvideo15.base.model import HunyuanVideo15Transformer3DModel as Ref
    from oracle.hunyuan15 import HunyuanVideo15Transformer3DModel as Orc
    cfg = dict(TINY_HY15, use_meanflow=True)
    ref = Ref(**cfg).eval()
    orc = Orc(**cfg).eval()
    sd = synthetic_state_dict(orc, 17)
    assert sorted(sd.keys()) == sorted(ref.state_dict().keys()), set(sd) ^ set(ref.state_dict())
    ref.load_state_dict(sd, strict=True)
    ref = ref.double()
    inp = hy15_inputs()
    img = seeded((1, 3, 64), 64)
    outs = {}
    for name, tr in (("r300", torch.tensor([300.0])), ("none", None)):
        with torch.no_grad():
            outs[name] = ref(hidden_states=inp["hidden_states"].double(), timestep=inp["timestep"].double(),
                             timestep_r=None if tr is None else tr.double(),
                             encoder_hidden_states=inp["encoder_hidden_states"].double(),
                             encoder_attention_mask=inp["encoder_attention_mask"],
                             encoder_hidden_states_2=inp["encoder_hidden_states_2"].double(),
                             encoder_attention_mask_2=inp["encoder_attention_mask_2"],
                             image_embeds=img.double(), return_dict=False)[0].float()
        print("hunyuan15_meanflow", name, tuple(outs[name].shape), float(outs[name].abs().mean()))
    torch.save(dict(config=cfg, seed=17, inputs=inp, image_embeds=img, timestep_r=torch.tensor([300.0]), out=outs,
                    keys=sorted(sd.keys())), os.path.join(OUT, "hunyuan15_meanflow.pt"))


def install_converter_stubs():
    """What the reference's converters import besides themselves, for gen_lora and gen_convert alike: the two diffusers rename
    tables of the legacy LoRA formats (not exercised) and src.quantize.ggml_ops, whose cat / chunk / split are torch's for plain
    tensors (their GGML branches need the gguf package).  Every `src.converters.*` module an earlier generator imported is
    dropped first, so that the reference's `from src.quantize.ggml_ops import ...` binds to THESE stubs whatever ran before
    (gen_convert after gen_lora saw gen_lora's `ggml_chunk = None`)."""
    for name in [n for n in sys.modules if n == "src.converters" or n.startswith("src.converters.")]:
        sys.modules.pop(name, None)
    _mod("diffusers.utils.state_dict_utils", DIFFUSERS_TO_PEFT={}, DIFFUSERS_OLD_TO_PEFT={})
    q = _mod("src.quantize")
    q.__path__ = []
    _mod("src.quantize.ggml_ops", ggml_cat=lambda ts, dim=0: torch.cat(list(ts), dim=dim),
         ggml_chunk=lambda t, n, dim=0: torch.chunk(t, n, dim=dim), ggml_split=lambda t, s, dim=0: torch.split(t, s, dim=dim))
    import diffusers
    if not hasattr(diffusers, "ModelMixin"):
        diffusers.ModelMixin = type("ModelMixin", (), {})
    cp = _mod("src.converters")
    cp.__path__ = [os.path.join(REF, "src/converters")]


def gen_lora():
    """Reference LoraConverter on seeded state dicts (stubs: install_converter_stubs)."""
    install_converter_stubs()
    mod = load_by_path("ref_lora_converter", "src/lora/lora_converter.py")
    conv = mod.LoraConverter()
    cases = {}
    # PEFT keys with alpha, "transformer." prefix, two modules of different rank
    peft = {
        "transformer.blocks.0.attn1.to_q.lora_A.weight": seeded((8, 32), 901),
        "transformer.blocks.0.attn1.to_q.lora_B.weight": seeded((64, 8), 902),
        "transformer.blocks.0.attn1.to_q.alpha": torch.tensor(4.0),
        "transformer.blocks.1.ffn.net.2.lora_A.weight": seeded((4, 128), 903),
        "transformer.blocks.1.ffn.net.2.lora_B.weight": seeded((32, 4), 904),
        "transformer.blocks.1.ffn.net.2.alpha": torch.tensor(0.25),
        "transformer.blocks.1.attn1.to_k.lora_A.weight": seeded((16, 32), 905),   # no alpha
        "transformer.blocks.1.attn1.to_k.lora_B.weight": seeded((64, 16), 906),
    }
    base = {
        "diffusion_model.blocks.0.ffn.net.0.proj.lora_down.weight": seeded((4, 32), 911),
        "diffusion_model.blocks.0.ffn.net.0.proj.lora_up.weight": seeded((128, 4), 912),
        "diffusion_model.blocks.0.ffn.net.0.proj.alpha": torch.tensor(8.0),
        "diffusion_model.blocks.0.attn2.to_out.0.lora_down.weight": seeded((16, 32), 913),
        "diffusion_model.blocks.0.attn2.to_out.0.lora_up.weight": seeded((32, 16), 914),
        "diffusion_model.blocks.0.attn2.to_out.0.alpha": torch.tensor(1.0),
    }
    model_keys = ["blocks.0.attn1.to_q.weight", "blocks.1.ffn.net.2.weight", "blocks.1.attn1.to_k.weight",
                  "blocks.0.ffn.net.0.proj.weight", "blocks.0.attn2.to_out.0.weight"]
    for name, sd in (("peft", peft), ("base", base)):
        out = conv.convert({k: v.clone() for k, v in sd.items()}, model_keys=list(model_keys))
        cases[name] = dict(inp=sd, out={k: v.clone() for k, v in out.items()})
        print("lora", name, sorted(out.keys())[:3], "...")
    torch.save(dict(cases=cases, model_keys=model_keys,
                    alpha_scales={(r, a): conv.get_alpha_scales(torch.zeros(r, 1), a)
                                  for r in (4, 8, 16, 64) for a in (0.25, 1.0, 4.0, 8.0, 128.0)}),
               os.path.join(OUT, "lora_convert.pt"))


def gen_fp_scaled():
    """Reference FP-scaled dequantisation.  `FPScaledLinear` is instantiated only to reach the bound method
    `_scale_and_cast_weight`; transformer_engine is optional in the reference and absent here."""
    mod = load_by_path("ref_scaled_layer", "src/quantize/scaled_layer.py")
    lin = mod.FPScaledLinear(16, 8, bias=False, compute_dtype=torch.bfloat16)
    cases = {}
    for name, dt in (("e4m3fn", torch.float8_e4m3fn), ("e5m2", torch.float8_e5m2)):
        allcodes = torch.arange(256, dtype=torch.uint8).view(dt).reshape(8, 32)          # every code point once
        w = (seeded((24, 40), 951) * 3.0).to(dt)
        s_scalar = torch.tensor(0.0173)
        s_row = (seeded((24, 1), 952).abs() * 0.02 + 0.001)
        s_all = (seeded((8, 1), 953).abs() * 0.5 + 0.01)
        cases[name] = dict(
            w=w.view(torch.uint8), w_all=allcodes.view(torch.uint8), s_scalar=s_scalar, s_row=s_row, s_all=s_all,
            out_scalar=mod.fp8_activation_dequant(w, s_scalar, torch.bfloat16),
            out_row=mod.fp8_activation_dequant(w, s_row, torch.bfloat16),
            out_all=mod.fp8_activation_dequant(allcodes, s_all, torch.bfloat16),
            out_method=lin._scale_and_cast_weight(w, s_scalar, target_dtype=torch.bfloat16))
        print("fp_scaled", name, float(cases[name]["out_row"].float().abs().mean()))
    # (a non-fp8 weight WITH a scale_weight raises TypeError in the reference, scaled_layer.py:525 — no fixture)
    torch.save(cases, os.path.join(OUT, "fp_scaled.pt"))


TINY_T5 = dict(vocab_size=100, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2,
               relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
               feed_forward_proj="gated-gelu")
TINY_CLIP = dict(vocab_size=100, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                 max_position_embeddings=24, layer_norm_eps=1e-5, hidden_act="quick_gelu", eos_token_id=2)


def gen_text_encoders():
    """The text encoders the reference resolves by class name from `transformers` (text_encoder/text_encoder.py:24-82,
    :335-342): T5EncoderModel, UMT5EncoderModel, CLIPTextModel of the transformers build installed HERE, small configs,
    fp32, with and without a padding mask.  The third-party package itself is the reference for this row."""
    import transformers
    from oracle import text_encoders as OT
    out = dict(transformers_version=transformers.__version__, t5_config=TINY_T5, clip_config=TINY_CLIP)
    S = 21
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 100, (2, S), generator=g)
    mask = torch.ones(2, S, dtype=torch.long)
    mask[1, 13:] = 0
    ids[1, 13:] = 0                                            # pad id
    out["t5_ids"], out["t5_mask"] = ids, mask
    for name, hf_cls, cfg_cls, extra in (("t5", transformers.T5EncoderModel, transformers.T5Config, {}),
                                         ("t5_relu", transformers.T5EncoderModel, transformers.T5Config,
                                          dict(feed_forward_proj="relu")),
                                         ("umt5", transformers.UMT5EncoderModel, transformers.UMT5Config, {})):
        kw = {**TINY_T5, **extra}
        orc = OT.T5EncoderModel(**kw, per_layer_bias=(name == "umt5")).eval()
        sd = text_encoder_state_dict(orc, 23, 24, "layer_norm.weight")
        sd.pop("encoder.embed_tokens.weight", None)              # tied to `shared.weight`
        hf = hf_cls(cfg_cls(**kw, is_decoder=False, use_cache=False, dropout_rate=0.0)).eval().float()
        res = hf.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        assert all(k in ("encoder.embed_tokens.weight",) for k in res.missing_keys), res.missing_keys
        with torch.no_grad():
            a = hf(input_ids=ids, output_hidden_states=True)
            b = hf(input_ids=ids, attention_mask=mask, output_hidden_states=True)
        out[name] = dict(keys=sorted(sd.keys()), seed=23, last=a.last_hidden_state, hidden1=a.hidden_states[1],
                         n_hidden=len(a.hidden_states), last_masked=b.last_hidden_state)
        print("text", name, float(a.last_hidden_state.abs().mean()), float((a.last_hidden_state - b.last_hidden_state).abs().max()))
    S = 19
    ids = torch.randint(3, 99, (2, S), generator=g)
    ids[0, 11], ids[1, 16] = 99, 99                              # EOS = highest id (legacy argmax pooling, eos_token_id == 2)
    ids[0, 12:], ids[1, 17:] = 0, 0
    mask = (ids != 0).long()
    orc = OT.CLIPTextModel(**TINY_CLIP).eval()
    sd = text_encoder_state_dict(orc, 29, 30, "layer_norm")
    hf = transformers.CLIPTextModel(transformers.CLIPTextConfig(**TINY_CLIP, attention_dropout=0.0, bos_token_id=1,
                                                                pad_token_id=0)).eval().float()
    # checkpoints (and transformers 4.57, the reference's pin) carry a `text_model.` prefix; 5.x modules dropped it
    flat = not any(k.startswith("text_model.") for k in hf.state_dict())
    res = hf.load_state_dict({(k[len("text_model."):] if flat else k): v for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all("position_ids" in k for k in res.missing_keys), res.missing_keys
    with torch.no_grad():
        a = hf(input_ids=ids, output_hidden_states=True)
        b = hf(input_ids=ids, attention_mask=mask, output_hidden_states=True)
    out["clip_ids"], out["clip_mask"] = ids, mask
    out["clip"] = dict(keys=sorted(sd.keys()), seed=29, last=a.last_hidden_state, pooled=a.pooler_output,
                       hidden_m2=a.hidden_states[-2], n_hidden=len(a.hidden_states), last_masked=b.last_hidden_state,
                       pooled_masked=b.pooler_output)
    print("text clip", float(a.pooler_output.abs().mean()), float((a.pooler_output - b.pooler_output).abs().max()))
    torch.save(out, os.path.join(OUT, "text_encoders.pt"))


TINY_QWEN_VL_TEXT = dict(vocab_size=200, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                         num_key_value_heads=1, rms_norm_eps=1e-6, rope_theta=1000000.0)
TINY_QWEN_VL_VISION = dict(depth=2, hidden_size=320, intermediate_size=172, num_heads=4, in_channels=3, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1])


def gen_qwen2_5_vl():
    """transformers.Qwen2_5_VLForConditionalGeneration (the class the QwenImage manifests name) on a small config: a
    right-padded text-only batch, and one prompt with two images (vision tower with windowed + full attention blocks,
    head dim 80 and intermediate size 172 chosen to have the production model's divisibility quirks, image-embedding
    scatter, 3-D RoPE positions).  fp32."""
    import transformers
    from oracle.qwen2_5_vl import Qwen2_5_VLForConditionalGeneration as Orc
    IMG = 151
    orc = Orc(**TINY_QWEN_VL_TEXT, mrope_section=(16, 24, 24), image_token_id=IMG, vision_config=TINY_QWEN_VL_VISION).eval()
    sd = text_encoder_state_dict(orc, 51, 52, "norm")
    for k in [k for k in sd if k.endswith("ln_q.weight")]:
        sd[k] = 1.0 + 0.1 * seeded(sd[k].shape, 53).to(torch.bfloat16).float()
    cfg = transformers.Qwen2_5_VLConfig(
        text_config={**TINY_QWEN_VL_TEXT, "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]},
                     "max_position_embeddings": 512, "tie_word_embeddings": False, "pad_token_id": 0, "bos_token_id": 1,
                     "eos_token_id": 2},
        vision_config={**TINY_QWEN_VL_VISION, "out_hidden_size": 256, "hidden_act": "silu"},
        image_token_id=IMG, video_token_id=152, vision_start_token_id=149, vision_end_token_id=150)
    hf = transformers.Qwen2_5_VLForConditionalGeneration(cfg).eval().float()
    res = hf.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(8)
    # text only, right padded
    ids = torch.randint(3, 140, (2, 17), generator=g)
    mask = torch.ones(2, 17, dtype=torch.long)
    mask[1, 11:] = 0
    ids[1, 11:] = 0
    with torch.no_grad():
        a = hf(input_ids=ids, attention_mask=mask, output_hidden_states=True)
    out = dict(transformers_version=transformers.__version__, text_config=TINY_QWEN_VL_TEXT, vision_config=TINY_QWEN_VL_VISION,
               image_token_id=IMG, keys=sorted(sd.keys()), seed=51,
               text=dict(ids=ids, mask=mask, last=a.hidden_states[-1], hidden1=a.hidden_states[1], n_hidden=len(a.hidden_states)))
    # text + two images: grids (t, h, w) in patches; merged tokens 15 + 8
    grid = torch.tensor([[1, 6, 10], [1, 8, 4]])
    n1, n2 = 15, 8
    seq = [5, 6, 149] + [IMG] * n1 + [150, 7, 8, 149] + [IMG] * n2 + [150] + [9, 10, 11, 12, 13]
    ids = torch.tensor([seq])
    mask = torch.ones_like(ids)
    pix = seeded((60 + 32, 3 * 2 * 14 * 14), 61)
    with torch.no_grad():
        b = hf(input_ids=ids, attention_mask=mask, pixel_values=pix, image_grid_thw=grid, output_hidden_states=True,
               mm_token_type_ids=(ids == IMG).int())
        vis = hf.model.visual(pix, grid_thw=grid).pooler_output
        pos, _ = hf.model.get_rope_index(ids, (ids == IMG).int(), image_grid_thw=grid, attention_mask=mask)
    out["image"] = dict(ids=ids, mask=mask, pixel_values=pix, grid=grid, last=b.hidden_states[-1], vision=vis,
                        position_ids=pos, n_hidden=len(b.hidden_states))
    torch.save(out, os.path.join(OUT, "qwen2_5_vl.pt"))
    print("qwen2_5_vl", float(a.hidden_states[-1].abs().mean()), float(b.hidden_states[-1].abs().mean()), float(vis.abs().mean()))


def extract_method(rel, cls, name, ns):
    """One METHOD of a reference class compiled as a plain function (by AST) in namespace `ns`: for engine classes whose module
    cannot be imported here (the engine base pulls diffusers / loguru).  The reference's code is RUN from where it lies."""
    import ast
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), filename=path)
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    f = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name)
    env = dict(ns)
    exec(compile(ast.Module(body=[f], type_ignores=[]), path, "exec"), env)
    return env[name]


TINY_WAN_I2V = dict(TINY_WAN, in_channels=36)


def gen_wan_i2v():
    """Wan-2.2 A14B image-to-video, the two things the reference adds to the text-to-video path:
      (a) `WanI2VEngine.run` ITSELF (engine/wan/i2v.py:13-314, compiled from the class by AST and driven with a stand-in `self`:
          the text encoder, scheduler, latent noise and `denoise` are stubs that carry no arithmetic of this path; the aspect-ratio
          resize and the frame count are the reference's own `BaseEngine._aspect_ratio_resize` / `_parse_num_frames`, the VAE encode
          is the reference AutoencoderKLWan on tiny channels through the `vae_encode` recipe base_engine.py:2139-2160): what reaches
          `denoise` — the 20-channel `latent_condition` = [first-frame mask x 4 | normalised condition latents], the guidance
          scales and the CFG decision.  `video_processor.preprocess` is diffusers' (absent): stated as x / 127.5 - 1.
      (b) the reference WanTransformer3DModel with `in_channels` = 36 on cat([latents, latent_condition]), float64."""
    import types
    import typing
    import numpy as np
    from PIL import Image
    install_vae_stubs()
    ref_vae_mod = load_by_path("ref_vae_wan_i2v", "src/vae/wan/model.py")
    from oracle.vae_wan import AutoencoderKLWanEncoder
    vae = ref_vae_mod.AutoencoderKLWan(**TINY_VAE).eval()
    vsd = vae_synthetic_state_dict(AutoencoderKLWanEncoder(**TINY_VAE), 15)
    vae.load_state_dict(vsd, strict=False)
    base = extract_defs("src/engine/base_engine.py", [], {})      # (namespace only)
    ns = dict(torch=torch, np=np, Image=Image, Dict=typing.Dict, Any=typing.Any, Callable=typing.Callable, List=typing.List,
              Union=typing.Union, Optional=typing.Optional, InputImage=typing.Any,
              safe_emit_progress=lambda *a, **k: None, make_mapped_progress=lambda cb, a, b: None)
    run = extract_method("src/engine/wan/i2v.py", "WanI2VEngine", "run", ns)
    resize = extract_method("src/engine/base_engine.py", "BaseEngine", "_aspect_ratio_resize", ns)
    parse_frames = extract_method("src/engine/base_engine.py", "BaseEngine", "_parse_num_frames", ns)
    seen = {}

    class Sched:
        config = types.SimpleNamespace(num_train_timesteps=1000)

        def set_timesteps(self, n, device=None):
            self.timesteps = torch.linspace(999.0, 1.0, n)

    class Self:
        device = torch.device("cpu")
        text_encoder = types.SimpleNamespace(encode=lambda prompt, device=None, num_videos_per_prompt=1, **kw:
                                             seeded((num_videos_per_prompt, 20, 64), 42 if prompt == "a cat" else 43))
        scheduler = Sched()
        component_dtypes = {"transformer": torch.float32}
        helpers = {}
        vae_scale_factor_spatial, vae_scale_factor_temporal = 8, 4
        video_processor = types.SimpleNamespace(preprocess=lambda img, height, width:
                                                torch.from_numpy(np.asarray(img).astype(np.float32) / 255.0).permute(2, 0, 1)[None] * 2.0 - 1.0)

        def load_component_by_type(self, *a, **k):
            pass

        to_device = _offload = load_component_by_type

        def _load_image(self, image):
            return Image.fromarray(image)

        def _aspect_ratio_resize(self, *a, **k):
            return resize(self, *a, **k)

        def _parse_num_frames(self, *a, **k):
            return parse_frames(self, *a, **k)

        def _get_timesteps(self, scheduler, timesteps, timesteps_as_indices, num_inference_steps):
            return scheduler.timesteps, num_inference_steps

        def load_config_by_type(self, kind):
            return types.SimpleNamespace(scale_factor_spatial=8, scale_factor_temporal=4, z_dim=16)

        def _get_latents(self, height, width, duration, num_channels_latents, vae_scale_factor_spatial, vae_scale_factor_temporal,
                         fps, batch_size, seed, dtype, generator):
            nf = self._parse_num_frames(duration, fps)
            return seeded((batch_size, num_channels_latents, (nf - 1) // vae_scale_factor_temporal + 1,
                           height // vae_scale_factor_spatial, width // vae_scale_factor_spatial), 41).to(dtype)

        def vae_encode(self, video, offload=False, dtype=None, normalize_latents_dtype=None):
            """BaseEngine.vae_encode without its disk cache (base_engine.py:2139-2160)."""
            seen["video_condition"] = video.clone()
            vae.enable_tiling(tile_sample_min_height=48, tile_sample_min_width=48, tile_sample_stride_height=32,
                              tile_sample_stride_width=32)
            with torch.no_grad():
                # `encode(video)[0].mode()`: DiagonalGaussianDistribution is diffusers' (absent); its mode is the mean = the first
                # z_dim channels of what `_encode` returns (vae/wan/model.py:1319-1331)
                lat = vae._encode(video)[:, :16]
            return vae.normalize_latents(lat.to(normalize_latents_dtype)).to(dtype)

        def denoise(self, **kw):
            seen.update({k: kw[k] for k in ("latent_condition", "first_frame_mask", "guidance_scale", "use_cfg_guidance",
                                            "boundary_timestep", "expand_timesteps")})
            seen["transformer_kwargs"] = sorted(kw["transformer_kwargs"])
            seen["latents_shape"] = tuple(kw["latents"].shape)
            return kw["latents"]

        def vae_decode(self, *a, **k):
            raise AssertionError("return_latents=True")

    image = (seeded((50, 90, 3), 71) * 60 + 128).clamp(0, 255).to(torch.uint8).numpy()        # 50 x 90 RGB, resized by the engine
    out = run(Self(), image=image, prompt="a cat", negative_prompt="blurry", duration=9, height=64, width=96, num_inference_steps=4,
              seed=0, high_noise_guidance_scale=3.5, low_noise_guidance_scale=2.0, boundary_ratio=0.9, return_latents=True)
    cond = seen["latent_condition"]
    assert cond.shape[1] == 20 and tuple(out.shape) == seen["latents_shape"]
    run(Self(), image=image, prompt="a cat", negative_prompt="blurry", duration=9, height=64, width=96, num_inference_steps=4,
        boundary_ratio=0.9, return_latents=True)                                   # default scales 1.0 / 1.0: no CFG
    no_cfg = seen["use_cfg_guidance"]
    # (b) the 36-channel expert on [latents | condition]
    from src.transformer.wan.base.model import WanTransformer3DModel as RefWan
    from oracle.wan import WanTransformer3DModel as OracleWan
    ref = RefWan(**TINY_WAN_I2V, rope_max_seq_len=64).eval()
    sd = synthetic_state_dict(OracleWan(**TINY_WAN_I2V), 19)
    assert sorted(sd.keys()) == sorted(ref.state_dict().keys())
    ref.load_state_dict(sd, strict=True)
    ref = ref.double()
    lat = seeded(seen["latents_shape"], 41)
    x = torch.cat([lat, cond], dim=1)
    txt = seeded((1, 20, 64), 42)
    with torch.no_grad():
        fwd = ref(hidden_states=x.double(), timestep=torch.tensor([950.0], dtype=torch.float64), encoder_hidden_states=txt.double(),
                  return_dict=False)[0]
    torch.save(dict(image=torch.from_numpy(image), height=64, width=96, duration=9, vae_config=TINY_VAE, vae_seed=15,
                    tile=(48, 48, 32, 32), resized=tuple(seen["video_condition"].shape[-2:]),
                    video_condition_frame0=seen["video_condition"][:, :, 0].clone(),
                    video_condition_rest_abs_max=float(seen["video_condition"][:, :, 1:].abs().max()),
                    latent_condition=cond.float(), first_frame_mask=seen["first_frame_mask"].float(),
                    guidance_scale=[float(g) for g in (3.5, 2.0)], use_cfg_guidance=True, use_cfg_guidance_default_scales=bool(no_cfg),
                    boundary_timestep=float(900.0), transformer_kwargs=seen["transformer_kwargs"], latents_shape=seen["latents_shape"],
                    wan_config=TINY_WAN_I2V, wan_seed=19, latents_seed=41, txt_seed=42, timestep=950.0, wan_out=fwd.float()),
               os.path.join(OUT, "wan_i2v.pt"))
    print("wan_i2v.pt", tuple(cond.shape), seen["video_condition"].shape, float(cond[:, 4:].abs().mean()), tuple(fwd.shape), no_cfg)


# ---- leaf pins: the reference's IN-TREE copies of the diffusers leaves -----------------------------------------------
def extract_defs(rel, names, ns=None):
    """Execute only the named top-level functions / classes of a reference source file (by AST), in a namespace that
    supplies torch / nn / F / math / typing — for files whose module-level imports need packages absent here.  The
    reference's code is RUN from where it lies; nothing of it is written anywhere."""
    import ast
    import math
    import typing
    import numpy as np
    import torch.nn.functional as F
    from einops import rearrange
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), filename=path)
    want = set(names)
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    missing = want - {n.name for n in body}
    assert not missing, f"{rel}: {missing} not found"
    env = dict(torch=torch, nn=nn, F=F, math=math, np=np, rearrange=rearrange, Tensor=torch.Tensor,
               Optional=typing.Optional, Tuple=typing.Tuple, Union=typing.Union, Dict=typing.Dict, List=typing.List,
               Any=typing.Any, Callable=typing.Callable)
    env.update(ns or {})
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), env)
    return env


def gen_leaf_pins():
    """tests/golden/leaf_pins.pt: outputs of the reference's own in-tree copies of the un-vendored diffusers leaves the
    hot path uses, run here on seeded inputs / weights (weights are regenerated from seeds by the tests):
      timestep embedding stack   transformer/stepvideo/base/modules.py:204-330  (get_timestep_embedding, Timesteps,
                                 TimestepEmbedding), :525-550 (PixArtAlphaTextProjection), :657-712 (GELU, FeedForward),
                                 :121-175 (RMSNorm)
      rotary                     utils/models/hunyuan.py:138-185 (get_1d_rotary_pos_embed),
                                 transformer/flux2/control/base_model.py:71-132 (apply_rotary_emb)
      AdaLN family               transformer/hunyuanvideo/base/model.py:98-161 (linear(silu(emb)) -> chunk(6) -> formula),
                                 transformer/chroma/base/model.py:59-135 (chunk(6) / chunk(3) orders),
                                 converters/utils.py:82-85 (swap_scale_shift: AdaLayerNormContinuous is [scale, shift])
      2-D VAE blocks             vae/seedvr/modules/__model.py:73-142 (ResnetBlock2D),
                                 vae/hunyuanimage3/model.py:169-240 (AttnBlock, ResnetBlock), :297-308 (Upsample)
      FlowMatch-Euler            scheduler/flow.py:293-355 (FlowMatchDiscreteScheduler.step) + sd3_time_shift"""
    out = {}
    sv = extract_defs("src/transformer/stepvideo/base/modules.py",
                      ["get_timestep_embedding", "Timesteps", "TimestepEmbedding", "PixArtAlphaTextProjection", "GELU",
                       "FeedForward", "RMSNorm"],
                      dict(get_activation=lambda name: {"silu": nn.SiLU(), "swish": nn.SiLU(), "gelu": nn.GELU()}[name]))
    t = torch.tensor([0.0, 1.0, 37.5, 500.0, 718.75, 999.0, 1000.0])
    out["timestep_embedding"] = [dict(t=t / sc, dim=256, flip=True, shift=sh, scale=sc,
                                      out=sv["get_timestep_embedding"](t / sc, 256, flip_sin_to_cos=True,
                                                                       downscale_freq_shift=sh, scale=sc))
                                 for sc, sh in ((1.0, 0.0), (1000.0, 0.0), (1.0, 1.0))]

    def run(mod, seed, *xs):
        sd = synthetic_state_dict(mod, seed)
        mod.load_state_dict(sd, strict=True)
        with torch.no_grad():
            return mod.eval()(*xs), sorted(sd.keys())

    x = seeded((2, 256), 101)
    o, keys = run(sv["TimestepEmbedding"](256, 192), 102, x)
    out["TimestepEmbedding"] = dict(seed=102, x_seed=101, x_shape=(2, 256), dims=(256, 192), out=o, keys=keys)
    x = seeded((2, 9, 96), 103)
    o, keys = run(sv["PixArtAlphaTextProjection"](96, 160), 104, x)
    out["PixArtAlphaTextProjection"] = dict(seed=104, x_seed=103, x_shape=(2, 9, 96), dims=(96, 160), out=o, keys=keys)
    x = seeded((2, 9, 128), 105)
    o, keys = run(sv["FeedForward"](128, inner_dim=320, bias=True), 106, x)
    out["FeedForward"] = dict(seed=106, x_seed=105, x_shape=(2, 9, 128), dims=(128, 320), out=o, keys=keys)
    x = seeded((2, 9, 128), 107) * 3
    o, keys = run(sv["RMSNorm"](128, eps=1e-6), 108, x)
    out["RMSNorm"] = dict(seed=108, x_seed=107, x_shape=(2, 9, 128), dim=128, eps=1e-6, out=o, keys=keys)

    hy = extract_defs("src/utils/models/hunyuan.py", ["get_1d_rotary_pos_embed"])
    pos = torch.arange(0, 77).float() * 1.5
    out["get_1d_rotary_pos_embed"] = [dict(dim=d, pos=pos, out=hy["get_1d_rotary_pos_embed"](d, pos, theta=10000.0, use_real=True))
                                      for d in (16, 56, 128)]
    f2 = extract_defs("src/transformer/flux2/control/base_model.py", ["apply_rotary_emb"])
    cos, sin = hy["get_1d_rotary_pos_embed"](128, torch.arange(40).float(), use_real=True)
    xs1, xs2 = seeded((2, 40, 3, 128), 109), seeded((2, 3, 40, 128), 110)
    out["apply_rotary_emb"] = dict(cos=cos, sin=sin, x1_seed=109, x1_shape=(2, 40, 3, 128), x2_seed=110, x2_shape=(2, 3, 40, 128),
                                   out1=f2["apply_rotary_emb"](xs1, (cos, sin), sequence_dim=1),
                                   out2=f2["apply_rotary_emb"](xs2, (cos, sin), sequence_dim=2))

    hv = extract_defs("src/transformer/hunyuanvideo/base/model.py", ["HunyuanVideoTokenReplaceAdaLayerNormZero"],
                      dict(FP32LayerNorm=OL.FP32LayerNorm))
    ada = hv["HunyuanVideoTokenReplaceAdaLayerNormZero"](64)
    x, emb = seeded((2, 11, 64), 111), seeded((2, 64), 112)
    sd = synthetic_state_dict(ada, 113)
    ada.load_state_dict(sd, strict=True)
    with torch.no_grad():
        r = ada.eval()(x, emb, emb, 0)      # no replaced tokens: plain AdaLayerNormZero
    out["AdaLayerNormZero"] = dict(seed=113, x_seed=111, emb_seed=112, dim=64, keys=sorted(sd.keys()),
                                   out=[v for v in r[:5]])
    ch = extract_defs("src/transformer/chroma/base/model.py",
                      ["ChromaAdaLayerNormZeroPruned", "ChromaAdaLayerNormZeroSinglePruned"],
                      dict(FP32LayerNorm=OL.FP32LayerNorm, CombinedTimestepLabelEmbeddings=None))
    e6, e3 = seeded((2, 6, 64), 114), seeded((2, 3, 64), 115)
    with torch.no_grad():
        out["AdaLN_chunk_orders"] = dict(x_seed=111, e6_seed=114, e3_seed=115, dim=64,
                                         zero=[v for v in ch["ChromaAdaLayerNormZeroPruned"](64)(x, emb=e6)],
                                         single=[v for v in ch["ChromaAdaLayerNormZeroSinglePruned"](64)(x, emb=e3)])
    cu = extract_defs("src/converters/utils.py", ["swap_scale_shift"], dict(ggml_chunk=torch.chunk, ggml_cat=torch.cat))
    w = seeded((2 * 48, 32), 116)
    out["swap_scale_shift"] = dict(w_seed=116, w_shape=(96, 32), out=cu["swap_scale_shift"](w, dim=0))

    sr = extract_defs("src/vae/seedvr/modules/__model.py", ["ResnetBlock2D"])
    for tag, cin, cout, seed in (("same", 64, 64, 120), ("widen", 64, 96, 121)):
        blk = sr["ResnetBlock2D"](in_channels=cin, out_channels=cout)
        sd = vae_synthetic_state_dict(blk, seed)
        blk.load_state_dict(sd, strict=True)
        xx = seeded((1, cin, 12, 10), seed + 10)
        with torch.no_grad():
            out[f"ResnetBlock2D_{tag}"] = dict(seed=seed, x_seed=seed + 10, x_shape=(1, cin, 12, 10), cin=cin, cout=cout,
                                               keys=sorted(sd.keys()), out=blk.eval()(xx))
    h3 = extract_defs("src/vae/hunyuanimage3/model.py", ["swish", "Conv3d", "AttnBlock", "ResnetBlock", "Upsample"])
    for name, mk, cin in (("AttnBlock", lambda: h3["AttnBlock"](64), 64), ("ResnetBlock", lambda: h3["ResnetBlock"](64, 96), 64),
                          ("Upsample", lambda: h3["Upsample"](64, add_temporal_upsample=False), 64)):
        blk = mk()
        sd = vae_synthetic_state_dict(blk, 130 + len(name))
        blk.load_state_dict(sd, strict=True)
        xx = seeded((1, cin, 1, 12, 10), 140 + len(name))
        with torch.no_grad():
            out["ldm_" + name] = dict(seed=130 + len(name), x_seed=140 + len(name), x_shape=(1, cin, 1, 12, 10),
                                      keys=sorted(sd.keys()), out=blk.eval()(xx))

    # FlowMatch-Euler: the in-tree scheduler (needs only its config plumbing, supplied by the stubs)
    _mod("src.scheduler.scheduler", SchedulerInterface=type("SchedulerInterface", (), {}))
    sys.modules["diffusers.utils"].BaseOutput = dict
    fl = load_by_path("ref_flow", "src/scheduler/flow.py")
    sch = fl.FlowMatchDiscreteScheduler(shift=3.0)
    sch.set_timesteps(6)
    xx = seeded((1, 4, 8, 8), 150)
    traj = []
    for i, tt in enumerate(sch.timesteps):
        xx = sch.step(seeded((1, 4, 8, 8), 151 + i), tt, xx, return_dict=False)[0]
        traj.append(xx.clone())
    out["flow_euler"] = dict(shift=3.0, steps=6, x_seed=150, shape=(1, 4, 8, 8), timesteps=sch.timesteps.clone(),
                             sigmas=sch.sigmas.clone(), traj=traj)
    torch.save(out, os.path.join(OUT, "leaf_pins.pt"))
    print("leaf_pins.pt", sorted(out))


from tests.golden.make_golden_specs import flux_original_spec, hunyuan15_original_spec, wan_original_spec  # noqa: E402


def gen_convert():
    """tests/golden/convert_keys.pt: the reference's OWN checkpoint / LoRA key converters run on seeded original-format state
    dicts (dict operations and torch.chunk / split / cat only):
      WanTransformerConverter, FluxTransformerConverter   R/src/converters/transformer_converters.py:134-198, 1372-1840
      LoraManager.maybe_convert_state_dict's pipeline      R/src/lora/manager.py:633-644 (LoraConverter -> model converter ->
                                                           prefix strip), incl. the Kohya un-flattening lora_converter.py:185-255
    Stubs: src.quantize.ggml_ops' cat / chunk / split are torch's for plain tensors (their GGML branches need the gguf
    package); the diffusers rename tables of the two legacy LoRA formats are not exercised.  The fixture holds input specs
    (key -> shape, seed base) and the converted keys with their tensors."""
    install_converter_stubs()
    tc = load_by_path("src.converters.transformer_converters", "src/converters/transformer_converters.py")
    lc = load_by_path("ref_lora_converter2", "src/lora/lora_converter.py")
    from oracle import flux as OF, wan as OW
    wan_cfg = dict(patch_size=(1, 2, 2), num_attention_heads=1, attention_head_dim=64, in_channels=16, out_channels=16, text_dim=32,
                   freq_dim=32, ffn_dim=128, num_layers=2, cross_attn_norm=True, eps=1e-6)
    flux_cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=1,
                    joint_attention_dim=96, pooled_projection_dim=48, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    wan_keys = sorted(OW.WanTransformer3DModel(**wan_cfg).state_dict().keys())
    flux_keys = sorted(OF.FluxTransformer2DModel(**flux_cfg).state_dict().keys())
    cases = {}

    def run(name, conv_fn, spec, seed0, model_keys, prefix="", extra=None):
        sd = {prefix + k: v for k, v in spec_tensors(spec, seed0).items()}
        if extra:
            sd.update(extra)
        inp_keys = list(sd)
        out = conv_fn({k: v.clone() for k, v in sd.items()}, None if model_keys is None else list(model_keys))
        cases[name] = dict(spec={k: tuple(v) for k, v in spec.items()}, seed0=seed0, prefix=prefix, model_keys=model_keys,
                           extra=extra, inp_keys=inp_keys, out={k: tensor_digest(v) for k, v in out.items()})
        print("convert", name, len(inp_keys), "->", len(out), sorted(out)[:2])

    # 1. Wan original file, with and without model keys; 2. the same behind `model.diffusion_model.` with fp8 markers
    run("wan_original", lambda sd, mk: tc.WanTransformerConverter().convert(sd, mk), wan_original_spec(), 1000, wan_keys)
    run("wan_original_no_model_keys", lambda sd, mk: tc.WanTransformerConverter().convert(sd, mk), wan_original_spec(), 1000, None)
    scales = {f"model.diffusion_model.blocks.{i}.self_attn.{n}.scale_weight": torch.tensor(0.5 + i + j)
              for i in range(2) for j, n in enumerate(("q", "k", "v", "o"))}
    scales["model.diffusion_model.scaled_fp8"] = torch.zeros(2)
    # reference quirk, pinned as it is: the ".diff" drop marker is a substring of ".diffusion_model", so a file whose keys are
    # wrapped in `model.diffusion_model.` loses EVERY key in the Wan converter's pre-pass (the Kijai files are not wrapped)
    run("wan_fp8_wrapped", lambda sd, mk: tc.WanTransformerConverter().convert(sd, mk), wan_original_spec(), 1100, wan_keys,
        prefix="model.diffusion_model.", extra=scales)
    run("wan_fp8_kijai", lambda sd, mk: tc.WanTransformerConverter().convert(sd, mk), wan_original_spec(), 1100, wan_keys,
        extra={k[len("model.diffusion_model."):]: v for k, v in scales.items()})
    run("wan_already_converted", lambda sd, mk: tc.WanTransformerConverter().convert(sd, mk),
        {k: tuple(v.shape) for k, v in OW.WanTransformer3DModel(**wan_cfg).state_dict().items()}, 1200, wan_keys)
    # 3. Flux BFL file (guidance group, fused qkv / linear1, final-layer swap); a shard that holds only half of a linear1 pair
    run("flux_bfl", lambda sd, mk: tc.FluxTransformerConverter().convert(sd, mk), flux_original_spec(), 2000, flux_keys)
    run("flux_bfl_no_model_keys", lambda sd, mk: tc.FluxTransformerConverter().convert(sd, mk), flux_original_spec(), 2000, None)
    half = {k: v for k, v in flux_original_spec().items() if k.startswith("single_blocks.1.") and not k.endswith("linear1.bias")}
    half.update({k: v for k, v in flux_original_spec().items() if k.startswith("guidance_in.in_layer")})
    run("flux_partial_shard", lambda sd, mk: tc.FluxTransformerConverter().convert(sd, mk), half, 2100, None)
    run("flux_already_converted", lambda sd, mk: tc.FluxTransformerConverter().convert(sd, mk),
        {k: tuple(v.shape) for k, v in OF.FluxTransformer2DModel(**flux_cfg).state_dict().items()}, 2200, flux_keys)

    # 4. LoRAs through LoraManager.maybe_convert_state_dict's pipeline
    def lora_pipeline(conv_cls):
        def f(sd, mk):
            l = lc.LoraConverter()
            l.convert(sd, mk)
            conv_cls().convert(sd, mk)
            l._strip_known_prefixes_inplace(sd, model_keys=mk)
            return sd
        return f
    r = 4
    lx = {}
    for i in range(2):
        for a, n in (("self_attn", "q"), ("self_attn", "o"), ("cross_attn", "k"), ("cross_attn", "v")):
            m = f"diffusion_model.blocks.{i}.{a}.{n}"
            lx[m + ".lora_down.weight"] = (r, 64)
            lx[m + ".lora_up.weight"] = (64, r)
            lx[m + ".alpha"] = ()
        lx[f"diffusion_model.blocks.{i}.ffn.0.lora_down.weight"] = (r, 64)
        lx[f"diffusion_model.blocks.{i}.ffn.0.lora_up.weight"] = (128, r)
        lx[f"diffusion_model.blocks.{i}.ffn.2.lora_down.weight"] = (r, 128)
        lx[f"diffusion_model.blocks.{i}.ffn.2.lora_up.weight"] = (64, r)
        lx[f"diffusion_model.blocks.{i}.cross_attn.k.diff_b"] = (64,)
        lx[f"diffusion_model.blocks.{i}.cross_attn.norm_k.diff"] = (64,)
    run("wan_lightx2v_lora", lora_pipeline(tc.WanTransformerConverter), lx, 3000, wan_keys)
    fl = {}
    D = 3072      # a LoRA file does not reveal the model width: the reference falls back to FLUX.1-dev's (3072, mlp ratio 4)
    for kind_a, kind_b, pre in (("lora_A", "lora_B", ""), ("lora_down", "lora_up", "unet.")):
        i = 0 if kind_a == "lora_A" else 1
        for s_ in ("img_attn", "txt_attn"):
            fl[f"{pre}double_blocks.{i}.{s_}.qkv.{kind_a}.weight"] = (r, D)
            fl[f"{pre}double_blocks.{i}.{s_}.qkv.{kind_b}.weight"] = (3 * D, r)
            fl[f"{pre}double_blocks.{i}.{s_}.proj.{kind_a}.weight"] = (r, D)
            fl[f"{pre}double_blocks.{i}.{s_}.proj.{kind_b}.weight"] = (D, r)
        fl[f"{pre}single_blocks.{i}.linear1.{kind_a}.weight"] = (r, D)
        fl[f"{pre}single_blocks.{i}.linear1.{kind_b}.weight"] = (7 * D, r)
        fl[f"{pre}single_blocks.{i}.linear2.{kind_a}.weight"] = (r, 5 * D)
        fl[f"{pre}single_blocks.{i}.linear2.{kind_b}.weight"] = (D, r)
    run("flux_bfl_lora_peft_keys", lora_pipeline(tc.FluxTransformerConverter), {k: v for k, v in fl.items() if "lora_A" in k or "lora_B" in k},
        3100, flux_keys)
    run("flux_bfl_lora_base_keys", lora_pipeline(tc.FluxTransformerConverter), {k: v for k, v in fl.items() if "lora_down" in k or "lora_up" in k},
        3200, flux_keys)
    ko = {}
    for m, (o, i_) in (("lora_unet_double_blocks_0_img_attn_qkv", (3 * D, D)), ("lora_unet_double_blocks_0_txt_attn_proj", (D, D)),
                       ("lora_unet_single_blocks_1_linear2", (D, 5 * D)), ("lora_unet_double_blocks_1_img_mlp_0", (4 * D, D)),
                       ("lora_unet_single_blocks_0_linear1", (7 * D, D)), ("lora_unet_final_layer_linear", (64, D))):
        ko[m + ".lora_down.weight"] = (r, i_)
        ko[m + ".lora_up.weight"] = (o, r)
        ko[m + ".alpha"] = ()
    run("flux_kohya_lora", lora_pipeline(tc.FluxTransformerConverter), ko, 3300, flux_keys)
    kw = {}
    for m, (o, i_) in (("lora_unet_blocks_0_self_attn_q", (64, 64)), ("lora_unet_blocks_1_cross_attn_o", (64, 64)),
                       ("lora_unet_blocks_1_ffn_0", (128, 64))):
        kw[m + ".lora_down.weight"] = (r, i_)
        kw[m + ".lora_up.weight"] = (o, r)
        kw[m + ".alpha"] = ()
    run("wan_kohya_lora", lora_pipeline(tc.WanTransformerConverter), kw, 3400, wan_keys)
    # 5. HunyuanVideo-1.5 (round 4; HunyuanVideo15TransformerConverter, transformer_converters.py:899-1110): an original-format
    #    checkpoint (fused img / txt / refiner qkv), the already-converted layout, and a lightx2v-style LoRA keyed on the
    #    original names (fused-qkv down factor + alpha shared by q, k, v; up factor split in thirds)
    from oracle import hunyuan15 as OH
    hy_cfg = dict(in_channels=65, out_channels=32, num_attention_heads=1, attention_head_dim=128, num_layers=2, num_refiner_layers=1,
                  text_embed_dim=64, text_embed_2_dim=96, image_embed_dim=48)
    hy_model = OH.HunyuanVideo15Transformer3DModel(**hy_cfg)
    hy_keys = sorted(hy_model.state_dict().keys())
    hy = lambda sd, mk: tc.HunyuanVideo15TransformerConverter().convert(sd, mk)        # noqa: E731
    run("hy15_original", hy, hunyuan15_original_spec(), 4000, hy_keys)
    run("hy15_original_no_model_keys", hy, hunyuan15_original_spec(), 4000, None)
    run("hy15_wrapped", hy, hunyuan15_original_spec(), 4100, hy_keys, prefix="model.diffusion_model.")
    run("hy15_already_converted", hy, {k: tuple(v.shape) for k, v in hy_model.state_dict().items()}, 4200, hy_keys)
    hl = {}
    for m, (o, i_) in (("double_blocks.0.img_attn_qkv", (384, 128)), ("double_blocks.1.txt_attn_qkv", (384, 128)),
                       ("double_blocks.0.img_attn_proj", (128, 128)), ("double_blocks.1.img_mlp.fc1", (512, 128)),
                       ("double_blocks.1.txt_mlp.fc2", (128, 512)), ("double_blocks.0.img_mod.linear", (768, 128)),
                       ("txt_in.individual_token_refiner.blocks.0.self_attn_qkv", (384, 128)), ("final_layer.linear", (32, 128))):
        hl[f"diffusion_model.{m}.lora_down.weight"] = (r, i_)
        hl[f"diffusion_model.{m}.lora_up.weight"] = (o, r)
        hl[f"diffusion_model.{m}.alpha"] = ()
    run("hy15_original_key_lora", lora_pipeline(tc.HunyuanVideo15TransformerConverter), hl, 4300, hy_keys)
    torch.save(dict(cases=cases, wan_cfg=wan_cfg, flux_cfg=flux_cfg, hy_cfg=hy_cfg), os.path.join(OUT, "convert_keys.pt"))


def gen_leaf_pins2():
    """tests/golden/leaf_pins2.pt (round 3): the two remaining cheap pins of DESIGN.md §1.
      dynamic time shift   scheduler/rf.py:94-95 `time_shift`; scheduler/unipc.py:275-276 (the same closed form as a
                           method); scheduler/flow_match_pair.py:41-60 `FlowMatchScheduler.set_timesteps` with
                           `exponential_shift` + `shift_terminal` (:118-129 `calculate_shift`) — the Flux / QwenImage
                           FlowMatch-Euler schedule: linspace -> exp(mu) shift -> terminal stretch
      2-D VAE decoder      preprocess/diffusion_edge/taming/modules/diffusionmodules/model.py:462-580, the LDM `Decoder`
      TOPOLOGY             (the class the FLUX.1 autoencoder is an instance of: ch 128, ch_mult (1, 2, 4, 4), 2 res blocks,
                           z 16) with its ResnetBlock / AttnBlock / Upsample (:38-203).  The weights are drawn in the
                           DIFFUSERS key space (what oracle/vae_flux.py and the HIP class load) and renamed to the LDM
                           keys by `to_ldm` below, the published correspondence of the two layouts: mid.block_i <->
                           mid_block.resnets.i-1, mid.attn_1.{norm,q,k,v,proj_out} <-> mid_block.attentions.0.{group_norm,
                           to_q,to_k,to_v,to_out.0} (1x1 conv <-> linear), up.L <-> up_blocks.(n-1-L), nin_shortcut <->
                           conv_shortcut, norm_out <-> conv_norm_out."""
    out = {}
    rf = extract_defs("src/scheduler/rf.py", ["time_shift"])
    t = torch.linspace(1.0, 1.0 / 28, 28, dtype=torch.float64)
    out["time_shift"] = [dict(mu=mu, sigma=sg, t=t.clone(), out=rf["time_shift"](mu, sg, t)) for mu, sg in
                         ((0.5, 1.0), (1.15, 1.0), (0.8266, 1.0), (0.9, 2.0))]
    fm = extract_defs("src/scheduler/flow_match_pair.py", ["FlowMatchScheduler"])
    cases = []
    for steps, seq_len, terminal in ((8, 4096, 0.02), (28, 1024, None), (4, 8192 + 4096, 0.02), (50, 256, 0.02)):
        sch = fm["FlowMatchScheduler"](num_inference_steps=steps, exponential_shift=True, shift_terminal=terminal,
                                       sigma_max=1.0, sigma_min=1.0 / steps, exponential_shift_mu=0.8)
        sch.set_timesteps(steps, dynamic_shift_len=seq_len)
        cases.append(dict(steps=steps, seq_len=seq_len, shift_terminal=terminal, mu=sch.calculate_shift(seq_len),
                          sigmas=sch.sigmas.clone(), timesteps=sch.timesteps.clone()))
    out["flow_match_pair"] = cases

    from oracle.vae_flux import AutoencoderKLDecoder
    tm = extract_defs("src/preprocess/diffusion_edge/taming/modules/diffusionmodules/model.py",
                      ["nonlinearity", "Normalize", "Upsample", "ResnetBlock", "AttnBlock", "Decoder"])

    def to_ldm(sd, n_up):
        ldm = {}
        for k, v in sd.items():
            assert k.startswith("decoder."), k
            k2 = k[len("decoder."):]
            k2 = k2.replace("mid_block.resnets.0.", "mid.block_1.").replace("mid_block.resnets.1.", "mid.block_2.")
            if k2.startswith("mid_block.attentions.0."):
                k2 = k2.replace("mid_block.attentions.0.", "mid.attn_1.").replace("group_norm.", "norm.") \
                       .replace("to_q.", "q.").replace("to_k.", "k.").replace("to_v.", "v.").replace("to_out.0.", "proj_out.")
                if k2.endswith("weight") and v.dim() == 2:
                    v = v[:, :, None, None]            # nn.Linear <-> 1x1 convolution
            if k2.startswith("up_blocks."):
                i = int(k2.split(".")[1])
                rest = k2.split(".", 2)[2]
                rest = rest.replace("resnets.", "block.").replace("upsamplers.0.conv.", "upsample.conv.")
                k2 = f"up.{n_up - 1 - i}.{rest}"
            k2 = k2.replace("conv_shortcut.", "nin_shortcut.").replace("conv_norm_out.", "norm_out.")
            ldm[k2] = v
        return ldm

    dec = []
    for tag, cfg, seed, hw in (("flux_shape", dict(latent_channels=16, block_out_channels=(32, 64, 128, 128), layers_per_block=2), 161, (12, 10)),
                               ("three_levels", dict(latent_channels=8, block_out_channels=(32, 64, 64), layers_per_block=1), 162, (9, 14))):
        orc = AutoencoderKLDecoder(**cfg).eval()
        sd = vae_synthetic_state_dict(orc, seed)
        ch = cfg["block_out_channels"]
        ref = tm["Decoder"](ch=ch[0], out_ch=3, ch_mult=tuple(c // ch[0] for c in ch), num_res_blocks=cfg["layers_per_block"],
                            attn_resolutions=[], in_channels=3, resolution=64, z_channels=cfg["latent_channels"]).eval()
        missing, unexpected = ref.load_state_dict(to_ldm(sd, len(ch)), strict=True)
        z = seeded((1, cfg["latent_channels"]) + hw, seed + 10)
        with torch.no_grad():
            dec.append(dict(tag=tag, cfg=cfg, seed=seed, z_seed=seed + 10, z_shape=tuple(z.shape), out=ref(z),
                            ldm_keys=sorted(ref.state_dict().keys())))
    out["ldm_decoder"] = dec
    torch.save(out, os.path.join(OUT, "leaf_pins2.pt"))
    print("leaf_pins2.pt", sorted(out))


# Every fixture this script owns, in generation order (one generator each; a generator may write more than one file).
FIXTURES = ["attention", "efficiency", "flux_hybrid", "flux_controlnet", "flux_ip_adapter", "wan_hybrid", "wan_easycache", "wan_i2v", "qwen_hybrid", "qwen_variants", "hunyuan15_hybrid", "hunyuan15_meanflow",
            "vae_wan", "vae_wan_encode", "vae_hunyuan15", "vae_hunyuan15_encode", "vae_taehv", "vae_taehv_encode", "unipc", "lora",
            "fp_scaled", "text_encoders", "qwen2_5_vl", "leaf_pins", "leaf_pins2", "convert"]
# the generators that finish in seconds: `--check fast` (tests/test_oracle_golden.py runs it where /root/reference exists)
FAST = ["attention", "efficiency", "flux_hybrid", "flux_controlnet", "flux_ip_adapter", "wan_hybrid", "wan_easycache", "wan_i2v", "qwen_hybrid", "qwen_variants", "unipc", "lora", "fp_scaled", "leaf_pins", "leaf_pins2",
        "convert"]


def generate(names, out_dir):
    """Run the named generators in ONE process, in the given order, writing into `out_dir`."""
    global OUT
    OUT = out_dir
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    for name in names:
        globals()["gen_" + name]()


def _same(a, b, path, bad):
    """Deep bit-equality of two loaded fixtures (NaN == NaN: the fp8 tables hold NaN code points); mismatching paths -> bad."""
    if torch.is_tensor(a) or torch.is_tensor(b):
        ok = (torch.is_tensor(a) and torch.is_tensor(b) and a.dtype == b.dtype and a.shape == b.shape
              and bool(torch.equal(a.view(torch.uint8) if a.dtype.itemsize == 1 else a, b.view(torch.uint8) if b.dtype.itemsize == 1 else b)
                       or (a.is_floating_point() and torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
                           and torch.equal(a.float().isnan(), b.float().isnan()))))
        if not ok:
            bad.append(path)
    elif isinstance(a, dict) and isinstance(b, dict):
        if set(a) != set(b):
            bad.append(f"{path} (keys differ)")
        for k in a:
            if k in b:
                _same(a[k], b[k], f"{path}[{k!r}]", bad)
    elif isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        if len(a) != len(b):
            bad.append(f"{path} (length differs)")
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]", bad)
    elif isinstance(a, float) and isinstance(b, float):
        if not (a == b or (a != a and b != b)):
            bad.append(path)
    elif a != b:
        bad.append(path)


def check(names, committed=None):
    """Regenerate the named fixtures into a temporary directory and compare every file written with the committed one, bit for
    bit.  Returns (files compared, mismatches)."""
    import tempfile
    committed = committed or os.path.join(REPO, "tests", "golden")
    bad, files = [], []
    with tempfile.TemporaryDirectory() as tmp:
        generate(names, tmp)
        for fn in sorted(os.listdir(tmp)):
            files.append(fn)
            ref = os.path.join(committed, fn)
            if not os.path.exists(ref):
                bad.append(f"{fn}: not committed")
                continue
            _same(torch.load(os.path.join(tmp, fn), weights_only=False), torch.load(ref, weights_only=False), fn, bad)
    return files, bad


def main(argv):
    """make_golden.py                     regenerate ALL fixtures in place
       make_golden.py NAME [NAME ...]     regenerate the named ones (generator names: see FIXTURES)
       make_golden.py --check [fast|all|NAME ...]   regenerate to a temporary directory, assert bit-equality with the committed files"""
    if argv and argv[0] == "--check":
        sel = argv[1:] or ["all"]
        names = FIXTURES if sel == ["all"] else FAST if sel == ["fast"] else sel
        files, bad = check(names)
        print(f"[check] {len(names)} generators -> {len(files)} files compared, {len(bad)} mismatches")
        for b in bad[:50]:
            print("  MISMATCH", b)
        return 1 if bad else 0
    names = argv or FIXTURES
    unknown = [n for n in names if "gen_" + n not in globals()]
    if unknown:
        print("unknown fixtures:", unknown, "known:", FIXTURES)
        return 2
    generate(names, OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
