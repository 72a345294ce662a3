"""Drop-in boundary: registry semantics mirror the reference's FunctionRegister, and the backend
registers into the REFERENCE-style registry with the reference's calling convention."""
import pytest
import torch


def test_function_register_contract():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.register import FunctionRegister
    reg = FunctionRegister()

    @reg("double")
    def double(x):
        return 2 * x

    assert reg.call(3, key="double") == 6
    with pytest.raises(KeyError):
        reg("double")(lambda x: x)                 # duplicate key
    reg("double", overwrite=True)(lambda x: 3 * x)
    assert reg.call(3, key="double") == 9
    reg("gone", available=False)(lambda: 1)
    with pytest.raises(RuntimeError):
        reg.call(key="gone")                        # registered but unavailable
    reg.set_default("double")
    assert reg.call(2) == 6 and reg.get_default() == "double"
    assert set(reg.all()) == {"double", "gone"} and set(reg.all_available()) == {"double"}


def test_backend_registration_and_argument_errors():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import attention_backend as ab
    from apex_studio_amd.register import FunctionRegister, ClassRegister
    reg = ab.register(FunctionRegister())
    assert ab.KEY in reg.all()
    # no GPU in the CPU suite -> registered but unavailable, call() raises like the reference does
    if not torch.cuda.is_available():
        assert not reg.is_available(ab.KEY)
        with pytest.raises(RuntimeError):
            reg.call(torch.zeros(1, 2, 8, 64), torch.zeros(1, 2, 8, 64), torch.zeros(1, 2, 8, 64), key=ab.KEY)
    q = torch.zeros(1, 2, 8, 64)
    from apex_studio_amd.lib import ApexMIError
    for kw in (dict(attn_mask=torch.ones(8, 8)), dict(dropout_p=0.1), dict(is_causal=True)):
        with pytest.raises(ApexMIError):
            ab.hip_mfma(q, q, q, **kw)
    vreg = ClassRegister()
    creg = ab.register_models(ClassRegister(), vreg)
    assert {"flux.mi355", "wan.mi355", "qwenimage.mi355"} <= set(creg.all())
    assert {"auto_mi355", "wan_mi355", "qwenimage_mi355", "hunyuanvideo15_mi355"} <= set(vreg.all())


REF_REGISTER = "/root/reference/apps/api/src/register/__init__.py"


def _reference_register_module():
    """The reference's own registry module, loaded BY PATH (it has no imports beyond typing / functools).  The reference never
    travels to the GPU box: there this returns None and the tests that need it skip."""
    import importlib.util
    import os
    if not os.path.exists(REF_REGISTER):
        return None
    spec = importlib.util.spec_from_file_location("apex_reference_register", REF_REGISTER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_backend_registers_into_the_REAL_reference_registry():
    """The mirror (apex-studio_amd/register.py) is pinned to what it mirrors: `hip_mfma` and the component classes are registered
    into the reference's real `FunctionRegister` / `ClassRegister` (R/src/register/__init__.py:8-290) and dispatched through
    `attention_register.call(q, k, v, ..., key=)` exactly as R/src/attention/functions.py:84 / the processors do."""
    ref = _reference_register_module()
    if ref is None:
        pytest.skip("the reference tree is not on this machine (GPU box)")
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import attention_backend as ab
    from apex_studio_amd import register as mirror
    from apex_studio_amd.lib import ApexMIError
    reg = ref.FunctionRegister()

    @reg("sdpa")                                    # what the reference registers first (attention/functions.py:338)
    def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, softmax_scale=None, **kw):
        return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p,
                                                                is_causal=is_causal, scale=softmax_scale)
    reg.set_default("sdpa")
    ab.register(reg)
    assert ab.KEY in reg.all() and reg.get(ab.KEY) is ab.hip_mfma and reg.get_default() == "sdpa"
    q = torch.randn(1, 2, 8, 64)
    assert reg.call(q, q, q).shape == q.shape                       # the default backend still serves
    if not torch.cuda.is_available():
        assert not reg.is_available(ab.KEY) and ab.KEY not in reg.all_available()
        with pytest.raises(RuntimeError):
            reg.call(q, q, q, key=ab.KEY)                           # registered but unavailable: the reference's own error
        reg.set_availability(ab.KEY, True)                          # forced: the call reaches hip_mfma with the reference's kwargs
        with pytest.raises(ApexMIError):
            reg.call(q, q, q, attn_mask=None, dropout_p=0.0, is_causal=True, softmax_scale=None, key=ab.KEY)
        with pytest.raises(Exception) as ei:                        # CPU tensors: the product path fails loudly, no fallback
            reg.call(q, q, q, attn_mask=None, dropout_p=0.0, is_causal=False, softmax_scale=None, key=ab.KEY)
        assert not isinstance(ei.value, (KeyError, TypeError)), ei.value
    ab.register(reg, set_default=True)                              # overwrite=True: a second registration replaces, no KeyError
    assert reg.get_default() == ab.KEY
    vreg = ref.ClassRegister()
    creg = ab.register_models(ref.ClassRegister(), vreg)
    assert {"auto_mi355", "wan_mi355", "qwenimage_mi355", "hunyuanvideo15_mi355"} <= set(vreg.all())
    assert {"flux.mi355", "wan.mi355", "qwenimage.mi355", "hunyuanvideo15.mi355"} <= set(creg.all())
    # the mirror and the original behave alike on the whole contract (same calls, same results / exception types)
    def script(R):
        out = []
        r = R.FunctionRegister()
        r("a")(lambda x: x + 1)
        try:
            r("a")(lambda x: x)
        except Exception as e:
            out.append(type(e).__name__)
        r("a", overwrite=True)(lambda x: x + 2)
        r("b", available=False)(lambda x: x)
        out.append(r.call(1, key="a"))
        try:
            r.call(1, key="b")
        except Exception as e:
            out.append(type(e).__name__)
        try:
            r.get("zzz")
        except Exception as e:
            out.append(type(e).__name__)
        r.set_default("a")
        out += [r.call(5), r.get_default(), sorted(r.all()), sorted(r.all_available()), r.is_available("b"), r.is_available("zzz")]
        r.set_availability("b", True)
        out.append(sorted(r.all_available()))

        @r
        def named(x):
            return -x
        out.append(r.call(3, key="named"))
        return out
    assert script(ref) == script(mirror)


def test_flux_class_contract_on_meta_device():
    """What LoaderMixin._load_model demands (reference mixins/loader_mixin.py:219-531): from_config under
    empty weights, diffusers-style state-dict keys, load_state_dict(assign=True), .config access."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.flux import FluxTransformer2DModel
    cfg = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=128,
               pooled_projection_dim=64, guidance_embeds=True)
    m = FluxTransformer2DModel.from_config(cfg, device="meta")
    keys = set(m.state_dict().keys())
    for k in ("transformer_blocks.0.norm1.linear.weight", "transformer_blocks.0.attn.norm_q.weight",
              "transformer_blocks.0.attn.to_add_out.bias", "transformer_blocks.0.ff.net.0.proj.weight",
              "transformer_blocks.0.ff_context.net.2.bias", "single_transformer_blocks.0.proj_out.weight",
              "single_transformer_blocks.0.norm.linear.bias", "time_text_embed.guidance_embedder.linear_2.weight",
              "time_text_embed.text_embedder.linear_1.weight", "norm_out.linear.weight", "proj_out.bias",
              "x_embedder.weight", "context_embedder.bias"):
        assert k in keys, k
    sd = {k: torch.zeros(v.shape, dtype=torch.bfloat16) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=False, assign=True)
    assert m.x_embedder.weight.device.type == "cpu" and m.dtype == torch.bfloat16
    assert m.config.guidance_embeds and m.config.get("in_channels") == 64 and m.config.get("nope", 5) == 5
    with m.cache_context("cond"):
        pass
    from apex_studio_amd.lib import ApexMIError
    with pytest.raises(ApexMIError):     # CPU weights: the product path refuses, it does not fall back
        m.pack()


def test_wan_and_qwen_class_contracts_on_meta_device():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.wan import WanTransformer3DModel
    from apex_studio_amd.qwenimage import QwenImageTransformer2DModel
    w = WanTransformer3DModel.from_config(dict(num_attention_heads=2, ffn_dim=512, num_layers=1, text_dim=64),
                                          device="meta")
    keys = set(w.state_dict())
    for k in ("patch_embedding.weight", "condition_embedder.time_embedder.linear_1.weight",
              "condition_embedder.time_proj.bias", "condition_embedder.text_embedder.linear_2.weight",
              "blocks.0.attn1.norm_q.weight", "blocks.0.attn2.to_k.bias", "blocks.0.norm2.bias",
              "blocks.0.ffn.net.0.proj.weight", "blocks.0.scale_shift_table", "scale_shift_table", "proj_out.weight"):
        assert k in keys, k
    assert w.config.patch_size == (1, 2, 2) and w.config.get("in_channels") == 16
    w.set_chunking_profile("balanced")          # reference memory knob: accepted, no-op
    q = QwenImageTransformer2DModel.from_config(dict(num_attention_heads=2, num_layers=1, joint_attention_dim=64),
                                                device="meta")
    keys = set(q.state_dict())
    for k in ("transformer_blocks.0.img_mod.1.weight", "transformer_blocks.0.txt_mod.1.bias",
              "transformer_blocks.0.attn.add_q_proj.weight", "transformer_blocks.0.attn.norm_added_k.weight",
              "transformer_blocks.0.img_mlp.net.0.proj.weight", "transformer_blocks.0.txt_mlp.net.2.bias",
              "time_text_embed.timestep_embedder.linear_2.weight", "txt_norm.weight", "img_in.weight", "txt_in.bias",
              "norm_out.linear.weight", "proj_out.bias"):
        assert k in keys, k
    with pytest.raises(NotImplementedError):
        WanTransformer3DModel(image_dim=1280, device="meta")
    with pytest.raises(NotImplementedError):
        QwenImageTransformer2DModel(use_layer3d_rope=True, device="meta")
    # the Edit-2511-style switches are served since round 6: the extra embedding table carries the reference's key
    v = QwenImageTransformer2DModel(num_layers=1, zero_cond_t=True, use_additional_t_cond=True, device="meta")
    assert "time_text_embed.addition_t_embedding.weight" in v.state_dict() and v.config.zero_cond_t


def test_text_encoder_class_contracts_match_transformers():
    """The text-encoder drop-ins are resolved by class name from a module (text_encoder/text_encoder.py:24-82) and built
    through `_from_config(config_dict)` (loader_mixin.py:249-257): same class names, same parameter names and shapes as
    the `transformers` classes (keys of the 4.57 layout the checkpoints use), and no CPU fallback."""
    import transformers
    from apex_studio_amd import text_encoders as TE
    from apex_studio_amd.lib import ApexMIError
    t5 = dict(vocab_size=64, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2, feed_forward_proj="gated-gelu")
    for name, cfg_cls in (("T5EncoderModel", transformers.T5Config), ("UMT5EncoderModel", transformers.UMT5Config)):
        mine = getattr(TE, name)._from_config(t5, device="meta")
        hf = getattr(transformers, name)(cfg_cls(**t5))
        ref = {k: tuple(v.shape) for k, v in hf.state_dict().items()}
        got = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        assert got == ref, (name, set(got) ^ set(ref))
    clip = dict(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                max_position_embeddings=16)
    mine = TE.CLIPTextModel.from_config(clip, device="meta")
    hf = transformers.CLIPTextModel(transformers.CLIPTextConfig(**clip, bos_token_id=1, pad_token_id=0, eos_token_id=2))
    ref = {k if k.startswith("text_model.") else "text_model." + k: tuple(v.shape) for k, v in hf.state_dict().items()
           if "position_ids" not in k}
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == ref
    assert mine.config.hidden_act == "quick_gelu" and mine.config.get("eos_token_id") == 2
    cpu = TE.T5EncoderModel(t5, dtype=torch.bfloat16)
    with pytest.raises(ApexMIError):
        cpu(input_ids=torch.zeros(1, 4, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        TE.T5EncoderModel(dict(t5, feed_forward_proj="relu"), device="meta")


def test_qwen2_5_vl_class_contract_matches_transformers():
    """Same parameter names and shapes as transformers.Qwen2_5_VLForConditionalGeneration (the 4.57 module layout the
    reference's converter targets, converters/text_encoder_converters.py:30-45), both config spellings accepted."""
    import transformers
    from apex_studio_amd.qwen2_5_vl import Qwen2_5_VLForConditionalGeneration as Mine
    text = dict(vocab_size=160, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=1, rms_norm_eps=1e-6, rope_theta=1000000.0,
                rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]})
    vis = dict(depth=2, hidden_size=320, intermediate_size=172, num_heads=4, in_channels=3, patch_size=14,
               spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1], out_hidden_size=256)
    hf = transformers.Qwen2_5_VLForConditionalGeneration(transformers.Qwen2_5_VLConfig(
        text_config={**text, "tie_word_embeddings": False, "pad_token_id": 0, "bos_token_id": 1, "eos_token_id": 2},
        vision_config=vis, image_token_id=151, video_token_id=152, vision_start_token_id=149, vision_end_token_id=150))
    ref = {k: tuple(v.shape) for k, v in hf.state_dict().items()}
    nested = Mine._from_config(dict(text_config=text, vision_config=vis, image_token_id=151), device="meta")
    assert {k: tuple(v.shape) for k, v in nested.state_dict().items()} == ref
    flat = Mine.from_config({**text, "vision_config": vis, "image_token_id": 151}, device="meta")      # 4.x config.json
    assert {k: tuple(v.shape) for k, v in flat.state_dict().items()} == ref
    assert flat.config.mrope_section == (16, 24, 24) and flat.config.vision_config.fullatt_block_indexes == [1]
    with pytest.raises(NotImplementedError):
        Mine(dict(text_config={**text, "num_attention_heads": 4}), device="meta")          # head dim 64: sections do not fit
