"""Token ids / pixels in, uint8 frames out, through the ENGINES' own `run()`: every stage on the HIP classes —
text encoders (T5 + CLIP, UMT5, Qwen2.5-VL with a condition image), VAE encode, denoise loop, VAE decode, frame
post-processing — as `engine.run(prompt=…)` is called by the reference's render queue (R/src/api/ray_tasks.py:2775-2812), from the
point where the CPU tokenizer / image processor has produced ids and pixel patches.

Each chain is compared with the same chain run stage by stage through the public pieces (encode with `prompt.TextEncoder` /
`qwen_prompt_embeds`, then `run(prompt_embeds=…)`): identical frames, byte for byte.  The stages themselves are held to the
oracle elsewhere (test_gpu_text.py, test_gpu_qwen_vl.py, test_gpu_end_to_end.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import flux as OF
from oracle import qwenimage as OQ
from oracle import wan as OW
from tests.golden.seeded import seeded, synthetic_state_dict, text_encoder_state_dict, vae_synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _init(m, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    for n, p in m.named_parameters():
        if "norm" in n or n.endswith("ln_q.weight"):
            p.data.fill_(1.0)
        elif n.endswith("bias"):
            p.data.zero_()
        else:
            p.data.copy_((torch.randn(p.shape, generator=g, device=DEV) * (0.125 if n.endswith(".q.weight") else 1.0)
                          / p.shape[-1] ** 0.5).to(p.dtype))
    return m


def _frames_ok(fr, shape):
    assert isinstance(fr, np.ndarray) and fr.dtype == np.uint8 and fr.shape == shape, (fr.dtype, fr.shape)
    assert fr.std() > 5.0, "degenerate frames"


def test_flux_ids_to_frames():
    from apex_studio_amd import text_encoders as TE
    from apex_studio_amd.engine_flux import FluxT2IEngine
    from apex_studio_amd.flux import FluxTransformer2DModel
    from apex_studio_amd.prompt import TextEncoder
    from tests.test_gpu_end_to_end import _flux_vae_pair
    cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
               joint_attention_dim=128, pooled_projection_dim=128, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    m = FluxTransformer2DModel(**cfg, device=DEV, dtype=BF)
    m.load_state_dict({k: v.to(BF) for k, v in synthetic_state_dict(OF.FluxTransformer2DModel(**cfg), 7).items()}, strict=True)
    _, vae = _flux_vae_pair(dict(latent_channels=16, block_out_channels=(32, 64, 128, 128), layers_per_block=1), 19)
    t5 = _init(TE.T5EncoderModel(dict(vocab_size=120, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2), device=DEV), 1)
    clip = _init(TE.CLIPTextModel(dict(vocab_size=90, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                       num_attention_heads=2, max_position_embeddings=77, eos_token_id=89), device=DEV), 2)
    eng = FluxT2IEngine(m, decode_fn=lambda z: vae.decode(vae.denormalize_latents(z.float()).to(vae.dtype), return_dict=False)[0],
                        text_encoder=clip, text_encoder_2=t5)
    ids5 = torch.randint(3, 120, (1, 40), generator=torch.Generator().manual_seed(3))
    idsc = torch.randint(3, 88, (1, 77), generator=torch.Generator().manual_seed(4))
    idsc[0, 30:] = 89
    kw = dict(height=128, width=128, num_inference_steps=3, guidance_scale=3.5, seed=11, output_type="np",
              text_encoder_2_kwargs=dict(max_sequence_length=40))
    frames = eng.run(prompt_ids=idsc, prompt_2_ids=ids5, **kw)
    _frames_ok(frames, (1, 128, 128, 3))
    # the same, stage by stage
    pooled = TextEncoder(clip).encode(input_ids=idsc, max_sequence_length=77, pad_with_zero=False, output_type="pooler_output")
    emb = TextEncoder(t5).encode(input_ids=ids5, max_sequence_length=40, pad_with_zero=False)
    assert emb.shape == (1, 40, 128) and pooled.shape == (1, 128)
    assert np.array_equal(frames, eng.run(emb, pooled, **kw))
    assert not np.array_equal(frames, eng.run(prompt_ids=idsc, prompt_2_ids=(ids5 + 1) % 120, **kw)), "the prompt must matter"
    # as the host calls it: UniversalEngine.run is @torch.inference_mode() (R/src/engine/registry.py:196)
    with torch.inference_mode():
        assert np.array_equal(frames, eng.run(prompt_ids=idsc, prompt_2_ids=ids5, **kw))


def test_wan_ids_to_frames():
    from apex_studio_amd import text_encoders as TE
    from apex_studio_amd.engine_wan import WanT2VEngine
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    from apex_studio_amd.wan import WanTransformer3DModel
    cfg = dict(patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=128,
               freq_dim=256, ffn_dim=512, num_layers=2, cross_attn_norm=True, eps=1e-6)
    experts = []
    for seed in (9, 10):
        h = WanTransformer3DModel(**cfg, device=DEV, dtype=BF)
        h.load_state_dict({k: v.to(BF) for k, v in synthetic_state_dict(OW.WanTransformer3DModel(**cfg), seed).items()}, strict=True)
        experts.append(h)
    vae = AutoencoderKLWan(base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=1, temperal_downsample=[False, True, True],
                           device=DEV, dtype=BF)
    vae.load_state_dict({k: v.to(BF) for k, v in vae_synthetic_state_dict(vae, 23).items()}, strict=True)
    umt5 = _init(TE.UMT5EncoderModel(dict(vocab_size=150, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2), device=DEV), 5)
    eng = WanT2VEngine(experts[0], experts[1], vae=vae, text_encoder=umt5)
    g = torch.Generator().manual_seed(6)
    ids, nids = torch.randint(3, 150, (1, 24), generator=g), torch.randint(3, 150, (1, 24), generator=g)
    mask, nmask = torch.ones(1, 24, dtype=torch.long), torch.ones(1, 24, dtype=torch.long)
    mask[0, 17:], nmask[0, 5:] = 0, 0
    kw = dict(height=64, width=96, duration=5, num_inference_steps=3, guidance_scale=(4.0, 3.0), seed=12, output_type="np",
              text_encoder_kwargs=dict(max_sequence_length=24))
    frames = eng.run(prompt_ids=(ids, mask), negative_prompt_ids=(nids, nmask), **kw)
    _frames_ok(frames, (1, 5, 64, 96, 3))
    pe = eng.encode_prompt(prompt_ids=(ids, mask), text_encoder_kwargs=dict(max_sequence_length=24))
    ne = eng.encode_prompt(prompt_ids=(nids, nmask), text_encoder_kwargs=dict(max_sequence_length=24))
    assert pe.shape == (1, 24, 128) and float(pe[0, 17:].abs().sum()) == 0.0 and float(pe[0, :17].abs().sum()) > 0.0
    assert np.array_equal(frames, eng.run(prompt_embeds=pe, negative_prompt_embeds=ne, **kw))
    with torch.inference_mode():                       # as the host's UniversalEngine.run calls it
        assert np.array_equal(frames, eng.run(prompt_ids=(ids, mask), negative_prompt_ids=(nids, nmask), **kw))


def test_qwen_edit_ids_and_pixels_to_frames(golden_dir):
    from apex_studio_amd.engine_qwenimage import QwenImageEditPlusEngine
    from apex_studio_amd.prompt import qwen_prompt_embeds
    from apex_studio_amd.qwenimage import QwenImageTransformer2DModel
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    from tests.test_gpu_qwen_vl import _models
    g = torch.load(os.path.join(golden_dir, "qwen2_5_vl.pt"), weights_only=False)
    _, vl = _models(g)
    cfg = dict(patch_size=2, in_channels=64, out_channels=16, num_layers=2, attention_head_dim=128, num_attention_heads=2,
               joint_attention_dim=g["text_config"]["hidden_size"], axes_dims_rope=(16, 56, 56))
    m = QwenImageTransformer2DModel(**cfg, device=DEV, dtype=BF)
    m.load_state_dict({k: v.to(BF) for k, v in synthetic_state_dict(OQ.QwenImageTransformer2DModel(**cfg), 11).items()}, strict=True)
    vae = AutoencoderKLWan(base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=1, temperal_downsample=[False, True, True],
                           device=DEV, dtype=BF)
    vae.load_state_dict({k: v.to(BF) for k, v in vae_synthetic_state_dict(vae, 31).items()}, strict=True)
    eng = QwenImageEditPlusEngine(m, vae=vae, text_encoder=vl)
    im = g["image"]
    inputs = dict(input_ids=im["ids"], attention_mask=im["mask"], pixel_values=im["pixel_values"].to(BF), image_grid_thw=im["grid"])
    cond = seeded((1, 3, 96, 64), 71).clamp(-1, 1)
    kw = dict(images=cond.to(DEV), height=128, width=96, num_inference_steps=2, seed=13, output_type="np", drop_idx=4)
    frames = eng.run(prompt_inputs=inputs, **kw)
    _frames_ok(frames, (1, 128, 96, 3))
    emb, msk = qwen_prompt_embeds(vl, im["ids"], im["mask"], im["pixel_values"].to(BF), im["grid"], drop_idx=4, dtype=BF)
    n_valid = int(im["mask"].sum())
    assert emb.shape == (1, n_valid - 4, cfg["joint_attention_dim"]) and int(msk.sum()) == n_valid - 4
    assert np.array_equal(frames, eng.run(prompt_embeds=emb, **kw))
    with torch.inference_mode():                       # as the host's UniversalEngine.run calls it
        assert np.array_equal(frames, eng.run(prompt_inputs=inputs, **kw))
    other = dict(inputs, pixel_values=inputs["pixel_values"] * 0.5)
    assert not np.array_equal(frames, eng.run(prompt_inputs=other, **kw)), "the condition image must reach the prompt embedding"


def test_hunyuan15_ids_to_frames(golden_dir):
    """HunyuanVideo-1.5 `run(prompt_ids=…, prompt_2_ids=…)`: the MLLM (Qwen2.5-VL class, hidden state -3, the
    template tokens cropped) and the glyph ByT5 (T5-v1.1 encoder, 256-token padding with its mask) run inside the engine as
    `encode_prompt` does in the reference (R/src/engine/hunyuanvideo15/shared/__init__.py:145-283, 344-437); a prompt without
    quoted text gets zero glyph embeddings with an all-zero mask.  Compared with the stage-by-stage chain, byte for byte."""
    from apex_studio_amd import text_encoders as TE
    from apex_studio_amd.engine_hunyuan15 import HunyuanVideo15T2VEngine
    from apex_studio_amd.hunyuan15 import HunyuanVideo15Transformer3DModel
    from apex_studio_amd.prompt import TextEncoder
    from apex_studio_amd.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    from oracle import hunyuan15 as OH
    from oracle.vae_hunyuan15 import AutoencoderKLHunyuanVideo15 as OVae
    from tests.test_gpu_qwen_vl import _models
    g = torch.load(os.path.join(golden_dir, "qwen2_5_vl.pt"), weights_only=False)
    _, vl = _models(g)
    d_txt = g["text_config"]["hidden_size"]
    byt5 = _init(TE.T5EncoderModel(dict(vocab_size=384, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2), device=DEV), 7)
    cfg = dict(in_channels=65, out_channels=32, num_attention_heads=2, attention_head_dim=128, num_layers=2,
               num_refiner_layers=1, text_embed_dim=d_txt, text_embed_2_dim=128, image_embed_dim=64)
    m = HunyuanVideo15Transformer3DModel(**cfg, device=DEV, dtype=BF)
    m.load_state_dict({k: v.to(BF) for k, v in synthetic_state_dict(OH.HunyuanVideo15Transformer3DModel(**cfg), 21).items()}, strict=True)
    vcfg = dict(in_channels=3, out_channels=3, latent_channels=32, block_out_channels=(32, 64, 64, 128, 128),
                layers_per_block=1, spatial_compression_ratio=16, temporal_compression_ratio=4)
    vae = AutoencoderKLHunyuanVideo15(**vcfg, device=DEV, dtype=BF)
    vae.load_state_dict({k: v.to(BF) for k, v in vae_synthetic_state_dict(OVae(**vcfg), 23).items()}, strict=True)
    crop, L2 = 6, 24
    eng = HunyuanVideo15T2VEngine(m, vae=vae, vision_num_semantic_tokens=3, vision_states_dim=64, text_encoder=vl,
                                  text_encoder_2=byt5, tokenizer_2_max_length=L2, prompt_template_encode_start_idx=crop)
    t = g["text"]
    ids, mask = t["ids"][:1], t["mask"][:1]
    nids, nmask = t["ids"][1:2] if t["ids"].shape[0] > 1 else (ids + 1) % 50, t["mask"][1:2] if t["ids"].shape[0] > 1 else mask
    gen = torch.Generator().manual_seed(8)
    gids = torch.randint(3, 380, (1, L2), generator=gen)
    gmask = torch.ones(1, L2, dtype=torch.long)
    gmask[0, 15:] = 0
    kw = dict(height=96, width=128, num_frames=5, num_inference_steps=2, guidance_scale=3.0, seed=14, output_type="np")
    frames = eng.run(prompt_ids=(ids, mask), prompt_2_ids=(gids, gmask), negative_prompt_ids=(nids, nmask), **kw)
    _frames_ok(frames, (1, 5, 96, 128, 3))
    # stage by stage
    hs = vl(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), output_hidden_states=True).hidden_states
    pe, pm = hs[-3][:, crop:].to(BF), mask[:, crop:].to(DEV)
    pe2, pm2 = TextEncoder(byt5).encode(input_ids=gids, attention_mask=gmask, max_sequence_length=L2, pad_to_max_length=True,
                                        use_attention_mask=True, return_attention_mask=True, pad_with_zero=False)
    got = eng.encode_prompt((ids, mask), (gids, gmask))
    assert got[0].shape == (1, ids.shape[1] - crop, d_txt) and torch.equal(got[0], pe) and torch.equal(got[1].cpu(), pm.to(BF).cpu())
    assert got[2].shape == (1, L2, 128) and torch.equal(got[2], pe2.to(DEV, BF)) and torch.equal(got[3].cpu(), pm2.to(BF))
    nhs = vl(input_ids=nids.to(DEV), attention_mask=nmask.to(DEV), output_hidden_states=True).hidden_states
    zeros2, zmask2 = torch.zeros(1, L2, 128, device=DEV, dtype=BF), torch.zeros(1, L2, device=DEV)
    manual = eng.run(prompt_embeds=pe, prompt_embeds_mask=pm, prompt_embeds_2=pe2.to(DEV, BF), prompt_embeds_mask_2=pm2.to(DEV),
                     negative_prompt_embeds=nhs[-3][:, crop:].to(BF), negative_prompt_embeds_mask=nmask[:, crop:].to(DEV),
                     negative_prompt_embeds_2=zeros2, negative_prompt_embeds_mask_2=zmask2, **kw)
    assert np.array_equal(frames, manual)
    # no quoted text: zero glyph embeddings, all-zero mask (shared/__init__.py:252-262); and the glyph text must matter
    no_glyph = eng.encode_prompt((ids, mask), None)
    assert float(no_glyph[2].abs().sum()) == 0.0 and float(no_glyph[3].abs().sum()) == 0.0 and no_glyph[2].shape == (1, L2, 128)
    assert not np.array_equal(frames, eng.run(prompt_ids=(ids, mask), prompt_2_ids=None, negative_prompt_ids=(nids, nmask), **kw))
    with torch.inference_mode():                       # as the host's UniversalEngine.run calls it
        assert np.array_equal(frames, eng.run(prompt_ids=(ids, mask), prompt_2_ids=(gids, gmask), negative_prompt_ids=(nids, nmask), **kw))
