"""Host-side sampler loops (stay in Python): expert switching, guidance selection, progress protocol,
CFG renorm — exercised on CPU with stand-in transformers (the HIP models are tested in test_gpu_*)."""
import contextlib
from types import SimpleNamespace

import torch


class _FakeWan:
    def __init__(self, tag):
        self.tag, self.calls = tag, []
        self.config = SimpleNamespace(in_channels=16)
        self.device, self.dtype = torch.device("cpu"), torch.bfloat16

    def __call__(self, hidden_states, timestep, encoder_hidden_states, return_dict=False):
        self.calls.append(float(timestep[0]))
        return (hidden_states.float() * 0.1 + self.tag,)


def test_wan_moe_denoise_switches_expert_at_boundary():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.engine_wan import WanT2VEngine
    hi, lo = _FakeWan(1.0), _FakeWan(2.0)
    eng = WanT2VEngine(hi, lo, vae=None)
    seen = []
    out = eng.run(prompt_embeds=torch.zeros(1, 4, 8), height=64, width=64, duration=9, num_inference_steps=8,
                  seed=0, generator=torch.Generator().manual_seed(0), return_latents=True,
                  progress_callback=lambda p, m: seen.append((p, m)))
    assert out.shape == (1, 16, 3, 8, 8) and out.dtype == torch.float32
    assert hi.calls and lo.calls and min(hi.calls) >= 875 > max(lo.calls)
    assert len(hi.calls) + len(lo.calls) == 8
    ps = [p for p, _ in seen]
    assert ps == sorted(ps) and ps[0] == 0.2 and ps[-1] == 1.0 and any("Denoising step 8/8" in m for _, m in seen)
    assert eng._select_dual_noise_guidance_scale(torch.tensor(900), 875.0, [4.0, 3.0]) == 4.0
    assert eng._select_dual_noise_guidance_scale(torch.tensor(100), 875.0, [4.0, 3.0]) == 3.0


class _FakeQwen:
    def __init__(self):
        self.device, self.dtype = torch.device("cpu"), torch.float32
        self.seen_shapes = []

    @contextlib.contextmanager
    def cache_context(self, name):
        yield

    def __call__(self, hidden_states, timestep, encoder_hidden_states, img_shapes, txt_seq_lens,
                 encoder_hidden_states_mask=None, return_dict=False):
        self.seen_shapes.append((hidden_states.shape[1], img_shapes, txt_seq_lens))
        return (hidden_states * (1.0 + encoder_hidden_states.mean()),)


def test_qwen_edit_plus_loop_slices_target_tokens_and_renorms_cfg():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.engine_qwenimage import QwenImageEditPlusEngine
    m = _FakeQwen()
    eng = QwenImageEditPlusEngine(m)
    cond = torch.randn(1, 16, 64)
    out = eng.run(prompt_embeds=torch.ones(1, 5, 8), image_latents=cond, image_shapes=[(64, 64)], height=64,
                  width=64, num_inference_steps=3, negative_prompt_embeds=torch.zeros(1, 7, 8), true_cfg_scale=4.0,
                  seed=1)
    assert out.shape == (1, 16, 64)
    n_tok, shapes, lens = m.seen_shapes[0]
    assert n_tok == 32 and shapes == [[(1, 4, 4), (1, 4, 4)]] and lens == [5]
    assert m.seen_shapes[1][2] == [7] and len(m.seen_shapes) == 6      # cond + uncond per step
    ts = eng.scheduler.timesteps
    assert float(ts[0]) <= 1000.0 and float(eng.scheduler.sigmas[-1]) == 0.0


def test_flux_engine_progress_and_preview_protocol():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.engine_flux import FluxT2IEngine, pack_latents, unpack_latents

    class _FakeFlux:
        config = SimpleNamespace(in_channels=64, guidance_embeds=True)
        device, dtype = torch.device("cpu"), torch.float32

        @contextlib.contextmanager
        def cache_context(self, name):
            yield

        def __call__(self, hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids,
                     img_ids, return_dict=False):
            assert guidance.shape == (1,) and float(guidance[0]) == 3.5 and float(timestep[0]) <= 1.0
            assert img_ids.shape == (hidden_states.shape[1], 3) and txt_ids.shape == (encoder_hidden_states.shape[1], 3)
            return (hidden_states * 0.5,)

    previews, prog = [], []
    eng = FluxT2IEngine(_FakeFlux(), decode_fn=lambda z: z.mean())
    img = eng.run(prompt_embeds=torch.zeros(1, 6, 32), pooled_prompt_embeds=torch.zeros(1, 16), height=64, width=64,
                  num_inference_steps=4, seed=0, generator=torch.Generator().manual_seed(0),
                  progress_callback=lambda p, m: prog.append(p), render_on_step=True,
                  render_on_step_callback=previews.append, render_on_step_interval=2)
    assert img.ndim == 0 and len(previews) == 2 and prog[-1] == 1.0 and prog == sorted(prog)
    x = torch.randn(1, 16, 8, 8)
    assert torch.equal(unpack_latents(pack_latents(x), 64, 64, 8), x)


class _FakeHunyuan:
    def __init__(self):
        self.config = SimpleNamespace(in_channels=65)
        self.device, self.dtype = torch.device("cpu"), torch.float32
        self.calls = []

    @contextlib.contextmanager
    def cache_context(self, name):
        self.calls.append(name)
        yield

    def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, encoder_hidden_states_2,
                 encoder_attention_mask_2, image_embeds, return_dict=False):
        assert hidden_states.shape[1] == 65 and float(hidden_states[:, 32:].abs().sum()) == 0.0   # zero cond + mask
        assert image_embeds.shape[1:] == (729, 1152) and float(image_embeds.abs().sum()) == 0.0      # t2v marker
        assert 0.0 < float(timestep[0]) <= 1000.0
        return (hidden_states[:, :32] * 0.1 + encoder_hidden_states.mean(),)


def test_hunyuan15_t2v_loop_cfg_rescale_and_progress():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.engine_hunyuan15 import HunyuanVideo15T2VEngine
    m = _FakeHunyuan()
    eng = HunyuanVideo15T2VEngine(m)
    seen = []
    kw = dict(prompt_embeds=torch.ones(1, 6, 8), prompt_embeds_mask=torch.ones(1, 6), prompt_embeds_2=torch.ones(1, 4, 8),
              prompt_embeds_mask_2=torch.ones(1, 4), height=64, width=96, num_frames=9, num_inference_steps=4,
              generator=torch.Generator().manual_seed(0), return_latents=True)
    out = eng.run(progress_callback=lambda p, m_: seen.append(p), guidance_scale=1.0, **kw)
    assert out.shape == (1, 32, 3, 4, 6) and m.calls == ["pred_cond"] * 4
    assert seen == sorted(seen) and seen[0] == 0.15 and seen[-1] == 1.0
    m.calls.clear()
    neg = dict(negative_prompt_embeds=torch.zeros(1, 6, 8), negative_prompt_embeds_mask=torch.ones(1, 6),
               negative_prompt_embeds_2=torch.zeros(1, 4, 8), negative_prompt_embeds_mask_2=torch.ones(1, 4))
    a = eng.run(guidance_scale=6.0, guidance_rescale=0.0, **kw, **neg)
    assert m.calls == ["pred_uncond", "pred_cond"] * 4
    kw["generator"] = torch.Generator().manual_seed(0)
    b = eng.run(guidance_scale=6.0, guidance_rescale=0.7, **kw, **neg)
    assert not torch.equal(a, b) and torch.isfinite(b).all()


def test_hunyuan15_i2v_condition_latents_mask_and_image_embeds():
    """reference engine/hunyuanvideo15/i2v.py:20-58, :262-264: cond = image latents on latent frame 0 / zeros after it,
    mask = 1 on frame 0, SigLIP embeddings forwarded; the first frame comes from vae.encode(...).mode() normalised."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.engine_hunyuan15 import HunyuanVideo15I2VEngine
    seen = {}

    class _Fake(_FakeHunyuan):
        def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, encoder_hidden_states_2,
                     encoder_attention_mask_2, image_embeds, return_dict=False):
            seen["cond"], seen["mask"] = hidden_states[:, 32:64].clone(), hidden_states[:, 64:].clone()
            seen["img"] = image_embeds.clone()
            return (hidden_states[:, :32] * 0.1,)

    class _FakeVAE:
        dtype = torch.float32
        tiling = 0

        def enable_tiling(self):
            self.tiling += 1

        def encode(self, x, return_dict=False):
            assert x.shape == (1, 3, 1, 64, 96)
            mean = torch.full((1, 32, 1, 4, 6), 0.5)
            return (SimpleNamespace(mode=lambda: mean, sample=lambda generator=None: mean + 1),)

        def normalize_latents(self, z):
            return z * 2.0

    vae = _FakeVAE()
    eng = HunyuanVideo15I2VEngine(_Fake(), vae=vae)
    kw = dict(prompt_embeds=torch.ones(1, 6, 8), prompt_embeds_mask=torch.ones(1, 6), prompt_embeds_2=torch.ones(1, 4, 8),
              prompt_embeds_mask_2=torch.ones(1, 4), height=64, width=96, num_frames=9, num_inference_steps=2,
              generator=torch.Generator().manual_seed(0), return_latents=True)
    emb = torch.full((1, 729, 1152), 0.25)
    out = eng.run(image=torch.zeros(1, 3, 64, 96), image_embeds=emb, **kw)
    assert out.shape == (1, 32, 3, 4, 6) and vae.tiling >= 1
    assert torch.equal(seen["cond"][:, :, 0], torch.full((1, 32, 4, 6), 1.0)) and float(seen["cond"][:, :, 1:].abs().sum()) == 0
    assert torch.equal(seen["mask"][:, :, 0], torch.ones(1, 1, 4, 6)) and float(seen["mask"][:, :, 1:].abs().sum()) == 0
    assert torch.equal(seen["img"], emb)
    # first-frame latents given directly (no VAE call), and the image is mandatory
    eng2 = HunyuanVideo15I2VEngine(_Fake(), vae=None)
    eng2.run(image=torch.full((1, 32, 1, 4, 6), 3.0), **dict(kw, generator=torch.Generator().manual_seed(0)))
    assert torch.equal(seen["cond"][:, :, 0], torch.full((1, 32, 4, 6), 3.0))
    import pytest
    with pytest.raises(ValueError):
        eng2.run(**kw)


def test_hunyuan15_meanflow_timestep_r_schedule():
    """reference engine/hunyuanvideo15/i2v.py:281-288: with `config.use_meanflow` the i2v loop passes timestep_r = the NEXT
    timestep (0 after the last), expanded to the batch in the latents' dtype; its t2v loop never passes one."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.engine_hunyuan15 import HunyuanVideo15I2VEngine, HunyuanVideo15T2VEngine
    seen = []

    class _Fake(_FakeHunyuan):
        def __init__(self, meanflow):
            super().__init__()
            self.config = SimpleNamespace(in_channels=65, use_meanflow=meanflow)

        def __call__(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, encoder_hidden_states_2,
                     encoder_attention_mask_2, image_embeds, return_dict=False, **kw):
            seen.append((float(timestep[0]), None if "timestep_r" not in kw else float(kw["timestep_r"][0])))
            if "timestep_r" in kw:
                assert kw["timestep_r"].shape == timestep.shape and kw["timestep_r"].dtype == hidden_states.dtype
            return (hidden_states[:, :32] * 0.1,)

    kw = dict(prompt_embeds=torch.ones(1, 6, 8), prompt_embeds_mask=torch.ones(1, 6), prompt_embeds_2=torch.ones(1, 4, 8),
              prompt_embeds_mask_2=torch.ones(1, 4), height=64, width=96, num_frames=9, num_inference_steps=3,
              generator=torch.Generator().manual_seed(0), return_latents=True)
    first = torch.zeros(1, 32, 1, 4, 6)
    HunyuanVideo15I2VEngine(_Fake(True)).run(image=first, **kw)
    ts = [t for t, _ in seen]
    assert len(seen) == 3 and [r for _, r in seen] == ts[1:] + [0.0] and ts == sorted(ts, reverse=True)
    seen.clear()
    HunyuanVideo15I2VEngine(_Fake(False)).run(image=first, **kw)
    assert [r for _, r in seen] == [None] * 3
    seen.clear()
    HunyuanVideo15T2VEngine(_Fake(True)).run(**kw)
    assert [r for _, r in seen] == [None] * 3
