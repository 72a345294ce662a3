"""Wan-2.2 A14B image-to-video (`engine_wan.WanI2VEngine`, R/src/engine/wan/i2v.py:13-314) against tests/golden/wan_i2v.pt — the
reference's own `WanI2VEngine.run` executed on a stand-in engine object (make_golden.py::gen_wan_i2v): what reaches `denoise`
(the 20-channel latent condition = first-frame mask x 4 | normalised condition latents, the guidance scales, the CFG decision),
and the reference 36-channel WanTransformer3DModel on [latents | condition].  CPU: the host logic on stand-in experts / VAE;
GPU: the HIP VAE encode, the HIP expert and the engine end to end against the oracle chain."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import apex_studio_amd  # noqa: F401
from tests.conftest import measured
from tests.golden.seeded import seeded, synthetic_state_dict

try:
    from tests.golden.seeded import vae_synthetic_state_dict
except ImportError:                                   # pragma: no cover
    vae_synthetic_state_dict = None

BF = torch.bfloat16


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


@pytest.fixture(scope="module")
def g(golden_dir):
    return torch.load(os.path.join(golden_dir, "wan_i2v.pt"), weights_only=False)


class _FakeVae:
    """CPU stand-in with the VAE surface the engine touches; `encode` hands back the fixture's (de-normalised) condition latents."""

    def __init__(self, cond_norm):
        self.cond = cond_norm
        self.dtype, self.device = torch.float32, torch.device("cpu")
        self.seen = {}

    def enable_tiling(self, *a, **k):
        self.seen["tiling"] = True

    def encode(self, video, return_dict=False):
        self.seen["video"] = video.clone()
        return (SimpleNamespace(mode=lambda: self.cond * 2.0 + 0.5),)

    def normalize_latents(self, lat):
        return (lat - 0.5) / 2.0


class _FakeExpert:
    def __init__(self, name, log):
        self.name, self.log = name, log
        self.config = SimpleNamespace(in_channels=36, out_channels=16)
        self.device, self.dtype = torch.device("cpu"), torch.float32

    def __call__(self, hidden_states, timestep, encoder_hidden_states, return_dict=False):
        self.log.append((self.name, float(timestep[0]), hidden_states.clone(), float(encoder_hidden_states.mean())))
        return (hidden_states[:, :16].float() * 0.1,)


def test_reference_sizes_mask_and_cfg_rule_on_cpu(g):
    from apex_studio_amd.engine_wan import WanI2VEngine
    log = []
    cond16 = g["latent_condition"][:, 4:]
    vae = _FakeVae(cond16)
    eng = WanI2VEngine(_FakeExpert("hi", log), _FakeExpert("lo", log), vae=vae, boundary_ratio=0.9)
    img = g["image"].numpy()
    # the reference's aspect-preserving size (BaseEngine._aspect_ratio_resize) and its resize + x / 127.5 - 1
    assert eng.aspect_ratio_size(img.shape[0], img.shape[1], g["height"] * g["width"], 16) == tuple(g["resized"])
    px, h, w = eng.preprocess_image(img, g["height"], g["width"])
    assert (h, w) == tuple(g["resized"]) and torch.equal(px, g["video_condition_frame0"])
    # the mask of the reference run, bit for bit
    B, _, T, hl, wl = g["latents_shape"]
    assert torch.equal(eng.first_frame_mask(B, g["duration"], hl, wl), g["latent_condition"][:, :4])
    # the whole run: what the experts receive is cat([latents, reference latent_condition]); CFG on with both scales > 1
    pe, ne = seeded((1, 20, 64), 42), seeded((1, 20, 64), 43)
    lat0 = seeded(g["latents_shape"], g["latents_seed"])
    out = eng.run(image=img, prompt_embeds=pe, negative_prompt_embeds=ne, height=g["height"], width=g["width"], duration=g["duration"],
                  num_inference_steps=4, high_noise_guidance_scale=3.5, low_noise_guidance_scale=2.0, latents=lat0, return_latents=True)
    assert tuple(out.shape) == tuple(g["latents_shape"]) and vae.seen["tiling"]
    v = vae.seen["video"]
    assert tuple(v.shape) == (1, 3, g["duration"], h, w) and torch.equal(v[:, :, 0], g["video_condition_frame0"]) \
        and float(v[:, :, 1:].abs().max()) == g["video_condition_rest_abs_max"] == 0.0
    first = log[0][2]
    assert first.shape[1] == 36 and torch.equal(first[:, :16], lat0) and torch.allclose(first[:, 16:], g["latent_condition"], atol=1e-6)
    assert len(log) == 8 and [e[0] for e in log[:2]] == ["hi", "hi"] and log[-1][0] == "lo", "cond + uncond per step, experts by t >= 900"
    assert all(torch.equal(e[2][:, 16:], first[:, 16:]) for e in log), "the condition rides along unchanged; only 16 channels are stepped"
    assert {round(e[3], 6) for e in log} == {round(float(pe.mean()), 6), round(float(ne.mean()), 6)}
    # the reference's default scales (1.0 / 1.0) switch CFG off even with a negative prompt (i2v.py:56-64)
    log.clear()
    eng.run(image=img, prompt_embeds=pe, negative_prompt_embeds=ne, height=g["height"], width=g["width"], duration=g["duration"],
            num_inference_steps=4, latents=lat0, return_latents=True)
    assert len(log) == 4 and g["use_cfg_guidance"] is True and g["use_cfg_guidance_default_scales"] is False
    assert g["transformer_kwargs"] == sorted(["encoder_hidden_states", "encoder_hidden_states_image", "attention_kwargs",
                                              "enhance_kwargs", "rope_on_cpu"])
    with pytest.raises(NotImplementedError):
        eng.run(image=img, prompt_embeds=pe, expand_timesteps=True)
    with pytest.raises(ValueError):
        eng.run(prompt_embeds=pe)
    with pytest.raises(ValueError):
        WanI2VEngine(SimpleNamespace(config=SimpleNamespace(in_channels=16, out_channels=16), device=torch.device("cpu"),
                                     dtype=torch.float32), vae=vae).run(image=img, prompt_embeds=pe)


def test_oracle_wan_36_channels_matches_the_reference_run(g):
    """oracle.wan with `in_channels` = 36 against the reference model's float64 forward on [latents | condition]."""
    from oracle import wan as OW
    orc = OW.WanTransformer3DModel(**g["wan_config"]).eval()
    orc.load_state_dict(synthetic_state_dict(orc, g["wan_seed"]), strict=True)
    x = torch.cat([seeded(g["latents_shape"], g["latents_seed"]), g["latent_condition"]], dim=1)
    with torch.no_grad():
        out = orc(x, torch.tensor([g["timestep"]]), seeded((1, 20, 64), g["txt_seed"]))
    assert _rel(out, g["wan_out"]) < 1e-5


# --------------------------------------------------------------------------------------------------------------- GPU
DEV = "cuda"


def _hip_vae(cfg, sd):
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    vae = AutoencoderKLWan(**cfg, device=DEV, dtype=BF)
    res = vae.load_state_dict({k: v.to(BF) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys
    return vae


@pytest.mark.gpu
def test_hip_condition_and_expert_match_the_reference_run(g):
    """(a) `prepare_latent_condition` on the HIP VAE (tiled encode of [image | zeros], mode, normalise, mask) against the latent
    condition the reference's own run() handed to denoise; (b) the HIP expert with 36 input channels against the reference model."""
    from oracle.vae_wan import AutoencoderKLWanEncoder
    from apex_studio_amd.engine_wan import WanI2VEngine
    from apex_studio_amd.wan import WanTransformer3DModel
    from oracle import wan as OW
    orc_v = AutoencoderKLWanEncoder(**g["vae_config"]).eval()
    vsd = vae_synthetic_state_dict(orc_v, g["vae_seed"])
    vae = _hip_vae(g["vae_config"], vsd)
    vae.enable_tiling(*g["tile"])
    sd = synthetic_state_dict(OW.WanTransformer3DModel(**g["wan_config"]), g["wan_seed"])
    m = WanTransformer3DModel(**g["wan_config"], device=DEV, dtype=BF)
    m.load_state_dict({k: v.to(BF) for k, v in sd.items()}, strict=True)
    eng = WanI2VEngine(m, m, vae=vae, boundary_ratio=0.9)
    px, h, w = eng.preprocess_image(g["image"].numpy(), g["height"], g["width"])
    cond = eng.prepare_latent_condition(px, g["duration"], 1).float().cpu()
    assert cond.shape == g["latent_condition"].shape and torch.equal(cond[:, :4], g["latent_condition"][:, :4])
    e = measured("wan_i2v.latent_condition.bf16_vs_reference_run", _rel(cond[:, 4:], g["latent_condition"][:, 4:]), 9e-3)   # measured 4.5e-3
    x = torch.cat([seeded(g["latents_shape"], g["latents_seed"]), g["latent_condition"]], dim=1)
    out = m(hidden_states=x.to(DEV).to(BF), timestep=torch.tensor([g["timestep"]], device=DEV),
            encoder_hidden_states=seeded((1, 20, 64), g["txt_seed"]).to(DEV).to(BF), return_dict=False)[0].float().cpu()
    e2 = measured("wan_i2v.expert36.bf16_vs_reference_run", _rel(out, g["wan_out"]), 1.1e-2)   # measured 5.1e-3
    print(f"[wan i2v] condition latents vs the reference run {e:.2e}; 36-channel expert vs the reference model {e2:.2e}")


@pytest.mark.gpu
def test_wan_i2v_image_to_frames_end_to_end():
    """`WanI2VEngine.run(image=...)` through the HIP VAE encode, two HIP experts (CFG, expert switch), UniPC and the tiled HIP decode
    against the same chain on the oracle (bf16-storage policy and fp32): decoded frames at the end-to-end bars."""
    from oracle import layers as OL
    from oracle import wan as OW
    from oracle.postprocess import video_to_uint8_frames
    from oracle.vae_wan import AutoencoderKLWanDecoder, AutoencoderKLWanEncoder
    from apex_studio_amd.engine_wan import WanI2VEngine
    from apex_studio_amd.postprocess import tensor_to_frames
    from apex_studio_amd.schedulers import UniPCMultistepScheduler
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    from apex_studio_amd.wan import WanTransformer3DModel
    from tests.test_gpu_end_to_end import _frame_report, _wan_vae_cfg
    cfg = dict(patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=128, in_channels=36, out_channels=16,
               text_dim=64, freq_dim=256, ffn_dim=512, num_layers=2, cross_attn_norm=True, eps=1e-6)
    height, width, duration, steps = 96, 128, 9, 4
    experts_o, experts_h = [], []
    for seed in (29, 30):
        o = OW.WanTransformer3DModel(**cfg).eval()
        sd = synthetic_state_dict(o, seed)
        o.load_state_dict(sd, strict=True)
        h = WanTransformer3DModel(**cfg, device=DEV, dtype=BF)
        h.load_state_dict({k: v.to(BF) for k, v in sd.items()}, strict=True)
        experts_o.append(o)
        experts_h.append(h)
    vae = AutoencoderKLWan(**_wan_vae_cfg(), device=DEV, dtype=BF)
    mean, std = list(vae.config.latents_mean), list(vae.config.latents_std)
    dec_o = AutoencoderKLWanDecoder(**_wan_vae_cfg(), latents_mean=mean, latents_std=std).eval()
    enc_o = AutoencoderKLWanEncoder(**_wan_vae_cfg()).eval()
    dsd, esd = vae_synthetic_state_dict(dec_o, 23), vae_synthetic_state_dict(enc_o, 24)
    dec_o.load_state_dict(dsd, strict=True)
    enc_o.load_state_dict(esd, strict=True)
    res = vae.load_state_dict({k: v.to(BF) for k, v in {**dsd, **esd}.items()}, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, (res.unexpected_keys, res.missing_keys)
    tile = (64, 64, 48, 48)
    for v in (vae, dec_o, enc_o):
        v.enable_tiling(*tile)
    px = (seeded((1, 3, height, width), 77) * 0.5).clamp(-1, 1).to(BF).float()      # pixels as the engine receives them
    lat0 = seeded((1, 16, (duration - 1) // 4 + 1, height // 8, width // 8), 41)
    pe, ne = seeded((1, 20, 64), 42).to(BF), seeded((1, 20, 64), 43).to(BF)
    gs = (3.5, 2.0)
    eng = WanI2VEngine(experts_h[0], experts_h[1], vae=vae, scheduler=UniPCMultistepScheduler(shift=3.0), boundary_ratio=0.875)
    kw = dict(image=px.to(DEV), prompt_embeds=pe.to(DEV), negative_prompt_embeds=ne.to(DEV), height=height, width=width,
              duration=duration, num_inference_steps=steps, high_noise_guidance_scale=gs[0], low_noise_guidance_scale=gs[1],
              latents=lat0.to(DEV))
    lat_hip = eng.run(return_latents=True, **kw)
    dec_hip = eng.run(**kw)
    frames_hip = tensor_to_frames(dec_hip, "np")
    used = []

    def chain(pol):
        st = (lambda x: x.to(BF)) if pol.emulate_bf16 else (lambda x: x.float())
        video = torch.cat([px[:, :, None], torch.zeros(1, 3, duration - 1, height, width)], dim=2)
        post = enc_o.encode(st(video).float(), policy=pol)
        lat_c = (st(post[:, :16]).float() - torch.tensor(mean).view(1, 16, 1, 1, 1)) * (1.0 / torch.tensor(std).view(1, 16, 1, 1, 1))
        mask = torch.zeros(1, 4, lat0.shape[2], lat0.shape[3], lat0.shape[4])
        mask[:, :, 0] = 1.0
        cond = torch.cat([mask, lat_c], dim=1)
        sch = UniPCMultistepScheduler(shift=3.0)
        ts = sch.set_timesteps(steps)
        used[:] = [bool(t >= 875.0) for t in ts]
        lat = lat0.clone()
        for t in ts:
            orc, scale = (experts_o[0], gs[0]) if bool(t >= 875.0) else (experts_o[1], gs[1])
            x = st(torch.cat([lat, cond], dim=1)).float()
            c = st(orc(x, t.expand(1).float(), pe.float(), policy=pol))
            u = st(orc(x, t.expand(1).float(), ne.float(), policy=pol))
            lat = sch.step((u + scale * (c - u)).to(torch.float32), t, lat, return_dict=False)[0]
        z = st(dec_o.denormalize_latents(lat.float())).float()
        dec = st(dec_o.decode(z, policy=pol))
        return lat, dec, video_to_uint8_frames(dec)

    ref16, ref32 = chain(OL.BF16_STORAGE), chain(OL.FP32)
    assert used[0] and not used[-1]
    _frame_report(f"wan i2v 4 steps, experts high/low = {used}", (lat_hip, dec_hip, frames_hip), ref16, ref32)
