"""Host-side contract of the TAEHV light VAE classes (no GPU): state-dict keys against the reference's key list
(tests/golden/vae_taehv.pt), checkpoint handling, the `use_light_vae` switch of the HunyuanVideo-1.5 VAE class."""
import os

import pytest
import torch


def test_light_vae_keys_equal_the_reference_decoder_keys(golden_dir):
    from apex_studio_amd.vae_taehv import AutoencoderKLHunyuanVideo15Light, TAEHV
    g = torch.load(os.path.join(golden_dir, "vae_taehv.pt"), weights_only=False)
    m = AutoencoderKLHunyuanVideo15Light(device="meta")
    assert sorted(m.state_dict().keys()) == g["keys"]
    assert m.taehv.frames_to_trim == 3 and m.taehv.patch_size == 2 and m.taehv.slope == 0.2
    t = TAEHV(checkpoint_path=None, device="meta")                     # the reference's defaults: wan21, patch 1, ReLU
    assert t.slope == 0.0 and t.patch_size == 1 and t.decoder[22].weight.shape == (3, 64, 3, 3)
    e = torch.load(os.path.join(golden_dir, "vae_taehv_encode.pt"), weights_only=False)
    full = TAEHV(checkpoint_path=None, model_type="hy15", patch_size=2, device="meta")        # encoder + decoder, as the reference
    assert sorted(k for k in full.state_dict() if k.startswith("encoder.")) == e["keys"]
    assert sorted("taehv." + k for k in full.state_dict() if k.startswith("decoder.")) == g["keys"]
    assert TAEHV(checkpoint_path=None, model_type="wan22", device="meta").decoder[1].weight.shape == (256, 48, 3, 3)
    assert TAEHV(checkpoint_path=None, decoder_time_upscale=(False, True), device="meta").frames_to_trim == 1


def test_checkpoint_handling(tmp_path):
    from safetensors.torch import save_file
    from apex_studio_amd.vae_taehv import AutoencoderKLHunyuanVideo15Light, TAEHV
    t = TAEHV(checkpoint_path=None, model_type="hy15", latent_channels=32, patch_size=2, with_encoder=False, device="cpu")
    sd = {k: torch.full_like(v, 0.5) for k, v in t.state_dict().items()}
    sd["encoder.0.weight"] = torch.zeros(64, 12, 3, 3, dtype=torch.bfloat16)       # dropped by a decode-only instance
    # a checkpoint trained with 4x temporal growth on a layer that is built with 2x: the LAST output channels are kept
    key = "decoder.13.conv.weight"
    big = torch.cat([torch.zeros(256, 128, 1, 1), torch.ones(256, 128, 1, 1)]).to(torch.bfloat16)
    sd[key] = big
    path = str(tmp_path / "taehv.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    light = AutoencoderKLHunyuanVideo15Light(taehv_checkpoint_path=path, device="cpu")
    assert float(light.taehv.decoder[13].conv.weight.float().mean()) == 1.0
    assert float(light.taehv.decoder[1].weight.float().mean()) == 0.5
    torch.save(sd, str(tmp_path / "taehv.pth"))
    t2 = TAEHV(checkpoint_path=str(tmp_path / "taehv.pth"), model_type="hy15", patch_size=2, with_encoder=False, device="cpu")
    assert torch.equal(t2.decoder[13].conv.weight, light.taehv.decoder[13].conv.weight)
    with pytest.raises(FileNotFoundError):
        AutoencoderKLHunyuanVideo15Light(taehv_checkpoint_path=str(tmp_path / "missing.safetensors"), device="cpu")
    with pytest.raises(ValueError):
        open(str(tmp_path / "x.bin"), "wb").close()
        AutoencoderKLHunyuanVideo15Light(taehv_checkpoint_path=str(tmp_path / "x.bin"), device="cpu")
    with pytest.raises(RuntimeError):
        t.encode_video(torch.zeros(1, 4, 3, 8, 8))                     # built without its encoder
    with pytest.raises(ValueError):
        t.decode_video(torch.zeros(1, 2, 16, 4, 4))                    # wrong latent channel count


def test_use_light_vae_switch_semantics():
    from apex_studio_amd.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    from apex_studio_amd.vae_taehv import AutoencoderKLHunyuanVideo15Light
    vcfg = dict(in_channels=3, out_channels=3, latent_channels=32, block_out_channels=(32, 64, 64, 128, 128),
                layers_per_block=1, spatial_compression_ratio=16, temporal_compression_ratio=4)
    vae = AutoencoderKLHunyuanVideo15(**vcfg, light_vae_path="/nonexistent/taehv.safetensors", device="meta")
    assert vae.light_vae is None and vae.use_light_vae is False        # meta build: nothing is read (reference model.py:813-819)
    vae.enable_tiling()
    assert vae.use_tiling and vae.use_light_vae is False
    with pytest.raises(FileNotFoundError):
        vae.enable_tiling(use_light_vae=True)                          # now it must exist
    assert vae.use_light_vae is False
    vae.set_light_vae(AutoencoderKLHunyuanVideo15Light(device="meta"))
    vae.enable_tiling(use_light_vae=True)
    assert vae.use_light_vae is True
    vae.enable_tiling()                                                # the engine's argument-free call keeps the switch
    assert vae.use_light_vae is True
    vae.enable_tiling(use_light_vae=False)
    assert vae.use_light_vae is False
