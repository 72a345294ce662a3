"""3-D causal VAE decode on HIP: conv / norm / resample ops against PyTorch fp32, and the whole decoder
against the reference's streaming output (tests/golden/vae_wan.pt) and the full-sequence oracle."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import layers as OL
from tests.golden.seeded import seeded, vae_synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("cin,cout,k,T,H,W", [(16, 16, (1, 1, 1), 3, 9, 11), (16, 128, (3, 3, 3), 3, 12, 10),
                                               (96, 96, (3, 3, 3), 5, 20, 24), (192, 384, (3, 1, 1), 4, 8, 8),
                                               (128, 64, (1, 3, 3), 3, 16, 16), (96, 3, (3, 3, 3), 2, 33, 17),
                                               (96, 96, (3, 3, 3), 1, 20, 24), (16, 128, (3, 3, 3), 1, 12, 10),
                                               (384, 384, (3, 3, 3), 1, 16, 16), (192, 96, (3, 1, 1), 1, 8, 8)])
def test_conv3d_cl(cin, cout, k, T, H, W):
    from apex_studio_amd import ops
    x = _bf(seeded((T, H, W, cin), 1))
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * k[0] * k[1] * k[2]) ** -0.5))
    b = _bf(seeded((cout,), 3) * 0.1)
    res = _bf(seeded((T, H, W, (cout + 3) // 4 * 4), 4))
    wp = ops.pack_conv_weight(w.to(DEV))
    bp = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    bp[:cout] = b.to(DEV)
    out = ops.conv3d_cl(x.to(DEV), wp, bp, k)
    xin = x.float().permute(3, 0, 1, 2)[None]                                     # [1, C, T, H, W]
    xin = F.pad(xin, ((k[2] - 1) // 2, (k[2] - 1) // 2, (k[1] - 1) // 2, (k[1] - 1) // 2, k[0] - 1, 0))
    ref = F.conv3d(xin, w.float(), b.float())[0].permute(1, 2, 3, 0)              # [T, H, W, Cout]
    assert _rel(out[..., :cout].cpu(), ref) < 4e-3
    assert torch.equal(out[..., cout:].cpu().float(), torch.zeros(T, H, W, wp.shape[0] - cout))
    out2 = ops.conv3d_cl(x.to(DEV), wp, bp, k, residual=res.to(DEV))
    assert _rel(out2[..., :cout].cpu(), ref + res.float()[..., :cout]) < 4e-3


def test_vae_elementwise_ops():
    from apex_studio_amd import ops
    for C in (96, 192, 384, 128, 1024, 640):
        x = _bf(seeded((2, 5, 7, C), 11) * 2)
        g = _bf(1 + 0.1 * seeded((C,), 12))
        for silu in (False, True):
            y = ops.rmsnorm_cl(x.to(DEV), g.to(DEV), silu=silu)
            ref = F.normalize(x.float(), dim=-1) * C ** 0.5 * g.float()
            if silu:
                ref = F.silu(ref)
            assert _rel(y.cpu(), ref) < 4e-3, (C, silu)
    x = _bf(seeded((3, 4, 5, 16), 13))
    up = ops.upsample2x_cl(x.to(DEV)).cpu()
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=(2.0, 2.0), mode="nearest-exact").permute(0, 2, 3, 1)
    assert torch.equal(up.float(), ref)
    x = _bf(seeded((3, 4, 5, 32), 14))
    il = ops.time_interleave_cl(x.to(DEV)).cpu()
    ref = torch.stack((x[..., :16], x[..., 16:]), dim=1).reshape(6, 4, 5, 16)
    assert torch.equal(il, ref)
    a, b = _bf(seeded((2, 12, 9, 8), 15)), _bf(seeded((2, 12, 9, 8), 16))
    for dim in (1, 2):
        E = 5
        av = a.to(DEV).narrow(dim, a.shape[dim] - E, E)
        bb = b.to(DEV).clone()
        ops.crossfade_(av, bb.narrow(dim, 0, E), dim=dim)
        w = (torch.arange(E) / E).view([-1 if d == dim else 1 for d in range(4)])
        ref = b.float().clone()
        ref.narrow(dim, 0, E).copy_(a.float().narrow(dim, a.shape[dim] - E, E) * (1 - w) + b.float().narrow(dim, 0, E) * w)
        assert torch.allclose(bb.cpu().float(), ref, atol=2e-2, rtol=1e-2)
        assert torch.equal(bb.cpu().narrow(dim, E, b.shape[dim] - E), b.narrow(dim, E, b.shape[dim] - E))


def _hip_vae(cfg, sd):
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    vae = AutoencoderKLWan(**cfg, device=DEV, dtype=torch.bfloat16)
    res = vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=False)   # one half at a time
    assert not res.unexpected_keys
    halves = (("encoder.", "quant_conv."), ("decoder.", "post_quant_conv."))
    assert any(all(k.startswith(h) for k in res.missing_keys) for h in halves), res.missing_keys[:4]
    return vae


def test_vae_decode_matches_streaming_reference_and_oracle(golden_dir):
    from oracle.vae_wan import AutoencoderKLWanDecoder
    g = torch.load(os.path.join(golden_dir, "vae_wan.pt"), weights_only=False)
    cfg = g["config"]
    orc = AutoencoderKLWanDecoder(**cfg).eval()
    sd = vae_synthetic_state_dict(orc, g["seed"])
    orc.load_state_dict(sd, strict=True)
    z = seeded(g["z_shape"], g["z_seed"]).to(torch.bfloat16)
    vae = _hip_vae(cfg, sd)
    assert sorted(k for k in vae.state_dict() if k.startswith(("decoder.", "post_quant_conv."))) == g["keys"]
    for tiled in (False, True):
        if tiled:
            vae.enable_tiling(*g["tile"])
            orc.enable_tiling(*g["tile"])
        out = vae.decode(z.to(DEV), return_dict=False)[0].float().cpu()
        ref_stream = g["tiled" if tiled else "untiled"].float()            # reference class, streaming, fp32->bf16
        ref16 = orc.decode(z.float(), policy=OL.BF16_STORAGE)
        ref32 = orc.decode(z.float())
        assert out.shape == ref_stream.shape and torch.isfinite(out).all()
        e_like, e_ref, e_emul = _rel(out, ref16), _rel(out, ref_stream), _rel(ref16, ref32)
        print(f"[vae tiled={tiled}] hip vs bf16-storage oracle {e_like:.3e}; vs reference streaming {e_ref:.3e}; "
              f"emulation vs fp32 {e_emul:.3e}")
        assert e_like < 2e-2, e_like
        assert e_ref < 2 * e_emul + 1e-2, (e_ref, e_emul)
    zn = vae.denormalize_latents(z.to(DEV).float())
    assert torch.allclose(zn[0, :, 0, 0, 0].cpu(), g["denorm_sample"], atol=2e-2)


def test_vae_first_frame_only_and_determinism():
    """T = 1 (the QwenImage image VAE case): the temporal upsamplers pass the single frame through."""
    from oracle.vae_wan import AutoencoderKLWanDecoder
    cfg = dict(base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=1, temperal_downsample=[False, True, True])
    orc = AutoencoderKLWanDecoder(**cfg).eval()
    sd = vae_synthetic_state_dict(orc, 17)
    orc.load_state_dict(sd, strict=True)
    z = seeded((1, 16, 1, 12, 12), 62).to(torch.bfloat16)
    vae = _hip_vae(cfg, sd)
    out = vae.decode(z.to(DEV), return_dict=False)[0]
    assert out.shape == (1, 3, 1, 96, 96)
    ref = orc.decode(z.float(), policy=OL.BF16_STORAGE)
    assert _rel(out.float().cpu(), ref) < 2e-2
    assert torch.equal(out, vae.decode(z.to(DEV), return_dict=False)[0])


def test_groupnorm_cl():
    from apex_studio_amd import ops
    for C, hw in ((128, (40, 36)), (256, (17, 9)), (512, (33, 31))):
        x = _bf(seeded((1,) + hw + (C,), 21) * 2 + 0.3)
        g, b = _bf(1 + 0.1 * seeded((C,), 22)), _bf(0.1 * seeded((C,), 23))
        for silu in (False, True):
            y = ops.groupnorm_cl(x.to(DEV), g.to(DEV), b.to(DEV), silu=silu)
            ref = F.group_norm(x.float().permute(0, 3, 1, 2), 32, g.float(), b.float(), eps=1e-6).permute(0, 2, 3, 1)
            if silu:
                ref = F.silu(ref)
            assert _rel(y.cpu(), ref) < 4e-3, (C, silu)


def test_flux_vae_decode_matches_oracle():
    """2-D VAE decoder vs the CPU restatement of diffusers' Decoder (parity unpinned against diffusers)."""
    from oracle.vae_flux import AutoencoderKLDecoder
    from apex_studio_amd.vae_flux import AutoencoderKL
    cfg = dict(latent_channels=16, block_out_channels=(32, 64, 128, 128), layers_per_block=1)
    orc = AutoencoderKLDecoder(**cfg).eval()
    sd = vae_synthetic_state_dict(orc, 19)
    for k in list(sd):                     # GroupNorm affine: weight ~ 1, bias small (the generic rule made them N(0,..))
        if ".norm" in k or "group_norm" in k or "conv_norm_out" in k:
            sd[k] = ((torch.ones_like(sd[k]) if k.endswith("weight") else torch.zeros_like(sd[k])) + 0.05 * sd[k].sign()).to(torch.bfloat16).float()   # bf16-representable, like every other weight
    orc.load_state_dict(sd, strict=True)
    vae = AutoencoderKL(**cfg, device=DEV, dtype=torch.bfloat16)
    assert sorted(vae.state_dict().keys()) == sorted(sd.keys())
    vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    z = seeded((1, 16, 20, 24), 63).to(torch.bfloat16)
    out = vae.decode(z.to(DEV), return_dict=False)[0].float().cpu()
    ref16 = orc.decode(z.float(), policy=OL.BF16_STORAGE)
    ref32 = orc.decode(z.float())
    assert out.shape == ref32.shape == (1, 3, 160, 192) and torch.isfinite(out).all()
    e_like, e_true, e_emul = _rel(out, ref16), _rel(out, ref32), _rel(ref16, ref32)
    print(f"[flux vae] hip vs bf16-storage oracle {e_like:.3e}; vs fp32 {e_true:.3e}; emulation vs fp32 {e_emul:.3e}")
    assert e_like < 2e-2 and e_true < 2 * e_emul + 1e-2
    assert abs(float(vae.denormalize_latents(torch.tensor(0.3611))) - 1.1159) < 1e-4


# ---- HunyuanVideo-1.5 VAE ---------------------------------------------------------

@pytest.mark.parametrize("cin,cout,T,H,W", [(32, 128, 3, 10, 14), (64, 3, 2, 33, 17), (128, 128, 1, 8, 8)])
def test_conv3d_cl_replicate(cin, cout, T, H, W):
    """HunyuanVideo15CausalConv3d: F.pad(mode="replicate") by (1, 1, 1, 1, 2, 0) then a valid 3x3x3 conv."""
    from apex_studio_amd import ops
    x = _bf(seeded((T, H, W, cin), 1))
    w = _bf(seeded((cout, cin, 3, 3, 3), 2, scale=(cin * 27) ** -0.5))
    b = _bf(seeded((cout,), 3) * 0.1)
    res = _bf(seeded((T, H, W, (cout + 3) // 4 * 4), 4))
    wp = ops.pack_conv_weight(w.to(DEV))
    bp = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    bp[:cout] = b.to(DEV)
    out = ops.conv3d_cl(x.to(DEV), wp, bp, (3, 3, 3), replicate=True)
    xin = F.pad(x.float().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 2, 0), mode="replicate")
    ref = F.conv3d(xin, w.float(), b.float())[0].permute(1, 2, 3, 0)
    assert _rel(out[..., :cout].cpu(), ref) < 4e-3
    zero = F.conv3d(F.pad(x.float().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 2, 0)), w.float(), b.float())[0].permute(1, 2, 3, 0)
    assert _rel(out[..., :cout].cpu(), zero) > 5e-2                  # the padding mode is observable
    out2 = ops.conv3d_cl(x.to(DEV), wp, bp, (3, 3, 3), residual=res.to(DEV), replicate=True)
    assert _rel(out2[..., :cout].cpu(), ref + res.float()[..., :cout]) < 4e-3


@pytest.mark.parametrize("D,frames,per", [(128, 3, 80), (256, 4, 35), (1024, 5, 48), (128, 1, 64)])
def test_attention_framecausal(D, frames, per):
    """Mid-block attention of the HunyuanVideo-1.5 VAE (model.py:137-176): one head of the full channel width, token i
    sees the keys of frames <= its own."""
    from apex_studio_amd import ops
    S = frames * per
    q, k, v = (_bf(seeded((1, 1, S, D), 30 + i)) for i in range(3))
    out = ops.attention_framecausal(q.to(DEV), k.to(DEV), v.to(DEV), per).cpu()
    fr = torch.arange(S) // per
    mask = fr[:, None] >= fr[None, :]
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=mask)
    assert out.shape == ref.shape and _rel(out, ref) < 1e-2
    if frames > 1:
        assert _rel(out, F.scaled_dot_product_attention(q.float(), k.float(), v.float())) > 5e-2


def test_add_bf16():
    from apex_studio_amd import ops
    a, b = _bf(seeded((3, 5, 7, 64), 40)), _bf(seeded((3, 5, 7, 64), 41))
    assert torch.equal(ops.add(a.to(DEV), b.to(DEV)).cpu(), _bf(a.float() + b.float()))


def test_hunyuan15_vae_decode_matches_reference_and_oracle(golden_dir):
    from oracle.vae_hunyuan15 import AutoencoderKLHunyuanVideo15 as Orc
    from apex_studio_amd.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    g = torch.load(os.path.join(golden_dir, "vae_hunyuan15.pt"), weights_only=False)
    cfg = g["config"]
    orc = Orc(**cfg).eval()
    sd = vae_synthetic_state_dict(orc, g["seed"])
    orc.load_state_dict(sd, strict=True)
    z = seeded(g["z_shape"], g["z_seed"]).to(torch.bfloat16)
    vae = AutoencoderKLHunyuanVideo15(**cfg, device=DEV, dtype=torch.bfloat16)
    vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    assert sorted(k for k in vae.state_dict() if k.startswith("decoder.")) == g["keys"]
    for tiled in (False, True):
        if tiled:
            vae.enable_tiling()
            orc.enable_tiling()
        out = vae.decode(z.to(DEV), return_dict=False)[0].float().cpu()
        ref = g["tiled" if tiled else "untiled"].float()                     # the reference class, fp32 -> bf16
        ref16 = orc.decode(z.float(), policy=OL.BF16_STORAGE)
        ref32 = orc.decode(z.float())
        assert out.shape == ref.shape and torch.isfinite(out).all()
        e_like, e_ref, e_emul = _rel(out, ref16), _rel(out, ref), _rel(ref16, ref32)
        print(f"[hy15 vae tiled={tiled}] hip vs bf16-storage oracle {e_like:.3e}; vs reference {e_ref:.3e}; "
              f"emulation vs fp32 {e_emul:.3e}")
        assert e_like < 2e-2, e_like
        assert e_ref < 2 * e_emul + 1e-2, (e_ref, e_emul)
    assert torch.equal(vae.decode(z.to(DEV), return_dict=False)[0], vae.decode(z.to(DEV), return_dict=False)[0])
    zn = vae.denormalize_latents(z.to(DEV).float())
    assert torch.allclose(zn.cpu(), z.float() / cfg.get("scaling_factor", 1.03682), atol=1e-6)


def test_group_mean():
    from apex_studio_amd import ops
    for C, gs in ((64, 16), (32, 2), (256, 4), (96, 8)):
        x = _bf(seeded((3, 5, 7, C * gs), 45) * 2)
        out = ops.group_mean(x.to(DEV), C).cpu()
        ref = x.float().view(3, 5, 7, C, gs).mean(dim=-1)
        assert out.shape == (3, 5, 7, C) and torch.equal(out, _bf(ref)), (C, gs)


def test_hunyuan15_vae_encode_matches_reference_and_oracle(golden_dir):
    """HunyuanVideo-1.5 VAE ENCODE (image-to-video conditioning): one frame, a 5-frame clip (the first-frame rule of the
    temporal downsamplers), and a tiled encode, against the reference class run in the build container and the oracle."""
    from oracle.vae_hunyuan15 import AutoencoderKLHunyuanVideo15 as Orc
    from apex_studio_amd.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    g = torch.load(os.path.join(golden_dir, "vae_hunyuan15_encode.pt"), weights_only=False)
    cfg = g["config"]
    orc = Orc(**cfg).eval()
    sd = vae_synthetic_state_dict(orc, g["seed"])
    orc.load_state_dict(sd, strict=True)
    vae = AutoencoderKLHunyuanVideo15(**cfg, device=DEV, dtype=torch.bfloat16)
    vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    assert sorted(vae.state_dict().keys()) == g["keys"]
    for name in ("image", "clip", "tiled"):
        c = g[name]
        x = seeded(c["shape"], c["seed"]).clamp(-1, 1).to(torch.bfloat16)
        if name == "tiled":
            vae.enable_tiling(tile_sample_min_height=c["tile"], tile_sample_min_width=c["tile"],
                              tile_latent_min_height=c["tile"] // 16, tile_latent_min_width=c["tile"] // 16)
            orc.enable_tiling()
        post = vae.encode(x.to(DEV), return_dict=False)[0]
        out = post.parameters.float().cpu()
        kw = dict(tile_sample_min=c["tile"]) if name == "tiled" else {}
        ref16 = orc.encode(x.float(), policy=OL.BF16_STORAGE, **kw)
        ref32 = orc.encode(x.float(), **kw)
        ref = c["moments"].float()
        assert out.shape == ref.shape and torch.isfinite(out).all()
        e_like, e_ref, e_emul = _rel(out, ref16), _rel(out, ref), _rel(ref16, ref32)
        print(f"[hy15 vae encode {name}] hip vs bf16-storage oracle {e_like:.3e}; vs reference {e_ref:.3e}; "
              f"emulation vs fp32 {e_emul:.3e}")
        assert e_like < 2e-2, e_like
        assert e_ref < 2 * e_emul + 1e-2, (e_ref, e_emul)
        assert torch.equal(post.mode(), post.parameters[:, :cfg["latent_channels"]])
    assert torch.allclose(vae.normalize_latents(torch.tensor(2.0)), torch.tensor(2.0 * 1.03682))


def test_hunyuan15_vae_tiles_on_side_streams_are_bit_identical(golden_dir):
    """`decode_streams` tiles of the tiled decode run on their own HIP streams (the low-resolution stages launch far fewer
    workgroups than the chip has slots): same kernels on the same data, so 1, 2 and 5 streams must give the same bits, repeated
    calls included (stream reuse, allocator hand-over between streams)."""
    from apex_studio_amd.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    cfg = torch.load(os.path.join(golden_dir, "vae_hunyuan15.pt"), weights_only=False)["config"]
    vae = AutoencoderKLHunyuanVideo15(**cfg, device=DEV, dtype=torch.bfloat16)
    vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in vae_synthetic_state_dict(vae, 3).items()}, strict=True)
    vae.enable_tiling(tile_sample_min_height=64, tile_sample_min_width=64, tile_latent_min_height=4, tile_latent_min_width=4)
    z = seeded((1, cfg["latent_channels"], 3, 10, 14), 5).to(DEV).to(torch.bfloat16)          # 4 x 5 = 20 tiles
    outs = {}
    vae.batch_head_blocks = 0
    for ns in (1, 2, 5, 2, 1):
        vae.decode_streams = ns
        o = vae.decode(z, return_dict=False)[0]
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all()
        outs.setdefault(ns, o)
        assert torch.equal(o, outs[1]), ns
    # ... and with the lowest-resolution stages of all equally shaped tiles in ONE launch per layer (clips stacked along T,
    # apexmi_conv3d_cl_clips; the first-frame rule of the temporal upsamplers per clip; frame-causal attention per clip)
    for nb in (1, 2, 4):
        vae.batch_head_blocks, vae.decode_streams = nb, 2
        o = vae.decode(z, return_dict=False)[0]
        torch.cuda.synchronize()
        assert torch.equal(o, outs[1]), nb


def test_wan_vae_tiles_on_side_streams_are_bit_identical():
    """`tile_streams` video tiles of the tiled Wan decode / encode run on their own HIP streams: 1, 2 and 4 streams must give the
    same bits, repeated calls included."""
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    vae = AutoencoderKLWan(base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=1, temperal_downsample=[False, True, True],
                           device=DEV, dtype=torch.bfloat16)
    vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in vae_synthetic_state_dict(vae, 9).items()}, strict=True)
    vae.enable_tiling(64, 64, 48, 48)
    z = seeded((1, 16, 3, 20, 26), 4).to(DEV).to(torch.bfloat16)             # 3 x 4 latent tiles of 8 x 8, stride 6
    x = seeded((1, 3, 5, 160, 176), 6).clamp(-1, 1).to(DEV).to(torch.bfloat16)
    dec, enc = {}, {}
    for ns in (1, 2, 4, 2, 1):
        vae.tile_streams = ns
        d = vae.decode(z, return_dict=False)[0]
        e = vae.encode(x, return_dict=False)[0].parameters
        torch.cuda.synchronize()
        assert torch.isfinite(d.float()).all() and torch.isfinite(e.float()).all()
        dec.setdefault(ns, d)
        enc.setdefault(ns, e)
        assert torch.equal(d, dec[1]) and torch.equal(e, enc[1]), ns


# ---- Wan / QwenImage VAE encode (the B-model `.encode` contract) ------------------------------------

@pytest.mark.parametrize("cin,cout,T,H,W", [(32, 32, 3, 16, 20), (64, 64, 1, 33, 18), (96, 96, 2, 64, 48)])
def test_conv2d_cl_down2(cin, cout, T, H, W):
    """WanResample "downsample2d": ZeroPad2d((0, 1, 0, 1)) + Conv2d(dim, dim, 3, stride=2), per frame."""
    from apex_studio_amd import ops
    x = _bf(seeded((T, H, W, cin), 1))
    w = _bf(seeded((cout, cin, 3, 3), 2, scale=(cin * 9) ** -0.5))
    b = _bf(seeded((cout,), 3) * 0.1)
    wp = ops.pack_conv_weight(w.to(DEV))
    out = ops.conv2d_cl_down2(x.to(DEV), wp, b.to(DEV))
    ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), w.float(), b.float(), stride=2).permute(0, 2, 3, 1)
    assert out.shape == ref.shape and _rel(out.cpu(), ref) < 4e-3


def test_wan_vae_encode_matches_streaming_reference_and_oracle(golden_dir):
    from oracle.vae_wan import AutoencoderKLWanEncoder
    g = torch.load(os.path.join(golden_dir, "vae_wan_encode.pt"), weights_only=False)
    cfg = g["config"]
    orc = AutoencoderKLWanEncoder(**cfg).eval()
    sd = vae_synthetic_state_dict(orc, g["seed"])
    orc.load_state_dict(sd, strict=True)
    vae = _hip_vae(cfg, sd)
    assert sorted(k for k in vae.state_dict() if k.startswith(("encoder.", "quant_conv."))) == g["keys"]
    x = seeded(g["x_shape"], g["x_seed"]).to(torch.bfloat16)
    for tiled in (False, True):
        if tiled:
            vae.enable_tiling(*g["tile"])
            orc.enable_tiling(*g["tile"])
        for name, xin in (("video", x), ("image", x[:, :, :1])):
            post = vae.encode(xin.to(DEV), return_dict=False)[0]
            got = post.parameters.float().cpu()
            ref = g[name + ("_tiled" if tiled else "")]                       # the reference class, streaming, fp32
            ref16, ref32 = orc.encode(xin.float(), policy=OL.BF16_STORAGE), orc.encode(xin.float())
            assert got.shape == ref.shape and torch.isfinite(got).all()
            e_like, e_ref, e_emul = _rel(got, ref16), _rel(got, ref), _rel(ref16, ref32)
            print(f"[vae encode {name} tiled={tiled}] hip vs bf16-storage oracle {e_like:.3e}; vs reference streaming {e_ref:.3e}; "
                  f"emulation vs fp32 {e_emul:.3e}")
            assert e_like < 2e-2 and e_ref < 2 * e_emul + 1e-2
            assert torch.equal(post.mode(), post.parameters[:, :cfg["z_dim"]])
    gen = torch.Generator(device=DEV).manual_seed(5)
    smp = post.sample(gen)
    assert smp.shape == post.mode().shape and not torch.equal(smp, post.mode())
    lat = post.mode().float()
    assert torch.allclose(vae.denormalize_latents(vae.normalize_latents(lat)), lat, atol=1e-4)
    with pytest.raises(ValueError):
        vae.encode(x[:, :, :3].to(DEV))


def test_single_frame_tiles_batched_pass_is_bit_identical():
    """An image (T = 1) decodes / encodes its spatial tiles in shape groups, every frame of a stacked tensor treated as
    an independent one-frame clip (`apexmi_conv3d_cl_frames`): same bits as one pass per tile, and the independent-frame
    convolution equals per-frame calls."""
    from apex_studio_amd import ops
    from oracle.vae_wan import AutoencoderKLWanDecoder
    cfg = dict(base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=1, temperal_downsample=[False, True, True])
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    vae = AutoencoderKLWan(**cfg, device=DEV, dtype=torch.bfloat16)
    vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in vae_synthetic_state_dict(vae, 23).items()}, strict=True)
    vae.enable_tiling(48, 48, 32, 32)
    z = seeded((1, 16, 1, 14, 10), 81).to(torch.bfloat16).to(DEV)
    x = seeded((1, 3, 1, 112, 80), 82).clamp(-1, 1).to(torch.bfloat16).to(DEV)
    outs = []
    for batched in (True, False):
        vae.batch_single_frame_tiles = batched
        outs.append((vae.decode(z, return_dict=False)[0], vae.encode(x, return_dict=False)[0].parameters))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].shape == (1, 3, 1, 112, 80) and outs[0][1].shape == (1, 32, 1, 14, 10)
    orc = AutoencoderKLWanDecoder(**cfg).eval()
    orc.load_state_dict({k: v for k, v in vae_synthetic_state_dict(vae, 23).items() if k in orc.state_dict()}, strict=True)
    orc.enable_tiling(48, 48, 32, 32)
    assert _rel(outs[0][0].float().cpu(), orc.decode(z.float().cpu(), policy=OL.BF16_STORAGE)) < 2e-2
    # the op itself: N frames in one launch == N single-frame launches
    for cin, cout in ((16, 96), (96, 96)):
        xs = _bf(seeded((5, 12, 10, cin), 83)).to(DEV)
        w = ops.pack_conv_weight(_bf(seeded((cout, cin, 3, 3, 3), 84, scale=(27 * cin) ** -0.5)).to(DEV))
        b = _bf(seeded((cout,), 85) * 0.1).to(DEV)
        one = torch.cat([ops.conv3d_cl(xs[i:i + 1].contiguous(), w, b, (3, 3, 3)) for i in range(5)], dim=0)
        assert torch.equal(ops.conv3d_cl(xs, w, b, (3, 3, 3), independent_frames=True), one)
        assert not torch.equal(ops.conv3d_cl(xs, w, b, (3, 3, 3)), one)            # as one clip the frames DO mix


@pytest.mark.parametrize("cin,cout,T,H,W,k", [(64, 64, 3, 9, 11, (1, 3, 3)), (96, 48, 1, 16, 8, (1, 3, 3)), (32, 32, 2, 5, 7, (3, 3, 3))])
def test_conv_reads_through_a_nearest_2x_upsample(cin, cout, T, H, W, k):
    """`apexmi_conv3d_cl_up2`: the convolution of the upsampled image without materialising it (WanUpsample + Conv2d,
    reference vae/wan/model.py:225-237, :264-273) must equal upsample2x_cl followed by the plain convolution, bit for bit."""
    from apex_studio_amd import ops
    x = _bf(seeded((T, H, W, cin), 1)).to(DEV)
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * 9) ** -0.5)).to(DEV)
    b = _bf(seeded((cout,), 3) * 0.1).to(DEV)
    wp = ops.pack_conv_weight(w)
    bp = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    bp[:cout] = b
    two = ops.conv3d_cl(ops.upsample2x_cl(x), wp, bp, k)
    one = ops.conv3d_cl(x, wp, bp, k, upsample2x=True)
    assert one.shape == two.shape == (T, 2 * H, 2 * W, wp.shape[0]) and torch.equal(one, two)
    ref = F.conv3d(F.pad(F.interpolate(x.float().cpu().permute(3, 0, 1, 2)[None], scale_factor=(1, 2, 2), mode="nearest"),
                         (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] - 1, 0)), w.float().cpu(), b.float().cpu())[0]
    assert _rel(one[..., :cout].cpu(), ref.permute(1, 2, 3, 0)) < 4e-3


@pytest.mark.parametrize("cin,cout,T,H,W,k,up", [(96, 96, 5, 128, 128, (3, 3, 3), False), (192, 192, 3, 150, 160, (3, 3, 3), False),
                                                 (192, 96, 2, 96, 100, (1, 3, 3), True), (384, 384, 9, 96, 96, (3, 3, 3), False),
                                                 (96, 3, 3, 160, 144, (3, 3, 3), False), (256, 256, 1, 300, 260, (1, 3, 3), False),
                                                 (64, 64, 2, 200, 180, (3, 3, 3), False), (96, 96, 1, 300, 300, (3, 3, 3), False)])
def test_conv_v2_tiles_are_bit_identical_to_the_128x128_kernel(cin, cout, T, H, W, k, up):
    """The conv-shaped tilings (512x96, 256x192, 256x256, 512x32/64; `conv.v2`) gather through buffer_load ... lds with a
    per-piece tap-validity mask; same K order per output element as the 128x128 kernel, so the results must be equal
    bit for bit — ragged M, the upsample fold, the single-frame tap skip and bias + residual included."""
    from apex_studio_amd import lib, ops
    x = _bf(seeded((T, H, W, cin), 1)).to(DEV)
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * k[0] * k[1] * k[2]) ** -0.5)).to(DEV)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    b[:cout] = _bf(seeded((cout,), 3) * 0.1).to(DEV)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    res = _bf(seeded((T, Ho, Wo, wp.shape[0]), 4)).to(DEV)
    outs = []
    try:
        lib.tune_set("conv.slab", 0)       # the direct-convolution kernels take most of these shapes otherwise (tests below)
        for v2 in (0, 1):
            lib.tune_set("conv.v2", v2)
            outs.append((ops.conv3d_cl(x, wp, b, k, upsample2x=up), ops.conv3d_cl(x, wp, b, k, residual=res, upsample2x=up)))
    finally:
        lib.tune_set("conv.v2", 1)
        lib.tune_set("conv.slab", 2)
    assert T * Ho * Wo >= 65536, "shape must be large enough for the v2 dispatch"
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


SLAB_CASES = [   # cin, cout, T, H, W, k, up, norm, res, independent
    (96, 96, 5, 128, 128, (3, 3, 3), False, False, True, False),       # Wan full-resolution stage
    (96, 96, 3, 131, 173, (3, 3, 3), False, True, False, False),       # ragged tiles, fused norm
    (96, 3, 3, 160, 144, (3, 3, 3), False, False, False, False),       # conv_out: Cout 3 (+1 pad), narrow epilogue
    (192, 192, 3, 150, 160, (3, 3, 3), False, True, True, False),      # 8 x 32 tiles, 192 channels in one wave, fused norm
    (384, 384, 9, 96, 96, (3, 3, 3), False, False, True, False),       # two N tiles (blockIdx.y), 8 channel slices
    (192, 96, 2, 96, 100, (1, 3, 3), True, False, False, False),       # read through the 2x upsample
    (384, 192, 3, 80, 72, (1, 3, 3), True, True, False, False),
    (96, 64, 7, 120, 101, (2, 3, 3), False, False, True, False),       # kT = 2
    (192, 192, 12, 96, 96, (3, 3, 3), False, False, False, True),      # independent single-frame clips
    (144, 160, 3, 150, 150, (3, 3, 3), False, False, False, False),    # Cout not a multiple of 32: masked columns
    (48, 32, 4, 200, 180, (3, 3, 3), False, False, False, False),      # one slice: the implicit GEMM's summation order
]


@pytest.mark.parametrize("cin,cout,T,H,W,k,up,norm,res,indep", SLAB_CASES)
def test_direct_convolution_kernels_against_the_implicit_gemm(cin, cout, T, H, W, k, up, norm, res, indep):
    """`conv.slab`: the direct convolution (haloed input slab staged once per temporal tap and 48-channel slice, spatial taps
    as shifted LDS reads) against the implicit-GEMM tiles on the same operands.  Same products, f32 sums in another order
    (temporal tap -> slice -> spatial tap): at most 1 bf16 ulp apart on a few 1e-4 of the outputs (the fused norm output may
    move by one more ulp), bit-identical when Cin is one slice and the temporal taps run oldest-first (`conv.torder=0`; shipped: by
    input frame mod 3, for L2 reuse between the workgroups of consecutive frames); deterministic; and the order-preserving 8 x 32 form
    (`conv.slab=1`, Cin = 96) is bit-identical to the implicit GEMM."""
    from apex_studio_amd import lib, ops
    x = _bf(seeded((T, H, W, cin), 1)).to(DEV)
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * k[0] * 9) ** -0.5)).to(DEV)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    b[:cout] = _bf(seeded((cout,), 3) * 0.1).to(DEV)
    g = _bf(1 + 0.1 * seeded((wp.shape[0],), 5)).to(DEV)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    r = _bf(seeded((T, Ho, Wo, wp.shape[0]), 4)).to(DEV) if res else None
    assert T * Ho * Wo >= 65536

    def run():
        if norm:
            raw, nrm = ops.conv3d_cl_norm(x, wp, b, k, g, silu=True, residual=r, upsample2x=up, independent_frames=indep)
            return torch.cat([raw.flatten(), nrm.flatten()])
        return ops.conv3d_cl(x, wp, b, k, residual=r, upsample2x=up, independent_frames=indep).flatten()

    outs = {}
    try:
        for v in (0, 1, 2):
            lib.tune_set("conv.slab", v)
            outs[v] = run()
        assert torch.equal(run(), outs[2])
    finally:
        lib.tune_set("conv.slab", 2)
    ref, got = outs[0].float(), outs[2].float()
    rel = float((got - ref).norm() / ref.norm())
    frac = float((outs[2] != outs[0]).float().mean())
    ulps = float(((got - ref).abs() / (ref.abs() * 2.0 ** -7 + 1e-3)).max())
    print(f"[slab] {cin}->{cout} k{k} up={up} norm={norm}: rel {rel:.2e}, {frac:.2e} of outputs differ, max {ulps:.2f} ulp")
    assert rel < 1e-4 and frac < 2e-3 and ulps <= (2.01 if norm else 1.01)
    if cin == 48:      # one slice: with the temporal taps oldest-first (conv.torder = 0) the order IS the implicit GEMM's
        try:
            lib.tune_set("conv.torder", 0)
            assert torch.equal(run(), outs[0])
        finally:
            lib.tune_set("conv.torder", 1)
    if cin == 96 and cout <= 96 and not up:
        assert torch.equal(outs[1], outs[0]), "the 8 x 32 form keeps the implicit GEMM's summation order"
    n = min(cout, 8)
    xin = x.float().cpu().permute(3, 0, 1, 2)[None]
    if up:
        xin = F.interpolate(xin, scale_factor=(1, 2, 2))
    if not norm and not indep:        # and against torch on a few output channels
        ref_t = F.conv3d(F.pad(xin, (1, 1, 1, 1, k[0] - 1, 0)), w[:n].float().cpu(), b[:n].float().cpu())[0].permute(1, 2, 3, 0)
        if res:
            ref_t = ref_t + r[..., :n].float().cpu()
        mine = outs[2].view(T, Ho, Wo, wp.shape[0])[..., :n].float().cpu()
        assert _rel(mine, ref_t) < 4e-3


@pytest.mark.parametrize("cin,cout,T,H,W,k,up,silu,res", [(96, 96, 5, 128, 128, (3, 3, 3), False, True, True),
                                                          (192, 192, 3, 150, 160, (3, 3, 3), False, True, False),
                                                          (192, 96, 2, 96, 100, (1, 3, 3), True, True, False),
                                                          (96, 96, 1, 300, 300, (3, 3, 3), False, False, True),
                                                          (384, 384, 3, 160, 160, (3, 3, 3), False, True, True),
                                                          (64, 64, 2, 20, 20, (3, 3, 3), False, True, False)])
def test_conv_with_fused_output_rmsnorm(cin, cout, T, H, W, k, up, silu, res):
    """`apexmi_conv3d_cl_norm`: the RMS norm (+ SiLU) of the conv's output produced in the conv's epilogue.  Raw output
    bit-identical to the plain conv; normed output against the separate rmsnorm_cl pass over that raw output: same
    values read, f32 sum in a different order -> at most isolated 1-ulp flips (bar 2e-5 rel L2, 1 ulp).  The last two
    shapes are NOT fusable (two N tiles / too small): the wrapper must give the same through its fallback."""
    from apex_studio_amd import ops
    x = _bf(seeded((T, H, W, cin), 1)).to(DEV)
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * k[0] * k[1] * k[2]) ** -0.5)).to(DEV)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    b[:cout] = _bf(seeded((cout,), 3) * 0.1).to(DEV)
    g = _bf(1 + 0.1 * seeded((cout,), 5)).to(DEV)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    r = _bf(seeded((T, Ho, Wo, cout), 4)).to(DEV) if res else None
    assert ops.conv3d_cl_norm_fusable(x, cout, up) == (cout <= 192 and T * Ho * Wo >= 65536)
    plain = ops.conv3d_cl(x, wp, b, k, residual=r, upsample2x=up)
    want = ops.rmsnorm_cl(plain, g, silu=silu)
    raw, normed = ops.conv3d_cl_norm(x, wp, b, k, g, silu=silu, residual=r, upsample2x=up)
    assert torch.equal(raw, plain)
    none, normed2 = ops.conv3d_cl_norm(x, wp, b, k, g, silu=silu, residual=r, upsample2x=up, want_raw=False)
    assert none is None and torch.equal(normed2, normed)
    rel = _rel(normed.cpu(), want.cpu())
    frac = float((normed != want).float().mean())
    print(f"[fused norm] {cin}->{cout} up={up} silu={silu}: vs the separate pass rel {rel:.2e}, {frac:.2e} of elements differ")
    assert rel < 2e-5 and frac < 1e-3
    d = (normed.float() - want.float()).abs()
    assert float((d / want.float().abs().clamp_min(1e-3)).max()) < 2.0 ** -6


@pytest.mark.parametrize("cin,cout,T,H,W", [(64, 64, 9, 10, 12), (96, 96, 5, 7, 9), (32, 32, 3, 16, 16)])
def test_conv3d_cl_temporal_stride(cin, cout, T, H, W):
    """WanResample "downsample3d" time_conv in full-sequence form: output j >= 1 = causal 3x1x1 conv ending at frame 2j —
    equal (bit for bit) to convolving every frame and keeping frames 2, 4, ..., and to Conv3d(stride=(2,1,1)) on frames 0.."""
    from apex_studio_amd import ops
    x = _bf(seeded((T, H, W, cin), 301))
    w = _bf(seeded((cout, cin, 3, 1, 1), 302, scale=(3 * cin) ** -0.5))
    b = _bf(seeded((cout,), 303) * 0.1)
    wp = ops.pack_conv_weight(w.to(DEV))
    bp = b.to(DEV)
    To = (T - 1) // 2
    out = ops.conv3d_cl_tstrided(x.to(DEV), wp, bp, (3, 1, 1), 2, 2, To)
    full = ops.conv3d_cl(x.to(DEV), wp, bp, (3, 1, 1))
    assert out.shape == (To, H, W, cout) and torch.equal(out, full[2::2])
    ref = F.conv3d(x.float().permute(3, 0, 1, 2)[None], w.float(), b.float(), stride=(2, 1, 1))[0].permute(1, 2, 3, 0)
    assert ref.shape == out.shape and _rel(out.cpu(), ref) < 4e-3
    with pytest.raises(RuntimeError):
        ops.conv3d_cl_tstrided(x.to(DEV), wp, bp, (3, 1, 1), 2, 2, To + 1)


@pytest.mark.parametrize("cin,cout,T,H,W", [(256, 256, 5, 120, 128), (128, 256, 3, 150, 161), (512, 512, 9, 96, 96),
                                             (64, 3, 3, 160, 144), (1024, 1024, 3, 160, 144)])
def test_conv_v2_tiles_with_replicate_padding(cin, cout, T, H, W):
    """HunyuanVideo15CausalConv3d (replicate padding = clamped tap coordinates) on the conv-shaped tiles: the gather clamps
    (t, y, x) per tap instead of testing a validity mask; bit-identical to the 128x128 kernel's replicate mode."""
    from apex_studio_amd import lib, ops
    k = (3, 3, 3)
    x = _bf(seeded((T, H, W, cin), 1)).to(DEV)
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * 27) ** -0.5)).to(DEV)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    b[:cout] = _bf(seeded((cout,), 3) * 0.1).to(DEV)
    res = _bf(seeded((T, H, W, wp.shape[0]), 4)).to(DEV)
    outs = []
    try:
        lib.tune_set("conv.slab", 0)       # (the 64-channel-slice direct convolution takes most of these otherwise)
        for v2 in (0, 1):
            lib.tune_set("conv.v2", v2)
            outs.append(ops.conv3d_cl(x, wp, b, k, residual=res, replicate=True))
    finally:
        lib.tune_set("conv.v2", 1)
        lib.tune_set("conv.slab", 2)
    assert T * H * W >= 65536 and torch.equal(outs[0], outs[1])
    n = min(cout, 8)
    xin = F.pad(x.float().cpu().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 2, 0), mode="replicate")
    ref = F.conv3d(xin, w[:n].float().cpu(), b[:n].float().cpu())[0].permute(1, 2, 3, 0) + res[..., :n].float().cpu()
    assert _rel(outs[1][..., :n].float().cpu(), ref) < 4e-3


@pytest.mark.parametrize("cin,cout,T,H,W,k,repl,res", [
    (128, 128, 9, 128, 128, (3, 3, 3), True, True),       # HunyuanVideo-1.5 last stage: replicate padding, Cout 128
    (256, 128, 5, 107, 131, (3, 3, 3), True, False),      # ragged tiles
    (256, 256, 5, 120, 128, (3, 3, 3), True, True),       # two N tiles
    (512, 1024, 3, 160, 144, (3, 3, 3), True, False),     # DCAE up-projection widths
    (128, 128, 1, 512, 512, (1, 3, 3), False, True),      # Flux 2-D VAE (zero padding, one frame)
    (512, 512, 1, 256, 256, (1, 3, 3), False, False),
    (256, 256, 6, 120, 106, (1, 3, 3), False, False),     # TAEHV stage
    (64, 128, 4, 160, 144, (2, 3, 3), False, True),       # kT = 2
])
def test_direct_convolution_with_64_channel_slices(cin, cout, T, H, W, k, repl, res):
    """The slab kernel's second slice width (64 channels, 128-byte pitch, chunk ^ ((p >> 1) & 7), 8 x 32 tiles x 128 output
    channels) for channel counts that are multiples of 64 but not of 48, with zero OR replicate padding — against the
    implicit-GEMM kernels on the same operands (f32 summation order differs: <= 1 bf16 ulp on a few 1e-4 of the outputs) and
    against torch on a few channels."""
    from apex_studio_amd import lib, ops
    x = _bf(seeded((T, H, W, cin), 1)).to(DEV)
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * k[0] * 9) ** -0.5)).to(DEV)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=DEV)
    b[:cout] = _bf(seeded((cout,), 3) * 0.1).to(DEV)
    r = _bf(seeded((T, H, W, wp.shape[0]), 4)).to(DEV) if res else None
    assert T * H * W >= 65536
    outs = {}
    try:
        for v in (0, 2):
            lib.tune_set("conv.slab", v)
            outs[v] = ops.conv3d_cl(x, wp, b, k, residual=r, replicate=repl)
        assert torch.equal(ops.conv3d_cl(x, wp, b, k, residual=r, replicate=repl), outs[2])
    finally:
        lib.tune_set("conv.slab", 2)
    ref, got = outs[0].float(), outs[2].float()
    rel = float((got - ref).norm() / ref.norm())
    frac = float((outs[2] != outs[0]).float().mean())
    ulps = float(((got - ref).abs() / (ref.abs() * 2.0 ** -7 + 1e-3)).max())
    print(f"[slab64] {cin}->{cout} k{k} replicate={repl}: rel {rel:.2e}, {frac:.2e} of outputs differ, max {ulps:.2f} ulp")
    assert rel < 1e-4 and frac < 2e-3 and ulps <= 1.01
    n = 8
    xin = F.pad(x.float().cpu().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, k[0] - 1, 0), mode="replicate" if repl else "constant")
    ref_t = F.conv3d(xin, w[:n].float().cpu(), b[:n].float().cpu())[0].permute(1, 2, 3, 0)
    if res:
        ref_t = ref_t + r[..., :n].float().cpu()
    assert _rel(outs[2][..., :n].float().cpu(), ref_t) < 4e-3
