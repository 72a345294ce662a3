"""The oracle's diffusers-leaf restatements (oracle/layers.py, oracle/vae_flux.py) and the FlowMatch-Euler step against
the reference's OWN in-tree copies of those leaves, run in the build container by tests/golden/make_golden.py
(`gen_leaf_pins`; the fixture holds seeds, shapes and outputs only).  diffusers itself is un-vendored and absent, but the
reference tree carries copies of nearly every leaf its hot path takes from it (file:line per entry in gen_leaf_pins'
docstring) — these tests move those leaves from "parity unpinned" to pinned.  CPU only."""
import os

import pytest
import torch
import torch.nn as nn

from oracle import layers as OL
from oracle import vae_flux as OV
from tests.golden.seeded import seeded, synthetic_state_dict, vae_synthetic_state_dict


@pytest.fixture(scope="module")
def pins(golden_dir):
    return torch.load(os.path.join(golden_dir, "leaf_pins.pt"), weights_only=False)


def _close(a, b, tol=2e-6):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=tol, rtol=tol), float((a - b).abs().max())


def _load(mod, seed, keys, gen=synthetic_state_dict):
    sd = gen(mod, seed)
    assert sorted(sd.keys()) == keys, "state-dict keys differ from the reference class"
    mod.load_state_dict(sd, strict=True)
    return mod.eval()


def test_timestep_embedding_is_the_reference_op_sequence(pins):
    for c in pins["timestep_embedding"]:
        out = OL.get_timestep_embedding(c["t"], c["dim"], flip_sin_to_cos=c["flip"], downscale_freq_shift=c["shift"],
                                        scale=c["scale"])
        assert torch.equal(out, c["out"]), "same f32 operations in the same order: bit-identical on one machine"


def test_timestep_mlp_text_projection_feedforward_rmsnorm(pins):
    c = pins["TimestepEmbedding"]
    m = _load(OL.TimestepEmbedding(*c["dims"]), c["seed"], c["keys"])
    _close(m(seeded(c["x_shape"], c["x_seed"])), c["out"])
    c = pins["PixArtAlphaTextProjection"]
    m = _load(OL.PixArtAlphaTextProjection(c["dims"][0], c["dims"][1], act_fn="gelu_tanh"), c["seed"], c["keys"])
    _close(m(seeded(c["x_shape"], c["x_seed"])), c["out"])
    c = pins["FeedForward"]
    m = _load(OL.FeedForward(c["dims"][0], inner_dim=c["dims"][1]), c["seed"], c["keys"])
    _close(m(seeded(c["x_shape"], c["x_seed"])), c["out"])
    c = pins["RMSNorm"]
    m = _load(OL.RMSNorm(c["dim"], c["eps"]), c["seed"], c["keys"])
    _close(m(seeded(c["x_shape"], c["x_seed"]) * 3), c["out"])


def test_rotary_table_and_application(pins):
    for c in pins["get_1d_rotary_pos_embed"]:
        for dt, tol in ((torch.float32, 1e-6), (torch.float64, 2e-5)):    # Flux asks for f64 angles; same formula
            cos, sin = OL.get_1d_rotary_pos_embed(c["dim"], c["pos"], theta=10000.0, freqs_dtype=dt)
            _close(cos, c["out"][0], tol)
            _close(sin, c["out"][1], tol)
    c = pins["apply_rotary_emb"]
    rope = (c["cos"], c["sin"])
    _close(OL.apply_rotary_emb(seeded(c["x1_shape"], c["x1_seed"]), rope, sequence_dim=1), c["out1"])
    _close(OL.apply_rotary_emb(seeded(c["x2_shape"], c["x2_seed"]), rope, sequence_dim=2), c["out2"])


def test_adaln_family(pins):
    c = pins["AdaLayerNormZero"]
    m = _load(OL.AdaLayerNormZero(c["dim"]), c["seed"], c["keys"])
    x, emb = seeded((2, 11, c["dim"]), c["x_seed"]), seeded((2, c["dim"]), c["emb_seed"])
    with torch.no_grad():
        got = m(x, emb)                      # x, gate_msa, shift_mlp, scale_mlp, gate_mlp
    for a, b in zip(got, c["out"]):
        _close(a, b)
    # chunk orders of the Zero (6) and Single (3) forms on a pre-projected embedding (Chroma's pruned classes)
    c = pins["AdaLN_chunk_orders"]
    d = c["dim"]
    x = seeded((2, 11, d), c["x_seed"])

    class _Id(nn.Module):                    # `linear(silu(emb))` replaced by the given projection
        def __init__(self, proj):
            super().__init__()
            self.proj = proj

        def forward(self, _):
            return self.proj

    z = OL.AdaLayerNormZero(d)
    z.linear = _Id(seeded((2, 6, d), c["e6_seed"]).flatten(1, 2))
    for a, b in zip(z(x, torch.zeros(2, d)), c["zero"]):
        _close(a, b)
    s1 = OL.AdaLayerNormZeroSingle(d)
    s1.linear = _Id(seeded((2, 3, d), c["e3_seed"]).flatten(1, 2))
    for a, b in zip(s1(x, torch.zeros(2, d)), c["single"]):
        _close(a, b)


def test_adaln_continuous_is_scale_then_shift(pins):
    """The reference converts BFL checkpoints (final_layer.adaLN_modulation = [shift, scale], as the original Flux code
    chunks it) with `swap_scale_shift` for the diffusers class (converters/utils.py:82-85, used at
    transformer_converters.py:1574-1582): so the class reads [scale, shift].  The oracle class fed the converted weight
    must equal the original-order formula."""
    c = pins["swap_scale_shift"]
    w = seeded(c["w_shape"], c["w_seed"])
    assert torch.equal(c["out"], torch.cat([w[48:], w[:48]], dim=0))
    m = OL.AdaLayerNormContinuous(48, 32)
    with torch.no_grad():
        m.linear.weight.copy_(c["out"])
        m.linear.bias.zero_()
        x, cond = seeded((2, 7, 48), 1), seeded((2, 32), 2)
        e = torch.nn.functional.silu(cond) @ w.T                   # original order: shift rows first, then scale rows
        shift, scale = e[:, :48], e[:, 48:]
        ref = torch.nn.functional.layer_norm(x, (48,), eps=1e-6) * (1 + scale)[:, None] + shift[:, None]
        _close(m(x, cond), ref)


def test_vae_blocks(pins):
    for tag in ("same", "widen"):
        c = pins[f"ResnetBlock2D_{tag}"]
        m = _load(OV.ResnetBlock2D(c["cin"], c["cout"]), c["seed"], c["keys"], vae_synthetic_state_dict)
        with torch.no_grad():
            _close(m(seeded(c["x_shape"], c["x_seed"]), OL.FP32), c["out"], 1e-5)
    # LDM-style blocks on a one-frame clip: a 3x3x3 convolution with symmetric padding 1 over ONE frame is the 2-D
    # convolution with its middle temporal tap; 1x1x1 convolutions are the Linear layers of the diffusers Attention
    def ref_sd(keys_shapes_module, seed):
        return vae_synthetic_state_dict(keys_shapes_module, seed)

    class _Shape(nn.Module):                 # parameter container with the reference block's names / shapes
        def __init__(self, spec):
            super().__init__()
            for k, shp in spec.items():
                mod, leaf = k.rsplit(".", 1)
                if not hasattr(self, mod):
                    setattr(self, mod, nn.Module())
                getattr(self, mod).register_parameter(leaf, nn.Parameter(torch.zeros(shp)))

    c = pins["ldm_ResnetBlock"]
    spec = {"norm1.weight": (64,), "norm1.bias": (64,), "conv1.weight": (96, 64, 3, 3, 3), "conv1.bias": (96,),
            "norm2.weight": (96,), "norm2.bias": (96,), "conv2.weight": (96, 96, 3, 3, 3), "conv2.bias": (96,),
            "nin_shortcut.weight": (96, 64, 1, 1, 1), "nin_shortcut.bias": (96,)}
    sd = ref_sd(_Shape(spec), c["seed"])
    assert sorted(sd) == c["keys"]
    m = OV.ResnetBlock2D(64, 96)
    m.load_state_dict({k.replace("nin_shortcut", "conv_shortcut"): (v[:, :, 1] if v.dim() == 5 and v.shape[2] == 3 else
                                                                    v[:, :, 0] if v.dim() == 5 else v) for k, v in sd.items()})
    with torch.no_grad():
        _close(m.eval()(seeded(c["x_shape"], c["x_seed"])[:, :, 0], OL.FP32), c["out"][:, :, 0], 1e-5)

    c = pins["ldm_AttnBlock"]
    spec = {"norm.weight": (64,), "norm.bias": (64,)}
    for n in ("q", "k", "v", "proj_out"):
        spec[f"{n}.weight"], spec[f"{n}.bias"] = (64, 64, 1, 1, 1), (64,)
    sd = ref_sd(_Shape(spec), c["seed"])
    assert sorted(sd) == c["keys"]
    m = OV.AttnBlock(64)
    ren = {"norm": "group_norm", "q": "to_q", "k": "to_k", "v": "to_v", "proj_out": "to_out.0"}
    m.load_state_dict({ren[k.rsplit(".", 1)[0]] + "." + k.rsplit(".", 1)[1]: (v.reshape(64, 64) if v.dim() == 5 else v)
                       for k, v in sd.items()})
    with torch.no_grad():
        _close(m.eval()(seeded(c["x_shape"], c["x_seed"])[:, :, 0], OL.FP32), c["out"][:, :, 0], 1e-5)

    c = pins["ldm_Upsample"]
    sd = ref_sd(_Shape({"conv.weight": (64, 64, 3, 3, 3), "conv.bias": (64,)}), c["seed"])
    assert sorted(sd) == c["keys"]
    m = OV.Upsample2D(64)
    m.load_state_dict({"conv.weight": sd["conv.weight"][:, :, 1], "conv.bias": sd["conv.bias"]})
    with torch.no_grad():
        _close(m.eval()(seeded(c["x_shape"], c["x_seed"])[:, :, 0], OL.FP32), c["out"][:, :, 0], 1e-5)


def test_flowmatch_euler_matches_the_in_tree_scheduler(pins):
    """apex_studio_amd.schedulers.FlowMatchEulerDiscreteScheduler (static shift, explicit sigmas) against the reference's
    FlowMatchDiscreteScheduler: sd3 time shift, sigma -> timestep, x + (sigma_next - sigma) v in f32."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    c = pins["flow_euler"]
    n = c["steps"]
    s = FlowMatchEulerDiscreteScheduler(shift=c["shift"])
    ts = s.set_timesteps(sigmas=torch.linspace(1, 0, n + 1)[:-1].tolist())
    _close(ts, c["timesteps"], 1e-4)
    _close(s.sigmas, c["sigmas"], 1e-6)
    x = seeded(c["shape"], c["x_seed"])
    s.set_begin_index(0)
    for i, t in enumerate(ts):
        x = s.step(seeded(c["shape"], c["x_seed"] + 1 + i), t, x, return_dict=False)[0]
        _close(x, c["traj"][i], 1e-6)


# ---- round 3: the remaining cheap pins (tests/golden/leaf_pins2.pt, `gen_leaf_pins2`) ---------------------------------
@pytest.fixture(scope="module")
def pins2(golden_dir):
    return torch.load(os.path.join(golden_dir, "leaf_pins2.pt"), weights_only=False)


def test_dynamic_time_shift_is_the_in_tree_closed_form(pins2):
    """e^mu / (e^mu + (1/t - 1)^sigma): reference scheduler/rf.py:94-95 (= unipc.py:275-276) against the product
    scheduler's `_time_shift` and the oracle's numpy restatement."""
    import numpy as np
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from oracle.schedulers import flow_sigmas
    s = FlowMatchEulerDiscreteScheduler(use_dynamic_shifting=True)
    for c in pins2["time_shift"]:
        got = s._time_shift(c["mu"], c["sigma"], c["t"])
        assert got.dtype == torch.float64
        _close(got, c["out"], 1e-12)
        if c["sigma"] == 1.0:
            assert np.allclose(flow_sigmas(c["t"].numpy(), mu=c["mu"])[:-1], c["out"].float().numpy(), atol=1e-7, rtol=0)


def test_dynamic_shift_schedule_matches_the_in_tree_flowmatch_scheduler(pins2):
    """linspace(1, 1/N, N) -> exponential shift with mu = calculate_shift(sequence length) -> terminal stretch -> x1000:
    reference scheduler/flow_match_pair.py:41-60, :118-129 against FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=, mu=)
    as the Flux / QwenImage engines call it (engine_flux.py, engine_qwenimage.py)."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.engine_flux import calculate_shift
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    for c in pins2["flow_match_pair"]:
        mu = calculate_shift(c["seq_len"], 256, 8192, 0.5, 0.9)
        assert abs(mu - c["mu"]) < 1e-12
        s = FlowMatchEulerDiscreteScheduler(shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9,
                                            base_image_seq_len=256, max_image_seq_len=8192,
                                            shift_terminal=c["shift_terminal"])
        n = c["steps"]
        ts = s.set_timesteps(sigmas=torch.linspace(1.0, 1.0 / n, n).tolist(), mu=mu)
        _close(s.sigmas[:-1], c["sigmas"], 2e-7)
        _close(ts, c["timesteps"], 2e-4)
        assert float(s.sigmas[-1]) == 0.0


def test_flux_vae_decoder_topology_is_the_ldm_decoder(pins2):
    """oracle/vae_flux.py's Decoder (the diffusers `Decoder` layout the FLUX.1 VAE checkpoint and the HIP class use) against
    the reference's in-tree LDM `Decoder` (preprocess/diffusion_edge/taming/modules/diffusionmodules/model.py:462-580) run
    on the same weights through the published key correspondence: block order, `layers_per_block + 1` resnets per level,
    upsampler on every level but the last, channel schedule, norm_out -> SiLU -> conv_out.  The strict load of the renamed
    state dict into the reference class in `gen_leaf_pins2` already pins names, shapes and counts."""
    for c in pins2["ldm_decoder"]:
        orc = OV.AutoencoderKLDecoder(**c["cfg"]).eval()
        sd = vae_synthetic_state_dict(orc, c["seed"])
        orc.load_state_dict(sd, strict=True)
        assert len(sd) == len(c["ldm_keys"])
        out = orc.decode(seeded(c["z_shape"], c["z_seed"]))
        assert out.shape == c["out"].shape
        rel = float((out - c["out"]).norm() / c["out"].norm())
        assert rel < 1e-5, (c["tag"], rel)


@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fp32_layernorm_is_torch_layer_norm_in_f32(affine, dtype):
    """`FP32LayerNorm` (diffusers.models.normalization, un-vendored; used by the reference's Wan blocks, transformer/wan/base
    /model.py norm1 / norm2 / norm3 / norm_out, and restated for MLX at mlx/modules/layers.py:7-17) is by definition
    `F.layer_norm(x.float(), shape, weight.float(), bias.float(), eps).to(x.dtype)`: pinned against torch's own layer_norm in
    f32 and against the written-out definition in float64 — the statistics are NOT taken in the storage dtype."""
    import torch.nn.functional as F
    dim, eps = 192, 1e-6
    ln = OL.FP32LayerNorm(dim, eps, elementwise_affine=affine)
    if affine:
        ln.weight.data = (1 + 0.1 * seeded((dim,), 3)).to(dtype)
        ln.bias.data = (0.05 * seeded((dim,), 4)).to(dtype)
    x = (seeded((2, 37, dim), 5) * 3 + 0.7).to(dtype)
    got = ln(x)
    w, b = (ln.weight.float(), ln.bias.float()) if affine else (None, None)
    assert got.dtype == dtype and torch.equal(got, F.layer_norm(x.float(), (dim,), w, b, eps).to(dtype))
    xd = x.double()
    ref = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + eps)
    if affine:
        ref = ref * w.double() + b.double()
    _close(got.float(), ref.to(dtype).float(), 2e-6 if dtype == torch.float32 else 8e-3)
    if dtype == torch.bfloat16:        # statistics in the storage dtype would differ: this is what "FP32" buys
        naive = F.layer_norm(x, (dim,), ln.weight if affine else None, ln.bias if affine else None, eps)
        assert (got.float() - ref.float()).abs().max() <= (naive.float() - ref.float()).abs().max() + 1e-6
