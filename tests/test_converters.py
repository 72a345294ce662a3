"""apex_studio_amd.converters / lora.convert_lora_state_dict against the reference's OWN converters run in the build container
(tests/golden/make_golden.py `gen_convert` -> tests/golden/convert_keys.pt: for every case the converted key set and a
digest of every tensor).  Original-format Wan / BFL-Flux checkpoints, fp8-scaled Kijai keys, lightx2v-keyed and BFL-keyed
LoRAs, Kohya single-file LoRAs, shards holding half of a fused pair, already-converted files.  CPU only.

Every case runs twice: on real tensors, and on `Src` placeholders (no tensor read until the plan is executed) — the form
`weights.load_checkpoint_into(converter=...)` streams weight files with."""
import os

import pytest
import torch

import apex_studio_amd  # noqa: F401
from apex_studio_amd import converters as CV
from apex_studio_amd import lora as LR
from tests.golden.seeded import spec_tensors, tensor_digest


@pytest.fixture(scope="module")
def gold(golden_dir):
    return torch.load(os.path.join(golden_dir, "convert_keys.pt"), weights_only=False)


def _inputs(c):
    sd = {c["prefix"] + k: v for k, v in spec_tensors(c["spec"], c["seed0"]).items()}
    if c["extra"]:
        sd.update(c["extra"])
    assert list(sd) == c["inp_keys"]
    return sd


CHECKPOINT_CASES = ["wan_original", "wan_original_no_model_keys", "wan_fp8_wrapped", "wan_fp8_kijai", "wan_already_converted",
                    "flux_bfl", "flux_bfl_no_model_keys", "flux_partial_shard", "flux_already_converted",
                    "hy15_original", "hy15_original_no_model_keys", "hy15_wrapped", "hy15_already_converted"]
LORA_CASES = ["wan_lightx2v_lora", "flux_bfl_lora_peft_keys", "flux_bfl_lora_base_keys", "flux_kohya_lora", "wan_kohya_lora",
              "hy15_original_key_lora"]


def _conv(name):
    return CV.get_transformer_converter("wan.base" if name.startswith("wan") else "hunyuanvideo15.base" if name.startswith("hy15")
                                        else "flux.base")


def _same(out, want, name):
    assert sorted(out) == sorted(want), (name, sorted(set(out) ^ set(want))[:8])
    for k, v in out.items():
        assert tensor_digest(v) == want[k], (name, k)


@pytest.mark.parametrize("name", CHECKPOINT_CASES)
def test_checkpoint_keys_and_tensors_match_the_reference_converter(gold, name):
    c = gold["cases"][name]
    sd = _inputs(c)
    out = _conv(name).convert(dict(sd), c["model_keys"])
    _same(out, c["out"], name)


@pytest.mark.parametrize("name", CHECKPOINT_CASES)
def test_placeholder_plan_reads_the_same_tensors(gold, name):
    """The conversion run on placeholders, then every target materialised from the 'file': same keys, same bytes — and a fused
    tensor is only ever touched through row ranges."""
    c = gold["cases"][name]
    sd = _inputs(c)
    plan = _conv(name).convert({k: CV.Src(k, tuple(v.shape), v.dtype) for k, v in sd.items()}, c["model_keys"])
    ranged = []

    def rows(key, a, b):
        ranged.append(key)
        return sd[key][a:b]
    out = {k: p.read(sd.__getitem__, rows) for k, p in plan.items()}
    _same(out, c["out"], name)
    if name == "flux_bfl":
        assert sum(k.endswith("qkv.weight") for k in set(ranged)) == 4 and all(tuple(p.shape) == tuple(out[k].shape) for k, p in plan.items())


@pytest.mark.parametrize("name", LORA_CASES)
def test_lora_pipeline_matches_the_reference(gold, name):
    """LoraManager.maybe_convert_state_dict: LoraConverter (PEFT / base / Kohya, alpha folding) -> the model's converter ->
    prefix strip."""
    c = gold["cases"][name]
    sd = _inputs(c)
    base = "wan.base" if name.startswith("wan") else "hunyuanvideo15.base" if name.startswith("hy15") else "flux.base"
    out = LR.convert_lora_state_dict(sd, base, c["model_keys"])
    _same(out, c["out"], name)
    assert all(torch.equal(sd[k], _inputs(c)[k]) for k in sd), "the caller's state dict must not be modified"


def test_converted_lightx2v_lora_splits_into_model_modules(gold):
    c = gold["cases"]["wan_lightx2v_lora"]
    out = LR.convert_lora_state_dict(_inputs(c), "wan.base", c["model_keys"])
    mods = LR.split_modules(out)
    assert sorted(mods) == sorted({k.rsplit(".lora_", 1)[0] for k in out if ".lora_" in k})
    assert "blocks.0.attn1.to_q" in mods and mods["blocks.1.ffn.net.2"]["A"].shape == (4, 128)
    for m in mods:
        assert m + ".weight" in c["model_keys"]


def test_registry_keys_pick_the_tables():
    assert isinstance(CV.get_transformer_converter("wan.mi355"), CV.WanKeyConverter)
    assert isinstance(CV.get_transformer_converter("flux.base"), CV.FluxKeyConverter)
    assert isinstance(CV.get_transformer_converter("qwenimage.base"), CV.NoOpKeyConverter)
    assert isinstance(CV.get_transformer_converter("hunyuanvideo15.base"), CV.Hunyuan15KeyConverter)
    assert isinstance(CV.get_transformer_converter(""), CV.NoOpKeyConverter)
    for other in ("wan.vace", "wan.s2v", "wan.animate", "flux.kontext"):      # the reference has other tables for these: not ported
        with pytest.raises(NotImplementedError):
            CV.get_transformer_converter(other)
    assert CV.kohya_unflatten("lora_unet_double_blocks_0_img_attn_qkv".replace("lora_unet", "unet")) == "unet.double_blocks.0.img_attn.qkv"
    assert CV.kohya_unflatten("unet_time_in_linear_1") == "unet.time.in.linear_1"


def test_placeholder_algebra():
    s = CV.Src("w", (12, 4))
    a, b, c = CV.rows3(s)
    assert (a.rows, b.rows, c.rows) == ((0, 4), (4, 8), (8, 12)) and a.shape == (4, 4)
    assert [p.rows for p in CV.rows_split(b, (1, 3))] == [(4, 5), (5, 8)]
    w = torch.arange(48.0).view(12, 4)
    assert torch.equal(CV.swap_halves(s).read({"w": w}.__getitem__), torch.cat([w[6:], w[:6]]))
    with pytest.raises(ValueError):
        CV.rows3(CV.swap_halves(s))
    with pytest.raises(ValueError):
        CV.rows_split(s, (5, 5))
