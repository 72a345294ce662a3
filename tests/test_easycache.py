"""EasyCache step skipping (`wan.mi355.enable_easy_cache`, apex-studio_amd/easycache.py): the host-side rule against the oracle's
restatement of the reference function (oracle/easycache.py, pinned to the reference by tests/golden/wan_easycache.pt in
tests/test_oracle_golden.py), the engine's enable / reset / disable choreography on CPU fakes, and — GPU — the HIP Wan model
driven through the fixture's own sampler sequence."""
import os
from types import SimpleNamespace

import pytest
import torch

import apex_studio_amd  # noqa: F401
from apex_studio_amd.easycache import EasyCache
from oracle.easycache import EasyCacheState, easycache_forward
from tests.golden.seeded import seeded, synthetic_state_dict


def _walk(n, thresh, ret_steps, step_scale, use_product, dtype=torch.float32):
    """A toy 'transformer' (a fixed nonlinear map of input, time and condition) under both implementations."""
    w = (seeded((16, 16), 5) * 0.3).to(dtype)
    conds = [seeded((1, 16, 1, 1, 1), 6).to(dtype), seeded((1, 16, 1, 1, 1), 7).to(dtype)]
    x = seeded((1, 20, 2, 4, 4), 8).to(dtype)             # 16 latent + 4 condition channels: only the first 16 are "raw input"
    st = EasyCacheState(n, thresh, ret_steps)
    ec = EasyCache(n, thresh, ret_steps)
    outs, flags = [], []
    for i in range(n):
        pair = []
        for c in conds:
            def fwd(x=x, c=c, i=i):
                y = torch.einsum("oc,bcfhw->bofhw", w, torch.tanh(x[:, :16] + c)) * (1.0 + 0.05 * i)
                return y
            if use_product:
                out = ec(x, 16, fwd)
                did = ec.computed[-1]
            else:
                out, did = easycache_forward(st, fwd, x, 16)
            assert out.dtype == torch.float32
            outs.append(out)
            flags.append(did)
            pair.append(out)
        x = torch.cat([(x[:, :16].float() - step_scale * (pair[1] + 2.0 * (pair[0] - pair[1]))).to(dtype), x[:, 16:]], dim=1)
    return outs, flags


@pytest.mark.parametrize("thresh,scale", [(0.4, 0.02), (0.05, 0.05), (3.0, 0.01)])
def test_product_rule_equals_the_oracle_restatement(thresh, scale):
    a_out, a_flags = _walk(12, thresh, 2, scale, use_product=True)
    b_out, b_flags = _walk(12, thresh, 2, scale, use_product=False)
    assert a_flags == b_flags, ("".join("C" if f else "-" for f in a_flags), "".join("C" if f else "-" for f in b_flags))
    assert a_flags[:4] == [True] * 4 and a_flags[-2:] == [True, True], "the first ret_steps pairs and the last pair always run"
    for u, v in zip(a_out, b_out):
        assert torch.allclose(u, v, rtol=1e-6, atol=1e-6)
    if thresh >= 0.4:
        assert not all(a_flags), "this threshold must skip something for the test to mean anything"


@pytest.mark.parametrize("thresh,scale", [(0.4, 0.02), (0.05, 0.05), (0.2, 0.03)])
def test_product_rule_keeps_the_reference_dtypes(thresh, scale):
    """bf16 latents (production): the reference keeps `raw_input`, the caches and the change statistics in the tensors' own dtype
    and casts only what it returns (R/src/transformer/wan/base/model.py:246, :265-283, :483-505); the oracle restates that natively.
    The product must take the same skip decisions on the same ROUNDED statistics and return the same bits."""
    a_out, a_flags = _walk(14, thresh, 2, scale, use_product=True, dtype=torch.bfloat16)
    b_out, b_flags = _walk(14, thresh, 2, scale, use_product=False, dtype=torch.bfloat16)
    assert a_flags == b_flags
    for u, v in zip(a_out, b_out):
        assert u.dtype == torch.float32 and torch.equal(u, v)
    f_out, f_flags = _walk(14, thresh, 2, scale, use_product=True, dtype=torch.float32)
    assert any(not torch.equal(u, v) for u, v in zip(a_out, f_out)), "bf16 statistics are not the f32 ones"


class _FakeExpert:
    """A CPU stand-in with the model's EasyCache surface and the REAL state object (apex-studio_amd/easycache.py)."""

    def __init__(self, name, log, gain):
        self.name, self.log, self.gain = name, log, gain
        self.config = SimpleNamespace(in_channels=16, out_channels=16)
        self.device, self.dtype = torch.device("cpu"), torch.float32
        self._easy_cache = None

    def enable_easy_cache(self, n, thresh, ret, should_reset_global_cache=True):
        self.log.append((self.name, "on", n, thresh, ret, should_reset_global_cache))
        if should_reset_global_cache or self._easy_cache is None:
            self._easy_cache = EasyCache(n, thresh, ret)
        else:
            self._easy_cache.num_steps, self._easy_cache.thresh, self._easy_cache.ret_steps = 2 * n, thresh, 2 * ret

    def share_easy_cache_state(self, other):
        self.log.append((self.name, "share", other.name))
        self._easy_cache = other._easy_cache

    def disable_easy_cache(self):
        self.log.append((self.name, "off"))
        self._easy_cache = None

    def __call__(self, hidden_states, timestep, encoder_hidden_states, return_dict=False):
        def fwd():
            self.log.append((self.name, "fwd", float(timestep[0])))
            return torch.tanh(hidden_states.float() * self.gain + encoder_hidden_states.mean())
        ec = self._easy_cache
        return ((ec(hidden_states, 16, fwd) if ec is not None else fwd()),)


def test_engine_shares_one_state_across_the_expert_switch_and_switches_off():
    """`WanT2VEngine.moe_denoise(easy_cache_thresh=…)` as the reference: the high-noise expert is enabled WITH a reset, the
    low-noise expert WITHOUT one (R/src/engine/wan/shared/__init__.py:372-381 vs :435-444) — the call count, K, the accumulated
    error and the caches run on across the switch: no second warm-up, the last pair always computed; off after the loop."""
    from apex_studio_amd.engine_wan import WanT2VEngine
    log = []
    hi, lo = _FakeExpert("hi", log, 0.9), _FakeExpert("lo", log, 1.1)
    eng = WanT2VEngine(hi, lo)
    n = 12
    ts = eng.scheduler.set_timesteps(n, device="cpu")
    n_hi = int((ts >= 875.0).sum())
    assert 0 < n_hi < n - 3
    lat = seeded((1, 16, 1, 4, 4), 1)
    states = []
    orig = lo.enable_easy_cache

    def spy(*a, **k):
        orig(*a, **k)
        states.append((lo._easy_cache, lo._easy_cache.cnt))
    lo.enable_easy_cache = spy
    # a huge threshold: after the warm-up every pair that MAY be skipped is skipped
    eng.moe_denoise(latents=lat, timesteps=ts, prompt_embeds=seeded((1, 4, 8), 2), negative_prompt_embeds=seeded((1, 4, 8), 3),
                    guidance_scale=[4.0, 3.0], boundary_timestep=875.0, easy_cache_thresh=1e9, easy_cache_ret_steps=3)
    ons = [e for e in log if e[1] in ("on", "share")]
    assert ons == [("hi", "on", n, 1e9, 3, True), ("lo", "share", "hi"), ("lo", "on", n, 1e9, 3, False)]
    (state, cnt_at_switch), = states
    assert cnt_at_switch == 2 * n_hi, "the low-noise expert continues the high-noise expert's call count"
    assert state.cnt == 2 * n and len(state.computed) == 2 * n
    # warm-up pairs (3) computed once, NOT again after the switch; K needs one more computed pair; the last pair is always computed
    assert state.computed[:6] == [True] * 6 and state.computed[-2:] == [True, True]
    assert not any(state.computed[2 * max(n_hi, 4):-2]), "".join("C" if c else "-" for c in state.computed)
    fwd = [e for e in log if e[1] == "fwd"]
    assert [e[0] for e in fwd[-2:]] == ["lo", "lo"] and fwd[-1][2] == float(ts[-1]), "the final denoise pair runs on the low-noise expert"
    assert sorted(log[-2:]) == [("hi", "off"), ("lo", "off")] and hi._easy_cache is None and lo._easy_cache is None
    log.clear()
    ts = eng.scheduler.set_timesteps(4, device="cpu")
    eng.moe_denoise(latents=lat, timesteps=ts, prompt_embeds=seeded((1, 4, 8), 2), boundary_timestep=875.0)
    assert not [e for e in log if e[1] in ("on", "off", "share")], "off by default (every BASELINE config)"


@pytest.mark.gpu
def test_hip_wan_easycache_follows_the_reference_run(golden_dir):
    """The HIP Wan model with `enable_easy_cache` driven through the fixture's sampler sequence (the reference's
    `easycache_forward_` on the reference model, float64): the same calls run / are served from the cache, outputs within the
    production bf16 bar of the reference's, the latent after ten steps within it too; `disable_easy_cache` restores the plain forward."""
    from apex_studio_amd.wan import WanTransformer3DModel
    from oracle import wan as OW
    g = torch.load(os.path.join(golden_dir, "wan_easycache.pt"), weights_only=False)
    dev = "cuda"
    m = WanTransformer3DModel(**g["config"], device=dev, dtype=torch.bfloat16)
    sd = synthetic_state_dict(OW.WanTransformer3DModel(**g["config"]), g["seed"])
    m.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    x = seeded((1, 16, 3, 8, 12), g["x_seed"]).to(dev)
    txts = [seeded((1, 20, 64), s).to(dev).to(torch.bfloat16) for s in g["txt_seeds"]]
    plain = m(hidden_states=x.to(torch.bfloat16), timestep=torch.tensor([g["timesteps"][0]], device=dev), encoder_hidden_states=txts[0],
              return_dict=False)[0].float()
    m.enable_easy_cache(g["n"], g["thresh"], g["ret_steps"])
    k, worst = 0, 0.0
    for i in range(g["n"]):
        pair = []
        for txt in txts:
            out = m(hidden_states=x.to(torch.bfloat16), timestep=torch.tensor([g["timesteps"][i]], device=dev), encoder_hidden_states=txt,
                    return_dict=False)[0]
            assert out.dtype == torch.float32
            ref = g["outs"][k].to(dev)
            worst = max(worst, float((out - ref).norm() / ref.norm()))
            pair.append(out)
            k += 1
        x = x - g["dt"] * (pair[1] + g["guidance"] * (pair[0] - pair[1]))
    flags = m._easy_cache.computed
    print(f"[easycache] computed {''.join('C' if c else '-' for c in flags)} (reference: {''.join('C' if c else '-' for c in g['computed'])}); "
          f"worst output rel L2 vs the reference run {worst:.2e}; final latent {float((x.cpu() - g['x_final']).norm() / g['x_final'].norm()):.2e}")
    assert flags == g["computed"]
    assert worst < 1e-2 and float((x.cpu() - g["x_final"]).norm() / g["x_final"].norm()) < 6e-3
    m.disable_easy_cache()
    again = m(hidden_states=seeded((1, 16, 3, 8, 12), g["x_seed"]).to(dev).to(torch.bfloat16), timestep=torch.tensor([g["timesteps"][0]], device=dev),
              encoder_hidden_states=txts[0], return_dict=False)[0].float()
    assert torch.equal(again, plain)
