"""FlowMatch-Euler sampler step (stays in Python) against the numpy closed form in oracle/."""
import numpy as np
import torch

from oracle import schedulers as OS
from oracle.flux import calculate_shift


def test_flux_sigmas_and_trajectory():
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    n = 28
    sig = np.linspace(1.0, 1.0 / n, n)
    mu = calculate_shift(4096)
    sch = FlowMatchEulerDiscreteScheduler.flux_dev()
    ts = sch.set_timesteps(sigmas=sig.tolist(), mu=mu)
    ref = OS.flow_sigmas(sig, mu=mu)
    assert np.allclose(sch.sigmas.numpy(), ref, atol=1e-7)
    assert np.allclose(ts.numpy(), ref[:-1] * 1000, atol=1e-4)
    assert abs(float(ts[0]) - 1000.0) < 1e-3 and float(sch.sigmas[-1]) == 0.0
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 16, 64, generator=g)
    vs = [torch.randn(1, 16, 64, generator=g) for _ in range(n)]
    traj = OS.euler_trajectory(x.numpy(), [v.numpy() for v in vs], ref)
    sch.set_begin_index(0)
    for i, t in enumerate(ts):
        x = sch.step(vs[i], t, x, return_dict=False)[0]
        assert np.allclose(x.numpy(), traj[i], atol=1e-6)


def test_static_shift_and_bf16_cast():
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    sch = FlowMatchEulerDiscreteScheduler(shift=3.0)
    sch.set_timesteps(4)
    # diffusers: without explicit sigmas the grid runs between the ends of the (already shifted) training schedule,
    # sigma_max = shift*1/(1+(shift-1)*1) = 1 and sigma_min = shift*1e-3/(1+(shift-1)*1e-3), and is shifted again
    ref = OS.flow_sigmas(np.linspace(1.0, 3.0 * 1e-3 / (1 + 2.0 * 1e-3), 4), shift=3.0)
    assert np.allclose(sch.sigmas.numpy(), ref, atol=1e-6)
    x = torch.ones(4, dtype=torch.bfloat16)
    out = sch.step(torch.ones(4, dtype=torch.bfloat16), sch.timesteps[0], x, return_dict=False)[0]
    assert out.dtype == torch.bfloat16


def test_unipc_matches_reference_trajectory(golden_dir):
    """UniPC restatement vs the reference's in-tree scheduler (scheduler/unipc.py) run here."""
    import os
    from apex_studio_amd.schedulers import UniPCMultistepScheduler
    from tests.golden.seeded import seeded
    g = torch.load(os.path.join(golden_dir, "unipc.pt"), weights_only=False)
    s = UniPCMultistepScheduler(shift=g["shift"])
    ts = s.set_timesteps(g["steps"])
    assert torch.equal(ts, g["timesteps"])
    assert torch.allclose(s.sigmas, g["sigmas"], atol=1e-7)
    x = seeded(g["shape"], g["seed"])
    for i, t in enumerate(ts):
        x = s.step(seeded(g["shape"], g["seed"] + 1 + i), t, x, return_dict=False)[0]
        assert torch.allclose(x, g["traj"][i], atol=2e-5, rtol=2e-5), (i, float((x - g["traj"][i]).abs().max()))
    assert x.dtype == torch.float32
