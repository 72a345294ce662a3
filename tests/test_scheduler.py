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
    ref = OS.flow_sigmas(np.linspace(1.0, 1.0 / 1000, 4), shift=3.0)
    assert np.allclose(sch.sigmas.numpy(), ref, atol=1e-6)
    x = torch.ones(4, dtype=torch.bfloat16)
    out = sch.step(torch.ones(4, dtype=torch.bfloat16), sch.timesteps[0], x, return_dict=False)[0]
    assert out.dtype == torch.bfloat16
