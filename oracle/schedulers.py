"""ORACLE — test infrastructure, not product code.

Closed-form numpy restatement of diffusers FlowMatchEulerDiscreteScheduler as the Flux / Qwen manifests
configure it: sigma' = e^mu / (e^mu + (1/sigma - 1)), timesteps = 1000 sigma',
x_{i+1} = x_i + (sigma'_{i+1} - sigma'_i) v_i with sigma'_N = 0.  "parity unpinned" against diffusers
itself (absent here); the in-tree sibling reference scheduler/flow.py:293-355 has the same update.
"""
import numpy as np


def flow_sigmas(sigmas, mu=None, shift=1.0):
    s = np.asarray(sigmas, dtype=np.float64)
    if mu is not None:
        s = np.exp(mu) / (np.exp(mu) + (1.0 / s - 1.0))
    else:
        s = shift * s / (1 + (shift - 1) * s)
    return np.concatenate([s, [0.0]]).astype(np.float32)


def euler_trajectory(x0, velocities, sigmas_shifted):
    x = np.asarray(x0, dtype=np.float32)
    out = []
    for i, v in enumerate(velocities):
        dt = np.float32(sigmas_shifted[i + 1]) - np.float32(sigmas_shifted[i])
        x = (x.astype(np.float32) + dt * np.asarray(v, dtype=np.float32)).astype(np.float32)
        out.append(x.copy())
    return out
