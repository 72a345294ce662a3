"""ORACLE — test infrastructure, not product code.

CPU restatement of the frame post-processing BaseEngine._tensor_to_frames / _tensor_to_frame perform
(apps/api/src/engine/base_engine.py:2945-2969) through the third-party diffusers `VideoProcessor.postprocess_video` /
`VaeImageProcessor.postprocess` (diffusers is un-pinned and absent from the container, so this leaf is "parity unpinned"
against diffusers; restated from its published code): denormalize `(x * 0.5 + 0.5).clamp(0, 1)` IN THE TENSOR'S DTYPE,
`.cpu().permute(...).float().numpy()`, then numpy_to_pil's `(images * 255).round().astype("uint8")`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np
import torch


def video_to_uint8_frames(video: torch.Tensor) -> np.ndarray:
    """video [B, C, T, H, W] in [-1, 1] (any float dtype) -> uint8 [B, T, H, W, C]."""
    out = []
    for b in range(video.shape[0]):
        frames = video[b].permute(1, 0, 2, 3)                          # [T, C, H, W]
        frames = (frames * 0.5 + 0.5).clamp(0, 1)                      # denormalize, in the decode dtype
        arr = frames.cpu().permute(0, 2, 3, 1).float().numpy()         # pt_to_numpy
        out.append((arr * 255).round().astype("uint8"))               # numpy_to_pil
    return np.stack(out)
