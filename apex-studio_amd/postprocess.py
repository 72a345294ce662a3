"""Frame post-processing on the GPU: what BaseEngine._tensor_to_frames / _tensor_to_frame do through
diffusers' VideoProcessor (engine/base_engine.py:2945-2969) — denormalise to [0, 1], channels-last, x255, round, uint8 —
as ONE HIP kernel over the decoded bf16 video, so bytes (1 B per sample instead of 2 B) are what crosses PCIe."""
from __future__ import annotations

from typing import List, Union

import torch

from . import lib as _l
from . import ops


def video_to_uint8(video: torch.Tensor) -> torch.Tensor:
    """video [B, C, T, H, W] bf16 on the GPU (float32 in the f32-storage verification mode), values in [-1, 1] -> uint8
    [B, T, H, W, C] on the GPU."""
    if video.device.type != "cuda" or video.dtype not in (torch.bfloat16, torch.float32):
        raise _l.ApexMIError("video_to_uint8 needs the bf16 decode output on a ROCm device (no CPU fallback)")
    if video.dim() != 5:
        raise ValueError(f"expected [B, C, T, H, W], got {tuple(video.shape)}")
    return torch.stack([ops.frames_to_u8(video[b]) for b in range(video.shape[0])], dim=0)


def tensor_to_frames(video: torch.Tensor, output_type: str = "pil") -> Union[torch.Tensor, List]:
    """`_tensor_to_frames(video, output_type)`: "pil" -> list (batch) of lists of PIL images, as
    VideoProcessor.postprocess_video returns; "np" -> uint8 numpy [B, T, H, W, C]; "uint8" -> the same on the device.
    (The float "pt" / "np" forms of diffusers are not offered: the kernel's product is bytes.)"""
    frames = video_to_uint8(video)
    if output_type == "uint8":
        return frames
    arr = frames.cpu().numpy()
    if output_type == "np":
        return arr
    if output_type == "pil":
        from PIL import Image
        return [[Image.fromarray(f) for f in clip] for clip in arr]
    raise ValueError(f"output_type {output_type!r} not in ('pil', 'np', 'uint8')")


def tensor_to_frame(image: torch.Tensor, output_type: str = "pil"):
    """`_tensor_to_frame`: [B, C, H, W] (or [B, C, 1, H, W]) -> one image per batch element."""
    if image.dim() == 5:
        if image.shape[2] != 1:
            raise ValueError(f"Expected 1 frame, got {image.shape[2]} frames with shape {tuple(image.shape)}")
        image = image[:, :, 0]
    out = tensor_to_frames(image.unsqueeze(2), output_type)
    if output_type == "pil":
        return [clip[0] for clip in out]
    return out[:, 0]
