"""Flux text-to-image engine surface on the HIP transformer — the part of B-engine
the denoise path touches.  Mirrors reference engine/flux/shared.py (pack/unpack :29-55,
calculate_shift :57-68, latent ids :197-215, base_denoise :504-619) and engine/flux/t2i.py:20-256
(`run`): same argument names, same progress-callback protocol `(progress: float, message: str)`,
same `render_on_step_callback(frame)` preview hook, `return_latents=True` to stop before decode.

Text encoders are outside this backend (prompt embeddings are inputs).  The 2-D VAE decode is
`vae_flux.AutoencoderKL` on the same HIP ops; the engine takes it as `decode_fn` (latents -> image) so the preview
hook and the final decode share one callable.  The sampler loop stays in Python by design.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .lora import EngineLoraMixin

from .schedulers import FlowMatchEulerDiscreteScheduler


def pack_latents(latents: torch.Tensor) -> torch.Tensor:
    b, c, h, w = latents.shape
    x = latents.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, (h // 2) * (w // 2), c * 4)


def unpack_latents(latents: torch.Tensor, height: int, width: int, vae_scale_factor: int = 8) -> torch.Tensor:
    b, n, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.view(b, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(b, ch // 4, h, w)


def latent_image_ids(h2: int, w2: int, device=None, dtype=torch.float32) -> torch.Tensor:
    ids = torch.zeros(h2, w2, 3, device=device, dtype=dtype)
    ids[..., 1] += torch.arange(h2, device=device, dtype=dtype)[:, None]
    ids[..., 2] += torch.arange(w2, device=device, dtype=dtype)[None, :]
    return ids.reshape(h2 * w2, 3)


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.15) -> float:
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


def compute_dtype(module) -> torch.dtype:
    """The dtype activations / latents travel in for `module`: its activation storage type — bf16 in production, float32
    when the module was put into the f32-storage verification mode (`set_storage_dtype`), where the engines keep the
    sampler state in float32 like the reference's CPU fp32 path."""
    return getattr(module, "storage_dtype", None) or module.dtype


def _emit(cb, p, msg):
    if cb is not None:
        try:
            cb(p, msg)
        except Exception:
            pass


class FluxT2IEngine(EngineLoraMixin):
    """engine.run(...) for Flux text-to-image with pre-computed prompt embeddings."""

    def __init__(self, transformer, scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None,
                 decode_fn: Optional[Callable[[torch.Tensor], object]] = None, vae_scale_factor: int = 8):
        self.transformer = transformer
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler.flux_dev()
        self.decode_fn = decode_fn
        self.vae_scale_factor = vae_scale_factor
        self.num_channels_latents = transformer.config.in_channels // 4

    @property
    def device(self):
        return self.transformer.device

    def base_denoise(self, latents, timesteps, guidance, prompt_embeds, pooled_prompt_embeds, text_ids,
                     latent_ids, negative_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                     negative_text_ids=None, true_cfg_scale: float = 1.0, use_cfg_guidance: bool = False,
                     render_on_step: bool = False, render_on_step_callback=None,
                     render_on_step_interval: int = 3, denoise_progress_callback=None,
                     preview_hw=None):
        _emit(denoise_progress_callback, 0.0, "Starting denoise")
        n = len(timesteps)
        for i, t in enumerate(timesteps):
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            with self.transformer.cache_context("cond"):
                noise_pred = self.transformer(
                    hidden_states=latents, timestep=timestep / 1000, guidance=guidance,
                    pooled_projections=pooled_prompt_embeds, encoder_hidden_states=prompt_embeds,
                    txt_ids=text_ids, img_ids=latent_ids, return_dict=False)[0]
            if use_cfg_guidance:
                with self.transformer.cache_context("uncond"):
                    neg = self.transformer(
                        hidden_states=latents, timestep=timestep / 1000, guidance=guidance,
                        pooled_projections=negative_pooled_prompt_embeds,
                        encoder_hidden_states=negative_prompt_embeds, txt_ids=negative_text_ids,
                        img_ids=latent_ids, return_dict=False)[0]
                noise_pred = neg + true_cfg_scale * (noise_pred - neg)
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            if (render_on_step and render_on_step_callback and self.decode_fn is not None
                    and ((i + 1) % render_on_step_interval == 0 or i == 0) and i != n - 1
                    and preview_hw is not None):
                try:
                    render_on_step_callback(self.decode_fn(
                        unpack_latents(latents, preview_hw[0], preview_hw[1], self.vae_scale_factor)))
                except Exception:
                    pass
            _emit(denoise_progress_callback, float(i + 1) / n, f"Denoise {i + 1}/{n}")
        _emit(denoise_progress_callback, 1.0, "Denoise finished")
        return latents

    @torch.no_grad()
    def run(self, prompt_embeds: torch.Tensor, pooled_prompt_embeds: torch.Tensor, height: int = 1024,
            width: int = 1024, num_inference_steps: int = 28, guidance_scale: float = 3.5,
            latents: Optional[torch.Tensor] = None, seed: Optional[int] = None,
            generator: Optional[torch.Generator] = None, return_latents: bool = False,
            progress_callback=None, render_on_step: bool = False, render_on_step_callback=None,
            render_on_step_interval: int = 3, sigmas=None, output_type: Optional[str] = None, **_ignored):
        dev, dt = self.device, compute_dtype(self.transformer)
        B = prompt_embeds.shape[0]
        h = 2 * (int(height) // (self.vae_scale_factor * 2))
        w = 2 * (int(width) // (self.vae_scale_factor * 2))
        if latents is None:
            if generator is None:
                generator = torch.Generator(device=dev)
                if seed is not None:
                    generator.manual_seed(seed)
            raw = torch.randn((B, self.num_channels_latents, h, w), generator=generator,
                              device=generator.device, dtype=torch.float32).to(device=dev, dtype=dt)
            latents = pack_latents(raw)
        else:
            latents = latents.to(device=dev, dtype=dt)
        latent_ids = latent_image_ids(h // 2, w // 2, device=dev, dtype=dt)
        text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=dev, dtype=dt)
        _emit(progress_callback, 0.2, "Prepared latents")
        if sigmas is None:
            sigmas = torch.linspace(1.0, 1.0 / num_inference_steps, num_inference_steps).tolist()
        mu = calculate_shift(latents.shape[1])
        timesteps = self.scheduler.set_timesteps(sigmas=sigmas, mu=mu, device=dev)
        self.scheduler.set_begin_index(0)
        guidance = None
        if self.transformer.config.guidance_embeds:
            guidance = torch.full([1], guidance_scale, device=dev, dtype=torch.float32).expand(B)

        def mapped(p, msg):
            _emit(progress_callback, 0.5 + 0.4 * p, msg)

        latents = self.base_denoise(
            latents=latents, timesteps=timesteps, guidance=guidance, prompt_embeds=prompt_embeds.to(dev, dt),
            pooled_prompt_embeds=pooled_prompt_embeds.to(dev, dt), text_ids=text_ids, latent_ids=latent_ids,
            render_on_step=render_on_step, render_on_step_callback=render_on_step_callback,
            render_on_step_interval=render_on_step_interval, denoise_progress_callback=mapped,
            preview_hw=(height, width))
        if return_latents or self.decode_fn is None:
            _emit(progress_callback, 1.0, "Returning latents")
            return latents
        _emit(progress_callback, 0.92, "Decoding")
        out = self.decode_fn(unpack_latents(latents, height, width, self.vae_scale_factor))
        if output_type is not None and torch.is_tensor(out):   # t2i.py:254 `self._tensor_to_frame(image)`: uint8 frames on the GPU
            from .postprocess import tensor_to_frame
            out = tensor_to_frame(out, output_type)
        _emit(progress_callback, 1.0, "Completed text-to-image pipeline")
        return out
