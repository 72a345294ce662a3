"""QwenImage-Edit-Plus engine surface on the HIP transformer (reference engine/qwenimage/shared.py:346-477
`base_denoise` incl. the true-CFG norm rescale :422-429, and engine/qwenimage/edit_plus.py:112-426 `run`:
condition-image latents are concatenated to the noise latents along the sequence axis and the prediction
is cut back to the target tokens, :405-406; `img_shapes` :287-303).  Prompt embeddings (Qwen2.5-VL) and
the VAE-encoded condition image latents are inputs; the sampler loop stays in Python."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from .lora import EngineLoraMixin

from .engine_flux import calculate_shift
from .schedulers import FlowMatchEulerDiscreteScheduler


def _emit(cb, p, msg):
    if cb is not None:
        try:
            cb(p, msg)
        except Exception:
            pass


class QwenImageEditPlusEngine(EngineLoraMixin):
    def __init__(self, transformer, scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None, decode_fn=None):
        self.transformer = transformer
        # Qwen-Image scheduler_config.json: dynamic exponential shifting, base/max shift 0.5/0.9,
        # base/max seq 256/8192, shift_terminal 0.02
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler(
            shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, base_image_seq_len=256,
            max_image_seq_len=8192, shift_terminal=0.02)
        self.decode_fn = decode_fn

    @property
    def device(self):
        return self.transformer.device

    def base_denoise(self, latents, timesteps, prompt_embeds, img_shapes, image_latents=None,
                     negative_prompt_embeds=None, true_cfg_scale: float = 1.0, use_cfg_guidance: bool = False,
                     denoise_progress_callback=None):
        _emit(denoise_progress_callback, 0.0, "Starting denoise")
        n = len(timesteps)
        n_tgt = latents.shape[1]
        for i, t in enumerate(timesteps):
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            x = latents if image_latents is None else torch.cat([latents, image_latents], dim=1)
            kw = dict(hidden_states=x, timestep=timestep / 1000, encoder_hidden_states_mask=None,
                      img_shapes=img_shapes, return_dict=False)
            with self.transformer.cache_context("cond"):
                noise_pred = self.transformer(encoder_hidden_states=prompt_embeds,
                                              txt_seq_lens=[prompt_embeds.shape[1]], **kw)[0][:, :n_tgt]
            if use_cfg_guidance and negative_prompt_embeds is not None:
                with self.transformer.cache_context("uncond"):
                    neg = self.transformer(encoder_hidden_states=negative_prompt_embeds,
                                           txt_seq_lens=[negative_prompt_embeds.shape[1]], **kw)[0][:, :n_tgt]
                comb = neg + true_cfg_scale * (noise_pred - neg)
                cond_norm = torch.norm(noise_pred, dim=-1, keepdim=True)
                noise_norm = torch.norm(comb, dim=-1, keepdim=True)
                noise_pred = comb * (cond_norm / noise_norm)
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            _emit(denoise_progress_callback, float(i + 1) / n, f"Denoise {i + 1}/{n}")
        return latents

    @torch.no_grad()
    def run(self, prompt_embeds: torch.Tensor, image_latents: Optional[torch.Tensor] = None,
            image_shapes: Sequence[Tuple[int, int]] = (), height: int = 1024, width: int = 1024,
            num_inference_steps: int = 8, negative_prompt_embeds: Optional[torch.Tensor] = None,
            true_cfg_scale: float = 1.0, latents: Optional[torch.Tensor] = None, seed: Optional[int] = None,
            return_latents: bool = True, progress_callback=None, **_ignored):
        dev, dt = self.device, self.transformer.dtype
        h2, w2 = height // 16, width // 16
        B = prompt_embeds.shape[0]
        if latents is None:
            g = torch.Generator(device=dev)
            if seed is not None:
                g.manual_seed(seed)
            latents = torch.randn((B, h2 * w2, 64), generator=g, device=dev, dtype=torch.float32).to(dt)
        else:
            latents = latents.to(dev, dt)
        img_shapes = [[(1, h2, w2)] + [(1, ih // 16, iw // 16) for ih, iw in image_shapes]] * B
        sigmas = torch.linspace(1.0, 1.0 / num_inference_steps, num_inference_steps).tolist()
        c = self.scheduler.config
        mu = calculate_shift(latents.shape[1], c["base_image_seq_len"], c["max_image_seq_len"], c["base_shift"],
                             c["max_shift"])
        timesteps = self.scheduler.set_timesteps(sigmas=sigmas, mu=mu, device=dev)
        self.scheduler.set_begin_index(0)
        cfg = negative_prompt_embeds is not None and true_cfg_scale > 1.0

        def mapped(p, msg):
            _emit(progress_callback, 0.5 + 0.4 * p, msg)

        latents = self.base_denoise(latents, timesteps, prompt_embeds.to(dev, dt), img_shapes,
                                    image_latents=None if image_latents is None else image_latents.to(dev, dt),
                                    negative_prompt_embeds=None if not cfg else negative_prompt_embeds.to(dev, dt),
                                    true_cfg_scale=true_cfg_scale, use_cfg_guidance=cfg,
                                    denoise_progress_callback=mapped)
        if return_latents or self.decode_fn is None:
            return latents
        return self.decode_fn(latents)
