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
    """engine.run(...) for Flux text-to-image: prompts (token ids, or strings when the text encoders carry tokenizers) or
    pre-computed prompt embeddings in, image out."""

    def __init__(self, transformer, scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None,
                 decode_fn: Optional[Callable[[torch.Tensor], object]] = None, vae_scale_factor: int = 8,
                 text_encoder=None, text_encoder_2=None):
        from .prompt import TextEncoder
        self.transformer = transformer
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler.flux_dev()
        self.decode_fn = decode_fn
        self.vae_scale_factor = vae_scale_factor
        self.num_channels_latents = transformer.config.in_channels // 4
        # CLIP-L (pooled) and T5-XXL (sequence), as the manifest names them (flux-dev-text-to-image yml:59-90)
        wrap = lambda m: m if m is None or isinstance(m, TextEncoder) else TextEncoder(m)      # noqa: E731
        self.text_encoder, self.text_encoder_2 = wrap(text_encoder), wrap(text_encoder_2)

    def encode_prompt(self, prompt=None, prompt_2=None, prompt_ids=None, prompt_2_ids=None, num_images: int = 1,
                      text_encoder_kwargs=None, text_encoder_2_kwargs=None):
        """`FluxShared.encode_prompt` (R/src/engine/flux/shared.py:215-345): pooled CLIP embedding + T5 sequence embedding.
        Returns (prompt_embeds, pooled_prompt_embeds, text_ids).  Manifest defaults: 77 / 512 tokens, `pad_with_zero` false."""
        from .prompt import split_ids
        if self.text_encoder is None or self.text_encoder_2 is None:
            raise RuntimeError("FluxT2IEngine: prompts need text_encoder (CLIP) and text_encoder_2 (T5); or pass prompt_embeds")
        k1 = {"max_sequence_length": 77, "pad_with_zero": False, **(text_encoder_kwargs or {})}
        k2 = {"max_sequence_length": 512, "pad_with_zero": False, **(text_encoder_2_kwargs or {})}
        if prompt_2 is None and prompt_2_ids is None:            # `if not prompt_2: prompt_2 = prompt` — same TEXT for both encoders
            if prompt is None:
                raise ValueError("token ids are per tokenizer: pass prompt_2_ids (T5) next to prompt_ids (CLIP)")
            prompt_2 = prompt
        a = dict(text=prompt) if prompt_ids is None else dict(zip(("input_ids", "attention_mask"), split_ids(prompt_ids)))
        b = dict(text=prompt_2) if prompt_2_ids is None else dict(zip(("input_ids", "attention_mask"), split_ids(prompt_2_ids)))
        pooled = self.text_encoder.encode(num_videos_per_prompt=num_images, output_type="pooler_output", **a, **k1)
        embeds = self.text_encoder_2.encode(num_videos_per_prompt=num_images, output_type="hidden_states", **b, **k2)
        text_ids = torch.zeros(embeds.shape[1], 3, device=embeds.device, dtype=embeds.dtype)
        return embeds, pooled, text_ids

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
        # The sampler's timesteps are known here: every step's AdaLN modulation vectors in one pass over the stacked projection
        # weights instead of one 6.4 GB GEMV per step (`begin_schedule`, bit-identical rows); a transformer without the hook (the
        # reference's own class behind this engine) just runs as before.
        scheduled = None          # this clip's schedule handle: two clips through one resident model never share a table
        if hasattr(self.transformer, "begin_schedule") and n > 0:
            B = latents.shape[0]
            scheduled = self.transformer.begin_schedule(
                torch.stack([t.expand(B).to(latents.dtype) / 1000 for t in timesteps]), guidance,
                [pooled_prompt_embeds] + ([negative_pooled_prompt_embeds] if use_cfg_guidance else []))
        try:
            latents = self._denoise_loop(latents, timesteps, guidance, prompt_embeds, pooled_prompt_embeds, text_ids, latent_ids,
                                         negative_prompt_embeds, negative_pooled_prompt_embeds, negative_text_ids, true_cfg_scale,
                                         use_cfg_guidance, render_on_step, render_on_step_callback, render_on_step_interval,
                                         denoise_progress_callback, preview_hw, scheduled)
        finally:
            if scheduled is not None:
                self.transformer.end_schedule(scheduled)
        _emit(denoise_progress_callback, 1.0, "Denoise finished")
        return latents

    def _denoise_loop(self, latents, timesteps, guidance, prompt_embeds, pooled_prompt_embeds, text_ids, latent_ids,
                      negative_prompt_embeds, negative_pooled_prompt_embeds, negative_text_ids, true_cfg_scale, use_cfg_guidance,
                      render_on_step, render_on_step_callback, render_on_step_interval, denoise_progress_callback, preview_hw,
                      scheduled):
        n = len(timesteps)
        for i, t in enumerate(timesteps):
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            jkw = {"joint_attention_kwargs": {"modulation_step": i, "modulation_schedule": scheduled}} if scheduled is not None else {}
            with self.transformer.cache_context("cond"):
                noise_pred = self.transformer(
                    hidden_states=latents, timestep=timestep / 1000, guidance=guidance,
                    pooled_projections=pooled_prompt_embeds, encoder_hidden_states=prompt_embeds,
                    txt_ids=text_ids, img_ids=latent_ids, return_dict=False, **jkw)[0]
            if use_cfg_guidance:
                with self.transformer.cache_context("uncond"):
                    neg = self.transformer(
                        hidden_states=latents, timestep=timestep / 1000, guidance=guidance,
                        pooled_projections=negative_pooled_prompt_embeds,
                        encoder_hidden_states=negative_prompt_embeds, txt_ids=negative_text_ids,
                        img_ids=latent_ids, return_dict=False, **jkw)[0]
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
        return latents

    @torch.no_grad()
    def run(self, prompt_embeds: Optional[torch.Tensor] = None, pooled_prompt_embeds: Optional[torch.Tensor] = None,
            height: int = 1024, width: int = 1024, num_inference_steps: int = 28, guidance_scale: float = 3.5,
            latents: Optional[torch.Tensor] = None, seed: Optional[int] = None,
            generator: Optional[torch.Generator] = None, return_latents: bool = False,
            progress_callback=None, render_on_step: bool = False, render_on_step_callback=None,
            render_on_step_interval: int = 3, sigmas=None, output_type: Optional[str] = None,
            prompt=None, prompt_2=None, prompt_ids=None, prompt_2_ids=None, num_images: int = 1,
            text_encoder_kwargs=None, text_encoder_2_kwargs=None, **_ignored):
        """`engine.run(prompt=..., height=..., ...)` as the render queue calls it (R/src/api/ray_tasks.py:2775-2812,
        R/src/engine/flux/t2i.py:15-260).  Either prompts — `prompt` / `prompt_2` strings (text encoders with tokenizers) or
        `prompt_ids` (CLIP) + `prompt_2_ids` (T5) — or the pre-computed `prompt_embeds` + `pooled_prompt_embeds`."""
        dev, dt = self.device, compute_dtype(self.transformer)
        if prompt_embeds is None:
            _emit(progress_callback, 0.05, "Encoding prompt")
            prompt_embeds, pooled_prompt_embeds, _ = self.encode_prompt(prompt, prompt_2, prompt_ids, prompt_2_ids, num_images,
                                                                        text_encoder_kwargs, text_encoder_2_kwargs)
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
