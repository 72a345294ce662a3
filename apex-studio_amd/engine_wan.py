"""Wan 2.2 A14B text-to-video engine surface on the HIP transformer + HIP VAE.

Mirrors reference engine/wan/t2v.py:12-247 (`run`) and engine/wan/shared/__init__.py:478-608
(`moe_denoise`), :464-476 (`_select_dual_noise_guidance_scale`), :309-462 (expert selection by
`t >= boundary_timestep`): same argument names, the `(progress, message)` callback protocol and the
`render_on_step_callback(frames)` preview hook.  Both 14B experts stay resident (57 GB of 288 GB), so
the reference's load/offload choreography between experts has no counterpart.  Text encoding (UMT5) is
outside the hot path: prompt embeddings are inputs.  The sampler loop stays in Python.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch

from .lora import EngineLoraMixin

from .engine_flux import compute_dtype
from .schedulers import UniPCMultistepScheduler


def _emit(cb, p, msg):
    if cb is not None:
        try:
            cb(p, msg)
        except Exception:
            pass


class WanT2VEngine(EngineLoraMixin):
    def __init__(self, high_noise_transformer, low_noise_transformer=None, vae=None,
                 scheduler: Optional[UniPCMultistepScheduler] = None, boundary_ratio: Optional[float] = 0.875,
                 vae_scale_factor_temporal: int = 4, vae_scale_factor_spatial: int = 8, text_encoder=None):
        from .prompt import TextEncoder
        self.text_encoder = text_encoder if text_encoder is None or isinstance(text_encoder, TextEncoder) \
            else TextEncoder(text_encoder)                      # UMT5-XXL (manifest wan-2.2-a14b-text-to-video yml)
        self.high_noise_transformer = high_noise_transformer
        self.low_noise_transformer = low_noise_transformer or high_noise_transformer
        self.vae = vae
        self.scheduler = scheduler or UniPCMultistepScheduler(shift=3.0)
        self.boundary_ratio = boundary_ratio
        self.vae_scale_factor_temporal = vae_scale_factor_temporal
        self.vae_scale_factor_spatial = vae_scale_factor_spatial
        # the latent the scheduler steps (the VAE's z_dim): `out_channels` — an image-to-video expert takes 36 input channels
        self.num_channels_latents = getattr(high_noise_transformer.config, "out_channels", None) or high_noise_transformer.config.in_channels

    @property
    def device(self):
        return self.high_noise_transformer.device

    @staticmethod
    def _select_dual_noise_guidance_scale(t, boundary_timestep, guidance_scale) -> float:
        high = boundary_timestep is not None and bool(t >= boundary_timestep)
        if isinstance(guidance_scale, (list, tuple)):
            return float(guidance_scale[0] if high else guidance_scale[1])
        return float(guidance_scale)

    def _select_dual_noise_transformer(self, t, boundary_timestep):
        if boundary_timestep is None or bool(t >= boundary_timestep):
            return self.high_noise_transformer
        return self.low_noise_transformer

    def vae_decode(self, latents: torch.Tensor) -> torch.Tensor:
        """reference engine/base_engine.py:2030-2059: denormalize -> enable tiling -> decode."""
        z = self.vae.denormalize_latents(latents.to(torch.float32)).to(compute_dtype(self.vae))
        self.vae.enable_tiling()
        return self.vae.decode(z, return_dict=False)[0]

    def moe_denoise(self, latents, timesteps, prompt_embeds, negative_prompt_embeds=None,
                    guidance_scale: Union[float, List[float]] = 5.0, boundary_timestep=None,
                    use_cfg_guidance: bool = True, transformer_dtype=None, render_on_step: bool = False,
                    render_on_step_callback=None, render_on_step_interval: int = 3,
                    denoise_progress_callback=None, easy_cache_thresh: float = 0.0, easy_cache_ret_steps: int = 10,
                    latent_condition: Optional[torch.Tensor] = None):
        """`easy_cache_thresh` > 0: EasyCache step skipping (R/src/engine/wan/shared/__init__.py:372-381, :435-444, :502-504).  The
        reference enables it on the high-noise expert WITH a reset of its (module-global) state and on the low-noise expert WITHOUT
        one, so the call count, the rate K, the accumulated error and the caches run on across the expert switch: no second
        warm-up of `ret_steps` pairs, and the last pair of the clip (`cnt >= 2n - 2`) is always computed.  Both experts are
        resident here: the first expert of the loop gets a fresh state, every later one continues it
        (`share_easy_cache_state`), and the cache is switched off when the loop ends."""
        _emit(denoise_progress_callback, 0.0, "Starting denoise")
        try:
            return self._moe_loop(latents, timesteps, prompt_embeds, negative_prompt_embeds, guidance_scale, boundary_timestep,
                                  use_cfg_guidance, transformer_dtype, render_on_step, render_on_step_callback, render_on_step_interval,
                                  denoise_progress_callback, easy_cache_thresh, easy_cache_ret_steps, latent_condition)
        finally:
            if easy_cache_thresh > 0.0:
                for tr in {id(self.high_noise_transformer): self.high_noise_transformer,
                           id(self.low_noise_transformer): self.low_noise_transformer}.values():
                    if hasattr(tr, "disable_easy_cache"):
                        tr.disable_easy_cache()

    def _moe_loop(self, latents, timesteps, prompt_embeds, negative_prompt_embeds, guidance_scale, boundary_timestep, use_cfg_guidance,
                  transformer_dtype, render_on_step, render_on_step_callback, render_on_step_interval, denoise_progress_callback,
                  easy_cache_thresh, easy_cache_ret_steps, latent_condition=None):
        n = len(timesteps)
        current = None
        for i, t in enumerate(timesteps):
            timestep = t.expand(latents.shape[0])
            transformer = self._select_dual_noise_transformer(t, boundary_timestep)
            if easy_cache_thresh > 0.0 and transformer is not current and hasattr(transformer, "enable_easy_cache"):
                first = current is None
                if not first and hasattr(transformer, "share_easy_cache_state"):
                    transformer.share_easy_cache_state(current)
                transformer.enable_easy_cache(n, easy_cache_thresh, easy_cache_ret_steps, should_reset_global_cache=first)
            current = transformer
            dt_x = transformer_dtype or compute_dtype(transformer)
            # image-to-video: [noisy latents | first-frame mask | encoded condition video] along channels (moe_denoise,
            # R/src/engine/wan/shared/__init__.py:515-520); the scheduler steps the 16 latent channels only
            x = latents.to(dt_x) if latent_condition is None else torch.cat([latents, latent_condition.to(latents.dtype)], dim=1).to(dt_x)
            scale = self._select_dual_noise_guidance_scale(t, boundary_timestep, guidance_scale)
            noise_pred = transformer(hidden_states=x, timestep=timestep, encoder_hidden_states=prompt_embeds,
                                     return_dict=False)[0]
            if use_cfg_guidance and negative_prompt_embeds is not None:
                uncond = transformer(hidden_states=x, timestep=timestep,
                                     encoder_hidden_states=negative_prompt_embeds, return_dict=False)[0]
                noise_pred = uncond + scale * (noise_pred - uncond)
            latents = self.scheduler.step(noise_pred.to(torch.float32), t, latents, return_dict=False)[0]
            if (render_on_step and render_on_step_callback and self.vae is not None
                    and ((i + 1) % render_on_step_interval == 0 or i == 0) and i != n - 1):
                try:
                    render_on_step_callback(self.vae_decode(latents))
                except Exception:
                    pass
            _emit(denoise_progress_callback, float(i + 1) / n, f"Denoising step {i + 1}/{n}")
        return latents

    def encode_prompt(self, prompt=None, prompt_ids=None, num_videos: int = 1, text_encoder_kwargs=None):
        """`self.text_encoder.encode(prompt, num_videos_per_prompt=..., **text_encoder_kwargs)` of R/src/engine/wan/t2v.py:77-95
        with the manifest's `use_attention_mask: true`: 512 tokens, masked encoder, embeddings past the prompt's length zeroed."""
        from .prompt import split_ids
        if self.text_encoder is None:
            raise RuntimeError("WanT2VEngine: prompts need a text_encoder (UMT5); or pass prompt_embeds")
        kw = {"use_attention_mask": True, **(text_encoder_kwargs or {})}
        a = dict(text=prompt) if prompt_ids is None else dict(zip(("input_ids", "attention_mask"), split_ids(prompt_ids)))
        return self.text_encoder.encode(num_videos_per_prompt=num_videos, **a, **kw)

    @torch.no_grad()
    def run(self, prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
            height: int = 720, width: int = 1280, duration: int = 81, num_inference_steps: int = 30,
            guidance_scale: Union[float, List[float]] = (4.0, 3.0), seed: Optional[int] = None,
            generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
            return_latents: bool = False, progress_callback=None, render_on_step: bool = False,
            render_on_step_callback=None, render_on_step_interval: int = 3, output_type: Optional[str] = None,
            prompt=None, negative_prompt=None, prompt_ids=None, negative_prompt_ids=None, num_videos: int = 1,
            text_encoder_kwargs=None, easy_cache_thresh: float = 0.0, easy_cache_ret_steps: int = 10, **_ignored):
        """`engine.run(prompt=..., negative_prompt=..., ...)` (R/src/engine/wan/t2v.py:12-247): prompts as strings (text
        encoder with a tokenizer) or token ids `(input_ids, attention_mask)`, or pre-computed embeddings."""
        dev = self.device
        if prompt_embeds is None:
            _emit(progress_callback, 0.05, "Encoding prompt")
            prompt_embeds = self.encode_prompt(prompt, prompt_ids, num_videos, text_encoder_kwargs)
            if negative_prompt is not None or negative_prompt_ids is not None:
                negative_prompt_embeds = self.encode_prompt(negative_prompt, negative_prompt_ids, num_videos, text_encoder_kwargs)
        B = prompt_embeds.shape[0]
        num_latent_frames = (duration - 1) // self.vae_scale_factor_temporal + 1
        shape = (B, self.num_channels_latents, num_latent_frames, height // self.vae_scale_factor_spatial,
                 width // self.vae_scale_factor_spatial)
        if latents is None:
            if generator is None:
                generator = torch.Generator(device=dev)
                if seed is not None:
                    generator.manual_seed(seed)
            latents = torch.randn(shape, generator=generator, device=generator.device, dtype=torch.float32).to(dev)
        else:
            latents = latents.to(device=dev, dtype=torch.float32)
        _emit(progress_callback, 0.2, "Prepared latents")
        timesteps = self.scheduler.set_timesteps(num_inference_steps, device=dev)
        boundary = None
        if self.boundary_ratio is not None:
            boundary = self.boundary_ratio * self.scheduler.config["num_train_timesteps"]
        cfg = negative_prompt_embeds is not None
        dt = compute_dtype(self.high_noise_transformer)
        pe = prompt_embeds.to(dev, dt)
        ne = negative_prompt_embeds.to(dev, dt) if cfg else None

        def mapped(p, msg):
            _emit(progress_callback, 0.5 + 0.4 * p, msg)

        latents = self.moe_denoise(latents=latents, timesteps=timesteps, prompt_embeds=pe,
                                   negative_prompt_embeds=ne, guidance_scale=list(guidance_scale)
                                   if isinstance(guidance_scale, (list, tuple)) else guidance_scale,
                                   boundary_timestep=boundary, use_cfg_guidance=cfg,
                                   render_on_step=render_on_step, render_on_step_callback=render_on_step_callback,
                                   render_on_step_interval=render_on_step_interval,
                                   denoise_progress_callback=mapped, easy_cache_thresh=easy_cache_thresh,
                                   easy_cache_ret_steps=easy_cache_ret_steps)
        if return_latents or self.vae is None:
            _emit(progress_callback, 1.0, "Returning latents")
            return latents
        _emit(progress_callback, 0.92, "Decoding video")
        video = self.vae_decode(latents)
        _emit(progress_callback, 1.0, "Completed text-to-video pipeline")
        if output_type is not None:      # t2v.py: `self._tensor_to_frames(video)` — uint8 frames made on the GPU
            from .postprocess import tensor_to_frames
            return tensor_to_frames(video, output_type)
        return video


class WanI2VEngine(WanT2VEngine):
    """Wan-2.2 A14B image-to-video (R/src/engine/wan/i2v.py:13-314, the `boundary_ratio` branch: two experts with
    `in_channels` = 36, no CLIP image embeddings — `moe_denoise` drops `encoder_hidden_states_image`, shared/__init__.py:494-495).
    The first frame conditions the clip through the latent path only:

        video_condition = [image | zeros x (num_frames - 1)]                           (i2v.py:186-198)
        latent_condition = normalize_latents(vae.encode(video_condition).mode())       (BaseEngine.vae_encode, base_engine.py:2062-2165)
        mask = 1 on the first frame's four sub-frames, 0 after -> [B, 4, T_lat, h, w]  (i2v.py:220-249)
        every step:  hidden_states = cat([latents, mask, latent_condition], dim=1)     (36 channels)

    `image`: a PIL image / HWC uint8 array (resized to the aspect-preserving size of area height x width on the reference's
    rule, `_aspect_ratio_resize`, base_engine.py:501-514, then x / 127.5 - 1) or pixels [B|1, 3, H, W] in [-1, 1] (used as they
    are: H, W multiples of 16).  The TI2V-5B form (`expand_timesteps`: per-token timesteps) is a different model and raises."""

    @staticmethod
    def aspect_ratio_size(h0: int, w0: int, max_area: int, mod_value: int = 16):
        """`_aspect_ratio_resize` (base_engine.py:501-514): the size of area <= max_area with the image's aspect ratio, floored to
        multiples of mod_value."""
        import numpy as np
        aspect = h0 / w0
        return (int(round(np.sqrt(max_area * aspect))) // mod_value * mod_value,
                int(round(np.sqrt(max_area / aspect))) // mod_value * mod_value)

    def preprocess_image(self, image, height: int, width: int):
        """-> (pixels [B, 3, H, W] float32 in [-1, 1] on the device, H, W)."""
        if isinstance(image, torch.Tensor):
            x = image if image.dim() == 4 else image[None]
            if x.shape[1] != 3 or x.shape[-2] % 16 or x.shape[-1] % 16:
                raise ValueError(f"wan i2v: pixel tensors must be [B, 3, H, W] with H, W multiples of 16, got {tuple(image.shape)}")
            return x.to(self.device, torch.float32), int(x.shape[-2]), int(x.shape[-1])
        import numpy as np
        from PIL import Image
        img = image if isinstance(image, Image.Image) else Image.fromarray(np.asarray(image))
        img = img.convert("RGB")
        h, w = self.aspect_ratio_size(img.height, img.width, height * width, 16)
        img = img.resize((w, h), Image.Resampling.LANCZOS)
        x = torch.from_numpy(np.asarray(img).astype(np.float32) / 255.0).permute(2, 0, 1)[None]      # VideoProcessor.preprocess:
        return (2.0 * x - 1.0).to(self.device), h, w                                                  # [0, 1] -> [-1, 1]

    def vae_encode(self, video: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """BaseEngine.vae_encode (base_engine.py:2139-2160): tiling on, `encode(...).mode()`, normalised in `dtype`."""
        self.vae.enable_tiling()
        lat = self.vae.encode(video.to(self.device, compute_dtype(self.vae)), return_dict=False)[0].mode()
        return self.vae.normalize_latents(lat.to(dtype))

    def first_frame_mask(self, batch: int, num_frames: int, latent_height: int, latent_width: int) -> torch.Tensor:
        """i2v.py:220-249: ones on pixel frame 0, zeros after; frame 0 repeated to the temporal factor so every latent frame owns
        `vae_scale_factor_temporal` mask channels -> [B, 4, T_lat, h, w]."""
        f = self.vae_scale_factor_temporal
        m = torch.ones(batch, 1, num_frames, latent_height, latent_width, device=self.device)
        m[:, :, 1:] = 0
        m = torch.cat([torch.repeat_interleave(m[:, :, 0:1], dim=2, repeats=f), m[:, :, 1:]], dim=2)
        return m.view(batch, -1, f, latent_height, latent_width).transpose(1, 2)

    def prepare_latent_condition(self, pixels: torch.Tensor, num_frames: int, batch: int) -> torch.Tensor:
        B0, _, H, W = pixels.shape
        first = pixels[:, :, None]
        video = torch.cat([first, first.new_zeros(B0, 3, num_frames - 1, H, W)], dim=2)
        cond = self.vae_encode(video, dtype=torch.float32)
        if cond.shape[0] != batch:
            cond = cond.repeat(batch // cond.shape[0], 1, 1, 1, 1)
        mask = self.first_frame_mask(batch, num_frames, cond.shape[-2], cond.shape[-1]).to(cond.dtype)
        return torch.cat([mask, cond], dim=1)

    @torch.no_grad()
    def run(self, image=None, prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
            height: int = 480, width: int = 832, duration: int = 81, num_inference_steps: int = 30,
            guidance_scale: Union[float, List[float], None] = None, high_noise_guidance_scale: Optional[float] = 1.0,
            low_noise_guidance_scale: Optional[float] = 1.0, seed: Optional[int] = None,
            generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None, return_latents: bool = False,
            progress_callback=None, render_on_step: bool = False, render_on_step_callback=None, render_on_step_interval: int = 3,
            output_type: Optional[str] = None, prompt=None, negative_prompt=None, prompt_ids=None, negative_prompt_ids=None,
            num_videos: int = 1, text_encoder_kwargs=None, easy_cache_thresh: float = 0.0, easy_cache_ret_steps: int = 10,
            expand_timesteps: bool = False, ip_image=None, **_ignored):
        if image is None:
            raise ValueError("wan i2v: `image` is required")
        if expand_timesteps or ip_image is not None:
            raise NotImplementedError("wan i2v: the TI2V-5B (`expand_timesteps`) and IP-image forms are other models; this engine "
                                      "serves the Wan-2.2 A14B two-expert image-to-video path")
        if self.vae is None:
            raise RuntimeError("WanI2VEngine needs the VAE (the condition video is encoded with it)")
        if self.high_noise_transformer.config.in_channels != 36:
            raise ValueError(f"wan i2v: the experts must take 36 input channels (16 latent + 4 mask + 16 condition), got "
                             f"{self.high_noise_transformer.config.in_channels}")
        dev = self.device
        _emit(progress_callback, 0.0, "Starting image-to-video pipeline")
        if high_noise_guidance_scale is not None and low_noise_guidance_scale is not None and guidance_scale is None:
            guidance_scale = [high_noise_guidance_scale, low_noise_guidance_scale]        # i2v.py:46-53
        if prompt_embeds is None:
            _emit(progress_callback, 0.05, "Encoding prompt")
            prompt_embeds = self.encode_prompt(prompt, prompt_ids, num_videos, text_encoder_kwargs)
            if negative_prompt is not None or negative_prompt_ids is not None:
                negative_prompt_embeds = self.encode_prompt(negative_prompt, negative_prompt_ids, num_videos, text_encoder_kwargs)
        # i2v.py:56-64: CFG only with a negative prompt AND both scales above 1
        gs = list(guidance_scale) if isinstance(guidance_scale, (list, tuple)) else guidance_scale
        cfg = negative_prompt_embeds is not None and (all(g > 1.0 for g in gs) if isinstance(gs, list) else gs > 1.0)
        B = prompt_embeds.shape[0]
        pixels, height, width = self.preprocess_image(image, height, width)
        z_dim = 16
        num_latent_frames = (duration - 1) // self.vae_scale_factor_temporal + 1
        shape = (B, z_dim, num_latent_frames, height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if latents is None:
            if generator is None:
                generator = torch.Generator(device=dev)
                if seed is not None:
                    generator.manual_seed(seed)
            latents = torch.randn(shape, generator=generator, device=generator.device, dtype=torch.float32).to(dev)
        else:
            latents = latents.to(device=dev, dtype=torch.float32)
        _emit(progress_callback, 0.3, "Initialized latent noise")
        latent_condition = self.prepare_latent_condition(pixels, duration, B)
        if tuple(latent_condition.shape[2:]) != tuple(latents.shape[2:]):
            raise ValueError(f"wan i2v: condition latents {tuple(latent_condition.shape)} do not match the video latents {tuple(latents.shape)}")
        timesteps = self.scheduler.set_timesteps(num_inference_steps, device=dev)
        boundary = None if self.boundary_ratio is None else self.boundary_ratio * self.scheduler.config["num_train_timesteps"]
        dt = compute_dtype(self.high_noise_transformer)
        pe = prompt_embeds.to(dev, dt)
        ne = negative_prompt_embeds.to(dev, dt) if cfg else None
        _emit(progress_callback, 0.45, "Starting denoise phase")

        def mapped(p, msg):
            _emit(progress_callback, 0.5 + 0.4 * p, msg)

        latents = self.moe_denoise(latents=latents, timesteps=timesteps, prompt_embeds=pe, negative_prompt_embeds=ne,
                                   guidance_scale=gs, boundary_timestep=boundary, use_cfg_guidance=cfg,
                                   render_on_step=render_on_step, render_on_step_callback=render_on_step_callback,
                                   render_on_step_interval=render_on_step_interval, denoise_progress_callback=mapped,
                                   easy_cache_thresh=easy_cache_thresh, easy_cache_ret_steps=easy_cache_ret_steps,
                                   latent_condition=latent_condition)
        _emit(progress_callback, 0.92, "Denoising complete")
        if return_latents:
            _emit(progress_callback, 1.0, "Returning latents")
            return latents
        video = self.vae_decode(latents)
        _emit(progress_callback, 1.0, "Completed image-to-video pipeline")
        if output_type is not None:
            from .postprocess import tensor_to_frames
            return tensor_to_frames(video, output_type)
        return video
