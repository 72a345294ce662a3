"""QwenImage-Edit-Plus engine surface on the HIP transformer (reference engine/qwenimage/shared.py:346-477
`base_denoise` incl. the true-CFG norm rescale :422-429, and engine/qwenimage/edit_plus.py:112-426 `run`:
condition-image latents are concatenated to the noise latents along the sequence axis and the prediction
is cut back to the target tokens, :405-406; `img_shapes` :287-303).  Prompt embeddings (Qwen2.5-VL) and
the VAE-encoded condition image latents are inputs; the sampler loop stays in Python."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from .lora import EngineLoraMixin

from .engine_flux import calculate_shift, compute_dtype
from .schedulers import FlowMatchEulerDiscreteScheduler


def _emit(cb, p, msg):
    if cb is not None:
        try:
            cb(p, msg)
        except Exception:
            pass


class QwenImageEditPlusEngine(EngineLoraMixin):
    def __init__(self, transformer, scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None, decode_fn=None,
                 vae=None, text_encoder=None):
        self.transformer = transformer
        self.vae = vae
        self.text_encoder = text_encoder        # Qwen2_5_VLForConditionalGeneration (prompt + condition images -> embeddings)
        # Qwen-Image scheduler_config.json: dynamic exponential shifting, base/max shift 0.5/0.9,
        # base/max seq 256/8192, shift_terminal 0.02
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler(
            shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, base_image_seq_len=256,
            max_image_seq_len=8192, shift_terminal=0.02)
        self.decode_fn = decode_fn
        self.cfg_streams = True          # true CFG: the unconditional forward on a side HIP stream (base_denoise)
        self._cfg_stream = None

    @property
    def device(self):
        return self.transformer.device

    @staticmethod
    def _pack_latents(latents):
        """[B, C, 1, H, W] -> [B, (H/2)(W/2), 4 C] (QwenImage `_pack_latents`, engine/qwenimage/shared.py)."""
        B, Cc, _, H, W = latents.shape
        x = latents.view(B, Cc, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(B, (H // 2) * (W // 2), Cc * 4)

    @staticmethod
    def _unpack_latents(latents, height, width):
        B, _, ch = latents.shape
        h, w = 2 * (height // 16), 2 * (width // 16)
        x = latents.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
        return x.reshape(B, ch // 4, 1, h, w)

    @torch.no_grad()
    def vae_encode(self, image: torch.Tensor, sample_mode: str = "mode", generator=None) -> torch.Tensor:
        """BaseEngine.vae_encode (engine/base_engine.py:2061-2165): tiled encode, posterior mode / sample, normalise."""
        x = image.to(self.device, compute_dtype(self.vae))
        if x.dim() == 4:
            x = x.unsqueeze(2)
        self.vae.enable_tiling()
        post = self.vae.encode(x, return_dict=False)[0]
        if sample_mode not in ("mode", "sample"):
            raise ValueError(f"Invalid sample mode: {sample_mode}")
        lat = post.mode() if sample_mode == "mode" else post.sample(generator=generator)
        return self.vae.normalize_latents(lat.to(compute_dtype(self.vae)))

    def prepare_image_latents(self, images, batch_size: int = 1):
        """`_prepare_image_latents` (engine/qwenimage/edit_plus.py:26-110) for condition images given as pixels in [-1, 1]
        ([B, 3, H, W]) or as latents: encode (posterior mode), repeat to the batch, pack, concatenate along the sequence.
        Returns (image_latents [B, S, 64], image_shapes [(H, W) in pixels, ...])."""
        if not isinstance(images, (list, tuple)):
            images = [images]
        packed, shapes = [], []
        for image in images:
            lat = self.vae_encode(image) if image.shape[1] != 16 else image.to(self.device)
            if lat.dim() == 4:
                lat = lat.unsqueeze(2)
            if batch_size > lat.shape[0]:
                if batch_size % lat.shape[0]:
                    raise ValueError(f"Cannot duplicate `image` of batch size {lat.shape[0]} to {batch_size} text prompts.")
                lat = torch.cat([lat] * (batch_size // lat.shape[0]), dim=0)
            shapes.append((lat.shape[3] * 8, lat.shape[4] * 8))
            packed.append(self._pack_latents(lat))
        return torch.cat(packed, dim=1), shapes

    @torch.no_grad()
    def vae_decode(self, latents: torch.Tensor, height: int, width: int) -> torch.Tensor:
        z = self._unpack_latents(latents, height, width)
        z = self.vae.denormalize_latents(z.to(torch.float32)).to(compute_dtype(self.vae))
        self.vae.enable_tiling()
        return self.vae.decode(z, return_dict=False)[0][:, :, 0]

    def _render_step(self, latents, render_on_step_callback, height, width):
        """Latent preview (reference engine/qwenimage/shared.py:319-342 `_render_step`): unpack -> denormalise -> decode the
        CURRENT latents and hand the image to the callback; a failing preview never interrupts the denoise."""
        try:
            img = self.decode_fn(latents) if self.decode_fn is not None else self.vae_decode(latents, height, width)
            render_on_step_callback(img)
        except Exception:
            pass

    def base_denoise(self, latents, timesteps, prompt_embeds, img_shapes, image_latents=None,
                     negative_prompt_embeds=None, true_cfg_scale: float = 1.0, use_cfg_guidance: bool = False,
                     denoise_progress_callback=None, render_on_step: bool = False, render_on_step_callback=None,
                     render_on_step_interval: int = 3, preview_hw=None):
        _emit(denoise_progress_callback, 0.0, "Starting denoise")
        n = len(timesteps)
        n_tgt = latents.shape[1]
        if hasattr(self.transformer, "pack"):
            self.transformer.pack()          # on the calling stream, before any forward forks off onto the side stream
        # every step's modulation vectors from one pass over the projection weights (qwenimage.py `begin_schedule`); built on the
        # calling stream before any forward forks off; a transformer without the hook runs as before
        scheduled = None          # this clip's schedule handle (schedule.py): never shared with another clip in flight
        if hasattr(self.transformer, "begin_schedule") and n > 0:
            scheduled = self.transformer.begin_schedule(torch.stack([t.expand(latents.shape[0]).to(latents.dtype) / 1000 for t in timesteps]))
        try:
            return self._denoise_loop(latents, timesteps, prompt_embeds, img_shapes, image_latents, negative_prompt_embeds,
                                      true_cfg_scale, use_cfg_guidance, denoise_progress_callback, render_on_step,
                                      render_on_step_callback, render_on_step_interval, preview_hw, scheduled)
        finally:
            if scheduled is not None:
                self.transformer.end_schedule(scheduled)

    def _denoise_loop(self, latents, timesteps, prompt_embeds, img_shapes, image_latents, negative_prompt_embeds, true_cfg_scale,
                      use_cfg_guidance, denoise_progress_callback, render_on_step, render_on_step_callback, render_on_step_interval,
                      preview_hw, scheduled):
        n = len(timesteps)
        n_tgt = latents.shape[1]
        for i, t in enumerate(timesteps):
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            x = latents if image_latents is None else torch.cat([latents, image_latents], dim=1)
            kw = dict(hidden_states=x, timestep=timestep / 1000, encoder_hidden_states_mask=None,
                      img_shapes=img_shapes, return_dict=False)
            if scheduled is not None:
                kw["attention_kwargs"] = {"modulation_step": i, "modulation_schedule": scheduled}
            cfg = use_cfg_guidance and negative_prompt_embeds is not None
            side = None
            if cfg and self.cfg_streams and x.is_cuda:
                # true CFG = two independent forwards per step: the unconditional one runs on a side HIP stream next to the
                # conditional one (per-stream workspaces in the model) and fills the tile-quantisation gaps of a B = 1 forward
                # (+5.5 % per image measured for two side-by-side forwards, profiles/r03_qwen_batch_streams_ab.json); same
                # kernels on the same data, bit-identical to running them back to back (`cfg_streams = False`)
                if self._cfg_stream is None:
                    self._cfg_stream = torch.cuda.Stream(device=x.device)
                side, main = self._cfg_stream, torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side), self.transformer.cache_context("uncond"):
                    neg = self.transformer(encoder_hidden_states=negative_prompt_embeds,
                                           txt_seq_lens=[negative_prompt_embeds.shape[1]], **kw)[0][:, :n_tgt]
                    neg.record_stream(main)
            with self.transformer.cache_context("cond"):
                noise_pred = self.transformer(encoder_hidden_states=prompt_embeds,
                                              txt_seq_lens=[prompt_embeds.shape[1]], **kw)[0][:, :n_tgt]
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            if cfg:
                if side is None:
                    with self.transformer.cache_context("uncond"):
                        neg = self.transformer(encoder_hidden_states=negative_prompt_embeds,
                                               txt_seq_lens=[negative_prompt_embeds.shape[1]], **kw)[0][:, :n_tgt]
                comb = neg + true_cfg_scale * (noise_pred - neg)
                cond_norm = torch.norm(noise_pred, dim=-1, keepdim=True)
                noise_norm = torch.norm(comb, dim=-1, keepdim=True)
                noise_pred = comb * (cond_norm / noise_norm)
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            # shared.py:455-457: every `render_on_step_interval` steps (and after the first), never after the last
            if (render_on_step and render_on_step_callback is not None and preview_hw is not None
                    and (self.decode_fn is not None or self.vae is not None)
                    and ((i + 1) % render_on_step_interval == 0 or i == 0) and i != n - 1):
                self._render_step(latents, render_on_step_callback, preview_hw[0], preview_hw[1])
            _emit(denoise_progress_callback, float(i + 1) / n, f"Denoise {i + 1}/{n}")
        return latents

    def encode_prompt(self, prompt_inputs, drop_idx: int = 64, num_images_per_prompt: int = 1):
        """`_get_qwen_prompt_embeds` with condition images (R/src/engine/qwenimage/shared.py:100-282, edit_plus.py:190-230):
        `prompt_inputs` = what the Qwen2.5-VL processor returns for the edit template + images (input_ids, attention_mask,
        pixel_values, image_grid_thw; the processor is a CPU `transformers` object); the 64 template tokens are dropped."""
        from .prompt import qwen_prompt_embeds
        if self.text_encoder is None:
            raise RuntimeError("QwenImageEditPlusEngine: prompts need a text_encoder (Qwen2.5-VL); or pass prompt_embeds")
        get = prompt_inputs.get if isinstance(prompt_inputs, dict) else (lambda k, d=None: getattr(prompt_inputs, k, d))
        return qwen_prompt_embeds(self.text_encoder, get("input_ids"), get("attention_mask"), get("pixel_values"),
                                  get("image_grid_thw"), drop_idx=drop_idx, num_images_per_prompt=num_images_per_prompt,
                                  dtype=compute_dtype(self.transformer))

    @torch.no_grad()
    def run(self, prompt_embeds: Optional[torch.Tensor] = None, image_latents: Optional[torch.Tensor] = None,
            image_shapes: Sequence[Tuple[int, int]] = (), height: int = 1024, width: int = 1024,
            num_inference_steps: int = 8, negative_prompt_embeds: Optional[torch.Tensor] = None,
            true_cfg_scale: float = 1.0, latents: Optional[torch.Tensor] = None, seed: Optional[int] = None,
            return_latents: bool = False, progress_callback=None, images=None, output_type: Optional[str] = None,
            render_on_step: bool = False, render_on_step_callback=None, render_on_step_interval: int = 3,
            prompt_inputs=None, negative_prompt_inputs=None, drop_idx: int = 64, **_ignored):
        """`engine.run(prompt=..., image_list=..., ...)` of the EditPlus engine (R/src/engine/qwenimage/edit_plus.py:112-426) from
        the point where the processor has run: `prompt_inputs` / `negative_prompt_inputs` (token ids + pixel patches of the
        condition images) go through Qwen2.5-VL here; `images` (pixels) through the VAE encoder; or pass the embeddings."""
        dev, dt = self.device, compute_dtype(self.transformer)
        if prompt_embeds is None:
            _emit(progress_callback, 0.05, "Encoding prompt")
            prompt_embeds, _ = self.encode_prompt(prompt_inputs, drop_idx)
            if negative_prompt_inputs is not None:
                negative_prompt_embeds, _ = self.encode_prompt(negative_prompt_inputs, drop_idx)
        if images is not None:       # condition images as pixels (or latents): encode + pack here
            image_latents, image_shapes = self.prepare_image_latents(images, prompt_embeds.shape[0])
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
                                    denoise_progress_callback=mapped, render_on_step=render_on_step,
                                    render_on_step_callback=render_on_step_callback,
                                    render_on_step_interval=render_on_step_interval, preview_hw=(height, width))
        if return_latents or (self.decode_fn is None and self.vae is None):
            return latents
        out = self.decode_fn(latents) if self.decode_fn is not None else self.vae_decode(latents, height, width)
        if output_type is not None and torch.is_tensor(out):
            from .postprocess import tensor_to_frame
            out = tensor_to_frame(out, output_type)
        return out
