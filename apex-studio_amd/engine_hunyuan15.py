"""HunyuanVideo-1.5 text-to-video and image-to-video sampler loops (stay in Python) — mirrors the reference engine
(`apps/api/src/engine/hunyuanvideo15/t2v.py:107-361`): latents [B, 32, F, H/16, W/16], the transformer input is
`cat([latents, cond_latents (zeros), mask (zeros)], dim=1)` (65 channels, :20-42, :238-241), the timestep goes in as
`t.expand(B).to(latents.dtype)` on the 0-1000 scale (:243-245), manual CFG with optional std rescale (:248-306),
FlowMatch-Euler step, progress protocol 0.45 -> [0.50, 0.90] -> decode.  Prompts: either the four embedding tensors, or token
ids through the engine's own text encoders — the MLLM (Qwen2.5-VL, third-from-last hidden state, the 108 template tokens
dropped) and the glyph ByT5 (a T5-v1.1 encoder over the quoted text, 256 tokens; zeros when the prompt quotes nothing) —
as `encode_prompt` does in the reference (`shared/__init__.py:145-283, 344-437`); tokenizers stay CPU `transformers` objects.

`HunyuanVideo15I2VEngine` mirrors `engine/hunyuanvideo15/i2v.py:14-407`: the first frame is VAE-encoded (posterior mode,
normalised; `_get_image_latents`, shared/__init__.py:285-299 -> BaseEngine.vae_encode, base_engine.py:2061-2165), the
condition latents are that frame followed by zeros and the mask is one on the first latent frame (i2v.py:20-58); the
SigLIP image embeddings (`encode_image`, shared :325-341) are an input like the prompt embeddings.
"""
from __future__ import annotations

from typing import Optional

import torch

from .lora import EngineLoraMixin
from .schedulers import FlowMatchEulerDiscreteScheduler


def _emit(cb, p, msg):
    if cb is not None:
        try:
            cb(p, msg)
        except TypeError:
            cb(p, msg, None)


class HunyuanVideo15T2VEngine(EngineLoraMixin):
    _passes_timestep_r = False

    def __init__(self, transformer, vae=None, scheduler: Optional[FlowMatchEulerDiscreteScheduler] = None,
                 vae_scale_factor_temporal: int = 4, vae_scale_factor_spatial: int = 16,
                 vision_num_semantic_tokens: int = 729, vision_states_dim: int = 1152, decode_fn=None,
                 text_encoder=None, text_encoder_2=None, tokenizer_max_length: int = 1000, tokenizer_2_max_length: int = 256,
                 prompt_template_encode_start_idx: int = 108):
        self.transformer = transformer
        # MLLM (Qwen2.5-VL class, text only) and glyph ByT5 (T5EncoderModel with the ByT5 config), manifest names
        # `text_encoder` / `text_encoder_2`; lengths and the template crop as the reference engine sets them (shared/__init__.py:40-60)
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.tokenizer_max_length, self.tokenizer_2_max_length = tokenizer_max_length, tokenizer_2_max_length
        self.prompt_template_encode_start_idx = prompt_template_encode_start_idx
        self.vae = vae
        self.decode_fn = decode_fn
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler(shift=7.0)
        self.vae_scale_factor_temporal = vae_scale_factor_temporal
        self.vae_scale_factor_spatial = vae_scale_factor_spatial
        self.vision_num_semantic_tokens = vision_num_semantic_tokens
        self.vision_states_dim = vision_states_dim
        self.num_channels_latents = (transformer.config.in_channels - 1) // 2       # 65 = 32 + 32 + 1

    @property
    def device(self):
        return self.transformer.device

    # ---- prompts -----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _get_mllm_prompt_embeds(self, input_ids, attention_mask, num_hidden_layers_to_skip: int = 2, crop_start=None):
        """`_get_mllm_prompt_embeds` (shared/__init__.py:145-232) from the tokenizer's output: `apply_chat_template(system message +
        prompt, padding="max_length", max_length=tokenizer_max_length + crop_start)` ids and mask -> hidden state
        `-(num_hidden_layers_to_skip + 1)` of the MLLM, the `crop_start` template tokens dropped, mask as int64."""
        if self.text_encoder is None:
            raise RuntimeError("HunyuanVideo15 engine: prompts need a text_encoder (Qwen2.5-VL); or pass prompt_embeds*")
        crop = self.prompt_template_encode_start_idx if crop_start is None else crop_start
        dev = self.device
        ids, mask = input_ids.to(dev), attention_mask.to(dev)
        hidden = self.text_encoder(input_ids=ids, attention_mask=mask, output_hidden_states=True).hidden_states[
            -(num_hidden_layers_to_skip + 1)]
        if crop:
            hidden, mask = hidden[:, crop:], mask[:, crop:]
        return hidden.to(self.transformer.dtype), mask.to(torch.int64)

    @torch.no_grad()
    def _get_byt5_prompt_embeds(self, glyph_ids, batch: int):
        """`_get_byt5_prompt_embeds` (shared/__init__.py:234-283): the ByT5 encoding of the prompt's quoted ("glyph") text, padded
        to `tokenizer_2_max_length` with its mask; a prompt that quotes nothing gets zeros and an all-zero mask (`glyph_ids`
        None, or None at that sample's place in a list)."""
        from .prompt import TextEncoder, split_ids
        dev, dt, L = self.device, self.transformer.dtype, self.tokenizer_2_max_length
        per = glyph_ids if isinstance(glyph_ids, list) else [glyph_ids] * batch if glyph_ids is None else None
        if per is None:                                   # one (ids, mask) pair for the whole batch
            ids, mask = split_ids(glyph_ids)
            per = [(ids[b:b + 1], None if mask is None else mask[b:b + 1]) for b in range(ids.shape[0])]
        embs, masks = [], []
        for g in per:
            if g is None:
                if self.text_encoder_2 is None:
                    d_model = self.transformer.config.text_embed_2_dim
                else:
                    d_model = next(p for n, p in self.text_encoder_2.named_parameters() if n.endswith("shared.weight")).shape[1]
                embs.append(torch.zeros(1, L, d_model, device=dev, dtype=dt))
                masks.append(torch.zeros(1, L, device=dev, dtype=torch.int64))
                continue
            if self.text_encoder_2 is None:
                raise RuntimeError("HunyuanVideo15 engine: glyph text needs a text_encoder_2 (ByT5); or pass prompt_embeds_2")
            ids, mask = split_ids(g)
            enc = self.text_encoder_2 if isinstance(self.text_encoder_2, TextEncoder) else TextEncoder(self.text_encoder_2)
            e, m = enc.encode(input_ids=ids, attention_mask=mask, max_sequence_length=L, pad_to_max_length=True,
                              use_attention_mask=True, return_attention_mask=True, output_type="hidden_states",
                              pad_with_zero=False)
            embs.append(e.to(dev, dt))
            masks.append(m.to(dev))
        return torch.cat(embs, dim=0), torch.cat(masks, dim=0)

    def encode_prompt(self, prompt_ids, prompt_2_ids=None, num_videos_per_prompt: int = 1):
        """`encode_prompt` (shared/__init__.py:344-437) from token ids: `prompt_ids` = (input_ids, attention_mask) of the MLLM chat
        template, `prompt_2_ids` = (input_ids, attention_mask) of the glyph text for ByT5, a per-sample list of those / None, or
        None (no quoted text).  Returns (prompt_embeds, prompt_embeds_mask, prompt_embeds_2, prompt_embeds_mask_2), repeated
        `num_videos_per_prompt` times and cast as the reference casts them (masks included)."""
        from .prompt import split_ids
        ids, mask = split_ids(prompt_ids)
        if mask is None:
            mask = torch.ones_like(ids)
        pe, pm = self._get_mllm_prompt_embeds(ids, mask)
        pe2, pm2 = self._get_byt5_prompt_embeds(prompt_2_ids, pe.shape[0])
        B, n, dt = pe.shape[0], num_videos_per_prompt, self.transformer.dtype

        def rep(e, m):
            L = e.shape[1]
            return (e.repeat(1, n, 1).view(B * n, L, -1).to(dt), m.repeat(1, n).view(B * n, L).to(dt))
        return (*rep(pe, pm), *rep(pe2, pm2))

    def vae_decode(self, latents: torch.Tensor) -> torch.Tensor:
        """BaseEngine.vae_decode (engine/base_engine.py:2030-2059) after t2v.py:350 `vae.enable_tiling()`: denormalise,
        tiled decode, [B, 3, F, H, W]."""
        z = self.vae.denormalize_latents(latents.to(torch.float32)).to(self.vae.dtype)
        self.vae.enable_tiling()
        return self.vae.decode(z, return_dict=False)[0]

    def prepare_cond_latents_and_mask(self, latents, dtype, device, image=None):
        b, c, f, h, w = latents.shape
        return (torch.zeros(b, c, f, h, w, dtype=dtype, device=device), torch.zeros(b, 1, f, h, w, dtype=dtype, device=device))

    def denoise(self, latents, timesteps, cond, uncond=None, guidance_scale: float = 6.0, guidance_rescale: float = 0.0,
                image_embeds=None, denoise_progress_callback=None, image=None):
        dt = self.transformer.dtype
        cond_latents, mask = self.prepare_cond_latents_and_mask(latents, dt, latents.device, image=image)
        n = len(timesteps)
        for i, t in enumerate(timesteps):
            x = torch.cat([latents.to(dt), cond_latents, mask], dim=1)
            timestep = t.expand(x.shape[0]).to(dt)
            extra = {}
            # i2v.py:281-288: r = the NEXT timestep (0 after the last); the reference's t2v loop never passes timestep_r
            if self._passes_timestep_r and getattr(self.transformer.config, "use_meanflow", False):
                tr = timesteps[i + 1] if i + 1 < n else torch.zeros((), device=t.device, dtype=t.dtype)
                extra["timestep_r"] = tr.expand(x.shape[0]).to(dt)
            pred = None
            if uncond is not None:
                with self.transformer.cache_context("pred_uncond"):
                    pred_u = self.transformer(hidden_states=x, image_embeds=image_embeds, timestep=timestep,
                                              return_dict=False, **extra, **uncond)[0]
            with self.transformer.cache_context("pred_cond"):
                pred_c = self.transformer(hidden_states=x, image_embeds=image_embeds, timestep=timestep,
                                          return_dict=False, **extra, **cond)[0]
            if uncond is not None:
                pred = pred_u + guidance_scale * (pred_c - pred_u)
                if guidance_rescale > 0.0:    # arXiv 2305.08891 §3.4, t2v.py:291-303
                    dims = list(range(1, pred_c.ndim))
                    rescaled = pred * (pred_c.std(dim=dims, keepdim=True) / pred.std(dim=dims, keepdim=True))
                    pred = guidance_rescale * rescaled + (1 - guidance_rescale) * pred
            else:
                pred = pred_c
            latents = self.scheduler.step(pred, t, latents, return_dict=False)[0]
            _emit(denoise_progress_callback, float(i + 1) / float(max(n, 1)), f"Denoising step {i + 1}/{n}")
        return latents

    @torch.no_grad()
    def run(self, prompt_embeds=None, prompt_embeds_mask=None, prompt_embeds_2=None, prompt_embeds_mask_2=None,
            negative_prompt_embeds=None, negative_prompt_embeds_mask=None, negative_prompt_embeds_2=None,
            negative_prompt_embeds_mask_2=None, height: int = 480, width: int = 832, num_frames: int = 121,
            num_inference_steps: int = 50, guidance_scale: float = 6.0, guidance_rescale: float = 0.0, sigmas=None,
            seed: Optional[int] = None, generator: Optional[torch.Generator] = None, latents=None,
            return_latents: bool = False, progress_callback=None, output_type: Optional[str] = None, image=None,
            image_embeds=None, use_light_vae: bool = False, prompt_ids=None, prompt_2_ids=None, negative_prompt_ids=None,
            negative_prompt_2_ids=None, num_videos_per_prompt: int = 1, **_ignored):
        """`engine.run(prompt=…, …)` (t2v.py:60-160) from the point where the CPU tokenizers have run: `prompt_ids` (+ optional
        `prompt_2_ids` glyph text, `negative_prompt_ids`, `negative_prompt_2_ids`) go through the MLLM and ByT5 here; or pass the
        four `prompt_embeds*` tensors (and their negative counterparts)."""
        dev, dt = self.device, self.transformer.dtype
        if prompt_embeds is None:
            if prompt_ids is None:
                raise ValueError("run() needs prompt_ids (token ids of the MLLM chat template) or prompt_embeds*")
            _emit(progress_callback, 0.05, "Encoding prompt")
            prompt_embeds, prompt_embeds_mask, prompt_embeds_2, prompt_embeds_mask_2 = self.encode_prompt(
                prompt_ids, prompt_2_ids, num_videos_per_prompt)
            if negative_prompt_ids is not None and guidance_scale > 1.0:
                (negative_prompt_embeds, negative_prompt_embeds_mask, negative_prompt_embeds_2,
                 negative_prompt_embeds_mask_2) = self.encode_prompt(negative_prompt_ids, negative_prompt_2_ids, num_videos_per_prompt)
        B = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0 and negative_prompt_embeds is not None
        if guidance_scale > 1.0 and negative_prompt_embeds is None and _ignored.get("negative_prompt") is not None:
            raise ValueError("CFG requested (guidance_scale > 1.0) but no negative prompt embeds were provided.")
        _emit(progress_callback, 0.15, "Preparing timesteps")
        if sigmas is None:
            sigmas = torch.linspace(1.0, 0.0, num_inference_steps + 1, dtype=torch.float64)[:-1]
        timesteps = self.scheduler.set_timesteps(num_inference_steps, device=dev, sigmas=sigmas)
        _emit(progress_callback, 0.20, "Preparing latents")
        shape = (B, self.num_channels_latents, (num_frames - 1) // self.vae_scale_factor_temporal + 1,
                 height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if latents is None:
            if generator is None:
                generator = torch.Generator(device=dev)
                if seed is not None:
                    generator.manual_seed(seed)
            latents = torch.randn(shape, generator=generator, device=generator.device, dtype=torch.float32).to(dev, dt)
        else:
            latents = latents.to(dev, dt)
        if image_embeds is None:      # t2v.py:181-187: zero vision states when there is no reference image
            image_embeds = torch.zeros(B, self.vision_num_semantic_tokens, self.vision_states_dim, dtype=dt, device=dev)
        else:
            image_embeds = image_embeds.to(dev, dt)
            if image_embeds.shape[0] != B:
                image_embeds = image_embeds.repeat(B // image_embeds.shape[0], 1, 1)
        cond = dict(encoder_hidden_states=prompt_embeds.to(dev, dt), encoder_attention_mask=prompt_embeds_mask.to(dev),
                    encoder_hidden_states_2=prompt_embeds_2.to(dev, dt), encoder_attention_mask_2=prompt_embeds_mask_2.to(dev))
        uncond = None
        if do_cfg:
            uncond = dict(encoder_hidden_states=negative_prompt_embeds.to(dev, dt),
                          encoder_attention_mask=negative_prompt_embeds_mask.to(dev),
                          encoder_hidden_states_2=negative_prompt_embeds_2.to(dev, dt),
                          encoder_attention_mask_2=negative_prompt_embeds_mask_2.to(dev))
        _emit(progress_callback, 0.45, f"Starting denoise (CFG: {'on' if do_cfg else 'off'})")

        def mapped(p, msg):
            _emit(progress_callback, 0.50 + 0.40 * p, msg)

        latents = self.denoise(latents, timesteps, cond, uncond, guidance_scale, guidance_rescale, image_embeds, mapped,
                               image=image)
        if return_latents:
            _emit(progress_callback, 1.0, "Returning latents")
            return latents
        if self.decode_fn is None and self.vae is None:
            raise RuntimeError("hunyuanvideo15: no decode_fn / VAE attached; pass return_latents=True")
        if self.decode_fn is None:
            import inspect
            if "use_light_vae" in inspect.signature(self.vae.enable_tiling).parameters:
                self.vae.enable_tiling(use_light_vae=use_light_vae)      # t2v.py:350 / i2v.py:396
            elif use_light_vae:
                raise TypeError(f"{type(self.vae).__name__}.enable_tiling has no light-VAE switch (use_light_vae=True)")
            else:
                self.vae.enable_tiling()
        _emit(progress_callback, 0.94, "Decoding latents to video with light VAE" if use_light_vae else "Decoding latents")
        video = self.decode_fn(latents) if self.decode_fn is not None else self.vae_decode(latents)
        _emit(progress_callback, 1.0, "Completed text-to-video pipeline")
        if output_type is not None:      # t2v.py: `self._tensor_to_frames(video)` — uint8 frames made on the GPU
            from .postprocess import tensor_to_frames
            return tensor_to_frames(video, output_type)
        return video



class HunyuanVideo15I2VEngine(HunyuanVideo15T2VEngine):
    """Image-to-video: `run(image=pixels [B|1, 3, H, W] in [-1, 1] (or first-frame latents [B|1, 32, 1, h, w]),
    image_embeds=SigLIP states [B|1, 729, 1152], ...)`; everything else as the text-to-video engine."""

    _passes_timestep_r = True

    def vae_encode(self, image: torch.Tensor, sample_mode: str = "mode", generator=None) -> torch.Tensor:
        """BaseEngine.vae_encode: tiling on, encode, posterior mode / sample, normalise in the VAE dtype."""
        if self.vae is None:
            raise RuntimeError("hunyuanvideo15 i2v: a VAE is needed to encode the first frame (or pass its latents)")
        x = image.to(self.device, self.vae.dtype)
        if x.dim() == 4:
            x = x.unsqueeze(2)
        self.vae.enable_tiling()
        post = self.vae.encode(x, return_dict=False)[0]
        if sample_mode not in ("mode", "sample"):
            raise ValueError(f"Invalid sample mode: {sample_mode}")
        lat = post.mode() if sample_mode == "mode" else post.sample(generator=generator)
        return self.vae.normalize_latents(lat.to(self.vae.dtype))

    def prepare_cond_latents_and_mask(self, latents, dtype, device, image=None):
        """i2v.py:20-58: condition = the image latents on latent frame 0 and zeros after it; mask = 1 on frame 0."""
        if image is None:
            raise ValueError("hunyuanvideo15 i2v: `image` (pixels or first-frame latents) is required")
        b, c, f, h, w = latents.shape
        lat = image if image.shape[1] == c else self.vae_encode(image)
        if lat.dim() == 4:
            lat = lat.unsqueeze(2)
        if lat.shape[-2:] != (h, w):
            raise ValueError(f"image latents {tuple(lat.shape[-2:])} do not match the video latents {(h, w)}")
        cond = torch.zeros(b, c, f, h, w, dtype=dtype, device=device)
        cond[:, :, 0] = lat[:, :, 0].to(device=device, dtype=dtype).expand(b, -1, -1, -1) if lat.shape[0] != b \
            else lat[:, :, 0].to(device=device, dtype=dtype)
        mask = torch.zeros(b, 1, f, h, w, dtype=dtype, device=device)
        mask[:, :, 0] = 1.0
        return cond, mask

    @torch.no_grad()
    def run(self, *args, image=None, image_embeds=None, **kwargs):
        """Same positional arguments as the text-to-video `run` (prompt embeddings first); `image` / `image_embeds` are
        keyword-only."""
        if image is None:
            raise ValueError("hunyuanvideo15 i2v: `image` is required")
        kwargs.setdefault("guidance_scale", 1.0)          # i2v.py:103: CFG off unless asked for
        return super().run(*args, image=image, image_embeds=image_embeds, **kwargs)
