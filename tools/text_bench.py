#!/usr/bin/env python
"""Time the text encoders at their production sizes (random weights): T5-XXL 512 tokens (Flux), UMT5-XXL 512 tokens
(Wan), CLIP-L 77 tokens (Flux pooled prompt), Qwen2.5-VL-7B with a 250-token prompt and two 392x392 images
(QwenImage-Edit-2509).  usage: text_bench.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import text_encoders as TE  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)


def init(m, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    for n, p in m.named_parameters():
        if "layer_norm" in n and n.endswith("weight"):
            p.data.fill_(1.0)
        elif "norm" in n or n.endswith("ln_q.weight"):
            p.data.fill_(1.0)
        elif n.endswith("bias"):
            p.data.zero_()
        else:
            p.data.copy_((torch.randn(p.shape, generator=g, device=dev) * (0.125 if n.endswith(".q.weight") else 1.0)
                          / p.shape[-1] ** 0.5).to(p.dtype))
    return m


def flops_t5(c, S):
    inner = c.num_heads * c.d_kv
    return c.num_layers * (2 * S * (4 * c.d_model * inner + 3 * c.d_model * c.d_ff) + 4 * S * S * inner)


cases = [("t5-xxl (flux)", TE.T5EncoderModel, dict(), 512),
         ("umt5-xxl (wan)", TE.UMT5EncoderModel, dict(), 512),
         ("clip-l (flux)", TE.CLIPTextModel, dict(), 77)]
for name, cls, cfg, S in cases:
    m = init(cls(cfg, device=dev, dtype=torch.bfloat16), 3)
    vocab = m.config.vocab_size
    ids = torch.randint(3, min(vocab, 30000), (1, S), device=dev)
    mask = torch.ones(1, S, dtype=torch.long, device=dev)
    mask[0, S * 2 // 3:] = 0
    m(input_ids=ids, attention_mask=mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = m(input_ids=ids, attention_mask=mask).last_hidden_state
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    extra = ""
    if hasattr(m.config, "d_ff"):
        extra = f"  {flops_t5(m.config, S) / dt / 1e12:.0f} TFLOP/s"
    print(f"{name}: {dt * 1e3:.2f} ms / encode of {S} tokens{extra}  out {tuple(out.shape)}")
    del m
    torch.cuda.empty_cache()

# Qwen2.5-VL-7B: 64 template tokens + text + two images of 28x28 patches (196 merged tokens each)
from apex_studio_amd.qwen2_5_vl import Qwen2_5_VLForConditionalGeneration  # noqa: E402
m = init(Qwen2_5_VLForConditionalGeneration({}, device=dev, dtype=torch.bfloat16), 4)
IMG = m.config.image_token_id
seq = list(range(100, 164)) + [IMG] * 196 + list(range(200, 210)) + [IMG] * 196 + list(range(300, 480))
ids = torch.tensor([seq], device=dev)
mask = torch.ones_like(ids)
grid = torch.tensor([[1, 28, 28], [1, 28, 28]])
pix = torch.randn(2 * 784, 1176, device=dev).to(torch.bfloat16)
m(input_ids=ids, attention_mask=mask, pixel_values=pix, image_grid_thw=grid, output_hidden_states=True)
torch.cuda.synchronize()
for what, kw in (("text + 2 images", dict(pixel_values=pix, image_grid_thw=grid)), ("vision tower only", None)):
    t0 = time.perf_counter()
    for _ in range(reps):
        if kw is None:
            out = m.get_image_features(pix, grid)
        else:
            out = m(input_ids=ids, attention_mask=mask, output_hidden_states=True, **kw).hidden_states[-1]
    torch.cuda.synchronize()
    print(f"qwen2.5-vl-7b ({what}): {(time.perf_counter() - t0) / reps * 1e3:.2f} ms for {ids.shape[1]} tokens  out {tuple(out.shape)}")
