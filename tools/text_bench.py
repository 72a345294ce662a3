#!/usr/bin/env python
"""Time the text encoders at their production sizes (random weights): T5-XXL 512 tokens (Flux), UMT5-XXL 512 tokens
(Wan), CLIP-L 77 tokens (Flux pooled prompt).  usage: text_bench.py [reps]"""
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
