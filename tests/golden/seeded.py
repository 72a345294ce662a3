"""Seeded input / weight generators shared by make_golden.py and the tests (data, not reference code)."""
import torch


def seeded(shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def synthetic_state_dict(model, seed: int = 7):
    """Deterministic, bf16-representable weights keyed by parameter name (sorted order)."""
    sd = {}
    g = torch.Generator().manual_seed(seed)
    for name, p in sorted(model.state_dict().items()):
        if name.endswith(("norm_q.weight", "norm_k.weight", "norm_added_q.weight", "norm_added_k.weight")):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif name.endswith("scale_shift_table"):
            w = torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5)
        elif name.endswith(("norm_q.weight", "norm_k.weight", "norm2.weight", "txt_norm.weight")) or ".norm_" in name:
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif name.endswith(".bias"):
            w = 0.02 * torch.randn(p.shape, generator=g)
        else:
            w = torch.randn(p.shape, generator=g) * (1.0 / (p.shape[-1] ** 0.5))
        sd[name] = w.to(torch.bfloat16).to(torch.float32)
    return sd


def vae_synthetic_state_dict(model, seed=13):
    """conv weights ~ N(0, 1/fan_in), gammas ~ 1 +- 0.1, small biases (deterministic, sorted keys)."""
    sd = {}
    g = torch.Generator().manual_seed(seed)
    for name, p in sorted(model.state_dict().items()):
        if name.endswith("gamma"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif name.endswith(".bias"):
            w = 0.02 * torch.randn(p.shape, generator=g)
        else:
            fan_in = p[0].numel()
            w = torch.randn(p.shape, generator=g) / fan_in ** 0.5
        sd[name] = w.to(torch.bfloat16).to(torch.float32)
    return sd


def text_encoder_state_dict(model, seed: int, norm_seed: int, norm_tag: str):
    """synthetic_state_dict with the (RMS / Layer) norm weights drawn around 1 — what make_golden.gen_text_encoders used."""
    sd = synthetic_state_dict(model, seed)
    for k in [k for k in sd if norm_tag in k and k.endswith("weight")]:
        sd[k] = 1.0 + 0.1 * seeded(sd[k].shape, norm_seed + len(k)).to(torch.bfloat16).float()
    # T5 attention has no 1/sqrt(d_kv): trained checkpoints carry that factor in q.  Without it random weights give
    # near one-hot softmaxes whose argmax flips under any rounding, and the comparison measures chaos, not kernels.
    for k in [k for k in sd if k.endswith("SelfAttention.q.weight")]:
        sd[k] = sd[k] * 0.125
    for k in [k for k in sd if k.endswith("embed_tokens.weight") or k.endswith("patch_embed.proj.weight")]:
        sd[k] = sd[k] * 16.0           # O(1) embeddings, as trained ones are (the synthetic draw is 1/sqrt(fan_in))
    return sd


def spec_tensors(spec, seed0):
    """key -> seeded tensor (seed = seed0 + index in sorted key order); 0-d entries become scalars."""
    return {k: (seeded(tuple(sh), seed0 + i) if len(sh) else torch.tensor(float(1 + (seed0 + i) % 7)))
            for i, (k, sh) in enumerate(sorted(spec.items()))}


def tensor_digest(t):
    """(shape, dtype, sha1 of the contiguous bytes): the converters only move, slice and re-join tensors, and the LoRA alpha
    folding multiplies by powers of two, so equality is exact and a digest pins it as well as the values would."""
    import hashlib
    t = t.detach().contiguous()
    return tuple(t.shape), str(t.dtype), hashlib.sha1(t.reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b"").hexdigest()
