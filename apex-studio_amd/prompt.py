"""Prompt encoding for the engines: token ids (or strings, when a tokenizer object is supplied) -> the embeddings the
denoise loop consumes, on the HIP text-encoder classes (text_encoders.py, qwen2_5_vl.py).

Mirrors the host-side wrapper the reference's engines call, `TextEncoder.encode` (`R/src/text_encoder/text_encoder.py:183-424`):
tokenise -> model(input_ids [, attention_mask]) -> pick `last_hidden_state` / `hidden_states[i]` / `pooler_output` ->
(`pad_with_zero`) cut every sequence at its true length and zero-fill back to `max_sequence_length` -> repeat per
`num_videos_per_prompt`, and the QwenImage variant `_get_qwen_prompt_embeds` (`R/src/engine/qwenimage/shared.py:100-282`:
`hidden_states[-1]` of a forward over text (+ pixels), masked tokens extracted per sample, the first `drop_idx` template
tokens dropped, zero-padded to the longest).  Per-family arguments as the manifests set them:

    Flux   CLIP-L  max_sequence_length 77,  pad_with_zero false, pooler_output          (manifest/image/flux-dev-…yml:134-141)
           T5-XXL  max_sequence_length 512, pad_with_zero false, hidden_states, no attention mask
    Wan    UMT5    use_attention_mask true, pad_with_zero (default), 512 tokens         (manifest/video/wan-2.2-a14b-…yml:297-298)

Tokenisers and the Qwen image processor stay `transformers` objects on the CPU: `text=` needs one to be
supplied; `input_ids=` / `attention_mask=` take already tokenised prompts (there are no vocabulary files in the build
container, so the GPU tests feed ids)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union

import torch

Ids = Union[torch.Tensor, Dict[str, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]


def split_ids(ids: Ids) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """`prompt_ids` argument of the engines: a [B, L] id tensor, (ids, attention_mask), or a tokenizer-output-like dict."""
    if isinstance(ids, dict) or hasattr(ids, "input_ids"):
        get = ids.get if isinstance(ids, dict) else (lambda k, d=None: getattr(ids, k, d))
        return get("input_ids"), get("attention_mask", None)
    if isinstance(ids, (tuple, list)):
        return ids[0], ids[1]
    return ids, None


_ftfy_warned = False


def prompt_clean(text: str, lower_case: bool = False) -> str:
    """`TextEncoder.prompt_clean` (R/src/text_encoder/text_encoder.py:117-131): ftfy repair (when the package is present — it is
    a pure-Python text fixer with no effect on clean ASCII), HTML entities unescaped twice, whitespace runs collapsed, stripped."""
    import html
    import re
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        # ftfy only ever changes non-ASCII text (mojibake, odd quotes, full-width forms): there the reference would have tokenised a
        # repaired string, so say so once instead of silently diverging; clean ASCII prompts are untouched either way
        global _ftfy_warned
        if not _ftfy_warned and not text.isascii():
            import warnings
            warnings.warn("apex_studio_amd.prompt: the `ftfy` package is not installed; the reference repairs non-ASCII prompt text "
                          "with ftfy.fix_text before tokenising (R/src/text_encoder/text_encoder.py:117-131) — this prompt contains "
                          "non-ASCII characters and is tokenised unrepaired", RuntimeWarning, stacklevel=2)
            _ftfy_warned = True
    text = html.unescape(html.unescape(text)).strip()
    text = re.sub(r"\s+", " ", text).strip()
    return text.lower() if lower_case else text


class TextEncoder:
    """`TextEncoder(model, tokenizer=None).encode(...)`: argument names and semantics of the reference's wrapper."""

    def __init__(self, model, tokenizer=None):
        self.model = model
        self.tokenizer = tokenizer

    @property
    def device(self):
        return next(self.model.parameters()).device

    @torch.no_grad()
    def encode(self, text: Union[str, List[str], None] = None, *, input_ids: Optional[torch.Tensor] = None,
               attention_mask: Optional[torch.Tensor] = None, max_sequence_length: int = 512, pad_to_max_length: bool = True,
               num_videos_per_prompt: int = 1, dtype: Optional[torch.dtype] = None, device=None,
               add_special_tokens: Optional[bool] = True, return_attention_mask: bool = False, use_attention_mask: bool = False,
               pad_with_zero: bool = True, output_type: str = "hidden_states", hidden_states_idx: int = -1,
               reshape_prompt_embeds: bool = True, clean_text: bool = True, lower_case: bool = False,
               use_position_ids: bool = False, use_token_type_ids: bool = False, arrange_attention_mask: bool = False):
        if use_position_ids or use_token_type_ids or arrange_attention_mask:
            raise NotImplementedError("TextEncoder.encode: use_position_ids / use_token_type_ids / arrange_attention_mask are not "
                                      "implemented (no encoder on the hot path's manifests sets them)")
        if input_ids is None:
            if text is None:
                raise ValueError("encode() needs `text` (with a tokenizer) or `input_ids`")
            if self.tokenizer is None:
                raise RuntimeError("encode(text=...) needs a tokenizer: pass one to TextEncoder(model, tokenizer), or pass "
                                   "input_ids / attention_mask (tokenisers stay `transformers` objects on the CPU)")
            text = [text] if isinstance(text, str) else list(text)
            if clean_text:                                  # the reference's default (text_encoder.py:210-211)
                text = [prompt_clean(t, lower_case=lower_case) for t in text]
            kw = dict(padding="max_length" if pad_to_max_length else "longest", max_length=max_sequence_length, truncation=True,
                      return_tensors="pt", return_attention_mask=True)
            if add_special_tokens is not None:
                kw["add_special_tokens"] = add_special_tokens
            tok = self.tokenizer(text, **kw)
            input_ids, attention_mask = tok.input_ids, tok.attention_mask
        if input_ids.dim() == 1:
            input_ids = input_ids.unsqueeze(0)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        elif attention_mask.dim() == 1:
            attention_mask = attention_mask.unsqueeze(0)
        batch = input_ids.shape[0]
        dev = self.device
        mask = attention_mask.to("cpu")
        seq_lens = mask.gt(0).sum(dim=1).long()
        inputs: Dict[str, Any] = {"input_ids": input_ids.to(dev)}
        if use_attention_mask:
            inputs["attention_mask"] = attention_mask.to(dev)
        want_hidden = output_type in ("hidden_states", "raw", "hidden_states_all")
        result = self.model(**inputs, output_hidden_states=want_hidden)
        if output_type == "raw":
            return (result, mask) if return_attention_mask else result
        if output_type == "hidden_states":
            if getattr(result, "hidden_states", None) is not None and (hidden_states_idx != -1 or
                                                                        getattr(result, "last_hidden_state", None) is None):
                emb = result.hidden_states[hidden_states_idx]
            else:
                emb = result.last_hidden_state
        elif output_type == "pooler_output":
            emb = result.pooler_output
        else:
            raise ValueError(f"Invalid output type: {output_type}")
        emb = emb.to(dtype=dtype or emb.dtype, device=device or emb.device)
        if output_type == "pooler_output":
            emb = emb.repeat(1, num_videos_per_prompt).view(batch * num_videos_per_prompt, -1)
        else:
            if pad_with_zero:       # tokens past a prompt's true length carry no information: exact zeros, not encoder outputs
                rows = [u[:int(v)] for u, v in zip(emb, seq_lens)]
                emb = torch.stack([torch.cat([u, u.new_zeros(max_sequence_length - u.size(0), u.size(1))]) if pad_to_max_length
                                   else u for u in rows], dim=0)
            if reshape_prompt_embeds:
                L = emb.shape[1]
                emb = emb.repeat(1, num_videos_per_prompt, 1).view(batch * num_videos_per_prompt, L, -1)
                mask = mask.repeat(1, num_videos_per_prompt).view(batch * num_videos_per_prompt, -1)
        return (emb, mask) if return_attention_mask else emb


@torch.no_grad()
def qwen_prompt_embeds(model, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                       pixel_values: Optional[torch.Tensor] = None, image_grid_thw: Optional[torch.Tensor] = None,
                       drop_idx: int = 64, num_images_per_prompt: int = 1, max_sequence_length: Optional[int] = None,
                       dtype: Optional[torch.dtype] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`_get_qwen_prompt_embeds` (R/src/engine/qwenimage/shared.py:100-282) from processor outputs: last hidden state of the
    Qwen2.5-VL forward, the attended tokens of every sample, minus the `drop_idx` tokens of the chat template (34 for
    text-to-image, 64 for the edit templates), zero-padded to the longest.  Returns (prompt_embeds, prompt_embeds_mask)."""
    dev = next(model.parameters()).device
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    kw: Dict[str, Any] = dict(input_ids=input_ids.to(dev), attention_mask=attention_mask.to(dev), output_hidden_states=True)
    if pixel_values is not None:
        kw.update(pixel_values=pixel_values.to(dev), image_grid_thw=image_grid_thw)
    hidden = model(**kw).hidden_states[-1]
    keep = attention_mask.to(hidden.device).bool()
    parts = torch.split(hidden[keep], keep.sum(dim=1).tolist(), dim=0)
    parts = [e[drop_idx:] for e in parts]
    longest = max(e.size(0) for e in parts)
    embeds = torch.stack([torch.cat([u, u.new_zeros(longest - u.size(0), u.size(1))]) for u in parts])
    masks = torch.stack([torch.cat([torch.ones(e.size(0), dtype=torch.long, device=e.device),
                                    torch.zeros(longest - e.size(0), dtype=torch.long, device=e.device)]) for e in parts])
    if dtype is not None:
        embeds = embeds.to(dtype)
    if pixel_values is None and max_sequence_length is not None:
        embeds, masks = embeds[:, :max_sequence_length], masks[:, :max_sequence_length]
    B, L, _ = embeds.shape
    embeds = embeds.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, L, -1)
    masks = masks.repeat(1, num_images_per_prompt).view(B * num_images_per_prompt, L)
    return embeds, masks
