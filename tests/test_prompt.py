"""apex_studio_amd.prompt.TextEncoder / qwen_prompt_embeds: the host-side semantics of the reference's wrapper
(`R/src/text_encoder/text_encoder.py:183-424`, `R/src/engine/qwenimage/shared.py:100-282`) with stand-in models on the CPU —
masking, `pad_with_zero`, pooled outputs, per-prompt repeats, template-token dropping — and that the engines' `run(prompt_ids=…)`
reaches the denoise loop with exactly those embeddings.  The HIP encoders behind it are tested in test_gpu_text*.py; the
ids -> uint8 frames chains on the GPU in tests/test_gpu_prompt_engines.py."""
from types import SimpleNamespace

import pytest
import torch

import apex_studio_amd  # noqa: F401
from apex_studio_amd.prompt import TextEncoder, qwen_prompt_embeds, split_ids


class _FakeEncoder(torch.nn.Module):
    """hidden[b, l] = id * (1 + 0.5 * [l attended]); pooled = hidden of the last token"""

    def __init__(self, dim=4):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(dim))
        self.seen = []

    def forward(self, input_ids, attention_mask=None, output_hidden_states=False, **kw):
        self.seen.append(dict(mask=None if attention_mask is None else attention_mask.clone(), extra=sorted(kw)))
        h = input_ids.float()[..., None] * self.w
        if attention_mask is not None:
            h = h * (1 + 0.5 * attention_mask.float()[..., None])
        return SimpleNamespace(last_hidden_state=h, hidden_states=(h * 0, h * 2, h), pooler_output=h[:, -1])


def test_encode_hidden_states_masking_and_zero_padding():
    m = _FakeEncoder()
    te = TextEncoder(m)
    ids = torch.tensor([[5, 6, 7, 0, 0], [1, 2, 3, 4, 9]])
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]])
    # Wan: masked encoder, embeddings past the true length are exact zeros, padded back to max_sequence_length
    emb, mk = te.encode(input_ids=ids, attention_mask=mask, max_sequence_length=5, use_attention_mask=True, return_attention_mask=True)
    assert emb.shape == (2, 5, 4) and torch.equal(mk, mask)
    assert torch.equal(m.seen[-1]["mask"], mask)
    assert torch.equal(emb[0, :3, 0], torch.tensor([7.5, 9.0, 10.5])) and float(emb[0, 3:].abs().sum()) == 0.0
    assert torch.equal(emb[1, :, 0], torch.tensor([1.5, 3.0, 4.5, 6.0, 13.5]))
    # Flux T5: no mask into the model, nothing zeroed
    emb = te.encode(input_ids=ids, attention_mask=mask, max_sequence_length=5, pad_with_zero=False)
    assert m.seen[-1]["mask"] is None and torch.equal(emb[0, :, 0], torch.tensor([5.0, 6, 7, 0, 0]))
    # an explicit layer (CLIP-skip style) and per-prompt repeats
    emb = te.encode(input_ids=ids, max_sequence_length=5, pad_with_zero=False, hidden_states_idx=1, num_videos_per_prompt=2)
    assert emb.shape == (4, 5, 4) and torch.equal(emb[0], emb[1]) and torch.equal(emb[0, :, 0], torch.tensor([10.0, 12, 14, 0, 0]))


def test_encode_pooled_and_argument_errors():
    te = TextEncoder(_FakeEncoder())
    pooled = te.encode(input_ids=torch.tensor([[3, 4, 8]]), max_sequence_length=3, output_type="pooler_output", num_videos_per_prompt=3)
    assert pooled.shape == (3, 4) and float(pooled[2, 0]) == 8.0
    with pytest.raises(RuntimeError, match="tokenizer"):
        te.encode("a prompt")
    with pytest.raises(ValueError, match="Invalid output type"):
        te.encode(input_ids=torch.tensor([[1]]), output_type="text_embeds")
    tok = lambda text, **kw: SimpleNamespace(input_ids=torch.tensor([[len(t), 1] for t in text]), attention_mask=torch.ones(len(text), 2, dtype=torch.long))  # noqa: E731
    assert TextEncoder(_FakeEncoder(), tokenizer=tok).encode(["ab", "abcd"], max_sequence_length=2).shape == (2, 2, 4)
    # the reference's default `clean_text=True` (text_encoder.py:117-131, 210-211): entities unescaped twice, whitespace collapsed
    from apex_studio_amd.prompt import prompt_clean
    assert prompt_clean("  a &amp;amp; b \n\t c  ") == "a & b c" and prompt_clean("A  B", lower_case=True) == "a b"
    te2 = TextEncoder(_FakeEncoder(), tokenizer=tok)
    assert float(te2.encode(["  ab   cd "], max_sequence_length=2, pad_with_zero=False)[0, 0, 0]) == 5.0          # "ab cd"
    assert float(te2.encode(["  ab   cd "], max_sequence_length=2, pad_with_zero=False, clean_text=False)[0, 0, 0]) == 10.0
    with pytest.raises(NotImplementedError):
        te.encode(input_ids=torch.tensor([[1]]), use_position_ids=True)
    ids, mask = split_ids({"input_ids": torch.ones(1, 2), "attention_mask": torch.zeros(1, 2)})
    assert mask is not None and split_ids(torch.ones(1, 2))[1] is None and split_ids((ids, mask))[1] is mask


def test_qwen_prompt_embeds_drops_the_template_and_pads():
    class VL(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.kw = None

        def forward(self, **kw):
            self.kw = kw
            h = kw["input_ids"].float()[..., None].repeat(1, 1, 3)
            return SimpleNamespace(hidden_states=(h * 0, h))
    m = VL()
    ids = torch.tensor([[0, 0, 10, 11, 12, 13], [20, 21, 22, 23, 24, 25]])
    mask = torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1]])           # the first prompt is left-padded
    emb, mk = qwen_prompt_embeds(m, ids, mask, pixel_values=torch.zeros(4, 8), image_grid_thw=torch.tensor([[1, 2, 2]]), drop_idx=2)
    assert set(m.kw) == {"input_ids", "attention_mask", "output_hidden_states", "pixel_values", "image_grid_thw"}
    assert emb.shape == (2, 4, 3) and mk.tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]]
    assert emb[0, :, 0].tolist() == [12.0, 13.0, 0.0, 0.0] and emb[1, :, 0].tolist() == [22.0, 23.0, 24.0, 25.0]
    emb, _ = qwen_prompt_embeds(m, ids, mask, drop_idx=2, max_sequence_length=3, num_images_per_prompt=2)
    assert emb.shape == (4, 3, 3) and "pixel_values" not in m.kw


def test_engines_encode_prompt_ids_before_the_loop():
    from apex_studio_amd.engine_flux import FluxT2IEngine
    from apex_studio_amd.engine_wan import WanT2VEngine
    from tests.test_engines import _FakeWan
    hi = _FakeWan(1.0)
    seen = {}
    orig = hi.__call__

    def spy(hidden_states, timestep, encoder_hidden_states, return_dict=False):
        seen.setdefault("enc", []).append(encoder_hidden_states.clone())
        return orig(hidden_states, timestep, encoder_hidden_states, return_dict)
    hi.__class__ = type("Spy", (_FakeWan,), {"__call__": staticmethod(spy)})
    eng = WanT2VEngine(hi, hi, vae=None, text_encoder=_FakeEncoder(dim=8))
    ids, mask = torch.tensor([[4, 5, 6, 0]]), torch.tensor([[1, 1, 0, 0]])
    out = eng.run(prompt_ids=(ids, mask), negative_prompt_ids=(ids * 0 + 1, mask), height=64, width=64, duration=5,
                  num_inference_steps=2, seed=0, generator=torch.Generator().manual_seed(0), return_latents=True,
                  text_encoder_kwargs=dict(max_sequence_length=4))
    assert out.shape == (1, 16, 2, 8, 8) and len(seen["enc"]) == 4                  # cond + uncond per step
    want = eng.encode_prompt(prompt_ids=(ids, mask), text_encoder_kwargs=dict(max_sequence_length=4))
    assert want.shape == (1, 4, 8) and float(want[0, 2:].abs().sum()) == 0.0 and float(want[0, 0, 0]) == 6.0
    assert torch.equal(seen["enc"][0].float(), want.to(seen["enc"][0].dtype).float())
    with pytest.raises(RuntimeError, match="text_encoder"):
        WanT2VEngine(hi, hi).run(prompt_ids=ids, height=64, width=64, duration=5, num_inference_steps=1, return_latents=True)
    with pytest.raises(ValueError, match="per tokenizer"):
        FluxT2IEngine(SimpleNamespace(config=SimpleNamespace(in_channels=64), device=torch.device("cpu"), dtype=torch.float32),
                      text_encoder=_FakeEncoder(), text_encoder_2=_FakeEncoder()).encode_prompt(prompt_ids=ids)


def test_prompt_clean_entities_whitespace_and_the_ftfy_notice():
    """`prompt_clean` (R/src/text_encoder/text_encoder.py:117-131): HTML entities unescaped twice, whitespace runs collapsed,
    stripped; ftfy is absent in this image — ASCII prompts are unaffected by that, a non-ASCII prompt gets ONE warning that the
    reference would have repaired it before tokenising."""
    import warnings
    from apex_studio_amd import prompt as P
    assert P.prompt_clean("  a &amp;amp; b \n\t c&lt;d  ") == "a & b c<d"
    assert P.prompt_clean("Hello  World", lower_case=True) == "hello world"
    try:
        import ftfy  # noqa: F401
        have = True
    except ImportError:
        have = False
    P._ftfy_warned = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        P.prompt_clean("plain ascii prompt")
        assert not w
        out = P.prompt_clean("cafÃ© au lait")
        P.prompt_clean("naïve again")
    if have:
        assert out == "café au lait" and not w
    else:
        assert out == "cafÃ© au lait" and len(w) == 1 and "ftfy" in str(w[0].message)
