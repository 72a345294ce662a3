"""CPU check of the teacher-forcing plans (tests/stage_parity.py): every storage point the oracle's forward records is
claimed by exactly one stage of the plan, with the shape the HIP workspace view will have (no GPU needed for that)."""
import torch

from oracle import flux as OF
from oracle import qwenimage as OQ
from oracle import wan as OW
from tests import stage_parity as SP
from tests.golden.seeded import seeded, synthetic_state_dict


def _count(plan):
    return sum(len(p) for _, p in plan)


def test_flux_plan_covers_the_oracle_trace():
    cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128,
               num_attention_heads=2, joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True,
               axes_dims_rope=(16, 56, 56))
    orc = OF.FluxTransformer2DModel(**cfg).eval()
    orc.load_state_dict(synthetic_state_dict(orc, 7))
    pol = SP.TracePolicy()
    out = orc(seeded((1, 64, 64), 1), seeded((1, 16, 128), 2), seeded((1, 64), 3), torch.tensor([0.5]),
              OF.latent_image_ids(8, 8), torch.zeros(16, 3), torch.tensor([4.0]), policy=pol)
    plan, po = SP.flux_plan(pol.points, cfg, 16)
    assert _count(plan) + 1 == len(pol.points) and torch.equal(po, out)
    assert [op for op, _ in plan].count("attention_prepared") == 4


def test_wan_plan_covers_the_oracle_trace():
    cfg = dict(patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=128, in_channels=16, out_channels=16,
               text_dim=64, freq_dim=256, ffn_dim=512, num_layers=2, cross_attn_norm=True, eps=1e-6)
    orc = OW.WanTransformer3DModel(**cfg).eval()
    orc.load_state_dict(synthetic_state_dict(orc, 9))
    pol = SP.TracePolicy()
    orc(seeded((1, 16, 3, 8, 12), 41), torch.tensor([500.0]), seeded((1, 20, 64), 42), policy=pol)
    plan, po = SP.wan_plan(pol.points, cfg)
    assert _count(plan) + 1 == len(pol.points)
    S = 3 * 4 * 6
    shapes = {lbl.split(" ", 1)[1]: tuple(ref.shape) for _, ps in plan for lbl, _, ref in ps if lbl.startswith("b0 ")}
    assert shapes["q rope"] == (2, S, 128) and shapes["cross k heads"] == (2, 20, 128) and shapes["cross v"] == (20, 256)


def test_qwen_plan_covers_the_oracle_trace():
    cfg = dict(patch_size=2, in_channels=64, out_channels=16, num_layers=2, attention_head_dim=128,
               num_attention_heads=2, joint_attention_dim=64, axes_dims_rope=(16, 56, 56))
    shapes = [(1, 6, 8), (1, 4, 6)]
    orc = OQ.QwenImageTransformer2DModel(**cfg).eval()
    orc.load_state_dict(synthetic_state_dict(orc, 11))
    pol = SP.TracePolicy()
    out = orc(seeded((1, 72, 64), 51), seeded((1, 13, 64), 52), torch.tensor([0.5]), shapes, policy=pol)
    plan, po = SP.qwen_plan(pol.points, cfg, 13)
    assert _count(plan) + 1 == len(pol.points) and torch.equal(po, out)
