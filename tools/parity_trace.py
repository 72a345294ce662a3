"""Stage-by-stage parity trace of a tiny Flux forward: HIP model vs the bf16-storage oracle (test tooling, GPU box).

Every `Policy.r` call of the oracle is a storage point, i.e. one kernel output of the HIP path.  The oracle's roundings
are recorded in call order, the HIP workspace is snapshotted after every op, and the pairs are compared, so the first
stage whose rounding points differ shows up as a jump from ~1e-5 to ~1e-3.
Usage: python tools/parity_trace.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402
from apex_studio_amd.flux import FluxTransformer2DModel  # noqa: E402
from oracle import flux as OF  # noqa: E402
from oracle import layers as OL  # noqa: E402
from tests.golden.seeded import seeded, synthetic_state_dict  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    H = 2
    cfg = dict(patch_size=1, in_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128,
               num_attention_heads=H, joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True,
               axes_dims_rope=(16, 56, 56))
    dim = H * 128
    s_txt, hw = 16, (8, 8)
    s_img = hw[0] * hw[1]
    orc = OF.FluxTransformer2DModel(**cfg).eval()
    sd = synthetic_state_dict(orc, 7)
    orc.load_state_dict(sd)
    x = seeded((1, s_img, 64), 31).to(torch.bfloat16)
    enc = seeded((1, s_txt, 128), 32).to(torch.bfloat16)
    pooled = seeded((1, 64), 33).to(torch.bfloat16)
    t, g = torch.tensor([0.5]), torch.tensor([4.0])
    img_ids, txt_ids = OF.latent_image_ids(*hw), torch.zeros(s_txt, 3)

    # ---- oracle trace: every storage rounding, in call order
    otrace = []

    class TracePolicy(OL.Policy):
        def r(self, v):
            out = v.to(torch.bfloat16).to(torch.float32)
            otrace.append(out.clone())
            return out

    ref = orc(x.float(), enc.float(), pooled.float(), t, img_ids, txt_ids, g, policy=TracePolicy(True))
    print("oracle storage points:", len(otrace))

    # ---- HIP trace: snapshot the workspace after every op
    m = FluxTransformer2DModel(**cfg, device=DEV, dtype=torch.bfloat16)
    m.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()})
    gtrace = []
    names = ["gemm", "gemm_grouped", "ln_modulate", "qkv_prepare", "attention_prepared", "gemv"]
    orig = {n: getattr(ops, n) for n in names}

    def wrap(n):
        def f(*a, **k):
            out = orig[n](*a, **k)
            torch.cuda.synchronize()
            ws = next(iter(m._ws.values())) if m._ws else None
            snap = {} if ws is None else {k2: getattr(ws, k2).clone() for k2 in ("X", "XN", "QKV", "Q", "K", "CAT", "FFH", "MOD", "TEMB")}
            gtrace.append((n, snap))
            return out
        return f

    for n in names:
        setattr(ops, n, wrap(n))
    d = DEV
    out = m(hidden_states=x.to(d), encoder_hidden_states=enc.to(d), pooled_projections=pooled.to(d), timestep=t.to(d),
            img_ids=img_ids.to(d), txt_ids=txt_ids.to(d), guidance=g.to(d), return_dict=False)[0]
    torch.cuda.synchronize()
    for n in names:
        setattr(ops, n, orig[n])
    print("hip ops:", [n for n, _ in gtrace])

    # conditioning vector and modulation (f32 on both sides)
    temb = orc.time_text_embed((t.to(torch.bfloat16) * 1000).float(), (g.to(torch.bfloat16) * 1000).float(), pooled.float())
    last = gtrace[-1][1]
    print(f"temb (f32)            rel {rel(last['TEMB'][0], temb[0]):.2e}")
    blk = orc.transformer_blocks[0]
    mod_i = blk.norm1.linear(torch.nn.functional.silu(temb))[0]
    off = m._mod_off[("d", 0, "img")]
    print(f"MOD img block0 (f32)  rel {rel(last['MOD'][0, off:off + 6 * dim], mod_i):.2e}")

    ops_only = [(n, s) for n, s in gtrace if n != "gemv"]
    it = iter(ops_only)

    def nxt(expect):
        n, s = next(it)
        assert n == expect, (n, expect)
        return s

    o = otrace
    T = s_txt

    def heads(v):           # oracle [1, S, H, 128] -> [H, S, 128]
        return v[0].permute(1, 0, 2)

    def show(label, got, want):
        print(f"{label:34s} rel {rel(got, want):.2e}")

    s = nxt("gemm"); show("x_embedder", s["X"][T:], o[0][0])
    s = nxt("gemm"); show("context_embedder", s["X"][:T], o[1][0])
    # double block: oracle order inside block.forward: r(nx)=2, r(nc)=3, q,k,v=4,5,6, cq,ck,cv=7,8,9, q,k rope=10,11, o=12,
    # x=13, n2=14, ffh=15, x=16, ctx=17, c2=18, ffch=19, ctx=20
    s = nxt("ln_modulate"); show("D ln1 img", s["XN"][T:], o[2][0]); show("D ln1 txt", s["XN"][:T], o[3][0])
    s = nxt("gemm_grouped")
    show("D q img", s["QKV"][T:, :dim], o[4][0].flatten(1)); show("D k img", s["QKV"][T:, dim:2 * dim], o[5][0].flatten(1))
    show("D v img", s["QKV"][T:, 2 * dim:], o[6][0].flatten(1)); show("D q txt", s["QKV"][:T, :dim], o[7][0].flatten(1))
    show("D k txt", s["QKV"][:T, dim:2 * dim], o[8][0].flatten(1)); show("D v txt", s["QKV"][:T, 2 * dim:], o[9][0].flatten(1))
    s = nxt("qkv_prepare"); show("D q norm+rope", s["Q"][0], heads(o[10])); show("D k norm+rope", s["K"][0], heads(o[11]))
    s = nxt("attention_prepared"); show("D attention out", s["CAT"][:, :dim], o[12][0])
    s = nxt("gemm_grouped"); show("D x after attn", s["X"][T:], o[13][0])
    x_txt_attn = s["X"][:T].clone()
    s = nxt("ln_modulate"); show("D ln2 img", s["XN"][T:], o[14][0])
    ln2_txt = s["XN"][:T].clone()
    s = nxt("gemm_grouped"); show("D ff hidden img", s["FFH"][T:], o[15][0])
    ffh_txt = s["FFH"][:T].clone()
    s = nxt("gemm_grouped"); show("D x after ff", s["X"][T:], o[16][0])
    show("D ctx after attn", x_txt_attn, o[17][0]); show("D ln2 txt", ln2_txt, o[18][0]); show("D ff hidden txt", ffh_txt, o[19][0])
    show("D ctx after ff", s["X"][:T], o[20][0])
    # single block: nh=21, mlp=22, q,k,v=23,24,25, q,k rope=26,27, o=28, h=29
    s = nxt("ln_modulate"); show("S ln", s["XN"], o[21][0])
    s = nxt("gemm_grouped"); show("S mlp gelu", s["CAT"][:, dim:], o[22][0]); show("S q", s["QKV"][:, :dim], o[23][0].flatten(1))
    show("S k", s["QKV"][:, dim:2 * dim], o[24][0].flatten(1)); show("S v", s["QKV"][:, 2 * dim:], o[25][0].flatten(1))
    s = nxt("qkv_prepare"); show("S q norm+rope", s["Q"][0], heads(o[26])); show("S k norm+rope", s["K"][0], heads(o[27]))
    s = nxt("attention_prepared"); show("S attention out", s["CAT"][:, :dim], o[28][0])
    s = nxt("gemm"); show("S h after proj_out", s["X"], o[29][0])
    s = nxt("ln_modulate"); show("norm_out", s["XN"][T:], o[30][0])
    show("proj_out (model output)", out[0], o[31][0])
    print(f"final vs oracle: {rel(out, ref):.3e}")


if __name__ == "__main__":
    main()
