import json, sys, os, torch
sys.path.insert(0, "/root/repo")
import apex_studio_amd
from apex_studio_amd import lib, ops
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
def tm(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
cases = {
    "wan N96 3x3x3": (96, 21, 256, 256, (3, 3, 3), False, False, False),
    "wan N96 3x3x3 fused norm + res": (96, 21, 256, 256, (3, 3, 3), True, True, False),
    "wan conv_out 96->3": (3, 21, 256, 256, (3, 3, 3), False, False, False),
    "ragged N96 norm": (96, 3, 131, 173, (3, 3, 3), True, False, False),
    "ragged N96 res": (96, 5, 135, 170, (3, 3, 3), False, True, False),
    "1x3x3 N96": (96, 9, 100, 90, (1, 3, 3), False, False, False),
    "kT=2 N64out": (64, 7, 120, 101, (2, 3, 3), False, True, False),
    "independent frames": (96, 12, 96, 96, (3, 3, 3), False, False, True),
    "N32 out": (32, 4, 200, 180, (3, 3, 3), False, False, False),
}
for name, (cout, T, H, W, k, norm, res, indep) in cases.items():
    cin = 96
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=DEV) * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], device=DEV, dtype=torch.bfloat16); b[:cout] = 0.1
    gam = (1 + 0.1 * torch.randn(wp.shape[0], generator=g, device=DEV)).to(torch.bfloat16)
    r = torch.randn(T, H, W, wp.shape[0], generator=g, device=DEV).to(torch.bfloat16) if res else None
    def run():
        if norm:
            return ops.conv3d_cl_norm(x, wp, b, k, gam, silu=True, residual=r, independent_frames=indep)
        return ops.conv3d_cl(x, wp, b, k, residual=r, independent_frames=indep)
    ms, outs = {}, {}
    for v in (0, 1):
        lib.tune_set("conv.slab", v)
        o = run()
        outs[v] = o if not norm else torch.cat([o[0].flatten(), o[1].flatten()])
        ms[v] = round(tm(run), 3)
    same = bool(torch.equal(outs[0], outs[1]))
    nd = int((outs[0] != outs[1]).sum())
    fl = 2.0 * T * H * W * cout * cin * k[0] * 9
    print(json.dumps({"case": name, "ms_v2": ms[0], "ms_slab": ms[1], "TF_slab": round(fl / ms[1] / 1e9, 1), "bit_identical": same, "differ": nd,
                      "maxdiff": float((outs[0].float() - outs[1].float()).abs().max())}), flush=True)
lib.tune_set("conv.slab", 1)
