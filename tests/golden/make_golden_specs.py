"""Key layouts ({key: shape}) of the ORIGINAL-format weight files the manifests name, shared by make_golden.py (which
feeds them to the reference's converters) and the tests (which rebuild the same seeded inputs).  Shapes only."""


def wan_original_spec(dim=64, ffn=128, text_dim=32, freq=32, layers=2, in_ch=16):
    """{key: shape} of an ORIGINAL-format Wan 2.x t2v transformer file (Wan-AI/Wan2.2-T2V-A14B key layout)."""
    sp = {"patch_embedding.weight": (dim, in_ch, 1, 2, 2), "patch_embedding.bias": (dim,),
          "text_embedding.0.weight": (dim, text_dim), "text_embedding.0.bias": (dim,),
          "text_embedding.2.weight": (dim, dim), "text_embedding.2.bias": (dim,),
          "time_embedding.0.weight": (dim, freq), "time_embedding.0.bias": (dim,),
          "time_embedding.2.weight": (dim, dim), "time_embedding.2.bias": (dim,),
          "time_projection.1.weight": (6 * dim, dim), "time_projection.1.bias": (6 * dim,),
          "head.head.weight": (in_ch * 4, dim), "head.head.bias": (in_ch * 4,), "head.modulation": (1, 2, dim)}
    for i in range(layers):
        b = f"blocks.{i}."
        for a in ("self_attn", "cross_attn"):
            for n in ("q", "k", "v", "o"):
                sp[b + f"{a}.{n}.weight"] = (dim, dim)
                sp[b + f"{a}.{n}.bias"] = (dim,)
            sp[b + f"{a}.norm_q.weight"] = (dim,)
            sp[b + f"{a}.norm_k.weight"] = (dim,)
        sp[b + "norm3.weight"] = (dim,)
        sp[b + "norm3.bias"] = (dim,)
        sp[b + "ffn.0.weight"] = (ffn, dim)
        sp[b + "ffn.0.bias"] = (ffn,)
        sp[b + "ffn.2.weight"] = (dim, ffn)
        sp[b + "ffn.2.bias"] = (dim,)
        sp[b + "modulation"] = (1, 6, dim)
    return sp


def flux_original_spec(dim=128, mlp=4, layers=2, single=2, in_ch=64, txt=96, pooled=48, guidance=True):
    """{key: shape} of a BFL-format Flux transformer file (black-forest-labs/FLUX.1-dev `flux1-dev.safetensors` key layout)."""
    sp = {"img_in.weight": (dim, in_ch), "img_in.bias": (dim,), "txt_in.weight": (dim, txt), "txt_in.bias": (dim,),
          "time_in.in_layer.weight": (dim, 256), "time_in.in_layer.bias": (dim,),
          "time_in.out_layer.weight": (dim, dim), "time_in.out_layer.bias": (dim,),
          "vector_in.in_layer.weight": (dim, pooled), "vector_in.in_layer.bias": (dim,),
          "vector_in.out_layer.weight": (dim, dim), "vector_in.out_layer.bias": (dim,),
          "final_layer.linear.weight": (in_ch, dim), "final_layer.linear.bias": (in_ch,),
          "final_layer.adaLN_modulation.1.weight": (2 * dim, dim), "final_layer.adaLN_modulation.1.bias": (2 * dim,)}
    if guidance:
        sp.update({"guidance_in.in_layer.weight": (dim, 256), "guidance_in.in_layer.bias": (dim,),
                   "guidance_in.out_layer.weight": (dim, dim), "guidance_in.out_layer.bias": (dim,)})
    for i in range(layers):
        b = f"double_blocks.{i}."
        for s_ in ("img", "txt"):
            sp[b + f"{s_}_mod.lin.weight"] = (6 * dim, dim)
            sp[b + f"{s_}_mod.lin.bias"] = (6 * dim,)
            sp[b + f"{s_}_attn.qkv.weight"] = (3 * dim, dim)
            sp[b + f"{s_}_attn.qkv.bias"] = (3 * dim,)
            sp[b + f"{s_}_attn.norm.query_norm.scale"] = (128,)
            sp[b + f"{s_}_attn.norm.key_norm.scale"] = (128,)
            sp[b + f"{s_}_attn.proj.weight"] = (dim, dim)
            sp[b + f"{s_}_attn.proj.bias"] = (dim,)
            sp[b + f"{s_}_mlp.0.weight"] = (mlp * dim, dim)
            sp[b + f"{s_}_mlp.0.bias"] = (mlp * dim,)
            sp[b + f"{s_}_mlp.2.weight"] = (dim, mlp * dim)
            sp[b + f"{s_}_mlp.2.bias"] = (dim,)
    for i in range(single):
        b = f"single_blocks.{i}."
        sp[b + "linear1.weight"] = ((3 + mlp) * dim, dim)
        sp[b + "linear1.bias"] = ((3 + mlp) * dim,)
        sp[b + "linear2.weight"] = (dim, (1 + mlp) * dim)
        sp[b + "linear2.bias"] = (dim,)
        sp[b + "norm.query_norm.scale"] = (128,)
        sp[b + "norm.key_norm.scale"] = (128,)
        sp[b + "modulation.lin.weight"] = (3 * dim, dim)
        sp[b + "modulation.lin.bias"] = (3 * dim,)
    return sp
