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


def hunyuan15_original_spec(dim=128, txt=64, byt5=96, vis=48, layers=2, refiner=1, in_ch=65, out_ch=32):
    """{key: shape} of an ORIGINAL-format HunyuanVideo-1.5 transformer file (tencent/HunyuanVideo-1.5 key layout: fused
    `img_attn_qkv` / `txt_attn_qkv` / refiner `self_attn_qkv`, `*_mlp.fc1/fc2`, `*_mod.linear`)."""
    sp = {"img_in.proj.weight": (dim, in_ch, 1, 1, 1), "img_in.proj.bias": (dim,),
          "vision_in.proj.0.weight": (vis,), "vision_in.proj.0.bias": (vis,),
          "vision_in.proj.1.weight": (vis, vis), "vision_in.proj.1.bias": (vis,),
          "vision_in.proj.3.weight": (dim, vis), "vision_in.proj.3.bias": (dim,),
          "vision_in.proj.4.weight": (dim,), "vision_in.proj.4.bias": (dim,),
          "txt_in.t_embedder.mlp.0.weight": (dim, 256), "txt_in.t_embedder.mlp.0.bias": (dim,),
          "txt_in.t_embedder.mlp.2.weight": (dim, dim), "txt_in.t_embedder.mlp.2.bias": (dim,),
          "txt_in.c_embedder.linear_1.weight": (dim, txt), "txt_in.c_embedder.linear_1.bias": (dim,),
          "txt_in.c_embedder.linear_2.weight": (dim, dim), "txt_in.c_embedder.linear_2.bias": (dim,),
          "txt_in.input_embedder.weight": (dim, txt), "txt_in.input_embedder.bias": (dim,),
          "byt5_in.layernorm.weight": (byt5,), "byt5_in.layernorm.bias": (byt5,),
          "byt5_in.fc1.weight": (2048, byt5), "byt5_in.fc1.bias": (2048,),
          "byt5_in.fc2.weight": (2048, 2048), "byt5_in.fc2.bias": (2048,),
          "byt5_in.fc3.weight": (dim, 2048), "byt5_in.fc3.bias": (dim,),
          "time_in.mlp.0.weight": (dim, 256), "time_in.mlp.0.bias": (dim,),
          "time_in.mlp.2.weight": (dim, dim), "time_in.mlp.2.bias": (dim,),
          "cond_type_embedding.weight": (3, dim),
          "final_layer.adaLN_modulation.1.weight": (2 * dim, dim), "final_layer.adaLN_modulation.1.bias": (2 * dim,),
          "final_layer.linear.weight": (out_ch, dim), "final_layer.linear.bias": (out_ch,)}
    for i in range(refiner):
        b = f"txt_in.individual_token_refiner.blocks.{i}."
        sp.update({b + "norm1.weight": (dim,), b + "norm1.bias": (dim,), b + "norm2.weight": (dim,), b + "norm2.bias": (dim,),
                   b + "self_attn_qkv.weight": (3 * dim, dim), b + "self_attn_qkv.bias": (3 * dim,),
                   b + "self_attn_proj.weight": (dim, dim), b + "self_attn_proj.bias": (dim,),
                   b + "mlp.fc1.weight": (4 * dim, dim), b + "mlp.fc1.bias": (4 * dim,),
                   b + "mlp.fc2.weight": (dim, 4 * dim), b + "mlp.fc2.bias": (dim,),
                   b + "adaLN_modulation.1.weight": (2 * dim, dim), b + "adaLN_modulation.1.bias": (2 * dim,)})
    for i in range(layers):
        b = f"double_blocks.{i}."
        for s_ in ("img", "txt"):
            sp.update({b + f"{s_}_mod.linear.weight": (6 * dim, dim), b + f"{s_}_mod.linear.bias": (6 * dim,),
                       b + f"{s_}_attn_qkv.weight": (3 * dim, dim), b + f"{s_}_attn_qkv.bias": (3 * dim,),
                       b + f"{s_}_attn_q_norm.weight": (128,), b + f"{s_}_attn_k_norm.weight": (128,),
                       b + f"{s_}_attn_proj.weight": (dim, dim), b + f"{s_}_attn_proj.bias": (dim,),
                       b + f"{s_}_mlp.fc1.weight": (4 * dim, dim), b + f"{s_}_mlp.fc1.bias": (4 * dim,),
                       b + f"{s_}_mlp.fc2.weight": (dim, 4 * dim), b + f"{s_}_mlp.fc2.bias": (dim,)})
    return sp
