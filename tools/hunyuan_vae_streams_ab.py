import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import apex_studio_amd
from bench import synth_vae_init
from apex_studio_amd.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
dev = torch.device("cuda", 0)
vae = synth_vae_init(AutoencoderKLHunyuanVideo15(device=dev, dtype=torch.bfloat16), 7)
vae.enable_tiling()
z = torch.randn(1, 32, 31, 30, 52, device=dev).to(torch.bfloat16)
outs = {}
vae.batch_head_blocks = 0
for ns, nb in ((1, 0), (2, 0), (1, 0), (2, 0), (2, 1), (2, 2), (2, 3), (1, 2), (2, 2)):
    vae.decode_streams, vae.batch_head_blocks = ns, nb
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    outs.setdefault(ns, o)
    print(f"decode_streams={ns} batch_head_blocks={nb}: {dt*1e3:.1f} ms  equal_to_sequential={torch.equal(o, outs[1])}  mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
