import json, os, sys, torch
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import apex_studio_amd
from apex_studio_amd import lib, ops
DEV="cuda"; g=torch.Generator(device=DEV).manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g, device=DEV)*scale).to(torch.bfloat16)
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters
for (M,N,K) in [(4096,4096,3072),(4096,8192,3072),(4096,16384,3072),(4096,4096,12288)]:
    a,w,b=rnd(M,K),rnd(N,K,scale=K**-0.5),rnd(N); out=torch.empty(M,N,device=DEV,dtype=torch.bfloat16)
    r={}
    for rep in range(2):
        for mode in (0,2):
            lib.tune_set("gemm.streamk", mode)
            ms=timeit(lambda: ops.gemm(a,w,b,out=out))
            r.setdefault(mode,[]).append(round(ms*1e3,1))
    print(json.dumps({"shape":[M,N,K],"tiles":(M//256)*(N//256),"us":{"tile launch":r[0],"persistent (no split)":r[2]}}), flush=True)
lib.tune_set("gemm.streamk",1)
