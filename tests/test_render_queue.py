"""N>1 path of the render queue on CPU: world_size-2 gloo processes (RCCL is the same code path with
backend "nccl" on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import render_queue as rq
    big = torch.arange(1 << 22, dtype=torch.int32) if rank == 0 else torch.zeros(1 << 22, dtype=torch.int32)
    small = torch.full((7,), 3.0) if rank == 0 else torch.zeros(7)
    odd = torch.arange(8 * 1024 * 1024 + 1, dtype=torch.uint8) if rank == 0 else \
        torch.zeros(8 * 1024 * 1024 + 1, dtype=torch.uint8)
    stats = rq.broadcast_shared([big, small, odd], src=0)
    ok = bool(torch.equal(big, torch.arange(1 << 22, dtype=torch.int32)) and
              torch.equal(small, torch.full((7,), 3.0)) and
              torch.equal(odd, torch.arange(8 * 1024 * 1024 + 1, dtype=torch.uint8)))
    clips = [{"id": i, "cost": c} for i, c in enumerate([5.0, 1.0, 1.0, 5.0, 1.0])]
    done = []
    res = rq.run_queue(clips, lambda c: done.append(c["id"]))
    q.put((rank, ok, stats["bytes"], sorted(done), sorted(res["clip_seconds"].keys())))
    dist.destroy_process_group()


def test_broadcast_and_queue_world2():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, ok0, n0, done0, all0), (r1, ok1, n1, done1, all1) = out
    assert ok0 and ok1 and n0 == n1 > 0
    # the two 5.0-cost clips land on different ranks; every clip rendered exactly once
    assert sorted(done0 + done1) == [0, 1, 2, 3, 4] and all0 == all1 == [0, 1, 2, 3, 4]
    assert (0 in done0) != (3 in done0)


def test_assign_clips_lpt():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd.render_queue import assign_clips
    # 4 video clips (cost 100) + 4 image clips (cost 2) on 8 GPUs: one clip each
    a = assign_clips([2, 2, 2, 2, 100, 100, 100, 100], 8)
    assert sorted(len(x) for x in a) == [1] * 8
    # on 4 GPUs each rank gets one video + one image
    a = assign_clips([2, 2, 2, 2, 100, 100, 100, 100], 4)
    assert all(len(x) == 2 and sum(1 for i in x if i >= 4) == 1 for x in a)
    assert assign_clips([], 2) == [[], []]


def _param_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import render_queue as rq
    from apex_studio_amd import text_encoders as TE
    from apex_studio_amd.vae_flux import AutoencoderKL
    # the shared components of a Flux queue at toy sizes: T5 (tied embedding), CLIP, 2-D VAE; constructed on every rank,
    # filled on rank 0 only — the receivers start from garbage
    t5 = TE.T5EncoderModel(dict(vocab_size=64, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2), dtype=torch.bfloat16)
    clip = TE.CLIPTextModel(dict(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                 num_attention_heads=2, max_position_embeddings=16), dtype=torch.bfloat16)
    vae = AutoencoderKL(latent_channels=16, block_out_channels=(32, 32, 64, 64), layers_per_block=1, dtype=torch.bfloat16)
    mods = [t5, clip, vae]
    g = torch.Generator().manual_seed(123)
    for m in mods:
        for p in m.parameters():
            p.data.copy_(torch.randn(p.shape, generator=g).to(p.dtype) if rank == 0 else torch.full(p.shape, float("nan")).to(p.dtype))
    vae._packed = {"stale": 1}                          # derived state on a receiver must be dropped
    stats = rq.broadcast_parameters(mods, src=0, bucket_bytes=1 << 18)   # several buckets
    g = torch.Generator().manual_seed(123)
    ok = True
    for m in mods:
        for p in m.parameters():
            ok = ok and torch.equal(p.data, torch.randn(p.shape, generator=g).to(p.dtype))
    n_unique = len(rq.shared_tensors(mods))
    q.put((rank, bool(ok), stats["bytes"], stats["buckets"], n_unique, dict(vae._packed)))
    dist.destroy_process_group()


def test_broadcast_parameters_world2():
    """The queue's one exchange step with REAL module parameters (text encoders + VAE), world 2 over gloo."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_param_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, ok0, n0, b0, u0, packed0), (_, ok1, n1, b1, u1, packed1) = out
    assert ok0 and ok1, "every parameter on every rank must equal rank 0's"
    assert n0 == n1 > 0 and b0 == b1 > 1 and u0 == u1
    assert packed0 == {"stale": 1} and packed1 == {}, "the receiver's packed-weight cache must be invalidated, the sender's kept"


def _run_world(target, world, timeout=240, extra=()):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


def test_broadcast_parameters_world4_uneven_tails():
    """The same exchange step at world 4 (the scatter + all-gather path with 4 peers; bucket sizes that are not multiples of
    the world size or of 16 bytes leave ragged tails in the staging buffer)."""
    out = _run_world(_param_worker, 4)
    assert all(o[1] for o in out), "every parameter on every rank must equal rank 0's"
    assert len({o[2] for o in out}) == 1 and len({o[3] for o in out}) == 1 and out[0][3] > 1
    assert out[0][5] == {"stale": 1} and all(o[5] == {} for o in out[1:])


def _ragged_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import render_queue as rq
    sizes = [1, 3, 17, 4097, (1 << 20) + 5, 9 << 20, (9 << 20) + 3, 33]        # odd byte counts, some past the scatter threshold
    gen = torch.Generator().manual_seed(7)
    want = [torch.randint(0, 255, (n,), generator=gen, dtype=torch.uint8) for n in sizes]

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for i, w in enumerate(want):
                self.register_buffer(f"b{i}", w.clone() if rank == 0 else torch.zeros_like(w))
    m = M()
    stats = rq.broadcast_parameters([m], src=0, bucket_bytes=(10 << 20) + 1)
    ok = all(torch.equal(getattr(m, f"b{i}"), w) for i, w in enumerate(want))
    # a rank that enumerates a different layout: EVERY rank raises, nobody is left waiting in a collective
    m2 = M()
    if rank == world - 1:
        m2.register_buffer("extra", torch.zeros(5))
    try:
        rq.broadcast_parameters([m2], src=0)
        raised = False
    except RuntimeError as e:
        raised = "layout" in str(e)
    q.put((rank, bool(ok), stats["buckets"], raised))
    dist.destroy_process_group()


def _dynamic_worker(rank, world, port, q, slow_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import render_queue as rq
    # 12 equal-cost clips; the "render" is a seeded function of the clip alone, the slow rank sleeps 5x longer per clip
    clips = [{"id": i, "cost": 1.0, "seed": 1000 + i} for i in range(12)]

    def make_runner(sink, nap):
        def run(c):
            time.sleep(nap * (5.0 if rank == slow_rank else 1.0))
            sink[c["id"]] = torch.randn(4, generator=torch.Generator().manual_seed(c["seed"]))
        return run
    out_s, out_d, out_d2 = {}, {}, {}
    stat = rq.run_queue(clips, make_runner(out_s, 0.0))                      # static LPT (the fallback), no naps
    dyn = rq.run_queue(clips, make_runner(out_d, 0.06), dynamic=True)
    dyn2 = rq.run_queue(clips, make_runner(out_d2, 0.0), dynamic=True)       # a second dynamic run: its own counter key
    gathered = [None] * world
    dist.all_gather_object(gathered, {"s": out_s, "d": out_d, "d2": out_d2})
    merged = {k: {} for k in ("s", "d", "d2")}
    for g in gathered:
        for k in merged:
            assert not set(merged[k]) & set(g[k]), "a clip rendered on two ranks"
            merged[k].update(g[k])
    same = all(sorted(merged[k]) == list(range(12)) for k in merged) and \
        all(torch.equal(merged["s"][i], merged["d"][i]) and torch.equal(merged["s"][i], merged["d2"][i]) for i in range(12))
    per_rank = [sum(1 for r in dyn["clip_rank"].values() if r == j) for j in range(world)]
    q.put((rank, stat["dispatch"], dyn["dispatch"], dyn2["dispatch"], bool(same), per_rank,
           [sum(1 for r in stat["clip_rank"].values() if r == j) for j in range(world)], sorted(dyn["clip_seconds"])))
    dist.destroy_process_group()


def _check_dynamic(out, world, slow):
    for rank, s_kind, d_kind, d2_kind, same, per_rank, per_rank_static, seen in out:
        assert (s_kind, d_kind, d2_kind) == ("static", "dynamic", "dynamic")
        assert same, "every clip exactly once, results identical to the static run"
        assert seen == list(range(12)) and sum(per_rank) == 12
        assert per_rank_static == [12 // world] * world
        others = [n for j, n in enumerate(per_rank) if j != slow]
        assert per_rank[slow] < min(others), (per_rank, "the slow rank must end with fewer clips than any other")
    assert len({tuple(o[5]) for o in out}) == 1, "every rank reports the same clip -> rank map"


def test_dynamic_dispatch_world2_slow_rank_gets_fewer_clips():
    """`run_queue(dynamic=True)`: ranks pull the next clip from an atomic counter in the rendezvous store
    (apps/api/src/api/ray_tasks.py:181-306 picks the GPU per job).  A rank whose runner is 5x slower ends with fewer clips,
    every clip is rendered exactly once, and the rendered results equal the static run's."""
    _check_dynamic(_run_world(_dynamic_worker, 2, extra=(1,)), 2, 1)


def test_dynamic_dispatch_world4():
    _check_dynamic(_run_world(_dynamic_worker, 4, extra=(2,)), 4, 2)


def test_dynamic_dispatch_order_and_world1_fallback():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import render_queue as rq
    assert rq.dispatch_order([2, 100, 2, 100, 50]) == [1, 3, 4, 0, 2]
    done = []
    res = rq.run_queue([{"id": i} for i in range(3)], lambda c: done.append(c["id"]), dynamic=True)
    assert res["dispatch"] == "static" and sorted(done) == [0, 1, 2] and res["clip_rank"] == {0: 0, 1: 0, 2: 0}


def test_broadcast_ragged_sizes_and_layout_mismatch_world4():
    out = _run_world(_ragged_worker, 4)
    assert all(o[1] for o in out) and out[0][2] >= 2
    assert all(o[3] for o in out), "a layout mismatch must raise on every rank"


def test_bench_refuses_to_fake_multi_gpu():
    """`python bench.py --gpus 2` must start 2 ranks or fail loudly — never one process reporting n_gpus 2."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "refusing to run" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout


def _nccl_world1_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import render_queue as rq
    from apex_studio_amd import text_encoders as TE
    dev = torch.device("cuda", 0)
    big = torch.arange(1 << 22, dtype=torch.int32, device=dev)            # 16 MiB: the scatter + all-gather route
    small = torch.full((7,), 3.0, device=dev)
    stats = rq.broadcast_shared([big, small], src=0)
    ok = bool(torch.equal(big.cpu(), torch.arange(1 << 22, dtype=torch.int32)) and torch.equal(small.cpu(), torch.full((7,), 3.0)))
    t5 = TE.T5EncoderModel(dict(vocab_size=64, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2), device=dev,
                           dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    want = []
    for p in t5.parameters():
        w = torch.randn(p.shape, generator=g).to(p.dtype)
        p.data.copy_(w)
        want.append(w)
    pst = rq.broadcast_parameters([t5], src=0, bucket_bytes=1 << 16)
    ok = ok and all(torch.equal(p.data.cpu(), w) for p, w in zip(t5.parameters(), want))
    done = []
    res = rq.run_queue([{"id": i, "cost": 1.0} for i in range(3)], lambda c: done.append(c["id"]), dynamic=True)
    q.put((rank, ok, stats["bytes"], stats["world"], pst["buckets"], pst["bytes"], sorted(done), res["dispatch"],
           dist.get_backend()))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_world1_carries_the_exchange_step():
    """RCCL initialisation and the queue's exchange step inside the GPU suite: backend `nccl` (= RCCL) at world 1 on the one
    visible MI355X — `broadcast_shared` (plain broadcast and the scatter + all-gather route) and `broadcast_parameters`
    (layout digest all-gather, bucketed staging) run THROUGH the backend, bytes unchanged.  N > 1 ranks need the 8-GPU node."""
    (rank, ok, nbytes, world, buckets, pbytes, done, dispatch, backend), = _run_world(_nccl_world1_worker, 1, timeout=600)
    assert backend == "nccl" and world == 1
    assert ok and nbytes == (1 << 24) + 28 and buckets > 1 and pbytes > 0
    assert done == [0, 1, 2] and dispatch == "static"
