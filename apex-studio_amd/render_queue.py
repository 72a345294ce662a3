"""One-clip-per-GPU render queue over an 8x MI355X node.

The reference's only multi-GPU mechanism is data parallelism over JOBS: one long-lived engine-runner
actor per GPU fed from a FIFO, GPU picked by free VRAM (apps/api/src/api/ray_tasks.py:181-306,
ray_resources.py:81-111, settings.py:8 MAX_JOBS_PER_GPU=1).  Here that is one process per GPU under
torch.distributed (backend "nccl" = RCCL over xGMI), clips PULLED by whichever rank is free (an atomic
counter in the rendezvous store; a deterministic longest-processing-time split when there is no
store), and exactly ONE exchange step: a broadcast of the weights every clip
shares (text encoders, VAE) and of shared prompt embeddings at queue start.  No collective inside a
denoise step.

xGMI is point-to-point (7 links per GPU), so a root->all broadcast of a large buffer is done as
scatter (root sends 1/N to each peer over its own link) + all-gather (full mesh), which keeps every
link busy instead of funnelling N-1 full copies through ring hops.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Sequence

import torch
import torch.distributed as dist


def _sync(t: torch.Tensor):
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def broadcast_shared(tensors: Sequence[torch.Tensor], src: int = 0, group=None,
                     big_bytes: int = 8 << 20) -> Dict[str, float]:
    """In-place broadcast of `tensors` from rank `src`.  Large contiguous tensors go
    scatter + all-gather, small ones a plain broadcast.  Returns bytes / seconds / GB/s."""
    if not dist.is_initialized():
        return {"bytes": 0, "seconds": 0.0, "gbps": 0.0, "world": 1}
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nbytes = 0
    if tensors:
        _sync(tensors[0])
    t0 = time.perf_counter()
    for t in tensors:
        assert t.is_contiguous(), "broadcast_shared needs contiguous tensors"
        n = t.numel() * t.element_size()
        nbytes += n
        flat = t.view(-1).view(torch.uint8)
        if n < big_bytes or n % world != 0:
            dist.broadcast(flat, src=src, group=group)
            continue
        chunk = n // world
        mine = torch.empty(chunk, dtype=torch.uint8, device=t.device)
        if rank == src:
            dist.scatter(mine, scatter_list=list(flat.split(chunk)), src=src, group=group)
        else:
            dist.scatter(mine, scatter_list=None, src=src, group=group)
        dist.all_gather_into_tensor(flat, mine, group=group)
    if tensors:
        _sync(tensors[0])
    dt = time.perf_counter() - t0
    return {"bytes": nbytes, "seconds": dt, "gbps": (nbytes / dt / 1e9) if dt > 0 else 0.0, "world": world}


def shared_tensors(modules) -> List[torch.Tensor]:
    """Every parameter and buffer of `modules` once (tied weights — T5's `shared` / `encoder.embed_tokens` — are one
    storage), as the flat list `broadcast_parameters` moves."""
    seen, out = set(), []
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            key = (t.data_ptr(), t.numel(), t.dtype)
            if t.numel() and key not in seen:
                seen.add(key)
                out.append(t.data)
    return out


def broadcast_parameters(modules, src: int = 0, group=None, bucket_bytes: int = 1 << 30) -> Dict[str, float]:
    """The one exchange step of the render queue (north_star: "RCCL over xGMI only for the broadcast of shared
    text-encoder/VAE weights"): rank `src` holds the loaded text encoders / VAE, every other rank has constructed the
    same classes on its device with uninitialised storage; the parameters travel in ~1 GiB buckets (one staging
    buffer, scatter + all-gather per bucket, see `broadcast_shared`) and are written in place on the receivers, whose
    derived state (fused / padded / packed copies) is then dropped through the `_weights_changed` hooks.
    Sizes to expect: Flux T5-XXL 9.5 GB + CLIP-L 0.25 GB + VAE 0.17 GB; Wan UMT5-XXL 11.4 GB + VAE 0.5 GB."""
    tensors = shared_tensors(modules)
    if not dist.is_initialized() or not tensors:      # an initialised group of ONE rank still goes through the backend below
        return {"bytes": sum(t.numel() * t.element_size() for t in tensors), "seconds": 0.0, "gbps": 0.0,
                "world": 1, "tensors": len(tensors), "buckets": 0}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = tensors[0].device
    # every rank must have built the same (order, size, dtype) list — the dedup of tied weights keys on data_ptr, so a tie that
    # exists on one rank only would shift every later offset and the receivers would silently get wrong bytes.  Compare a
    # digest of the layout on ALL ranks before anything moves, and fail on all of them together.
    import hashlib
    layout = hashlib.sha256(repr([(t.numel(), str(t.dtype)) for t in tensors]).encode()).digest()[:8]
    mine = torch.frombuffer(bytearray(layout), dtype=torch.uint8).to(dev)
    seen = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine, group=group)
    if any(not torch.equal(s.cpu(), mine.cpu()) for s in seen):
        raise RuntimeError(f"broadcast_parameters: rank {rank} enumerates {len(tensors)} tensors with a different (size, dtype) "
                           f"layout than another rank — construct the same classes with the same configuration on every rank")
    pad = 16 * world                                          # every bucket a multiple of the world size and 16 B
    buckets, cur, cur_n = [], [], 0
    for t in tensors:
        assert t.is_contiguous() and t.device == dev, "broadcast_parameters: contiguous tensors on one device"
        n = (t.numel() * t.element_size() + 15) // 16 * 16
        if cur and cur_n + n > bucket_bytes:
            buckets.append((cur, cur_n))
            cur, cur_n = [], 0
        cur.append((t, cur_n, t.numel() * t.element_size()))
        cur_n += n
    if cur:
        buckets.append((cur, cur_n))
    cap = (max(n for _, n in buckets) + pad - 1) // pad * pad
    stage = torch.empty(cap, dtype=torch.uint8, device=dev)
    _sync(stage)
    t0 = time.perf_counter()
    total = 0
    for items, n in buckets:
        npad = (n + pad - 1) // pad * pad
        if rank == src:
            for t, off, nb in items:
                stage[off:off + nb].copy_(t.view(-1).view(torch.uint8))
        broadcast_shared([stage[:npad]], src=src, group=group)
        if rank != src:
            for t, off, nb in items:
                t.view(-1).view(torch.uint8).copy_(stage[off:off + nb])
        total += n
    _sync(stage)
    dt = time.perf_counter() - t0
    if rank != src:
        for m in modules:
            for sub in m.modules():
                hook = getattr(sub, "_weights_changed", None)
                if callable(hook):
                    hook()
    return {"bytes": total, "seconds": dt, "gbps": total / dt / 1e9 if dt > 0 else 0.0, "world": world,
            "tensors": len(tensors), "buckets": len(buckets)}


def assign_clips(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of clip indices to ranks; deterministic, identical on
    every rank (ties broken by index).  Puts the long (video) clips on distinct GPUs first, which is
    what bounds the makespan of a mixed image/video queue."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += float(costs[i])
    return out


_queue_epoch = 0          # run_queue calls so far in this process: every rank calls it the same number of times, so the
                          # store key of a run is the same on all of them without any exchange


def _rendezvous_store(group=None):
    """The key-value store the process group was rendezvoused through (a TCPStore on rank 0's host for `env://` /
    torchrun), or None.  Only the default group has one we can reach."""
    if not dist.is_initialized() or group is not None:
        return None
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        return _get_default_store()
    except Exception:
        return None


def dispatch_order(costs: Sequence[float]) -> List[int]:
    """Pull order of the dynamic queue: longest clip first (ties by index) — the LPT seed; with a shared counter the
    i-th pull anywhere gets the i-th entry."""
    return sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))


def run_queue(clips: Sequence[dict], runner: Callable[[dict], object], costs: Sequence[float] = None,
              group=None, dynamic: bool = False) -> Dict[str, object]:
    """Each rank renders clips with `runner(clip)`; returns (on every rank) the per-clip seconds, which rank rendered
    which clip, the per-rank busy time and the makespan.

    `dynamic=False`: the static longest-processing-time split of `assign_clips` on the caller's `cost` guesses.
    `dynamic=True`: ranks PULL — the reference picks the GPU per job at submit time and feeds one FIFO per GPU actor
    (apps/api/src/api/ray_tasks.py:181-306, engine.py:89-122), i.e. a GPU that finishes early takes the next job.  Here
    a rank that becomes free takes the next index of `dispatch_order(costs)` from an atomic counter in the rendezvous
    store (`Store.add`, served by rank 0's TCPStore: a request / reply on the host network — no collective, nothing
    inside a denoise step), so a wrong cost guess, a data-dependent clip (EasyCache skips) or a slow board costs at
    most one clip of imbalance instead of a rank's whole static share.  Every clip is rendered exactly once; with
    exact costs and equal boards the pull order reproduces the LPT assignment.  Without a reachable store (world 1,
    a sub-group) the static split is used."""
    global _queue_epoch
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    costs = list(costs) if costs is not None else [float(c.get("cost", 1.0)) for c in clips]
    _queue_epoch += 1
    store = _rendezvous_store(group) if dynamic and world > 1 else None
    if store is not None:
        order, key = dispatch_order(costs), f"apexmi/render_queue/{_queue_epoch}/next"

        def take():
            while True:
                n = int(store.add(key, 1)) - 1
                if n >= len(order):
                    return
                yield order[n]
        mine = take()
    else:
        mine = iter(assign_clips(costs, world)[rank])
    if dist.is_initialized():
        dist.barrier(group=group)
    t_start = time.perf_counter()
    times = {}
    for i in mine:
        t0 = time.perf_counter()
        runner(clips[i])
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        times[i] = time.perf_counter() - t0
    busy = time.perf_counter() - t_start
    if dist.is_initialized():
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "times": times, "busy": busy}, group=group)
    else:
        gathered = [{"rank": 0, "times": times, "busy": busy}]
    clip_seconds, clip_rank = {}, {}
    for g in gathered:
        clip_seconds.update(g["times"])
        clip_rank.update({i: g["rank"] for i in g["times"]})
    if sorted(clip_seconds) != list(range(len(clips))) or sum(len(g["times"]) for g in gathered) != len(clips):
        raise RuntimeError(f"render queue: {len(clips)} clips queued, rendered {sorted(clip_seconds)} "
                           f"({sum(len(g['times']) for g in gathered)} renders)")
    makespan = max(g["busy"] for g in gathered)
    return {"clip_seconds": clip_seconds, "clip_rank": clip_rank, "busy": [g["busy"] for g in gathered],
            "makespan": makespan, "dispatch": "dynamic" if store is not None else "static",
            "clips_per_hour": 3600.0 * len(clips) / makespan if makespan > 0 else 0.0}
