"""Per-clip modulation schedules (the tables `begin_schedule` builds for Flux / QwenImage), scoped to the clip that made them.

A resident transformer can serve two clips at once on two HIP streams (per-stream workspaces, flux.py `forward`); the table of
one clip must then never be read by the other.  `begin_schedule` returns a `ModulationSchedule` HANDLE that the sampler loop
passes back with every step (`{"modulation_step": i, "modulation_schedule": handle}`); a call without a handle is served from a
table only while exactly ONE schedule is live on the model (the single-clip case), otherwise it computes its own vectors.

What a handle carries besides its tables:
  * the timesteps (and guidance) its rows were computed from — every scheduled step compares the values `forward` received with
    its row ON THE DEVICE (one tiny launch, no sync) into a mismatch flag that `release()` reads once per clip and raises on;
  * the event recorded on the stream that built the tables: a consumer on another stream waits on it once.
There is no counterpart in the reference (its AdaLN projections run per step inside each block,
R/src/transformer/flux/base/model.py:230-328); the tables are bit-identical to those per-step launches.
"""
from __future__ import annotations

import itertools
import sys
from typing import Any, Dict, Optional

import torch

from . import lib as _l

_ids = itertools.count(1)


class ModulationSchedule:
    def __init__(self, n: int, B: int, timesteps: torch.Tensor, guidance: Optional[torch.Tensor] = None):
        self.id = next(_ids)
        self.n, self.B = int(n), int(B)
        self.timesteps = timesteps.detach().float().reshape(self.n, -1).contiguous()      # [n, B] or [n, 1], as scheduled
        self.guidance = None if guidance is None else guidance.detach().float().reshape(-1).contiguous()
        self.tables: Dict[Any, torch.Tensor] = {}          # key (model-defined) -> [n * B, mod_total] f32
        # one mismatch counter per (step, image), bumped with a device ATOMIC add (`index_add_`): `row()` is called from concurrent
        # streams (the images of a batch under batch_streams, the cond / uncond passes under cfg_streams), and a plain `add_` on one
        # scalar is a read-modify-write that two streams can interleave — a lost increment would let an invalid clip pass
        self.mismatch = torch.zeros(self.n * self.B, dtype=torch.int32, device=self.timesteps.device)
        self._slot = torch.arange(self.n * self.B, dtype=torch.int64, device=self.timesteps.device)
        self.ready = None
        self.stream = None
        self._waited = set()
        self.live = True

    # -- producer side ---------------------------------------------------------------------------------------------------
    def publish(self):
        """Tables are complete in stream order of the current stream: record that point for consumers on other streams."""
        if self.timesteps.is_cuda:
            st = torch.cuda.current_stream(self.timesteps.device)
            self.stream = st.cuda_stream
            self.ready = torch.cuda.Event()
            self.ready.record(st)
        return self

    # -- consumer side ---------------------------------------------------------------------------------------------------
    def acquire(self):
        """Make the current stream see the tables (a no-op on the stream that built them; one wait per other stream)."""
        if self.ready is None:
            return
        st = torch.cuda.current_stream(self.timesteps.device)
        if st.cuda_stream != self.stream and st.cuda_stream not in self._waited:
            st.wait_event(self.ready)
            for t in self.tables.values():       # the caching allocator must not hand the rows out again while this stream reads them
                t.record_stream(st)
            self._waited.add(st.cuda_stream)

    def row(self, key, i: int, b: int, timestep: Optional[torch.Tensor] = None, guidance: Optional[torch.Tensor] = None,
            clamp_b: bool = False):
        """[1, mod_total] view of (step i, image b) of table `key`, or None when the handle does not cover the call.  `timestep`
        ([1], what forward received for image b) and `guidance` are checked against the scheduled values on the device.
        `clamp_b`: a schedule built for fewer images than the batch serves the rest with its last column (QwenImage: the
        conditioning depends on the timestep alone)."""
        t = self.tables.get(key)
        if not self.live or t is None or not (0 <= i < self.n):
            return None
        if b >= self.B:
            if not clamp_b:
                return None
            b = self.B - 1
        self.acquire()
        if timestep is not None:
            bad = timestep.detach().float().reshape(-1)[0] != self.timesteps[i, min(b, self.timesteps.shape[1] - 1)]
            if guidance is not None and self.guidance is not None:
                bad = bad | (guidance.detach().float().reshape(-1)[0] != self.guidance[min(b, self.guidance.numel() - 1)])
        r = i * self.B + b
        if timestep is not None:
            self.mismatch.index_add_(0, self._slot[r:r + 1], bad.to(torch.int32).reshape(1))
        return t[r:r + 1]

    def release(self, check: bool = True):
        """End of the clip: free the tables; ONE device read of the mismatch flag.  A forward that was handed a row computed for
        another timestep / guidance produced wrong modulation — that is reported, loudly, not absorbed."""
        self.live = False
        self.tables = {}
        bad = int(self.mismatch.sum().item()) if check and sys.exc_info()[0] is None else 0
        if bad != 0:
            raise _l.ApexMIError(f"modulation schedule #{self.id}: {bad} scheduled step(s) were called with a "
                                 "timestep / guidance different from the row they read (begin_schedule's timesteps must be the "
                                 "values forward receives); the clip's output is invalid")


class ScheduleRegistry:
    """The live schedules of ONE model (id -> handle)."""

    def __init__(self):
        self._live: Dict[int, ModulationSchedule] = {}

    def __len__(self):
        return len(self._live)

    def add(self, s: ModulationSchedule) -> ModulationSchedule:
        self._live[s.id] = s
        return s.publish()

    def clear(self):
        for s in self._live.values():
            s.release(check=False)
        self._live = {}

    def end(self, handle: Optional[ModulationSchedule] = None):
        if handle is None:                       # legacy form: every schedule of the model
            hs, self._live = list(self._live.values()), {}
            for s in hs:
                s.release()
            return
        if self._live.pop(handle.id, None) is not None:
            handle.release()

    def find(self, kw) -> Optional[ModulationSchedule]:
        """The schedule a forward's kwargs name; without a handle, the model's only live schedule (else none: ambiguous)."""
        if not self._live or not kw or kw.get("modulation_step") is None:
            return None
        h = kw.get("modulation_schedule")
        if h is not None:
            return h if self._live.get(getattr(h, "id", None)) is h else None
        if len(self._live) == 1:
            return next(iter(self._live.values()))
        return None

    def only(self) -> Optional[ModulationSchedule]:
        return next(iter(self._live.values())) if len(self._live) == 1 else None
