"""Frame sharding of a clip across the GPUs of one node (one process per GPU, RCCL over xGMI).

The hot path shards by frame: every rank runs the full 3-way batch for a contiguous range of frames.  FSAI and
structure injection have no cross-frame dependency; flow-guided smoothing reads exactly one
neighbour (F9): rank r needs rank r-1's LAST frame's fused q|k of chunk 1 at the level-0 hooked layers.  That is
the only exchange step on the path, and the only collective this module issues: per hooked level-0 layer and
DDIM step, a ``[n, 2d]`` 16-bit slab (5.2 MB at 512x512: 2.6 MB each of q and k) moves one hop down the chain -- point-to-point
``isend/irecv`` (xGMI is point-to-point, so a one-hop shift costs one link transfer), or an all-gather of the
slabs when ``mode="allgather"``.  The transfer is started right after chunk 1's fused projection and waited for
just before the warp, so it overlaps chunk 2's projection (see ``UNetEngine._attn1_sharded``).
"""
from __future__ import annotations

import datetime
import os
from typing import Optional

import torch


class ExchangeTimeout(RuntimeError):
    """A boundary exchange did not complete within the bound: the message names this rank and the peer(s) it waited for.
    Callers (bench.py, the entry script) exit non-zero on it instead of waiting for the driver's limit."""


def exchange_timeout_s() -> float:
    """Bound on one boundary exchange, seconds (``VFACE_EXCHANGE_TIMEOUT_S``, default 120).  Under gloo / host staging it is
    the ``Work.wait`` timeout; under RCCL the waits are stream-ordered (the host does not block), so the same figure is handed
    to ``init_process_group(timeout=..)`` (``process_group_timeout``) and RCCL's watchdog aborts the process when a peer never
    shows up; ``VFACE_EXCHANGE_BLOCKING=1`` polls ``Work.is_completed`` on the host instead and raises ``ExchangeTimeout``."""
    return float(os.environ.get("VFACE_EXCHANGE_TIMEOUT_S", "120"))


def process_group_timeout() -> datetime.timedelta:
    return datetime.timedelta(seconds=exchange_timeout_s())


def frame_range(rank: int, world: int, total: int):
    """Contiguous split of ``total`` frames; the first ``total % world`` ranks get one extra."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class FrameShard:
    def __init__(self, rank: int, world: int, total_frames: int, dist=None, mode: str = "p2p"):
        self.rank, self.world, self.total = rank, world, total_frames
        self.first, self.count = frame_range(rank, world, total_frames)
        self.dist = dist
        self.mode = mode
        if world > 1 and dist is None:
            raise ValueError("world_size > 1 needs torch.distributed")

    def set_index(self, k: int) -> None:
        """Ordinal of the NEXT exchange inside its UNet forward (0 = the first hooked flow layer, 1 = the second ..): the engine
        states it before every ``start_exchange`` -- eagerly and from the host calls between hipGraph segments alike -- so that an
        exchange object that keeps per-layer state (``StreamShard``) never has to infer it from call counts."""
        self.index = int(k)

    # ---- flow bookkeeping: global flow[i] maps frame i -> frame i+1 (temporal_flow.py:163-188, F-1 fields)
    def local_flow(self, global_flow: torch.Tensor) -> torch.Tensor:
        """Fields between consecutive frames INSIDE this shard: ``[count-1, 2, h, w]``."""
        return global_flow[self.first:self.first + self.count - 1]

    def halo_flow(self, global_flow: torch.Tensor) -> Optional[torch.Tensor]:
        """The field from the previous rank's last frame into this shard's first frame, or None on rank 0."""
        if self.first == 0:
            return None
        return global_flow[self.first - 1]

    # ---- the boundary exchange
    def start_exchange(self, tail: torch.Tensor, recv: Optional[torch.Tensor] = None):
        """Send this shard's last-frame slab to rank+1 and start receiving rank-1's.  Returns a handle.
        ``recv``: a persistent receive buffer shaped like ``tail`` (the hipGraph-segmented forward bakes its address into
        the captured warp launch); default: a fresh one per call."""
        if self.world == 1:
            return None
        d = self.dist
        if tail.is_cuda and d.get_backend() == "gloo":
            # rehearsal on a box without one GPU per rank (gloo cannot move device memory peer to peer): the same
            # protocol through host staging.  Production runs use backend "nccl" (RCCL over xGMI) below.
            host = tail.detach().to("cpu").contiguous()
            ops, halo = [], None
            if self.rank + 1 < self.world:
                ops.append(d.P2POp(d.isend, host, self.rank + 1))
            if self.rank > 0:
                halo = torch.empty_like(host)
                ops.append(d.P2POp(d.irecv, halo, self.rank - 1))
            works = d.batch_isend_irecv(ops) if ops else []
            return ("host", works, (halo, tail.device))
        if self.mode == "allgather":
            t = tail.contiguous()
            bufs = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            work = d.all_gather_into_tensor(bufs, t, async_op=True)  # rank r's slab = rows [r*n, (r+1)*n)
            return ("ag", work, bufs.view((self.world,) + tuple(t.shape)))
        ops, halo = [], None
        if self.rank + 1 < self.world:
            ops.append(d.P2POp(d.isend, tail.contiguous(), self.rank + 1))
        if self.rank > 0:
            halo = recv if recv is not None else torch.empty_like(tail)
            ops.append(d.P2POp(d.irecv, halo, self.rank - 1))
        works = d.batch_isend_irecv(ops) if ops else []
        return ("p2p", works, halo)

    def _peers(self) -> str:
        peers = ([f"rank {self.rank - 1} (its last-frame slab)"] if self.rank > 0 else []) + \
                ([f"rank {self.rank + 1} (to take ours)"] if self.rank + 1 < self.world else [])
        return " and ".join(peers) if self.mode != "allgather" else f"all {self.world} ranks (all-gather)"

    def _wait(self, works) -> None:
        """Bounded wait on the works of one exchange.  gloo (and any backend whose ``wait`` blocks the host) takes the timeout
        directly; RCCL's ``wait`` only orders the stream, so the bound there is the process group's own timeout (watchdog),
        unless ``VFACE_EXCHANGE_BLOCKING=1`` asks for host polling."""
        limit = exchange_timeout_s()
        backend = self.dist.get_backend() if hasattr(self.dist, "get_backend") else "gloo"
        poll = backend != "gloo" and os.environ.get("VFACE_EXCHANGE_BLOCKING") == "1"
        import time
        t0 = time.monotonic()
        for w in works:
            try:
                if poll:
                    while not w.is_completed():
                        if time.monotonic() - t0 > limit:
                            raise TimeoutError
                        time.sleep(1e-4)
                    w.wait()
                elif backend == "gloo":
                    left = max(limit - (time.monotonic() - t0), 1e-3)
                    if w.wait(datetime.timedelta(seconds=left)) is False:
                        raise TimeoutError
                else:
                    w.wait()
            except (TimeoutError, RuntimeError) as e:
                if isinstance(e, TimeoutError) or "time" in str(e).lower():
                    raise ExchangeTimeout(f"rank {self.rank} of {self.world}: the boundary exchange did not complete within "
                                          f"{limit:g} s -- waited for {self._peers()}") from e
                raise

    def finish_exchange(self, handle) -> Optional[torch.Tensor]:
        """Wait (bounded: ``exchange_timeout_s``) for the transfer; returns the previous rank's slab (None on rank 0 / single
        rank).  Raises ``ExchangeTimeout`` naming the peer when it does not arrive."""
        if handle is None:
            return None
        kind, work, buf = handle
        if kind == "host":
            self._wait(work)
            halo, dev = buf
            return halo.to(dev) if halo is not None else None
        if kind == "ag":
            self._wait([work])
            return buf[self.rank - 1] if self.rank > 0 else None
        self._wait(work)
        return buf

    # ---- one decision for all ranks of a clip (hipGraph capture outcome)
    def agree(self, ok: bool, over_budget: bool = False):
        """All-reduce two flags over the ranks of the clip: returns ``(every rank ok, any rank over budget)``.  A rank whose
        capture failed and a rank whose capture succeeded must take ONE decision, or their exchange counts stop pairing."""
        if self.world == 1:
            return bool(ok), bool(over_budget)
        d = self.dist
        dev = "cuda" if d.get_backend() == "nccl" else "cpu"
        t = torch.tensor([0 if ok else 1, 1 if over_budget else 0], dtype=torch.int32, device=dev)
        d.all_reduce(t, op=d.ReduceOp.MAX)
        bad, over = (int(v) for v in t.cpu())
        return bad == 0, over != 0

    def drain(self, pending, remaining) -> None:
        """Finish the paired exchanges of an aborted forward so that the neighbours' calls still match: ``pending`` = a handle
        that was started and not finished (or None), ``remaining`` = the slabs (tensors shaped like the tails the forward would
        have sent; contents irrelevant) of the exchanges not yet started."""
        if pending is not None:
            self.finish_exchange(pending)
        for tail in remaining:
            self.finish_exchange(self.start_exchange(tail))

    def slab_bytes_per_step(self, n: int, d: int, layers: int = 2, elem: int = 2) -> int:
        """Bytes this rank SENDS per DDIM step: one ``[n, 2d]`` slab per hooked level-0 layer (none from the last rank)."""
        return 0 if self.rank + 1 >= self.world else layers * n * 2 * d * elem

    def install(self, engine, global_flow: torch.Tensor, device) -> None:
        """Hook the exchange into a ``UNetEngine`` (used by flow_fix layers only)."""
        hf = self.halo_flow(global_flow)
        engine.halo_flow = hf.to(device=device, dtype=torch.float32).contiguous() if hf is not None else None
        engine.halo_exchange = self if self.world > 1 else None
        engine.halo_hw = (int(global_flow.shape[-2]), int(global_flow.shape[-1]))


class LoopbackShard(FrameShard):
    """All shards of a clip run ONE AFTER ANOTHER in one process on one GPU: rank r's boundary slabs are kept in memory
    (in call order -- one per hooked level-0 layer and UNet call) and handed to rank r+1 when it runs.  The chain has
    no cycle (rank 0 needs nothing), so running the ranks in order needs no concurrency.  Same engine code path as the
    RCCL exchange (``UNetEngine._attn1_sharded``), which makes it the strict bit-for-bit test of that path
    (tests/test_sharded_gpu.py) and a way to walk a clip that does not fit one batch through one GPU."""

    def __init__(self, rank: int, world: int, total_frames: int, store: dict):
        super().__init__(rank, world, total_frames, dist=object() if world > 1 else None)
        self.store = store          # {rank: [slab, ...]} shared by the shards of one clip
        self.store.setdefault(rank, [])
        self._next = 0

    def begin_forward(self):
        """Call before each UNet forward of this shard: its slabs are re-recorded, the predecessor's re-read."""
        self.store[self.rank] = []
        self._next = 0

    def start_exchange(self, tail: torch.Tensor, recv=None):
        if self.world == 1:
            return None
        self.store[self.rank].append(tail.detach().clone())
        i = self._next
        self._next += 1
        return ("loop", i, None)

    def finish_exchange(self, handle):
        if handle is None or self.rank == 0:
            return None
        return self.store[self.rank - 1][handle[1]]

    def agree(self, ok: bool, over_budget: bool = False):
        return bool(ok), bool(over_budget)      # one process: its own verdict is everybody's


class StreamShard(FrameShard):
    """The two frame halves of ONE batch on one GPU as two shards that run AT THE SAME TIME on two HIP streams (the engine's two
    launch streams under ``flow_fix``, whose warp reads the previous frame: UNetEngine._step_forward_split).  Half 0 hands the
    fused q|k of its last frame to half 1 at every hooked flow layer, through device memory and an event:

      half 0, on its stream:  slab -> slot[k] (device copy), then record event[k]            (start_exchange)
      half 1, on its stream:  wait for event[k]; the warp (or the graph's receive buffer) then reads slot[k]   (finish_exchange)

    What makes this hand-over safe, each a hazard the form withdrawn in round 4 may have had (its code was never committed):
    * one slot and one event PER exchange ordinal k (``set_index``), never one re-used slot: half 0 does not wait for half 1, so it
      reaches the second hooked layer and stores its slab there while half 1 may not yet have consumed the first;
    * the ordinal is STATED by the engine, not counted here: a capturing call runs the forward three times (warm-up, capture
      pass, first replay), a replay once -- a counter drifts between the two;
    * half 0's host calls of a step are all enqueued BEFORE half 1's (the engine walks the halves in order), so when half 1
      enqueues its wait the event's latest record IS this step's: a wait enqueued before that record would refer to the previous
      step's and return at once -- reading the previous step's slab (the ~1e-3-sized, frame-3-then-5 difference of
      gpurun_out/r4E_t.log is what a stale slab produces after three DDIM steps);
    * slot[k] is re-written only by the next step's half 0, which starts after this step's join (both streams waited for);
    * slots and events are allocated once and never freed (no caching-allocator re-use across streams);
    * every stream-ordered call here goes to ``torch.cuda.current_stream()``: the warm-up forward of a capture runs on a side
      stream, the capture pass's real exchanges on the capturing stream, replays on the half's launch stream."""

    def __init__(self, rank: int, world: int, total_frames: int, shared: dict):
        super().__init__(rank, world, total_frames, dist=object() if world > 1 else None)
        self.shared = shared            # {"slots": {(rank, k): tensor}, "events": {(rank, k): Event}} of the batch's shards
        self.shared.setdefault("slots", {})
        self.shared.setdefault("events", {})
        self.index = 0

    def start_exchange(self, tail: torch.Tensor, recv=None):
        if self.world == 1:
            return None
        k = self.index
        if self.rank + 1 < self.world:
            key = (self.rank, k)
            slot = self.shared["slots"].get(key)
            if slot is None or slot.shape != tail.shape or slot.dtype != tail.dtype:
                slot = self.shared["slots"][key] = torch.empty(tail.shape, dtype=tail.dtype, device=tail.device)
            slot.copy_(tail)
            ev = self.shared["events"].get(key)
            if ev is None:
                ev = self.shared["events"][key] = torch.cuda.Event()
            ev.record()                                   # on the current stream, behind the copy
        return ("stream", k, None)

    def finish_exchange(self, handle):
        if handle is None or self.rank == 0:
            return None
        key = (self.rank - 1, handle[1])
        ev = self.shared["events"].get(key)
        if ev is None:
            raise RuntimeError(f"StreamShard: half {self.rank} reached exchange {handle[1]} before half {self.rank - 1} issued it "
                               "(the halves must be enqueued in order)")
        torch.cuda.current_stream().wait_event(ev)
        return self.shared["slots"][key]

    def agree(self, ok: bool, over_budget: bool = False):
        return bool(ok), bool(over_budget)      # one process; a half that runs eagerly still makes the same exchange calls
