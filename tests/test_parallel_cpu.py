"""Frame sharding on CPU with gloo, world_size 2 (and a 4-rank chain with uneven shards): the boundary exchange moves the right slab, and per-shard
flow smoothing with the exchanged halo reproduces the unsharded result exactly (oracle flow functions as the
checker -- the exchange itself is the product code under test)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import flow as oflow
from vface_amd.parallel import FrameShard, frame_range
from vface_amd.utils import synth


def test_frame_range_partition():
    for total, world in ((64, 4), (13, 4), (8, 8), (5, 2), (256, 8)):
        seen = []
        for r in range(world):
            f, c = frame_range(r, world, total)
            seen += list(range(f, f + c))
        assert seen == list(range(total))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q, total=5):
    import torch.distributed as dist
    # `port` is a rendezvous FILE path: nothing to race for between choosing a port and binding it
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        C, h, w = 8, 16, 16
        x = synth.synth_normal("par.x", (total, C, h, w))
        gflow = synth.synth_flow(total - 1, h, w, seed=3)
        full = oflow.align_by_flow(x, [gflow[i] for i in range(total - 1)], 0.8)
        sh = FrameShard(rank, world, total, dist, mode=mode)
        mine = x[sh.first:sh.first + sh.count]
        tok = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], h * w, C).contiguous()
        tail = tok(mine)[-1]
        halo = sh.finish_exchange(sh.start_exchange(tail))
        if rank == 0:
            assert halo is None and sh.halo_flow(gflow) is None
            frames, flows = mine, [f for f in sh.local_flow(gflow)]
            out = oflow.align_by_flow(frames, flows, 0.8)
        else:
            prev = x[sh.first - 1]
            assert torch.equal(halo, tok(prev[None])[0])  # exactly the previous rank's last frame
            halo_img = halo.reshape(h, w, C).permute(2, 0, 1)[None]
            frames = torch.cat([halo_img, mine], 0)
            flows = [sh.halo_flow(gflow)] + [f for f in sh.local_flow(gflow)]
            out = oflow.align_by_flow(frames, flows, 0.8)[1:]
        ok = torch.equal(out, full[sh.first:sh.first + sh.count])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["p2p", "allgather"])
def test_halo_exchange_gloo_world2(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import tempfile
    port = os.path.join(tempfile.mkdtemp(prefix="vface_rdzv_"), "store")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


@pytest.mark.parametrize("mode", ["p2p", "allgather"])
def test_halo_exchange_gloo_world4_uneven_shards(mode):
    """The chain as the 4-GPU configuration runs it (BASELINE configs[3]), with shards of unequal length (13 frames: 4, 3, 3, 3): every
    rank but the first receives exactly its predecessor's last frame -- the middle ranks send and receive in one batched call -- and
    the per-shard smoothing with that halo is the unsharded result, bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import tempfile
    port = os.path.join(tempfile.mkdtemp(prefix="vface_rdzv_"), "store")
    procs = [ctx.Process(target=_worker, args=(r, 4, port, mode, q, 13)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(4))
    assert res == {0: True, 1: True, 2: True, 3: True}


def _spawn(target, world, *args, timeout=120):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import tempfile
    store = os.path.join(tempfile.mkdtemp(prefix="vface_rdzv_"), "store")
    procs = [ctx.Process(target=target, args=(r, world, store, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    return procs, q


def _silent_peer_worker(rank, world, store, q):
    """Rank 0 never takes part in the exchange; rank 1 must give up within the bound and exit non-zero, naming rank 0."""
    import sys
    import time
    import torch.distributed as dist
    from vface_amd.parallel import ExchangeTimeout
    os.environ["VFACE_EXCHANGE_TIMEOUT_S"] = "2"
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    sh = FrameShard(rank, world, 4, dist)
    if rank == 0:
        time.sleep(6)            # alive, but never sends
        q.put((0, "slept"))
        q.close(); q.join_thread()
        os._exit(0)              # (no destroy_process_group: the peer is gone by now)
    t0 = time.monotonic()
    try:
        sh.finish_exchange(sh.start_exchange(torch.zeros(16, 8)))
    except ExchangeTimeout as e:
        q.put((1, str(e), time.monotonic() - t0))
        q.close(); q.join_thread()
        os._exit(3)
    q.put((1, "no timeout", 0.0))
    q.close(); q.join_thread()
    os._exit(0)


def test_peer_that_never_sends_times_out_with_nonzero_exit():
    """``finish_exchange`` is bounded; a rank whose predecessor never sends raises ``ExchangeTimeout``
    (message names the peer) and its process exits non-zero instead of sitting in ``wait()`` until the driver's limit."""
    procs, q = _spawn(_silent_peer_worker, 2, timeout=60)
    msgs = dict((m[0], m[1:]) for m in (q.get(timeout=5) for _ in range(2)))
    assert procs[1].exitcode == 3 and procs[0].exitcode == 0
    text, waited = msgs[1]
    assert "rank 0" in text and "did not complete within 2 s" in text and waited < 10


def _capture_failure_worker(rank, world, store, q):
    """The collective capture decision (engine._capture) rehearsed on its protocol: a forward makes 2
    exchanges; rank 0's "capture pass" dies after the first exchange was STARTED.  It must finish that one, drain the second
    with a dummy slab, and both ranks must reach the same verdict -- after which a further (eager) forward still pairs."""
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    try:
        sh = FrameShard(rank, world, 4, dist)
        slab = lambda v: torch.full((8, 4), float(v))
        got = []
        ok = True
        if rank == 0:
            h = sh.start_exchange(slab(10))              # exchange 1 of the capture pass: started ..
            ok = False                                   # .. then the capture raises on this rank
            sh.drain(h, [slab(-1)])                      # finish the pending one, pair the remaining one with a dummy
        else:
            for i in range(2):
                got.append(sh.finish_exchange(sh.start_exchange(slab(20 + i))))
        all_ok, over = sh.agree(ok, False)
        # one decision everywhere: nobody replays a graph; the next forward (eager, 2 exchanges) still pairs up
        for i in range(2):
            got.append(sh.finish_exchange(sh.start_exchange(slab(100 * (rank + 1) + i))))
        vals = [None if g is None else float(g[0, 0]) for g in got]
        q.put((rank, all_ok, over, vals))
    finally:
        dist.destroy_process_group()


def test_capture_failure_on_one_rank_is_one_decision_for_all():
    procs, q = _spawn(_capture_failure_worker, 2)
    assert [p.exitcode for p in procs] == [0, 0]
    res = {m[0]: m[1:] for m in (q.get(timeout=5) for _ in range(2))}
    assert res[0][0] is False and res[1][0] is False and not res[0][1] and not res[1][1]
    assert res[0][2] == [None, None]                        # rank 0 receives nothing
    assert res[1][2] == [10.0, -1.0, 100.0, 101.0]          # rank 1: the real slab, the dummy, then the eager forward's two
