"""Frame sharding on CPU with gloo, world_size 2: the boundary exchange moves the right slab, and per-shard
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


def _worker(rank, world, port, mode, q):
    import torch.distributed as dist
    # `port` is a rendezvous FILE path: nothing to race for between choosing a port and binding it
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        total, C, h, w = 5, 8, 16, 16
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
