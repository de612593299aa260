#!/usr/bin/env python3
"""Round 6: what do the host-issued exchange calls BETWEEN the graph segments of a frame-sharded forward cost?

One GPU, world size 1, backend "nccl" (= RCCL).  The 859.5 M UNet on ONE RANK'S SHARE of a sharded clip -- 16 frames (48 samples), shipped
`flow_fix` schedule, this rank playing rank 1 of 2 (it has a predecessor: it sends its tail slab and receives a halo at both hooked
level-0 layers) -- as the engine runs it at N > 1: five hipGraph segments per forward with the exchange calls issued from the host
between them.  Timed per forward (graph replay, N forwards back to back, events on the launch stream):

  unsharded, one launch sequence        the same 16 frames as a whole clip (no exchange, one graph)
  sharded, in-memory loop-back          five segments, the slabs handed over by Python (parallel.LoopbackShard): the cost of cutting the graph
  sharded, RCCL p2p to self             ... with batch_isend_irecv (2.6 MB slab out, 2.6 MB in, to / from this rank) + Work.wait between segments
  sharded, RCCL all-gather (1 rank)     ... with all_gather_into_tensor

The difference between the last two and the first is what 4 host round trips per step cost a rank, short of the xGMI transfer itself
(one hop of 2.6 MB at ~50-150 GB/s: 20-50 us, overlapped with the projections issued between start and finish).
usage (GPU box): python tools/rccl_selfloop_overhead.py [--iters 20]"""
import argparse
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--frames", type=int, default=16)
    a = ap.parse_args()
    from vface_amd import hip
    from vface_amd.engine import Act
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler
    from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG, LatentDiffusion
    from vface_amd.ldm.models.pnp_utils import register_spa_attn_injection as reg
    from vface_amd.parallel import FrameShard, LoopbackShard, process_group_timeout
    from vface_amd.utils import synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=dev, timeout=process_group_timeout())

    class SelfLoop(FrameShard):
        """Rank 1 of 2, alone in this process: its tail slab goes to this rank and is dropped, its halo comes from this rank (a stored slab)."""

        def __init__(self, total, mode, halos):
            super().__init__(1, 2, total, dist=dist, mode=mode)
            self.halos, self.sink, self.index = halos, {}, 0

        def begin_forward(self):
            pass

        def start_exchange(self, tail, recv=None):
            k = self.index
            halo = recv if recv is not None else torch.empty_like(tail)
            sink = self.sink.setdefault(k, torch.empty_like(tail))
            if self.mode == "allgather":
                works = [dist.all_gather_into_tensor(sink, tail.contiguous(), async_op=True),
                         dist.all_gather_into_tensor(halo, self.halos[k], async_op=True)]
            else:
                works = dist.batch_isend_irecv([dist.P2POp(dist.isend, tail.contiguous(), 0), dist.P2POp(dist.irecv, sink, 0),
                                                dist.P2POp(dist.isend, self.halos[k], 0), dist.P2POp(dist.irecv, halo, 0)])
            return ("p2p", works, halo)

    F_, h = a.frames, 64
    total = 2 * F_
    ldm = LatentDiffusion(dict(FFHQ_UNET_CONFIG))
    synth.fill_module_(ldm.unet, seed=0)
    ldm = ldm.to(dev)
    sampler = DDIMSampler(ldm)
    eng = ldm.unet.engine
    gflow = synth.synth_flow(total - 1, h, h)
    x = torch.cat([synth.synth_normal(f"ovh.x.{0 if c == 1 else c}", (F_, 9, h, h)) for c in range(3)]).to(dev)     # the sampler's batch: chunk 1 = chunk 0
    ctx = synth.synth_normal("ovh.c", (3 * F_, 1, 768)).to(dev)
    tt = torch.full((3 * F_,), 481, dtype=torch.long, device=dev)
    xin = torch.empty(3 * F_ * h * h, 16, dtype=eng.dtype, device=dev)
    hip.nchw_to_nhwc(x.float().contiguous(), xin, N=3 * F_, C_=9, hw=h * h, cpad=16)
    keep = {}

    def setup(shard):
        shard.install(eng, gflow, dev)
        key = (shard.rank, shard.world)
        if key not in keep:
            keep[key] = [f[None].to(dev) for f in shard.local_flow(gflow)]
        reg(sampler, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True)
        reg(sampler, 1, switch_on=True, input_blocks=True, middle_block=False, output_blocks=False, chunks=3, flow=keep[key],
            block_indices=list(range(9)), fusion="flow_fix", split_ratio_fft=0.8, alpha=0.8)

    def timed(shard, streams):
        eng._graphs, eng._graph_failed, eng._split_state = {}, set(), {}
        eng.split_streams, eng.use_graph, eng.share_prefix = streams, True, True
        setup(shard)
        for _ in range(3):
            shard.begin_forward()
            eng.step_forward_nhwc(Act(xin, 3 * F_, h, h), tt, ctx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            shard.begin_forward()
            eng.step_forward_nhwc(Act(xin, 3 * F_, h, h), tt, ctx)
        e1.record()
        torch.cuda.synchronize()
        segs = max((len(g["segments"]) for g in eng._graphs.values()), default=0)
        return e0.elapsed_time(e1) / a.iters, segs

    # the halo slabs rank 1 receives: what rank 0 of the same clip sends (run once, in-memory)
    store = {}
    r0 = LoopbackShard(0, 2, total, store)
    setup(r0)
    eng.use_graph, eng.share_prefix = False, True
    x0 = torch.cat([synth.synth_normal(f"ovh.x0.{0 if c == 1 else c}", (F_, 9, h, h)) for c in range(3)]).to(dev)
    xin0 = torch.empty_like(xin)
    hip.nchw_to_nhwc(x0.float().contiguous(), xin0, N=3 * F_, C_=9, hw=h * h, cpad=16)
    r0.begin_forward()
    eng.step_forward_nhwc(Act(xin0, 3 * F_, h, h), tt, ctx)
    torch.cuda.synchronize()
    halos = {k: s.clone() for k, s in enumerate(store[0])}
    print(f"{F_} frames ({3 * F_} samples) per rank, flow_fix, 64 x 64 latents; {a.iters} forwards each; slab {halos[0].numel() * 2 / 1e6:.2f} MB")
    whole = LoopbackShard(0, 1, F_, {})
    rows = [("unsharded, two launch sequences (what N = 1 runs)", whole, 2), ("unsharded, one launch sequence", whole, 1)]
    lb = LoopbackShard(1, 2, total, {0: [halos[0], halos[1]] * 64})
    rows += [("sharded (rank 1 of 2), in-memory loop-back", lb, 1), ("sharded, RCCL batch_isend_irecv to self", SelfLoop(total, "p2p", halos), 1),
             ("sharded, RCCL all-gather (one rank)", SelfLoop(total, "allgather", halos), 1)]
    for name, shard, streams in rows:
        if isinstance(shard, LoopbackShard) and shard.world > 1:
            # (its predecessor's slabs: the same two, whatever the call count)
            shard.store[0] = [halos[0], halos[1]] * 4096
        ms, segs = timed(shard, streams)
        print(f"   {name:55s}: {ms:7.2f} ms per forward, {segs} graph segment(s)", flush=True)
    eng.halo_exchange = None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
