#!/usr/bin/env python3
"""GroupNorm-apply (csrc/pointwise.hip gn_apply_kernel) alone on the UNet's large maps: us per launch and TB/s of algorithmic bytes
(round 6: 4.4-5.45 TB/s with the 48-register large-map instantiation; four loads in flight per lane instead of one: no gain).
usage (GPU box): python tools/bench_gn_apply.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from vface_amd import hip
hip.load()
dev="cuda:0"
def timeit(fn, iters=20):
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(iters+1)]
    for _ in range(3): fn()
    ev[0].record()
    for i in range(iters):
        fn(); ev[i+1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i+1]) for i in range(iters))*1e3
for nimg,hw,C,f32 in ((96,4096,320,True),(48,4096,320,True),(96,4096,640,False),(48,4096,960,False),(96,1024,640,True),(48,1024,640,True),(96,1024,1280,False)):
    x=torch.randn(nimg*hw,C,device=dev,dtype=torch.float32 if f32 else torch.float16)
    st=torch.randn(nimg,32,2,device=dev).abs()+0.5
    g=torch.randn(C,device=dev); b=torch.randn(C,device=dev)
    y=torch.empty(nimg*hw,C,device=dev,dtype=torch.float16)
    fn=lambda: hip.groupnorm_apply(x,st,g,b,y,nimg=nimg,hw=hw,C_=C,ldx=C,ldy=C,silu=True)
    us=timeit(fn); by=nimg*hw*C*(x.element_size()+2)
    print(f"nimg={nimg} hw={hw} C={C} in={'f32' if f32 else 'f16'}: {us:7.1f} us  {by/us/1e6:6.2f} TB/s", flush=True)
