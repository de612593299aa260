"""Platform check with torch-only kernels: a victim doing fp32 divisions / reciprocals / exps on fixed data while another
process runs exp-heavy kernels.  Does the victim's result ever change?"""
import sys, os, time
import torch, torch.multiprocessing as mp

def victim(iters, q, op):
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(0)
    a = (torch.rand(1 << 20, generator=g) * 60 + 1).to(DEV); b = (torch.rand(1 << 20, generator=g) * 31 + 1).to(DEV)
    first, bad = None, 0
    for it in range(iters):
        if op == "div": y = a / b
        elif op == "rcp": y = torch.reciprocal(b)
        elif op == "exp": y = torch.exp2(-a)
        elif op == "fma": y = a * b + a
        if first is None: first = y.clone()
        else: bad += (not torch.equal(y, first))
    q.put(("victim", op, bad))

def noise(kind, seconds, q):
    DEV = "cuda:0"
    x = torch.randn(1 << 24, device=DEV)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            if kind == "exp": torch.exp2(x)
            elif kind == "softmax": torch.softmax(x.view(4096, 4096), -1)
            else: x * 1.5
        torch.cuda.synchronize()
    q.put(("noise", kind))

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    for op in ("div", "rcp", "exp", "fma"):
        for kind in ("exp", "softmax"):
            ps = [ctx.Process(target=victim, args=(3000, q, op)), ctx.Process(target=noise, args=(kind, 4, q))]
            for p in ps: p.start()
            for p in ps: p.join(300)
            print(op, kind, [q.get(timeout=5) for _ in ps][0 if True else 1], flush=True)
