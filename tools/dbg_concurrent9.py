"""Torch-only victims (division / reciprocal / exp2 / fma on fixed data) next to a process running the VFace attention kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp

def victim(iters, q, op):
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(0)
    a = (torch.rand(1 << 18, generator=g) * 60 + 1).to(DEV); b = (torch.rand(1 << 18, generator=g) * 31 + 1).to(DEV)
    first, bad = None, 0
    for it in range(iters):
        if op == "div": y = a / b
        elif op == "rcp": y = torch.reciprocal(b)
        elif op == "exp": y = torch.exp2(-a * 0.1)
        elif op == "fma": y = a * b + a
        elif op == "floor_div": y = torch.floor(a / b) + (a - b * torch.floor(a / b))
        if first is None: first = y.clone()
        else: bad += (not torch.equal(y, first))
    q.put(("victim", op, bad))

def noise(seconds, q):
    from vface_amd import hip
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(1)
    n, d, heads, B = 1024, 64, 8, 6
    qkv = torch.randn(B * n, 3 * d, generator=g).half().to(DEV); att = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=B, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d,
                          ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=(d // heads) ** -0.5)
        torch.cuda.synchronize()
    q.put(("noise",))

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    for op in ("div", "rcp", "exp", "fma", "floor_div"):
        ps = [ctx.Process(target=victim, args=(4000, q, op)), ctx.Process(target=noise, args=(5, q))]
        for p in ps: p.start()
        for p in ps: p.join(300)
        print([r for r in (q.get(timeout=5) for _ in ps) if r[0] == "victim"], flush=True)
