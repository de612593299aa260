"""Platform check, no VFace kernels in the victim: fill -> overwrite -> read with torch ops only, while another process
keeps the GPU busy.  Does the reader ever see the fill value?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp

def victim(iters, q, use_attn):
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(0)
    X = torch.randn(2048, 128, generator=g).half().to(DEV)
    idx = torch.randint(0, 2048, (2048,), generator=g).to(DEV)
    A = torch.empty_like(X)
    perm = torch.randperm(2048, generator=g).to(DEV); Xp = X[perm].contiguous()
    ref = None; bad = 0
    if use_attn:
        from vface_amd import hip
        n, d, heads, B = 1024, 64, 8, 6
        qkv = torch.randn(B * n, 3 * d, generator=g).half().to(DEV); att = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    for it in range(iters):
        A.fill_(float(it % 3))
        A.index_copy_(0, perm, Xp)       # overwrite row by row in a scrambled order: the writer of a line sits on another XCD than the fill's
        Bt = A[idx] * 0.25 + A * 0.75    # gather + elementwise reader
        if use_attn:
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=B, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d,
                          ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=(d // heads) ** -0.5)
        if ref is None: ref = Bt.clone()
        else: bad += (not torch.equal(Bt, ref))
    q.put(("victim", use_attn, bad))

def noise(iters, q):
    from vface_amd import hip
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(1)
    n, d, heads, B = 1024, 64, 8, 6
    qkv = torch.randn(B * n, 3 * d, generator=g).half().to(DEV); att = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    for it in range(iters):
        hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=B, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d,
                      ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=(d // heads) ** -0.5)
        if it % 50 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); q.put(("noise", iters))

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    for use_attn in (False, True):
        ps = [ctx.Process(target=victim, args=(6000, q, use_attn)), ctx.Process(target=noise, args=(12000, q))]
        for p in ps: p.start()
        for p in ps: p.join(500)
        for _ in ps: print(q.get(timeout=5), flush=True)
