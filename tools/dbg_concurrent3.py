"""GEMM -> flow_warp under contention: does the warp read what the GEMM just wrote?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp

def work(rank, iters, mode, q):
    from vface_amd import hip, packing
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(rank)
    F_, h, d = 2, 32, 64
    n = h * h
    x = torch.randn(3 * F_ * n, d, generator=g).half().to(DEV)
    wq, wk = (torch.randn(d, d, generator=g) / 8 for _ in range(2))
    wlin = packing.fold_fsai(wq, wk, 0.8).half().to(DEV)
    flow = (torch.randn(F_ - 1, 2, h, h, generator=g) * 2).to(DEV)
    Fn = F_ * n
    T = torch.empty(Fn, 2 * d, dtype=torch.float16, device=DEV)
    dst = torch.empty(Fn, 3 * d, dtype=torch.float16, device=DEV)
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    first, bad = None, 0
    for it in range(iters):
        T.fill_(float(it % 7))          # whatever the warp sees if it runs early / reads a stale line
        if mode == "evict":
            junk.fill_(it & 255)
        hip.gemm(x[Fn:], wlin, T, M=Fn, N=2 * d, K=2 * d, lda=d, ldc=2 * d, ldw=2 * d, a2=x, lda2=d, k1=d)
        if mode == "sync":
            torch.cuda.synchronize()
        hip.flow_warp(T, dst, flow, F=F_, h=h, w=h, C_=2 * d, ld_src=2 * d, fs_src=n * 2 * d, ld_dst=3 * d, fs_dst=n * 3 * d, alpha=0.8)
        cur = dst[:, :2 * d].clone()
        if first is None: first = cur
        else: bad += (not torch.equal(cur, first))
    q.put((rank, mode, bad))

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    for mode in ("plain", "sync", "evict"):
        ps = [ctx.Process(target=work, args=(r, 600, mode, q)) for r in range(2)]
        for p in ps: p.start()
        for p in ps: p.join(500)
        for _ in ps: print(q.get(timeout=5), flush=True)
