"""Which kind of neighbour makes the warp kernel misread?  Victim: fill -> GEMM -> warp x2 (no attention of its own).
Noise process: torch matmul / vface attention / vface GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp

def victim(iters, q, own_att):
    from vface_amd import hip, packing
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(0)
    F_, h, d, heads = 2, 32, 64, 8
    n = h * h; B = 3 * F_; Fn = F_ * n
    x = torch.randn(B * n, d, generator=g).half().to(DEV)
    wq, wk = (torch.randn(d, d, generator=g) / 8 for _ in range(2))
    wlin = packing.fold_fsai(wq, wk, 0.8).half().to(DEV)
    flow = (torch.randn(F_ - 1, 2, h, h, generator=g) * 2).to(DEV)
    T = torch.empty(Fn, 2 * d, dtype=torch.float16, device=DEV)
    d1 = torch.empty(Fn, 2 * d, dtype=torch.float16, device=DEV); d2 = torch.empty_like(d1)
    qkv = torch.randn(B * n, 3 * d, generator=g).half().to(DEV); att = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    first, bad, dumps = None, 0, []
    kw = dict(F=F_, h=h, w=h, C_=2 * d, ld_src=2 * d, fs_src=n * 2 * d, ld_dst=2 * d, fs_dst=n * 2 * d, alpha=0.8)
    for it in range(iters):
        T.fill_(float(it % 3))
        hip.gemm(x[Fn:], wlin, T, M=Fn, N=2 * d, K=2 * d, lda=d, ldc=2 * d, ldw=2 * d, a2=x, lda2=d, k1=d)
        hip.flow_warp(T, d1, flow, **kw); hip.flow_warp(T, d2, flow, **kw)
        c1, c2 = d1.clone(), d2.clone()
        if own_att:
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=B, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d,
                          ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=(d // heads) ** -0.5)
        if first is None: first = c1
        else:
            for cc in (c1, c2):
                if not torch.equal(cc, first):
                    bad += 1
                    if len(dumps) < 6:
                        dumps.append(cc.cpu())
    if dumps:
        os.makedirs("gpurun_out", exist_ok=True)
        torch.save({"T": T.cpu(), "flow": flow.cpu(), "ref": first.cpu(), "bad": dumps, "h": h, "d": d}, "gpurun_out/warp_misreads.pt")
    q.put(("victim", own_att, bad))

def noise(kind, seconds, q):
    import time
    from vface_amd import hip
    if os.environ.get("DBG_NOISE_LIB"):
        hip.LIB_PATH = os.environ["DBG_NOISE_LIB"]
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(1)
    n, d, heads, B = 1024, 64, 8, 6
    if os.environ.get("DBG_SHIFT_VA") == "1":
        pad = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)    # keep it alive: the noise tensors land at other VAs
    qkv = torch.randn(B * n, 3 * d, generator=g).half().to(DEV); att = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    a = torch.randn(4096, 4096, generator=g).half().to(DEV)
    w = torch.randn(1280, 1280, generator=g).half().to(DEV); o = torch.empty(B * n, 1280, dtype=torch.float16, device=DEV)
    xa = torch.randn(B * n, 1280, generator=g).half().to(DEV)
    t0 = time.time(); k = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            if kind == "matmul": torch.mm(a, a)
            elif kind == "attention":
                hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=B, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d,
                              ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=(d // heads) ** -0.5)
            elif kind == "gemm": hip.gemm(xa, w, o, M=B * n, N=1280, K=1280, lda=1280, ldc=1280)
        torch.cuda.synchronize(); k += 20
    q.put(("noise", kind, k))

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    for kind, own in (("attention", False),):
        ps = [ctx.Process(target=victim, args=(1500, q, own))]
        if kind != "none": ps.append(ctx.Process(target=noise, args=(kind, 6, q)))
        for p in ps: p.start()
        for p in ps: p.join(500)
        print(kind, [q.get(timeout=5) for _ in ps], flush=True)
