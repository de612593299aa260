"""Bisect the kernel sequence of vface_attn1_forward (flow_fix) for the contention-dependent mismatch in the warped rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp

def work(rank, iters, mode, q):
    from vface_amd import hip, packing
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(rank)
    F_, h, d, heads = 2, 32, 64, 8
    n = h * h; B = 3 * F_; Fn = F_ * n
    x = torch.randn(B * n, d, generator=g).half().to(DEV)
    wq, wk, wv = (torch.randn(d, d, generator=g) / 8 for _ in range(3))
    wqkv = packing.pack_qkv(wq, wk, wv).half().to(DEV)
    wlin = packing.fold_fsai(wq, wk, 0.8).half().to(DEV)
    wo = (torch.randn(d, d, generator=g) / 8).half().to(DEV); bo = torch.randn(d, generator=g).to(DEV)
    flow = (torch.randn(F_ - 1, 2, h, h, generator=g) * 2).to(DEV)
    qkv = torch.empty(B * n, 3 * d, dtype=torch.float16, device=DEV)
    T = torch.empty(Fn, 2 * d, dtype=torch.float16, device=DEV)
    att = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    out = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    first, bad, info = None, 0, []
    for it in range(iters):
        qkv.fill_(float(it % 5)); T.fill_(float(it % 3))
        if "p0" in mode:
            hip.gemm(x, wqkv, qkv, M=Fn, N=3 * d, K=d, lda=d, ldc=3 * d)
        if "pv" in mode:
            hip.gemm(x[Fn:], wqkv[2 * d:], qkv[Fn:, 2 * d:], M=B * n - Fn, N=d, K=d, lda=d, ldc=3 * d)
        hip.gemm(x[Fn:], wlin, T, M=Fn, N=2 * d, K=2 * d, lda=d, ldc=2 * d, ldw=2 * d, a2=x, lda2=d, k1=d)
        if "f2" in mode:
            hip.gemm(x[2 * Fn:], wlin, qkv[2 * Fn:, :2 * d], M=Fn, N=2 * d, K=2 * d, lda=d, ldc=3 * d, ldw=2 * d, a2=x, lda2=d, k1=d)
        hip.flow_warp(T, qkv[Fn:2 * Fn, :2 * d], flow, F=F_, h=h, w=h, C_=2 * d, ld_src=2 * d, fs_src=n * 2 * d, ld_dst=3 * d,
                      fs_dst=n * 3 * d, alpha=0.8)
        if "early" in mode:
            cur = qkv[Fn:2 * Fn, :2 * d].clone()
        if "att" in mode:
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=B, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d,
                          ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=(d // heads) ** -0.5,
                          variant=2 if "exact" in mode else 0)
        if "og" in mode:
            hip.gemm(att, wo, out, M=B * n, N=d, K=d, lda=d, ldc=d, bias=bo)
        if "early" not in mode:
            cur = qkv[Fn:2 * Fn, :2 * d].clone()
        if first is None: first = cur
        elif not torch.equal(cur, first):
            bad += 1
            if len(info) < 3:
                dif = (cur != first)
                rows = dif.any(1).nonzero().flatten()
                r0 = rows[0].item()
                cols = dif[r0].nonzero().flatten()
                info.append(dict(it=it, fillq=it % 5, fillT=it % 3, nrows=len(rows), row_range=(rows[0].item(), rows[-1].item()),
                                 ncols_row0=len(cols), col_range=(cols[0].item(), cols[-1].item()),
                                 got=cur[r0, cols[:4]].tolist(), want=first[r0, cols[:4]].tolist()))
    q.put((rank, mode, bad, info))

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    for mode in ("att",):
        ps = [ctx.Process(target=work, args=(r, 500, mode, q)) for r in range(2)]
        for p in ps: p.start()
        for p in ps: p.join(500)
        for _ in ps: print(q.get(timeout=5), flush=True)
