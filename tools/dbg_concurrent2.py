"""Where does the flow_fix path lose run-to-run determinism under GPU contention?  Loop vface_attn1_forward on fixed
inputs while a second process does the same; compare T (fused q|k of chunk 1), the warped q|k rows and the output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp

def work(rank, iters, q):
    from vface_amd import hip, packing
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(rank)
    F_, h, d, heads = 2, 32, 64, 8
    n, B = h * h, 3 * 2
    x = torch.randn(B * n, d, generator=g).half().to(DEV)
    wq, wk, wv = (torch.randn(d, d, generator=g) / 8 for _ in range(3))
    wqkv = packing.pack_qkv(wq, wk, wv).half().to(DEV)
    wlin = packing.fold_fsai(wq, wk, 0.8).half().to(DEV)
    wo = (torch.randn(d, d, generator=g) / 8).half().to(DEV); bo = torch.randn(d, generator=g).to(DEV)
    flow = (torch.randn(F_ - 1, 2, h, h, generator=g) * 2).to(DEV)
    out = torch.empty(B * n, d, dtype=torch.float16, device=DEV)
    nws = hip.attn1_workspace_bytes(B, n, d, 3)
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    a256 = lambda b: (b + 255) // 256 * 256
    offT = a256(B * n * 3 * d * 2)
    first, bad = None, {"T": 0, "warped": 0, "out": 0, "qkv_other": 0}
    for it in range(iters):
        ws.zero_()
        hip.attn1_forward(x, wqkv, wlin, wo, bo, out, B=B, n=n, d=d, heads=heads, chunks=3, fusion=hip.FUSION_LINEAR, ldx=d, ldo=d,
                          workspace=ws, flow=flow, h=h, w=h, alpha=0.8)
        qkv = ws[:B * n * 3 * d * 2].view(torch.float16).view(B * n, 3 * d).clone()
        T = ws[offT:offT + F_ * n * 2 * d * 2].view(torch.float16).view(F_ * n, 2 * d).clone()
        cur = {"T": T, "warped": qkv[F_ * n:2 * F_ * n, :2 * d].clone(), "out": out.clone(),
               "qkv_other": torch.cat([qkv[:F_ * n].flatten(), qkv[2 * F_ * n:].flatten(), qkv[F_ * n:2 * F_ * n, 2 * d:].flatten()])}
        if first is None:
            first = cur
        else:
            for k in bad:
                bad[k] += (not torch.equal(cur[k], first[k]))
    q.put((rank, bad))

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=work, args=(r, 400, q)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(500)
    for _ in ps: print(q.get(timeout=5))
