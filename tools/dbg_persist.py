import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (M, N, K, opts) in [(70000, 320, 320, "all"), (70000, 320, 320, "none"), (70016, 320, 320, "none"), (131072, 128, 64, "none"), (131072, 128, 128, "none"), (131072, 128, 256, "none")]:
    a = torch.randn(M, K, generator=g).half().to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).half().to(DEV) if opts == "all" else None
    outs = []
    for flags in (0, hip.TUNE_NO_PERSISTENT):
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        cs = torch.zeros((M + 63) // 64, N, 2, dtype=torch.float32, device=DEV) if opts == "all" else None
        hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=N, bias=b, residual=res, ldr=N, flags=flags, colstats=cs)
        outs.append((out, cs))
    d = (outs[0][0].float() - outs[1][0].float()).abs()
    bad = (d.amax(1) > 0).nonzero().flatten()
    ref = a.float() @ w.float().t() + b + (res.float() if res is not None else 0)
    e0 = ((outs[0][0].float() - ref).norm() / ref.norm()).item(); e1 = ((outs[1][0].float() - ref).norm() / ref.norm()).item()
    print(M, N, K, opts, "rows differing:", len(bad), bad[:8].tolist(), bad[-4:].tolist(), "err persist %.2e plain %.2e" % (e0, e1),
          "cs equal" if outs[0][1] is None else torch.equal(outs[0][1], outs[1][1]))
    if len(bad):
        r = bad[0].item(); cols = (d[r] > 0).nonzero().flatten()
        print("   row", r, "cols", cols[:10].tolist(), len(cols), "tile m", r // 128, "row in tile", r % 128)
