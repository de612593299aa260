import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
from vface_amd.packing import pack_conv3x3
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (H, cin, cout) in [(8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280), (16, 2560, 1280), (32, 640, 640)]:
    xs = torch.randn(96, H, H, cin, generator=g).half().to(DEV)
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    outs = {}
    for nimg in (24, 48, 96):
        out = torch.zeros(nimg, H, H, cout, dtype=torch.float16, device=DEV)
        cs = torch.zeros(nimg * H * H // 64, cout, 2, device=DEV)
        rb = torch.randn(96, cout, generator=torch.Generator().manual_seed(1)).to(DEV)[:nimg].contiguous()
        hip.conv3x3(xs[:nimg].contiguous(), w, out, nimg=nimg, H=H, W=H, cin=cin, cout=cout, ldx=cin, ldy=cout, bias=b, rowbias=rb, colstats=cs)
        outs[nimg] = (out, cs)
    n24 = 24 * H * H // 64
    print(H, cin, cout, "ws bytes", [hip.load().vface_splitk_workspace_bytes(n * H * H, cout, 9 * cin, 0, H * H) for n in (24, 48, 96)],
          "out equal:", [torch.equal(outs[24][0], outs[n][0][:24]) for n in (48, 96)],
          "cs equal:", [torch.equal(outs[24][1], outs[n][1][:n24]) for n in (48, 96)])
M, N, K = 1536, 1280, 2560
a = torch.randn(4 * M, K, generator=g).half().to(DEV); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().to(DEV)
o = {}
for mult in (1, 2, 4):
    out = torch.zeros(mult * M, N, dtype=torch.float16, device=DEV)
    hip.gemm(a[:mult * M], w, out, M=mult * M, N=N, K=K, lda=K, ldc=N, rows_per_sample=64)
    o[mult] = out
print("skip gemm equal:", [torch.equal(o[1], o[m][:M]) for m in (2, 4)])
