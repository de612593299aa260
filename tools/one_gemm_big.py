#!/usr/bin/env python3
"""Run one launch shape of the 256 x 320 tile (csrc/gemm_big.hip) a few times: target for rocprofv3 --pmc passes (tools/pmc.sh).
usage: tools/pmc.sh gemm256_kernel tools/one_gemm_big.py [M N K [geglu|res16|plain]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vface_amd import hip  # noqa: E402

hip.load()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (24576, 10240, 1280)
kind = sys.argv[4] if len(sys.argv) > 4 else "geglu"
g = torch.Generator().manual_seed(0)
a = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
bias = torch.zeros(N, device="cuda")
out = torch.empty(M, N // 2 if kind == "geglu" else N, dtype=torch.float16, device="cuda")
kw = {"geglu": dict(flags=hip.EPI_GEGLU | hip.TUNE_BIG_TILE), "plain": dict(flags=hip.TUNE_BIG_TILE),
      "res16": dict(flags=hip.TUNE_BIG_TILE, residual=torch.zeros(M, N, dtype=torch.float16, device="cuda"), ldr=N)}[kind]
for _ in range(5):
    hip.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldc=out.shape[1], bias=bias, split_k=False, **kw)
torch.cuda.synchronize()
print("done")
