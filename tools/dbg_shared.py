import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd import hip
DEV = "cuda:0"
def ref_attn(q, k, v, heads, scale):
    B, n, d = q.shape; dh = d // heads
    sp = lambda t: t.reshape(B, n, heads, dh).permute(0, 2, 1, 3).float()
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, -1) @ sp(v)
    return a.permute(0, 2, 1, 3).reshape(B, n, d)
for dh, sets, n in [(40, 3, 512), (40, 3, 500), (40, 3, 448), (40, 3, 450), (40, 2, 500), (32, 3, 500), (40, 3, 100)]:
    Fr, heads = 2, 8; B = sets * Fr; d = heads * dh
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B, n, 3 * d, generator=g).half()
    qd = qkv.to(DEV)
    out = torch.zeros(B, n, d, dtype=torch.float16, device=DEV)
    hip.attention(qd, qd[:, :, d:], qd[:, :, 2 * d:], out, B=Fr, v_sets=sets, set_stride=Fr, heads=heads, n=n, nk=n, dh=dh,
                  ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=dh ** -0.5)
    idx = torch.arange(B) % Fr
    ref = ref_attn(qkv[idx, :, :d], qkv[idx, :, d:2 * d], qkv[..., 2 * d:], heads, dh ** -0.5)
    err = (out.cpu().float() - ref)
    per_sample = [(err[b].norm() / ref[b].norm()).item() for b in range(B)]
    bad_rows = (err.abs().amax(-1) > 0.02).nonzero()
    print(dh, sets, n, ["%.1e" % e for e in per_sample], "bad rows:", bad_rows[:6].tolist(), len(bad_rows),
          "nan" if torch.isnan(out).any() else "")
    if len(bad_rows):
        b, r = bad_rows[0].tolist()
        cols = (err[b, r].abs() > 0.02).nonzero().flatten().tolist()
        print("   first bad row cols:", cols[:20], len(cols))
