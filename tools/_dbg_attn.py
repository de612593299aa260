import sys, torch
sys.path.insert(0, '/root/repo')
from vface_amd import hip as h
h.load()
DEV='cuda'; dt=torch.float16
def rnd(shape, seed, dt, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt)
def ref_attn(q,k,v,heads,scale,qk_map):
    B,n,d=v.shape; dh=d//heads
    q=q[qk_map]; k=k[qk_map]
    sp=lambda t: t.reshape(B,n,heads,dh).permute(0,2,1,3).double()
    P=(sp(q)@sp(k).transpose(-1,-2)*scale).softmax(-1)
    return (P@sp(v)).permute(0,2,1,3).reshape(B,n,d).float()
for dh in (40, 80, 160, 8, 16, 32):
  for sets in ((3, 2, 1) if dh in (8,16,32,40) else (1,)):
    for n in (512, 500):
        Fr, heads = 2, 8
        B=sets*Fr; d=heads*dh
        qkv = rnd((B, n, 3 * d), 11, dt); qkv[:, n // 2, d:2 * d] *= 4.0
        qd = qkv.to(DEV)
        kw = dict(heads=heads, n=n, nk=n, dh=dh, ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=dh ** -0.5)
        qk_map=(torch.arange(B)%Fr)
        ref=ref_attn(qkv[...,:d].float(),qkv[...,d:2*d].float(),qkv[...,2*d:].float(),heads,dh**-0.5,qk_map)
        res=[]
        for variant in (0, 1, 2, 8, 9, 16, 17, 24):
            out = torch.zeros(B, n, d, dtype=dt, device=DEV)
            try:
                if sets>1: h.attention(qd, qd[:, :, d:], qd[:, :, 2 * d:], out, B=Fr, v_sets=sets, set_stride=Fr, variant=variant, **kw)
                else: h.attention(qd, qd[:, :, d:], qd[:, :, 2 * d:], out, B=B, variant=variant, **kw)
                torch.cuda.synchronize()
                res.append(f"{float((out.cpu().float()-ref).norm()/ref.norm()):.1e}")
            except Exception as e: res.append("n/a")
        bad = any(r!="n/a" and float(r)>2e-3 for r in res)
        print(f"dh {dh} sets {sets} n {n}: variants (0,1,2,8,9,16,17,24) {res} {'  <-- BAD' if bad else ''}")
