"""First-stage KL-VAE on the GPU: the HIP path against the fixture the reference's own Encoder / Decoder
produced (tests/golden/vae.npz), against the oracle, and size-independent properties at 512x512."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from vface_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CFG = {"small": dict(embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32,
                                                ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0))}


def _vae(tag, dtype=torch.float16):
    from vface_amd.ldm.models.autoencoder import FFHQ_VAE_CONFIG, AutoencoderKL
    cfg = FFHQ_VAE_CONFIG if tag == "ffhq" else CFG[tag]
    m = AutoencoderKL(**cfg, compute_dtype=dtype)
    synth.fill_module_(m, seed=0, prefix="vae.")
    return m.to(DEV)


@pytest.mark.parametrize("tag,res,nb", [("small", 32, 2), ("ffhq", 64, 1)])
def test_vae_vs_reference_golden(tag, res, nb):
    g = load_golden("vae")
    m = _vae(tag)
    assert sum(p.numel() for p in m.parameters()) == int(g[f"{tag}.n_params"])
    x = synth.synth_normal(f"vae.{tag}.x", (nb, 3, res, res)).clamp(-1, 1).to(DEV)
    noise = synth.synth_normal(f"vae.{tag}.noise", (nb, 4, res // 8, res // 8)).to(DEV)
    post = m.encode(x)
    z_mode = post.mode(0.18215).cpu()
    z_samp = post.sample(noise, 0.18215).cpu()
    e1, e2 = rel_l2(z_mode, g[f"{tag}.z_mode"]), rel_l2(z_samp, g[f"{tag}.z_sample"])
    dec = m.decode(g[f"{tag}.z_sample"].to(DEV) / 0.18215).cpu()
    e3 = rel_l2(dec, g[f"{tag}.dec"])
    # whole-network fp16 figures (30+ rounded layers each way), same scale as the whole-UNet ones; per-kernel parity is
    # asserted at 1e-3 in test_kernels_gpu.py and below.  The bound is the reference's OWN fp16 behaviour on the same inputs
    # (tests/golden/vae_lowp.npz, made by make_golden.py::gen_vae from the reference under torch.autocast(fp16)): this build
    # must be at least as close to the reference's fp32 output as the reference's shipped arithmetic is (+10 %: the one
    # case where the two coincide, the 32-channel z_sample at 8e-4).
    lp = load_golden("vae_lowp")
    ea = [rel_l2(lp[f"{tag}.{k}_autocast_f16"], g[f"{tag}.{k}"]) for k in ("z_mode", "z_sample", "dec")]
    ew = [rel_l2(lp[f"{tag}.{k}_w16"], g[f"{tag}.{k}"]) for k in ("z_mode", "z_sample", "dec")]
    print(f"VAE {tag}: z_mode {e1:.2e}  z_sample {e2:.2e}  decode {e3:.2e}   (reference under fp16 autocast: "
          f"{ea[0]:.2e} {ea[1]:.2e} {ea[2]:.2e}; reference with fp16-rounded weights only: {ew[0]:.2e} {ew[1]:.2e} {ew[2]:.2e})")
    for e, a in zip((e1, e2, e3), ea):
        assert e < 1.1 * a, (e, a)


def test_vae_bf16_measured():
    """bf16 storage is selectable (3 fewer mantissa bits than the reference's fp16 autocast): measured, loosely bounded."""
    g = load_golden("vae")
    m = _vae("small", torch.bfloat16)
    x = synth.synth_normal("vae.small.x", (2, 3, 32, 32)).clamp(-1, 1).to(DEV)
    e1 = rel_l2(m.encode(x).mode(0.18215).cpu(), g["small.z_mode"])
    e3 = rel_l2(m.decode(g["small.z_sample"].to(DEV) / 0.18215).cpu(), g["small.dec"])
    print(f"VAE small bf16: z_mode {e1:.2e}  decode {e3:.2e}")
    assert e1 < 3e-2 and e3 < 3e-2


def test_vae_cpu_tensors_fail_loudly():
    from vface_amd import hip
    m = _vae("small")
    with pytest.raises(hip.VFaceHipError):
        m.encode(torch.zeros(1, 3, 32, 32))
    with pytest.raises(hip.VFaceHipError):
        m.decode(torch.zeros(1, 4, 4, 4))
    with pytest.raises(RuntimeError):
        m.encoder(torch.zeros(1, 3, 32, 32, device=DEV))     # containers have no forward of their own


def test_softmax_rows_kernel():
    from vface_amd import hip
    for (M, N) in [(7, 16), (64, 4096), (3, 9216)]:
        s = synth.synth_normal(f"sm.{M}.{N}", (M, N)) * 3
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        hip.softmax_rows(s.to(DEV), out, M=M, N=N, scale=0.37)
        ref = torch.softmax(s * 0.37, -1)
        assert rel_l2(out.float().cpu(), ref) < 1e-3
        assert torch.allclose(out.float().sum(-1).cpu(), torch.ones(M), atol=2e-3)


def test_trailing_pad_stride2_conv():
    """Downsample (model.py:72-77): F.pad(0,1,0,1) + stride-2 padding-0 conv."""
    import math
    import torch.nn.functional as F
    from vface_amd import hip
    from vface_amd.packing import pack_conv3x3
    for cin, cout, H in [(64, 64, 16), (128, 128, 32), (24, 32, 10)]:
        x = synth.synth_normal(f"ds.x.{cin}", (2, cin, H, H)).half()
        w = (synth.synth_normal(f"ds.w.{cin}", (cout, cin, 3, 3)) / math.sqrt(9 * cin)).half()
        b = synth.synth_normal(f"ds.b.{cin}", (cout,))
        ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b, stride=2, padding=0)
        out = torch.empty(2, H // 2, H // 2, cout, dtype=torch.float16, device=DEV)
        hip.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(DEV), pack_conv3x3(w).to(DEV), out, nimg=2, H=H, W=H, cin=cin,
                    cout=cout, ldx=cin, ldy=cout, stride=2, bias=b.to(DEV), flags=hip.CONV_PAD_TRAILING)
        assert rel_l2(out.float().cpu().permute(0, 3, 1, 2), ref) < 1e-3


def test_vae_full_size_properties():
    """The shipped configuration at 512x512 (no oracle finishes there in test time): frames are independent, so a
    3-frame batch equals the frames run alone bit for bit (encode and decode); decode(encode(x).mode()) of the
    synthetic-weight model is finite and deterministic."""
    from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG  # noqa: F401
    m = _vae("ffhq")
    x = torch.stack([synth.synth_normal(f"vae.full.x.{f}", (3, 512, 512)).clamp(-1, 1) for f in range(3)]).to(DEV)
    post = m.encode(x)
    z = post.mode()
    assert z.shape == (3, 4, 64, 64) and torch.isfinite(z).all()
    z1 = m.encode(x[1:2]).mode()
    assert torch.equal(z1[0], z[1])
    dec = m.decode(z)
    assert dec.shape == (3, 3, 512, 512) and torch.isfinite(dec).all()
    d1 = m.decode(z[2:3])
    assert torch.equal(d1[0], dec[2])
    assert torch.equal(m.decode(z), dec)


def test_latent_diffusion_first_stage_surface():
    from vface_amd.ldm.models.autoencoder import FFHQ_VAE_CONFIG
    from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    tiny_unet = dict(image_size=8, in_channels=9, out_channels=4, model_channels=32, attention_resolutions=[4, 2, 1],
                     num_res_blocks=1, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                     transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
    ldm = LatentDiffusion(tiny_unet, first_stage_config=CFG["small"])
    synth.fill_module_(ldm.first_stage_model, seed=0, prefix="vae.")
    ldm = ldm.to(DEV)
    g = load_golden("vae")
    x = synth.synth_normal("vae.small.x", (2, 3, 32, 32)).clamp(-1, 1).to(DEV)
    noise = synth.synth_normal("vae.small.noise", (2, 4, 4, 4)).to(DEV)
    z = ldm.get_first_stage_encoding(ldm.encode_first_stage(x), noise=noise)
    assert rel_l2(z.cpu(), g["small.z_sample"]) < 2e-3
    dec = ldm.decode_first_stage(g["small.z_sample"].to(DEV))
    assert rel_l2(dec.cpu(), g["small.dec"]) < 3e-3
    assert any(k.startswith("first_stage_model.encoder.down.0.block.0.norm1") for k in ldm.state_dict())
