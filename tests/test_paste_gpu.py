"""Paste-back on the GPU (REFace/scripts/VFace_inference_batch.py:597-636): every kernel of csrc/paste.hip against
Pillow / numpy / torch doing what the reference's lines do, bit for bit where the arithmetic is 8-bit, and the whole
`PasteBack.paste` against the reference's statement sequence run on the host."""
import numpy as np
import pytest
import torch

from oracle import paste as opaste

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _coeffs(src_quad, dst_quad):
    A, B = [], []
    for (x, y), (u, v) in zip(dst_quad, src_quad):
        A += [[x, y, 1, 0, 0, 0, -u * x, -u * y], [0, 0, 0, x, y, 1, -v * x, -v * y]]
        B += [u, v]
    return np.linalg.solve(np.array(A, float), np.array(B, float))


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
def test_frame_to_u8_is_the_references_clamp_scale_truncate(dt):
    from vface_amd import hip
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(3, 3, 37, 53, generator=g) * 0.9).to(dt)
    x[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, 1 / 255, -0.9999, 0.9999]).to(dt)
    got = hip.frame_to_u8(x.to(DEV)).cpu().numpy()
    xs = torch.clamp((x.float() + 1.0) / 2.0, min=0.0, max=1.0).permute(0, 2, 3, 1).numpy()        # :597-598
    ref = (255. * xs).astype(np.uint8)                                                              # :606-608
    assert np.array_equal(got, ref)


def test_frame_to_u8_half_arithmetic_is_the_autocast_paths_float16_rounding():
    """under the reference's default ``--precision autocast`` the decoded frames are float16 and :597-608 round every
    operation to float16; ``half_arithmetic=True`` restates that (against torch + numpy run on the float16 tensor, as the
    reference runs them, and against the oracle), and does differ from the fp32 arithmetic in some pixels."""
    from vface_amd import hip
    from oracle import paste as op
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(2, 3, 64, 96, generator=g) * 0.9).half()
    x[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, 1 / 255, -0.9999, 0.9999]).half()
    got = hip.frame_to_u8(x.to(DEV), half_arithmetic=True).cpu().numpy()
    xs = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0).permute(0, 2, 3, 1).numpy()                # float16 all the way
    assert xs.dtype == np.float16
    ref = (255. * xs).astype(np.uint8)
    assert np.array_equal(got, ref)
    assert np.array_equal(got, op.to_u8_half(op.clamp01_half(x.permute(0, 2, 3, 1).numpy())))
    full = hip.frame_to_u8(x.to(DEV)).cpu().numpy()
    d = np.abs(got.astype(int) - full.astype(int))
    assert d.max() == 1 and 0 < (d > 0).mean() < 0.5
    with pytest.raises(hip.VFaceHipError):
        hip.frame_to_u8(x.float().to(DEV), half_arithmetic=True)


@pytest.mark.parametrize("h,w,ow,oh", [(64, 64, 128, 128), (50, 70, 33, 91), (37, 41, 100, 17), (128, 96, 96, 128), (512, 512, 1024, 1024)])
def test_resize_u8_is_pillows_bilinear_resize(h, w, ow, oh):
    from PIL import Image
    from vface_amd.scripts.paste_back import PasteBack
    rng = np.random.default_rng(h + w)
    frames = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    pb = PasteBack(device=DEV)
    got = pb.resize_u8(torch.from_numpy(frames).to(DEV), ow, oh).cpu().numpy()
    for f in range(2):
        ref = np.asarray(Image.fromarray(frames[f]).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(got[f], ref)


@pytest.mark.parametrize("quad", [[(30.3, 20.1), (110.7, 25.2), (115.1, 100.9), (25.5, 95.5)],
                                  [(-20, -10), (100, 5), (90, 140), (10, 90)],
                                  [(0, 0), (160, 0), (160, 120), (0, 120)]])
def test_perspective_paste_is_pillows_transform_and_alpha_composite(quad):
    from PIL import Image
    from vface_amd import hip
    rng = np.random.default_rng(7)
    sw = sh = 96
    crop = rng.integers(0, 256, (2, sh, sw, 3), dtype=np.uint8)
    bg = rng.integers(0, 256, (2, 120, 160, 3), dtype=np.uint8)
    cos = [_coeffs([(0, 0), (sw, 0), (sw, sh), (0, sh)], quad),
           _coeffs([(0, 0), (sw, 0), (sw, sh), (0, sh)], [(x + 7.25, y - 3.5) for x, y in quad])]
    refs = []
    for f in range(2):
        s = Image.fromarray(crop[f]).convert("RGBA")
        pasted = Image.fromarray(bg[f]).convert("RGBA")
        s.putalpha(255)
        pasted.alpha_composite(s.transform((160, 120), Image.PERSPECTIVE, cos[f], Image.BILINEAR))
        refs.append(np.asarray(pasted)[..., :3])
    # device coefficients, both frames in one launch
    frame = torch.from_numpy(bg).to(DEV)
    hip.perspective_paste(torch.from_numpy(crop).to(DEV), frame, torch.from_numpy(np.stack(cos)).to(DEV))
    got = frame.cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], refs[f])
        assert np.array_equal(got[f], opaste.perspective_paste(crop[f], bg[f], cos[f]))
    # host coefficients, one frame
    one = torch.from_numpy(bg[:1]).to(DEV)
    hip.perspective_paste(torch.from_numpy(crop[:1]).to(DEV), one, cos[0])
    assert np.array_equal(one.cpu().numpy()[0], refs[0])
    with pytest.raises(hip.VFaceHipError):
        hip.perspective_paste(torch.from_numpy(crop).to(DEV), frame, cos[0])        # eight host numbers cannot serve two frames


def test_frame_normalise_resize_is_totensor_normalize_resize():
    from vface_amd import hip
    rng = np.random.default_rng(5)
    fr = rng.integers(0, 256, (2, 150, 201, 3), dtype=np.uint8)
    for oh, ow in ((64, 64), (150, 201), (300, 333)):
        got = hip.frame_normalise_resize(torch.from_numpy(fr).to(DEV), oh, ow).cpu()
        t = (torch.from_numpy(fr).permute(0, 3, 1, 2).float().div(255) - 0.5) / 0.5          # get_tensor() (:48-56)
        ref = torch.nn.functional.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=False)   # transforms.Resize on a tensor
        assert (got - ref).abs().max().item() <= 2e-6
        orc = np.stack([opaste.resize_bilinear_f32(opaste.normalise_frame(fr[f]), oh, ow) for f in range(2)])
        assert np.array_equal(got.numpy(), orc)              # the oracle's operation order, bit for bit


def test_paste_back_equals_the_references_host_sequence():
    """`PasteBack.paste` against VFace_inference_batch.py:597-636 executed statement by statement with Pillow on the host, the
    VAE round trip replaced on both sides by the same deterministic stand-in (a per-channel affine map: the first stage has
    its own parity tests), F = 3 square frames, a different quad per frame."""
    from PIL import Image
    from vface_amd.scripts.paste_back import PasteBack
    rng = np.random.default_rng(11)
    F_, H, W, S = 3, 64, 64, 200
    dec = (torch.from_numpy(rng.standard_normal((F_, 3, H, W)).astype(np.float32)) * 0.8)
    frames = rng.integers(0, 256, (F_, S, S, 3), dtype=np.uint8)
    canvas = 128
    quads = [[(20.5 + 3 * f, 30.25), (150.0, 25.0 + f), (160.5, 170.0), (15.0, 165.5 - 2 * f)] for f in range(F_)]
    cos = np.stack([_coeffs([(0, 0), (canvas, 0), (canvas, canvas), (0, canvas)], q) for q in quads])
    stand_in = lambda x: (x * 0.9 + 0.05).clamp(-1.2, 1.2)

    pb = PasteBack(H=H, W=W, canvas=canvas, device=DEV, encode_decode=stand_in)
    got = pb.paste(dec.to(DEV), torch.from_numpy(frames).to(DEV), cos).cpu().numpy()

    x_samples = torch.clamp((dec + 1.0) / 2.0, min=0.0, max=1.0).permute(0, 2, 3, 1).numpy()            # :597-598
    for i in range(F_):
        img = Image.fromarray((255. * x_samples[i]).astype(np.uint8)).resize((canvas, canvas), Image.BILINEAR)      # :606-608
        orig = Image.fromarray(frames[i])
        t = (torch.from_numpy(frames[i]).permute(2, 0, 1).float().div(255) - 0.5) / 0.5                 # :611
        t = torch.nn.functional.interpolate(t[None], size=(H, W), mode="bilinear", align_corners=False)   # :612
        rec = torch.clamp((stand_in(t) + 1.0) / 2.0, min=0.0, max=1.0).permute(0, 2, 3, 1).numpy()       # :615-619
        rec = Image.fromarray((255. * rec[0]).astype(np.uint8)).convert("RGB")                           # :620
        conv = rec.resize((S, S), Image.BILINEAR)                                                        # :621
        swapped = img.convert("RGBA")                                                                    # :629
        pasted = conv.convert("RGBA")
        swapped.putalpha(255)
        pasted.alpha_composite(swapped.transform(orig.size, Image.PERSPECTIVE, cos[i], Image.BILINEAR))  # :632-633
        ref = np.asarray(pasted)[..., :3]
        diff = np.abs(ref.astype(int) - got[i].astype(int))
        # the float resize before the stand-in differs from ATen's CPU kernel by <= 2 ulp, which can move a background value
        # across an 8-bit truncation boundary: at most a handful of +-1 counts there, none inside the pasted quad
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())


def test_paste_back_refuses_host_tensors_and_non_square_frames():
    from vface_amd import hip
    from vface_amd.scripts.paste_back import PasteBack
    pb = PasteBack(H=32, W=32, canvas=64, device=DEV, encode_decode=lambda x: x)
    dec = torch.zeros(1, 3, 32, 32)
    with pytest.raises(hip.VFaceHipError):
        pb.paste(dec, torch.zeros(1, 80, 80, 3, dtype=torch.uint8, device=DEV), np.zeros(8))
    with pytest.raises(ValueError, match="images do not match"):       # :621 swaps width and height (see PasteBack.background)
        pb.paste(dec.to(DEV), torch.zeros(1, 80, 100, 3, dtype=torch.uint8, device=DEV), np.array([1, 0, 0, 0, 1, 0, 0, 0.0]))


def test_kernels_match_the_committed_pillow_fixture():
    """The HIP kernels against tests/golden/paste.npz (Pillow's outputs, committed): resize and perspective paste, bit for bit."""
    import os
    from vface_amd import hip
    from vface_amd.scripts.paste_back import PasteBack
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "paste.npz"), allow_pickle=False)
    pb = PasteBack(device=DEV)
    for i in range(5):
        ow, oh = (int(v) for v in z[f"resize{i}.size"])
        got = pb.resize_u8(torch.from_numpy(z[f"resize{i}.in"])[None].to(DEV), ow, oh)[0].cpu().numpy()
        assert np.array_equal(got, z[f"resize{i}.out"])
    for i in range(3):
        frame = torch.from_numpy(z["persp.bg"])[None].contiguous().to(DEV)
        hip.perspective_paste(torch.from_numpy(z["persp.crop"])[None].contiguous().to(DEV), frame, z[f"persp{i}.coeffs"])
        assert np.array_equal(frame[0].cpu().numpy(), z[f"persp{i}.out"][..., :3])
