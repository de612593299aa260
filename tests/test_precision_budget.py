"""The whole-network bound above north_star's 1e-3 is DERIVED, not chosen: an emulation of this build's
rounding points on the CPU oracle (tests/precision_budget.py: fp16 weights, one rounding per matrix-core operand, fp32
accumulation / statistics / residual stream) must reproduce the error the MI355X measures on the 859.5 M-parameter UNet, and
splits it into its shares.  CPU only; the GPU side of the same statement is tests/test_unet_gpu.py (FULL_BOUND, bf16)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import precision_budget as pb  # noqa: E402
from oracle import unet as ounet  # noqa: E402
from vface_amd.utils import synth  # noqa: E402

# rel-L2 of the HIP path against the reference's fp32 fixture, full UNet, plain / flow_fix / replace: GPUTEST_r03.json and every
# run of tests/test_unet_gpu.py::test_full_unet_vs_reference_golden since round 2 (1.172e-3 .. 1.184e-3)
MEASURED_FULL = 1.18e-3
# the reference itself with nothing but its parameters rounded to fp16 (tests/golden/lowp.npz, pinned by test_oracle_golden.py)
REFERENCE_W16_FLOOR = 9.6e-4


@pytest.fixture(scope="module")
def full_case():
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    spec = ounet.UNetSpec()
    sd = synth.synth_state_dict(ounet.param_shapes(spec), seed=0)
    # one frame's three samples of the GPU test's inputs (hooks off: samples are independent)
    x = synth.synth_normal("full.x", (6, 9, 64, 64))[:3].contiguous()
    ctx = synth.synth_normal("full.ctx", (6, 1, 768))[:3].contiguous()
    t = torch.full((3,), 481, dtype=torch.long)
    with torch.no_grad():
        ref = ounet.unet_forward(sd, spec, x, t, ctx, {})
    return spec, sd, x, ctx, t, ref


def _emulate(case, **kw):
    spec, sd, x, ctx, t, ref = case
    with torch.no_grad():
        return pb.rel(pb.forward(pb.Emu(sd, **kw), spec, x, t, ctx, {}), ref)


def test_emulated_rounding_points_reproduce_the_measured_whole_unet_error(full_case):
    total = _emulate(full_case, w16=True, act16=True, stream16=False)        # this build: fp32 residual stream
    weights = _emulate(full_case, w16=True, act16=False, stream16=False)
    acts = _emulate(full_case, w16=False, act16=True, stream16=False)
    print(f"emulated whole-UNet error of this build's rounding points: {total:.3e} (measured on the MI355X: {MEASURED_FULL:.3e})")
    print(f"  fp16 weights alone                       {weights:.3e}  (the reference's own floor: {REFERENCE_W16_FLOOR:.2e})")
    print(f"  one rounding per matrix-core operand     {acts:.3e}")
    print(f"  quadrature of the two                    {(weights ** 2 + acts ** 2) ** 0.5:.3e}")
    # the emulation is the measurement, to 15 %: the bound asserted on the GPU (FULL_BOUND = 1.35e-3) is this figure + margin,
    # not a number picked to pass
    assert abs(total - MEASURED_FULL) / MEASURED_FULL < 0.15, total
    # the shares are independent noise sources: they add in quadrature (to 10 %)
    assert abs((weights ** 2 + acts ** 2) ** 0.5 - total) / total < 0.10
    # the weight share IS the reference's own floor: any implementation that feeds fp16 weights to the matrix cores starts there
    assert abs(weights - REFERENCE_W16_FLOOR) / REFERENCE_W16_FLOOR < 0.10
    # so north_star's 1e-3 would leave this much for ~120 operand roundings in series -- less than what ONE rounding per
    # operand costs: the literal tolerance is not reachable with 16-bit operands, with this or any other kernel set
    assert (1e-3 ** 2 - weights ** 2) ** 0.5 < acts


def test_sixteen_bit_residual_stream_would_cost_what_round_one_measured(full_case):
    """Round 1 (all activations 16-bit, the reference's autocast layout) measured 1.45e-3; the fp32 stream bought 1.18e-3."""
    r1 = _emulate(full_case, w16=True, act16=True, stream16=True)
    print(f"emulated 16-bit residual stream: {r1:.3e} (round 1 measured 1.45e-3)")
    assert abs(r1 - 1.45e-3) / 1.45e-3 < 0.15


def test_smoke_bound_is_the_emulations():
    """`__graft_entry__.smoke()` asserts 1.25 x EMULATED_REL_L2: that constant must be what the emulation gives on the smoke case
    itself (small UNet, 16 x 16 latent, flow_fix), not a round number."""
    from oracle import hooks as ohooks
    from vface_amd import smoke
    spec, F_, h, w, x, ctx, t, flow = smoke.smoke_case()
    sd = synth.synth_state_dict(ounet.param_shapes(spec), seed=0)

    def reg():
        r = {}
        ohooks.register_spa_attn_injection(r, ounet.attn1_names(spec), 1, switch_on=True, input_blocks=True, middle_block=False,
                                           output_blocks=False, flow=[flow[i][None] for i in range(F_ - 1)], chunks=3,
                                           block_indices=list(range(9)), fusion="flow_fix")
        return r
    with torch.no_grad():
        ref = ounet.unet_forward(sd, spec, x, t, ctx, reg())
        emu = pb.rel(pb.forward(pb.Emu(sd, w16=True, act16=True, stream16=False), spec, x, t, ctx, reg()), ref)
    print(f"emulated smoke case: {emu:.3e} (constant {smoke.EMULATED_REL_L2:.3e}, bound {smoke.SMOKE_BOUND:.3e})")
    assert abs(emu - smoke.EMULATED_REL_L2) / smoke.EMULATED_REL_L2 < 0.02
    # the bound follows the emulation only UNDER a fixed ceiling, and the emulated figure itself is held under a fixed one: a change that
    # adds rounding points cannot raise its own tolerance
    assert smoke.SMOKE_BOUND == min(smoke.SMOKE_CEILING, 1.25 * smoke.EMULATED_REL_L2) and smoke.SMOKE_CEILING == 1.5e-3
    assert emu < 1.30e-3, emu
