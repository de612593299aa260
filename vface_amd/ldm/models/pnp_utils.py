"""The VFace hook surface: ``register_spa_attn_injection`` with the reference's signature and selection rule
(``REFace/ldm/models/pnp_utils.py:57,289-339``).

As in the reference, registration walks ``model.model.model.diffusion_model.{input,output,middle}_blocks``
(``model`` is the *sampler*), finds sub-modules whose name ends in ``attn_component`` and overwrites the
instance attribute ``module.forward`` with a closure ``forward(x, context=None, mask=None,
feature_transfer=True)`` that captures the arguments; a module that is not selected by ``block_indices`` keeps
its previous closure and only gets ``injection_schedule`` set (pnp_utils.py:292-304).  The closure also leaves
its captured configuration on the module (``_vface_cfg``) so the UNet engine can execute the hook as fused
HIP kernels instead of calling back into Python per layer.  Re-registration every DDIM step
(ddim_w_inv.py:303,305) is cheap: folded weights are cached per (module, fusion, ratio).
"""
from __future__ import annotations

from typing import List, Tuple

from ...engine import HookCfg, _dev_flow


def find_all_modules_by_name(model, mod_name) -> Tuple[List, List[str]]:
    """pnp_utils.py:33-40."""
    modules, names = [], []
    for name, module in model.named_modules():
        if name.endswith(mod_name):
            modules.append(module)
            names.append(name)
    return modules, names


def _make_forward(module, cfg: HookCfg):
    def forward(x, context=None, mask=None, feature_transfer=True):
        use = cfg if (feature_transfer and cfg.switch_on) else None
        return type(module).forward(module, x, context=context, mask=mask, _cfg=use)
    forward._vface = True
    return forward


# Default of the extra ``flow_gate`` argument below ("reference" | "flow_hw", see engine.HookCfg.flow_gate): the reference's
# own `q.shape[1] == 4096` test.  A sampler may carry its own default as ``sampler.flow_gate``.
DEFAULT_FLOW_GATE = "reference"


def register_spa_attn_injection(model, injection_schedule, switch_on=True, input_blocks=False, output_blocks=True,
                                middle_block=False, attn_component='attn1', chunks=3, flow=None, block_indices=None,
                                fusion="replace", split_ratio_fft=0.8, alpha=0.8, flow_gate=None):
    """Positional / keyword arguments of the reference (pnp_utils.py:57); ``flow_gate`` is this package's only addition."""
    unet = model.model.model.diffusion_model
    dev = next(unet.parameters()).device
    cfg = HookCfg(switch_on=switch_on, chunks=chunks, fusion=fusion, flow=_dev_flow(flow, dev),
                  split_ratio_fft=split_ratio_fft, alpha=alpha,
                  flow_gate=flow_gate or getattr(model, "flow_gate", None) or DEFAULT_FLOW_GATE)
    processed_all = {}
    for enabled, group in ((input_blocks, "input_blocks"), (output_blocks, "output_blocks"),
                           (middle_block, "middle_block")):
        if not enabled:
            continue
        modules, names = find_all_modules_by_name(getattr(unet, group), attn_component)
        processed = []
        for i, module in enumerate(modules):
            if block_indices is None or i in block_indices:
                module._vface_cfg = cfg
                module.forward = _make_forward(module, cfg)
                processed.append(names[i])
                continue
            setattr(module, "injection_schedule", injection_schedule)
        processed_all[group] = processed
    return processed_all
