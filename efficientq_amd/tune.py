"""``tune_activation_range`` of the reference (src/ptqer.py:238-272) on the HIP library: Adam (lr 5e-4) on every
``alpha_act``, end-to-end MSE between the quantised network output and the FP output, straight-through gradient through
``discretize`` (layer_helper.py:13-22).  It is dead code in the reference (no call site); parity is against the function
run in isolation (tests/golden/g12_tune_act.npz).

What runs where: quantised forward = fused act-quant conv kernel; input gradient of a quantised conv = the same conv
kernel on the output gradient; quantiser backward + its alpha-gradient reduction = effq_act_quant_backward; the optimiser
= effq_adam_step on ONE flat buffer holding every alpha_act.  torch autograd only chains these through the glue ops
between the quantised convs (ReLU, pooling, up-sampling, skip adds, the plain auxiliary heads), as row a11 keeps the host
graph in PyTorch.  Data parallel: the per-step gradient vector (one float per quantised conv) is all-reduced - the
"tiny quantizer-parameter gradients each step" of north_star.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn.functional as F

from .calibrate import set_init_alpha, set_quantized
from .qconv import PTQConv, SumReducer, get_ops


def tune_activation_range(model, output_fp, data_batch, max_iter: int = 1000, need_init: bool = False,
                          lr: float = 5e-4) -> List[float]:
    if need_init:                                                   # ptqer.py:250-252
        set_init_alpha(model)
        with torch.no_grad():
            model(data_batch)
    set_quantized(model)
    mods = [m for m in model.modules() if isinstance(m, PTQConv)]
    dev = data_batch.device
    ops = get_ops(dev)
    red = SumReducer()
    # every alpha_act becomes a view into one flat buffer: one Adam launch per step
    flat = torch.stack([m.alpha_act.data.reshape(()).float() for m in mods]).to(dev).contiguous()
    saved = [(p, p.requires_grad) for p in model.parameters()]
    for p, _ in saved:
        p.requires_grad_(False)
    for i, m in enumerate(mods):
        m.alpha_act.data = flat[i]
        m.alpha_act.requires_grad_(True)
    alphas = [m.alpha_act for m in mods]
    m_buf, v_buf = torch.zeros_like(flat), torch.zeros_like(flat)
    loss_all = []
    world = red.dist.get_world_size() if red else 1
    try:
        for it in range(max_iter):
            out_q = model(data_batch)
            loss = F.mse_loss(out_q, output_fp)
            grads = torch.autograd.grad(loss, alphas, allow_unused=True)
            g = torch.stack([torch.zeros((), device=dev) if gi is None else gi.reshape(()).float() for gi in grads])
            lv = loss.detach().reshape(1).double()
            if red:                       # shards hold different volumes: the mean of the shard losses / gradients
                red(g)
                g /= world
                red(lv)
                lv /= world
            # parameters without a gradient keep their value (torch.optim.Adam skips them; a zero gradient moves nothing)
            ops.adam_step(flat, g.contiguous(), m_buf, v_buf, lr, it + 1)
            loss_all.append(lv.item())
    finally:
        for m in mods:
            m.alpha_act.requires_grad_(True)
        for p, rg in saved:
            p.requires_grad_(rg)
    return loss_all
