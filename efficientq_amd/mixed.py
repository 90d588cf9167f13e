"""Mixed-precision search harness (BASELINE config 5): per-layer weight levels
qlvl_w in {4, 8, 16} under an average-bits budget.  NEW relative to the reference, which only has one
global ``--qlvl_w`` plus the first/last overrides (model_blk.py:98-107); it reuses the calibrator unchanged:
``PTQConv.qlvl_w`` is read at calibration time, so a per-layer assignment is just an attribute map.

Procedure: (1) one uniform calibration per candidate level gives every layer's sensitivity (its
``layer_loss`` at that level, with the reference's sequential error compensation in effect);
(2) a greedy knapsack upgrades, starting from the lowest level everywhere, the layer with the best loss
reduction per extra stored bit until the budget is met; (3) the chosen map is calibrated and scored by
the agreement of the quantised with the FP prediction (the Dice proxy used throughout).
Different budgets are independent => "replicas only" across GPUs (no collective; one budget per rank).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Iterable, List, Sequence

import torch

from . import calibrate as K
from .qconv import PTQConv


def inner_layers(model) -> Dict[str, PTQConv]:
    """Quantised convs whose weight precision is searched (first/last keep their own 256-level setting)."""
    qs = [(n, m) for n, m in model.named_modules() if isinstance(m, PTQConv)]
    return {n: m for n, m in qs[1:-1]}


def apply_levels(model, level_map: Dict[str, int]):
    for n, m in inner_layers(model).items():
        if n in level_map:
            m.qlvl_w = int(level_map[n])


def prediction_agreement(res, task: str) -> float:
    q, f = res["output_q"][-1], res["output_fp"][-1]
    if task == "brats":
        return ((q > 0) == (f > 0)).float().mean().item()
    return (q.argmax(1) == f.argmax(1)).float().mean().item()


def sensitivities(build: Callable[[], torch.nn.Module], vols, task, init_stride, levels: Sequence[int]):
    """{level: {layer: layer_loss}} from one uniform calibration per level."""
    table = {}
    for L in levels:
        model = build()
        apply_levels(model, {n: L for n in inner_layers(model)})
        res = K.calibrate_model(model, vols, task, init_stride)
        table[L] = {l.split(":")[0].strip(): float(l.split(":")[1]) for l in res["layer_loss"]}
    return table


def greedy_assignment(sizes: Dict[str, int], table: Dict[int, Dict[str, float]], levels: Sequence[int],
                      avg_bits: float) -> Dict[str, int]:
    """Start at the lowest level; upgrade by best (loss drop)/(extra bits) while the average stays <= avg_bits."""
    levels = sorted(levels)
    bits = {L: math.log2(L) for L in levels}
    cur = {n: levels[0] for n in sizes}
    total = sum(sizes.values())
    used = sum(sizes[n] * bits[cur[n]] for n in sizes)
    budget = avg_bits * total
    while True:
        best, best_gain = None, 0.0
        for n in sizes:
            i = levels.index(cur[n])
            if i + 1 == len(levels):
                continue
            nxt = levels[i + 1]
            extra = sizes[n] * (bits[nxt] - bits[cur[n]])
            if used + extra > budget + 1e-9:
                continue
            gain = (table[cur[n]][n] - table[nxt][n]) / extra
            if gain > best_gain:
                best, best_gain = (n, nxt, extra), gain
        if best is None:
            return cur
        cur[best[0]] = best[1]
        used += best[2]


def search(build: Callable[[], torch.nn.Module], vols, task, init_stride, budgets: Iterable[float],
           levels: Sequence[int] = (4, 8, 16)) -> List[dict]:
    table = sensitivities(build, vols, task, init_stride, levels)
    sizes = {n: m.weight.numel() for n, m in inner_layers(build()).items()}
    out = []
    for b in budgets:
        lm = greedy_assignment(sizes, table, levels, b)
        model = build()
        apply_levels(model, lm)
        res = K.calibrate_model(model, vols, task, init_stride)
        total = sum(sizes.values())
        out.append(dict(budget_bits=b, avg_bits=sum(sizes[n] * math.log2(lm[n]) for n in sizes) / total,
                        agreement=prediction_agreement(res, task), levels=lm,
                        sum_layer_loss=sum(float(l.split(":")[1]) for l in res["layer_loss"]),
                        seconds=res["t2"] - res["t0"]))
    return out
