"""Mixed-precision search harness (BASELINE config 5): per-layer weight levels
qlvl_w in {4, 8, 16} under an average-bits budget.  NEW relative to the reference, which only has one
global ``--qlvl_w`` plus the first/last overrides (model_blk.py:98-107); it reuses the calibrator unchanged:
``PTQConv.qlvl_w`` (and ``qlvl_act``) are read at calibration time, so a per-layer assignment is just an attribute map.

Procedure:
 (1) SENSITIVITY, end to end: one calibration with every searched layer at the lowest level (the base), then one per
     (layer, higher level) with only that layer raised.  The score of a calibrated network is the relative MSE of its
     quantised output against the FP output (``output_error``) - NOT the sum of the per-layer ``layer_loss`` values,
     which are not comparable across level maps (every layer is calibrated against the FP target on the QUANTISED
     upstream's activations, EfficientQConv.py:41,68, so a better upstream changes what a layer's loss can reach).
     ``sensitivity="layer_loss"`` keeps the cheap variant (one uniform calibration per level).
 (2) a greedy knapsack upgrades, starting from the lowest level everywhere, the layer with the best error reduction per
     extra stored bit until the budget is met;
 (3) the chosen map is calibrated and scored (output error, FP-vs-Q prediction agreement = the Dice proxy).
``act_follows=True`` lets a layer's activation levels follow its weight levels.
Different budgets / sensitivity probes are independent => "replicas only" across GPUs (no collective).
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


def apply_levels(model, level_map: Dict[str, int], act_follows: bool = False):
    for n, m in inner_layers(model).items():
        if n in level_map:
            m.qlvl_w = int(level_map[n])
            if act_follows and m.q_act:
                m.qlvl_act = int(level_map[n])


def prediction_agreement(res, task: str) -> float:
    q, f = res["output_q"][-1], res["output_fp"][-1]
    if task == "brats":
        return ((q > 0) == (f > 0)).float().mean().item()
    return (q.argmax(1) == f.argmax(1)).float().mean().item()


def output_error(res) -> float:
    """Relative MSE of the quantised network's output (the head PTQ uses, ptqer.py:148) against the FP output."""
    q, f = res["output_q"][-1].double(), res["output_fp"][-1].double()
    return (((q - f) ** 2).mean() / (f ** 2).mean()).item()


def _calibrate(build, level_map, vols, task, init_stride, act_follows):
    model = build()
    apply_levels(model, level_map, act_follows)
    return K.calibrate_model(model, vols, task, init_stride)


def sensitivities(build: Callable[[], torch.nn.Module], vols, task, init_stride, levels: Sequence[int],
                  mode: str = "end_to_end", act_follows: bool = False):
    """{level: {layer: score}}: the score a layer contributes at that level (lower is better).
    end_to_end: output error of the network with ONLY that layer at the level, the others at the lowest one.
    layer_loss: the layer's own layer_loss in a uniform calibration at that level."""
    levels = sorted(levels)
    names = list(inner_layers(build()))
    table = {}
    if mode == "layer_loss":
        for L in levels:
            res = _calibrate(build, {n: L for n in names}, vols, task, init_stride, act_follows)
            table[L] = {l.split(":")[0].strip(): float(l.split(":")[1]) for l in res["layer_loss"]}
        return table
    if mode != "end_to_end":
        raise ValueError(f"unknown sensitivity mode {mode}")
    base = {n: levels[0] for n in names}
    e0 = output_error(_calibrate(build, base, vols, task, init_stride, act_follows))
    table[levels[0]] = {n: e0 for n in names}
    for L in levels[1:]:
        table[L] = {}
        for n in names:
            table[L][n] = output_error(_calibrate(build, dict(base, **{n: L}), vols, task, init_stride, act_follows))
    return table


def greedy_assignment(sizes: Dict[str, int], table: Dict[int, Dict[str, float]], levels: Sequence[int],
                      avg_bits: float) -> Dict[str, int]:
    """Start at the lowest level; upgrade by best (score drop)/(extra bits) while the average stays <= avg_bits."""
    levels = sorted(levels)
    bits = {L: math.log2(L) for L in levels}
    cur = {n: levels[0] for n in sizes}
    total = sum(sizes.values())
    used = sum(sizes[n] * bits[cur[n]] for n in sizes)
    budget = avg_bits * total
    while True:
        best, best_gain = None, None
        for n in sizes:
            i = levels.index(cur[n])
            if i + 1 == len(levels):
                continue
            nxt = levels[i + 1]
            extra = sizes[n] * (bits[nxt] - bits[cur[n]])
            if used + extra > budget + 1e-9:
                continue
            gain = (table[cur[n]][n] - table[nxt][n]) / extra
            # (a probe may come out marginally worse than the base - calibration is a discrete search - : such a layer is
            # upgraded last, but the budget is still spent: more bits never hurt the stored network)
            if best_gain is None or gain > best_gain:
                best, best_gain = (n, nxt, extra), gain
        if best is None:
            return cur
        cur[best[0]] = best[1]
        used += best[2]


def search(build: Callable[[], torch.nn.Module], vols, task, init_stride, budgets: Iterable[float],
           levels: Sequence[int] = (4, 8, 16), sensitivity: str = "end_to_end", act_follows: bool = False) -> List[dict]:
    table = sensitivities(build, vols, task, init_stride, levels, sensitivity, act_follows)
    sizes = {n: m.weight.numel() for n, m in inner_layers(build()).items()}
    total = sum(sizes.values())
    out = []
    for b in budgets:
        lm = greedy_assignment(sizes, table, levels, b)
        res = _calibrate(build, lm, vols, task, init_stride, act_follows)
        out.append(dict(budget_bits=b, avg_bits=sum(sizes[n] * math.log2(lm[n]) for n in sizes) / total,
                        agreement=prediction_agreement(res, task), output_error=output_error(res), levels=lm,
                        sum_layer_loss=sum(float(l.split(":")[1]) for l in res["layer_loss"]),
                        seconds=res["t2"] - res["t0"]))
    return out


def uniform(build: Callable[[], torch.nn.Module], vols, task, init_stride, level: int, act_follows: bool = False) -> dict:
    """The reference's own configuration space: one global qlvl_w (first / last layers keep theirs)."""
    names = list(inner_layers(build()))
    res = _calibrate(build, {n: level for n in names}, vols, task, init_stride, act_follows)
    return dict(level=level, avg_bits=math.log2(level), agreement=prediction_agreement(res, task),
                output_error=output_error(res), seconds=res["t2"] - res["t0"])
