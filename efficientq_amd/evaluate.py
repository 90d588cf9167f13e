"""Row f1: quantised inference over overlapped patches and the FP-vs-quantised Dice proxy.

Mirrors the reference's evaluation path for the calibrated network - ``validate_seg``'s split / per-patch
forward / stitch (``utils/validate.py:212-264``, ``utils/transforms.py:784-852``) and ``validate_vs_label``
(``utils/metrics.py:119-148``) - without the NIfTI / data-loader / Dice-table machinery, which stays out of
scope.  The per-patch forward is the calibrated ``UResQ`` in quantized mode, i.e. every
conv runs ``conv3d_quant_calib_step`` with the activation quantiser fused (``PTQConv.py:163-167``).
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(i) for i in v)


def window_starts(size: int, patch: int, overlap: int) -> List[int]:
    """Start offsets along one axis (transforms.py:797-800): a step of patch-overlap while a whole patch still
    ends strictly before the border, then one patch flush with the border."""
    if patch > size:
        raise RuntimeError(f"patch {patch} larger than the image extent {size}")
    if overlap >= patch:
        raise RuntimeError("overlap must be smaller than the patch")
    return list(range(0, size - patch, patch - overlap)) + [size - patch]


def image_to_patch3d(images: torch.Tensor, patch_sz, overlap) -> List[torch.Tensor]:
    """Overlapped patches of an N x C x D x H x W batch in (d, h, w) raster order (transforms.py:784-810)."""
    if patch_sz is None or overlap is None:
        return images
    p, o = _triple(patch_sz), _triple(overlap)
    d, h, w = images.shape[-3:]
    return [images[..., i:i + p[0], j:j + p[1], k:k + p[2]]
            for i in window_starts(d, p[0], o[0]) for j in window_starts(h, p[1], o[1])
            for k in window_starts(w, p[2], o[2])]


def patch_to_image3d(images: torch.Tensor, patch_list: Sequence[torch.Tensor], patch_sz, overlap) -> torch.Tensor:
    """Stitch per-patch outputs (any leading dims, e.g. heads x batch x channels) back to the image grid: sum of
    the patches over the number of patches covering each voxel (transforms.py:812-852)."""
    if patch_sz is None or overlap is None:
        return images
    p, o = _triple(patch_sz), _triple(overlap)
    d, h, w = images.shape[-3:]
    first = patch_list[0]
    acc = torch.zeros(tuple(first.shape[:-3]) + (d, h, w), dtype=first.dtype, device=first.device)
    cnt = torch.zeros((d, h, w), dtype=torch.int32, device=first.device)
    n = 0
    for i in window_starts(d, p[0], o[0]):
        for j in window_starts(h, p[1], o[1]):
            for k in window_starts(w, p[2], o[2]):
                acc[..., i:i + p[0], j:j + p[1], k:k + p[2]] += patch_list[n]
                cnt[i:i + p[0], j:j + p[1], k:k + p[2]] += 1
                n += 1
    if n != len(patch_list):
        raise RuntimeError(f"{len(patch_list)} patches for a grid of {n}")
    return acc / cnt


def dice(pred_b: torch.Tensor, target_b: torch.Tensor) -> torch.Tensor:
    """metrics.py:21-25."""
    eps = 1e-6
    return (2 * (pred_b * target_b).sum().float() + eps) / (pred_b.sum().float() + target_b.sum().float() + eps)


def validate_vs_label(output: torch.Tensor, target: torch.Tensor, task: str = "lits"):
    """Dice of the hard predictions of `output` (NCDHW logits, or M x NCDHW for M heads) against `target`
    (metrics.py:119-148): per class for lits, background + per channel for brats."""
    if output.dim() >= 6:
        return [validate_vs_label(o, target, task) for o in output]
    if task == "lits":
        pred = torch.max(output, 1)[1]
        return [dice(pred == c, target == c) for c in range(output.shape[1])]
    if task == "brats":
        pred = (torch.sigmoid(output) >= 0.5).int()
        m = [dice(pred.sum(dim=1) == 0, target.sum(dim=1) == 0)]
        return m + [dice(pred[:, c], target[:, c]) for c in range(output.shape[1])]
    raise RuntimeError(f"Unknown task {task}")


@torch.no_grad()
def sliding_window_forward(model, images: torch.Tensor, patch_size=64, overlap=16) -> torch.Tensor:
    """validate_seg's inner loop (validate.py:236-245): split, run the model on every patch, stitch.
    `model(patch)` may return one tensor or a list of heads; the result is heads x N x C x D x H x W."""
    patches = image_to_patch3d(images, patch_size, overlap)
    preds = []
    for pt in patches:
        out = model(pt.contiguous())
        if isinstance(out, (list, tuple)):
            out = torch.stack(list(out))
        elif out.dim() == images.dim():          # a single head
            out = out.unsqueeze(0)
        preds.append(out)                        # heads x N x C x d x h x w (UResQ returns its heads stacked)
    return patch_to_image3d(images, preds, patch_size, overlap)


@torch.no_grad()
def fp_vs_quantised_dice(model_q, images: torch.Tensor, task: str, fp_model=None, fp_logits=None, patch_size=64,
                         overlap=16):
    """FP-vs-Q Dice proxy (no labels offline): the hard predictions of the FP network are the target of
    the calibrated network's.  Calibration overwrites the weights in place, so the FP side must come from before
    it: either `fp_model` (a copy of the network taken before calibration, run through the same sliding window)
    or `fp_logits` (its stitched last-head logits, e.g. do_ptq's output_fp[-1]).
    Returns (dice list of the last head, stitched quantised logits, FP logits)."""
    from . import calibrate as K
    if (fp_model is None) == (fp_logits is None):
        raise ValueError("give exactly one of fp_model / fp_logits")
    if fp_logits is None:
        K.set_fp(fp_model)
        fp_logits = sliding_window_forward(fp_model, images, patch_size, overlap)[-1]
    K.set_quantized(model_q)
    out_q = sliding_window_forward(model_q, images, patch_size, overlap)[-1]
    if task == "lits":
        target = torch.max(fp_logits, 1)[1]
    else:
        target = (torch.sigmoid(fp_logits) >= 0.5).int()
    return validate_vs_label(out_q, target, task), out_q, fp_logits
