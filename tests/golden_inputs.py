"""Seeded inputs shared by the golden generator (tests/golden/make_goldens.py) and the tests that replay its cases.

Test infrastructure only.  The large-volume layer cases (g5e: one 32 -> 32 3^3 layer on V >> n voxels) do not store their
tensors: they are rebuilt here from a seed, and the fixture keeps checksums and sub-samples to prove the rebuild.
The FP target `y = conv3d(x_fp, w, b)` is EXACT in fp32 whatever the summation order, by construction: `x_fp` is a
multiple of 2^-4 (< 8), `w` and `b` multiples of 2^-10 / 2^-14 (|w| < 0.25), so every product is a multiple of 2^-14 and
every partial sum stays far below 2^10 - 24 significant bits are never exceeded, and any conv implementation (oneDNN with
1 or 8 threads, the HIP kernels, fp64) returns the same bits.  That removes the FP forward as a source of differences
between the reference's runs and leaves the calibration itself."""
import torch
import torch.nn.functional as F

G5E_CASES = {
    # V = S^3 output voxels of ONE volume for n = 32*27 + 1 = 865 unknowns: V / n = 38 and 128 (g5b: V / n = 4)
    "s32": dict(S=32, seed=3032),
    "s48": dict(S=48, seed=3048),
}


def wide_layer_inputs(S, seed, c1=32, c2=32):
    """-> dict(w, b, x_fp, x, y, mask): start weights, FP input, quantised-upstream stand-in (quirk Q9), exact FP target,
    integer-valued attention mask (quirk Q1) of one 3^3 layer on a 1 x c1 x S^3 volume."""
    gen = torch.Generator().manual_seed(seed)
    sigma = 1.0 / (c1 * 27) ** 0.5
    w = torch.round(torch.randn(c2, c1, 3, 3, 3, generator=gen) * sigma * 1024).clamp_(-255, 255) / 1024
    b = torch.round(torch.randn(c2, generator=gen) * 0.1 * 16384) / 16384
    x_fp = torch.round(torch.relu(torch.randn(1, c1, S, S, S, generator=gen)) * 16).clamp_(max=127) / 16
    x = torch.relu(x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen))
    mask = torch.randint(1, 4, (1, S, S, S), generator=gen).float()
    y = F.conv3d(x_fp, w, b, 1, 1)
    return dict(w=w, b=b, x_fp=x_fp, x=x, y=y, mask=mask)


def checksums(t):
    """(sum, sum of squares, weighted sum) in fp64: equal only for equal tensors, for all practical purposes."""
    d = t.double().flatten()
    ramp = torch.arange(d.numel(), dtype=torch.float64) % 1009
    return torch.stack([d.sum(), (d * d).sum(), (d * ramp).sum()])
