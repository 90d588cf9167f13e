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


def gauss_int(gen, *shape):
    """Near-normal variates (sum of four uniform integers, unit variance to 0.03 %) built from torch.randint and exact
    fp32 arithmetic only.  torch.randn is NOT bit-reproducible between hosts (its vectorised log / cos differ in the last
    bit on a few elements between CPU generations: the first version of this fixture failed its checksums on the GPU
    box's host for exactly that reason); the integer generator and IEEE + - * / are."""
    k = torch.zeros(shape, dtype=torch.int64)
    for _ in range(4):
        k += torch.randint(-1024, 1025, shape, generator=gen)
    return k.float() / 1183.5          # |k| <= 4096: exact in fp32; one correctly rounded division


def wide_layer_inputs(S, seed, c1=32, c2=32):
    """-> dict(w, b, x_fp, x, y, mask): start weights, FP input, quantised-upstream stand-in (quirk Q9), exact FP target,
    integer-valued attention mask (quirk Q1) of one 3^3 layer on a 1 x c1 x S^3 volume."""
    gen = torch.Generator().manual_seed(seed)
    sigma = 1.0 / (c1 * 27) ** 0.5
    w = torch.round(gauss_int(gen, c2, c1, 3, 3, 3) * sigma * 1024).clamp_(-255, 255) / 1024
    b = torch.round(gauss_int(gen, c2) * 0.1 * 16384) / 16384
    x_fp = torch.round(torch.relu(gauss_int(gen, 1, c1, S, S, S)) * 16).clamp_(max=127) / 16
    x = torch.relu(x_fp + 0.05 * gauss_int(gen, *x_fp.shape))
    mask = torch.randint(1, 4, (1, S, S, S), generator=gen).float()
    y = F.conv3d(x_fp, w, b, 1, 1)
    return dict(w=w, b=b, x_fp=x_fp, x=x, y=y, mask=mask)


def checksums(t):
    """Order-independent checksums of an fp32 tensor: the int64 sum of its bit patterns, and the same weighted by a ramp.
    (fp64 sums of the VALUES depend on the summation order, which differs between hosts with different vector widths:
    the first version of this function failed on the GPU box's host for equal tensors.)"""
    bits = t.contiguous().view(torch.int32).flatten().to(torch.int64)
    ramp = torch.arange(bits.numel(), dtype=torch.int64) % 1009
    return torch.stack([bits.sum(), (bits * ramp).sum()])
