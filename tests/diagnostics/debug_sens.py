"""Diagnostic (GPU): where does the per-iteration loss gap vs the oracle come from?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from oracle import effq_oracle as O
from efficientq_amd.hip_ops import get_ops, make_geom, to_ndhwc
torch.set_num_threads(1)
ops = get_ops("cuda:0"); dev = "cuda:0"
gen = torch.Generator().manual_seed(2024)
c, S, N = 32, 12, 2
w = torch.randn(c, c, 3, 3, 3, generator=gen) * (2.0 / (c * 27)) ** 0.5
b = torch.randn(c, generator=gen) * 0.1
x_fp = torch.relu(torch.randn(N, c, S, S, S, generator=gen))
y = F.conv3d(x_fp, w, b, 1, 1)
x = torch.relu(x_fp + 0.05 * torch.randn(x_fp.shape, generator=gen))
att = torch.randint(1, 3, (N, S, S, S), generator=gen).float()
fit = O.fit_scale(x, 4, 0, 1)
xq = fit.alpha * fit.b
xn = to_ndhwc(x.to(dev)); yn = to_ndhwc(y.to(dev))
a, it, st = ops.fit_scale(xn, 4, 0.0, 1.0)
xq_g, _, _ = ops.quant_dequant_f64path(xn, st, 4, 0.0, 1.0)
print("alpha", a, fit.alpha, "xq equal:", torch.equal(xq_g.permute(0, 4, 1, 2, 3).cpu(), xq))
geom = make_geom(x.shape, c, 3, 1, 1)
A0, B0 = ops.gram(xq_g, att.to(dev), yn, geom, True)
ps = O.ProxSystem(xq, y, (3, 3, 3), 1, 1, w, b, att)
cols = torch.from_numpy(np.ascontiguousarray(O.patch_matrix(xq.numpy(), (3, 3, 3), 1, 1, ones_row=True))).double()
xh = cols * att.reshape(1, -1).double()
ymat = torch.cat([s for s in y], dim=1).reshape(c, -1).double()
A64, B64 = 2 * cols @ xh.T, 2 * ymat @ xh.T
def rel(a, b): return ((a.double() - b).abs().max() / b.abs().max()).item()
print("A0 err: gpu %.3e  ref32(1thr) %.3e" % (rel(A0.cpu(), A64), rel(ps.A0, A64)))
print("B0 err: gpu %.3e  ref32(1thr) %.3e" % (rel(B0.cpu(), B64), rel(ps.B0, B64)))
my = y.double(); rho_scale = max(y.numel() * y.std().item() / (w.numel() * w.std().item()), 1.0) * att.mean().item()
rho, eta = 10 * rho_scale, rho_scale
def iter0(A0t, B0t, tag):
    Ainv = ops.spd_inverse(A0t, True, rho, eta)
    W0 = w.to(dev).contiguous(); b0 = b.to(dev)
    ws = torch.empty_like(W0); bs = torch.empty(c, device=dev)
    ops.prox_solve(B0t, Ainv, W0, b0, W0, torch.zeros_like(W0), rho, eta, ws, bs)
    return ws.cpu(), bs.cpu()
A = A64.clone(); d = torch.full((A.shape[0],), rho + eta, dtype=torch.float64); d[-1] = eta; A += torch.diag(d)
Bm = B64 + eta * torch.cat([w.reshape(c, -1), b[:, None]], 1).double(); Bm[:, :-1] += rho * w.reshape(c, -1).double()
W64 = torch.linalg.solve(A, Bm.T).T
for tag, (At, Bt) in {"gpu-gram": (A0, B0), "exact-gram": (A64.float().to(dev), B64.float().to(dev)),
                      "ref32-gram": (ps.A0.to(dev), ps.B0.to(dev))}.items():
    ws, bs = iter0(At, Bt, tag)
    got = torch.cat([ws.reshape(c, -1), bs[:, None]], 1).double()
    fw = O.fit_scale(ws, 4, -1, 1)
    G = fw.alpha * fw.b
    loss = F.mse_loss(F.conv3d(xq, G, bs, 1, 1), y).item()
    print(f"{tag:12s} w* err vs fp64 {((got - W64).abs().max() / W64.abs().max()).item():.3e}  iter0 loss {loss:.8f}")
fw = O.fit_scale(W64[:, :-1].float().reshape(w.shape), 4, -1, 1)
print("fp64 everything iter0 loss", F.mse_loss(F.conv3d(xq, fw.alpha * fw.b, W64[:, -1].float(), 1, 1), y).item())
print("cond(A) ~", (torch.linalg.eigvalsh(A)[-1] / torch.linalg.eigvalsh(A)[0]).item())
print("---- mixing")
for tag, (At, Bt) in {"gpuA+exactB": (A0, B64.float().to(dev)), "exactA+gpuB": (A64.float().to(dev), B0)}.items():
    ws, bs = iter0(At, Bt, tag)
    fw = O.fit_scale(ws, 4, -1, 1)
    loss = F.mse_loss(F.conv3d(xq, fw.alpha * fw.b, bs, 1, 1), y).item()
    print(f"{tag:12s} iter0 loss {loss:.8f}")
E = (A0.cpu().double() - A64); Er = (ps.A0.double() - A64)
print("A0: |E|max gpu %.4e ref %.4e ; fro gpu %.4e ref %.4e ; |A|max %.4e" % (E.abs().max(), Er.abs().max(), E.norm(), Er.norm(), A64.abs().max()))
i = E.abs().argmax().item(); r, cc = divmod(i, A64.shape[0]); print("worst entry", r, cc, "val", A64[r, cc].item(), "err", E[r, cc].item())
print("bias row err gpu", E[-1].abs().max().item(), "ref", Er[-1].abs().max().item(), " diag err gpu", E.diag().abs().max().item(), "ref", Er.diag().abs().max().item())
relE = (E.abs() / A64.abs().clamp_min(1e-30)); print("max per-entry rel err gpu", relE.max().item(), "ref", (Er.abs() / A64.abs().clamp_min(1e-30)).max().item())
EB = (B0.cpu().double() - B64); EBr = (ps.B0.double() - B64)
print("B0: |E|max gpu %.4e ref %.4e ; fro gpu %.4e ref %.4e ; |B|max %.4e" % (EB.abs().max(), EBr.abs().max(), EB.norm(), EBr.norm(), B64.abs().max()))
print("mean signed err A gpu %.4e ref %.4e ; B gpu %.4e ref %.4e" % (E.mean(), Er.mean(), EB.mean(), EBr.mean()))
