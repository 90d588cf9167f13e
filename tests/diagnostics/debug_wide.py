"""Diagnostic (GPU): stage-by-stage error of the first ADMM iteration of the wide golden layers (g5b) vs fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from oracle import effq_oracle as O
from efficientq_amd.hip_ops import get_ops, make_geom, to_ndhwc
T = lambda a: torch.from_numpy(np.array(a)).clone()
tag = sys.argv[1] if len(sys.argv) > 1 else "c64"
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/g5b_wide_layers.npz"))
x, y, w, b = T(g[f"{tag}_x"]), T(g[f"{tag}_y"]), T(g[f"{tag}_w_in"]), T(g[f"{tag}_b_in"])
att = T(g[f"{tag}_mask_full"]).float()
c = w.shape[0]
ops = get_ops("cuda:0"); dev = "cuda:0"
fit = O.fit_scale(x, 4, 0, 1); xq = fit.alpha * fit.b
xn = to_ndhwc(x.to(dev)); yn = to_ndhwc(y.to(dev))
a, it, st = ops.fit_scale(xn, 4, 0.0, 1.0)
xq_g, _, xidx = ops.quant_dequant_f64path(xn, st, 4, 0.0, 1.0, want_idx=True)
print("alpha_act", a, fit.alpha, "xq equal:", torch.equal(xq_g.permute(0, 4, 1, 2, 3).cpu(), xq))
geom = make_geom(x.shape, c, 3, 1, 1)
cols = torch.from_numpy(np.ascontiguousarray(O.patch_matrix(xq.numpy(), (3, 3, 3), 1, 1, ones_row=True))).double()
xh = cols * att.reshape(1, -1).double()
ymat = torch.cat([s for s in y], dim=1).reshape(c, -1).double()
A64, B64 = 2 * cols @ xh.T, 2 * ymat @ xh.T
def rel(p, q): return ((p.double() - q).abs().max() / q.abs().max()).item()
A0f, B0f = ops.gram(xq_g, att.to(dev), yn, geom, True)
alpha_t = torch.tensor(a, dtype=torch.float32, device=dev)
A0i, B0i = ops.gram_i8(xidx, ops.att_classes(att.to(dev)), yn, geom, True, alpha_t, 4)
ps = O.ProxSystem(xq, y, (3, 3, 3), 1, 1, w, b, att)
print("A0 err vs fp64: f32-kernel %.2e  i8-kernel %.2e  reference-f32 %.2e" % (rel(A0f.cpu(), A64), rel(A0i.cpu(), A64), rel(ps.A0, A64)))
print("B0 err vs fp64: f32-kernel %.2e  i8-kernel %.2e  reference-f32 %.2e" % (rel(B0f.cpu(), B64), rel(B0i.cpu(), B64), rel(ps.B0, B64)))
rho_scale = max(y.numel() * y.std().item() / (w.numel() * w.std().item()), 1.0) * att.mean().item()
rho, eta = 10 * rho_scale, rho_scale
n = A64.shape[0]
def sysm(r):
    A = A64.clone(); d = torch.full((n,), r + eta, dtype=torch.float64); d[-1] = eta; A += torch.diag(d)
    Bm = B64 + eta * torch.cat([w.reshape(c, -1), b[:, None]], 1).double(); Bm[:, :-1] += r * w.reshape(c, -1).double()
    return A, Bm
A, Bm = sysm(rho)
W64 = torch.linalg.solve(A, Bm.T).T
W0 = w.to(dev).contiguous(); b0 = b.to(dev); z = torch.zeros_like(W0)
def lossof(ws, bs):
    fw = O.fit_scale(ws, 4, -1, 1)
    return F.mse_loss(F.conv3d(xq, fw.alpha * fw.b, bs, 1, 1), y).item()
print("fp64 everything: iter0 loss %.8f" % lossof(W64[:, :-1].float().reshape(w.shape), W64[:, -1].float()))
for gname, (At, Bt) in {"i8 gram": (A0i, B0i), "f32 gram": (A0f, B0f), "exact gram": (A64.float().to(dev), B64.float().to(dev))}.items():
    # direct inverse at rho0
    Ainv = ops.spd_inverse(At, True, rho, eta)
    Ai64 = torch.linalg.inv(A)
    ld = Ainv.shape[1]
    ws = torch.empty_like(W0); bs = torch.empty(c, device=dev)
    ops.prox_solve(Bt, Ainv, W0, b0, W0, z, rho, eta, ws, bs)
    got = torch.cat([ws.reshape(c, -1).cpu(), bs.cpu()[:, None]], 1).double()
    # shifted solve through the inverse of A(2 rho)
    Ainv2 = ops.spd_inverse(At, True, 2 * rho, eta)
    ws2 = torch.empty_like(W0); bs2 = torch.empty(c, device=dev)
    ops.prox_solve_shifted(Bt, Ainv2, W0, b0, W0, z, rho, eta, 2 * rho, ws2, bs2)
    got2 = torch.cat([ws2.reshape(c, -1).cpu(), bs2.cpu()[:, None]], 1).double()
    e = lambda q: ((q - W64).abs().max() / W64.abs().max()).item()
    eb = lambda q: ((q[:, -1] - W64[:, -1]).abs().max() / W64[:, -1].abs().max()).item()
    print(f"{gname:10s} Ainv err {rel(Ainv[:, :n].cpu(), Ai64):.2e} | direct: w* err {e(got):.2e} b* err {eb(got):.2e} loss {lossof(ws.cpu(), bs.cpu()):.8f}"
          f" | shifted: w* err {e(got2):.2e} b* err {eb(got2):.2e} loss {lossof(ws2.cpu(), bs2.cpu()):.8f}")
print("golden iter0: t1 %.8f t8 %.8f" % (g[f"{tag}_t1_loss_hist"][0], g[f"{tag}_t8_loss_hist"][0]))
