import sys, os, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops
ops = get_ops("cuda:0")
n, L = 32769, 4
gen = torch.Generator().manual_seed(7 * n + L)
w = (torch.randn(n, generator=gen) * 0.05).cuda(); du = (torch.randn(n, generator=gen) * 0.005).cuda()
v = torch.empty(n, device="cuda:0")
prev = None
for call in range(5):
    st = ops.new_fp_state()
    ops.fixed_point_bucket(w, du, v, L, st)
    torch.cuda.synchronize()
    ws = ops._ws["fp_bucket"].cpu().numpy().tobytes()
    hint, hint_next, maxbits, t1, t2 = struct.unpack_from("<ffIII", ws, 0)
    tot = struct.unpack_from("<d", ws, 32)[0]
    base = 256 + 8 * 65536
    spre = torch.frombuffer(bytearray(ws[base: base + 8 * 4096]), dtype=torch.float64)
    base2 = base + 8 * (65536 + 2)
    stt = st.cpu()
    print(call, "alpha", float(stt[0]).hex(), "sums", float(stt[2]).hex(), float(stt[3]).hex(), "hint", hint, hint_next, maxbits, t1, t2, "tot", tot.hex(),
          "spre chk", float(spre.sum()).hex(), "same spre" if prev is not None and torch.equal(prev, spre) else "diff")
    prev = spre
