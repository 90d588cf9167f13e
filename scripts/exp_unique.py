import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientq_amd.hip_ops import get_ops
dev = "cuda:0"; ops = get_ops(dev)
g = torch.Generator().manual_seed(0)
for rep in range(3):
    att = torch.randint(1, 3, (16, 64, 64, 64), generator=g).float().to(dev)
    torch.cuda.synchronize(); t0 = time.time()
    flat = att.reshape(-1)
    vals, inv = torch.unique(flat, return_inverse=True); k = int(vals.numel()); torch.cuda.synchronize(); t1 = time.time()
    counts = torch.bincount(inv, minlength=k); padded = (counts + 127) // 128 * 128
    order = torch.argsort(inv, stable=True); torch.cuda.synchronize(); t2 = time.time()
    starts = torch.cumsum(counts, 0) - counts; pstarts = torch.cumsum(padded, 0) - padded
    cls_sorted = inv[order]
    dest = pstarts[cls_sorted] + (torch.arange(flat.numel(), device=flat.device) - starts[cls_sorted]); torch.cuda.synchronize(); t3 = time.time()
    lst = torch.full((int(padded.sum().item()),), -1, dtype=torch.int32, device=flat.device)
    lst[dest] = order.to(torch.int32); torch.cuda.synchronize(); t4 = time.time()
    chunk_cls = torch.repeat_interleave(torch.arange(k, dtype=torch.int32, device=flat.device), padded // 128); torch.cuda.synchronize(); t5 = time.time()
    print(f"unique {1e3*(t1-t0):.2f} sort {1e3*(t2-t1):.2f} dest {1e3*(t3-t2):.2f} scatter {1e3*(t4-t3):.2f} repeat {1e3*(t5-t4):.2f} ms")
    t0 = time.time(); r = ops.att_classes(att); torch.cuda.synchronize(); print(f"att_classes {1e3*(time.time()-t0):.2f} ms")
