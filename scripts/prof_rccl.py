"""Cost of the data-parallel collectives on ONE rank (a 1-rank RCCL group): the in-stream ncclAllReduce of rccl.py against
torch.distributed.all_reduce, tiny (16 B: the per-iteration sums) and large (the packed Gram system of n = 6913), and the
pack / unpack kernels around the large one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1)
from efficientq_amd import rccl
from efficientq_amd.hip_ops import get_ops
ops = get_ops(dev)
comm = rccl.get_comm(None)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); t_host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, t_host * 1e6


small = torch.zeros(2, dtype=torch.float64, device=dev)
filler = torch.zeros(1 << 20, device=dev)
for name, fn in (("direct 16 B", lambda: comm.all_reduce_sum_(small)), ("torch  16 B", lambda: dist.all_reduce(small))):
    g, h = timed(fn, 200)
    print(f"{name}: {g:8.1f} us on the stream per call, {h:8.1f} us of host time per call")
    # between two dependent kernels: kernel, all-reduce, kernel
    g2, h2 = timed(lambda: (filler.add_(1.0), fn(), filler.add_(1.0)), 200)
    g0, _ = timed(lambda: (filler.add_(1.0), filler.add_(1.0)), 200)
    print(f"   between two dependent kernels: +{g2 - g0:6.1f} us per call")
n, c2 = 6913, 256
A0 = torch.randn(n, n, device=dev); A0 = (A0 + A0.T).contiguous(); B0 = torch.randn(c2, n, device=dev)
buf = torch.empty(ops.lib.effq_gram_packed_elems(n, c2), dtype=torch.float32, device=dev)
print(f"packed Gram system of n = {n}: {buf.numel() * 4 / 1e6:.0f} MB")
for name, fn in (("direct", lambda: comm.all_reduce_sum_(buf)), ("torch ", lambda: dist.all_reduce(buf))):
    g, h = timed(fn, 10)
    print(f"{name} all-reduce of it: {g / 1e3:8.3f} ms ({buf.numel() * 4 / g / 1e3:.1f} GB/s on one rank)")
g, _ = timed(lambda: ops.gram_reduce(A0, B0, lambda t: t), 10)
print(f"pack + unpack alone: {g / 1e3:8.3f} ms")
rccl.close_all(); dist.destroy_process_group()
