"""All-reduce(SUM) on RCCL, enqueued DIRECTLY on the stream the kernels of the calibration run on.

The data-parallel path issues ~50 tiny all-reduces per layer (statistics, the two sums of every iteration of the
activation scale fit, the Gram system, the loss history), each BETWEEN two dependent kernels of one stream.
``torch.distributed.all_reduce`` runs a collective on the process group's own stream: two event hops (current stream ->
group stream -> current stream) around every call, each a barrier packet in the queue - measured on one MI355X with a
1-rank RCCL group (``EFFQ_DP_FORCE=1``): +135 ms per calibration for 1115 collectives, 121 us each, 21 % of the step,
before a single byte has crossed xGMI.  Here ``ncclAllReduce`` is called through the C ABI of the RCCL library the
framework itself has loaded, with ``torch.cuda.current_stream()`` as its stream argument: the collective is one more
kernel in the chain, ordered by the stream like every other.

The communicator is created once per process group: rank 0 draws the ``ncclUniqueId`` and broadcasts its 128 bytes
through ``torch.distributed`` (whatever backend the group has); ``ncclCommInitRank`` is collective.  Plumbing only:
device memory, streams and the rendezvous stay the framework's.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
from typing import Optional

import torch

NCCL_UNIQUE_ID_BYTES = 128
_DT = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
       torch.float64: 8, torch.bfloat16: 9}
NCCL_SUM = 0


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * NCCL_UNIQUE_ID_BYTES)]     # (c_char fields read back truncated at the first NUL)


_lib = None
_uid_seq = 0
NCCL_MAX = 2


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
    err = None
    for path in cands:
        try:
            lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as e:             # pragma: no cover
            err = e
            continue
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclGetUniqueId.restype = C.c_int
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclCommInitRank.restype = C.c_int
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclAllReduce.restype = C.c_int
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclAllGather.restype = C.c_int
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclCommDestroy.restype = C.c_int
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetErrorString.restype = C.c_char_p
        _lib = lib
        return lib
    raise OSError(f"librccl.so not found ({err})")


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_load().ncclGetErrorString(rc).decode(errors='replace')}")


def exchange_unique_id(raw, rank: int, world: int, group=None) -> bytes:
    """Rank 0's 128 bytes on every rank of `group` (no device work: runs on any backend; covered on gloo by
    tests/dp_worker.py)."""
    import torch.distributed as dist
    if world > 1:
        if group is None:
            # through the rendezvous store (plain TCP): the framework's own NCCL communicator - and the extra HIP
            # streams it brings, which alias the calibration's streams onto shared hardware queues (measured: +20 % per
            # calibration with an eagerly initialised "nccl" group that is never used) - is never created
            global _uid_seq
            store = dist.distributed_c10d._get_default_store()
            key = f"effq_rccl_uid_{_uid_seq}"
            _uid_seq += 1
            if rank == 0:
                store.set(key, raw)
            raw = store.get(key)
        else:
            box = [raw]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0), group=group)
            raw = box[0]
    raw = bytes(raw)
    if len(raw) != NCCL_UNIQUE_ID_BYTES:
        raise RuntimeError(f"exchange_unique_id: {len(raw)} bytes arrived, expected {NCCL_UNIQUE_ID_BYTES}")
    return raw


class DirectComm:
    """One RCCL communicator over the ranks of a torch.distributed group, used for in-stream SUM all-reduces."""

    def __init__(self, group=None):
        import torch.distributed as dist
        lib = _load()
        # the calibration's own streams first: a communicator brings streams too, and hardware queues go by creation order
        from . import hip_ops
        hip_ops.get_ops(torch.device("cuda", torch.cuda.current_device())).warm_streams()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES) if self.rank == 0 else None
        raw = exchange_unique_id(raw, self.rank, self.world, group)
        C.memmove(C.byref(uid), bytes(raw), NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        _check(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.calls = 0

    def all_reduce_sum_(self, t: torch.Tensor, op: int = NCCL_SUM) -> torch.Tensor:
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("DirectComm.all_reduce_sum_: contiguous device tensors only")
        st = torch.cuda.current_stream(t.device).cuda_stream
        _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DT[t.dtype], op, self.comm,
                                     C.c_void_p(st)), "ncclAllReduce")
        self.calls += 1
        return t

    def all_reduce_max_(self, t: torch.Tensor) -> torch.Tensor:
        return self.all_reduce_sum_(t, NCCL_MAX)

    def barrier(self, device) -> int:
        """Every rank has reached this point and the device has drained; returns the number of ranks that took part."""
        one = torch.ones(1, dtype=torch.float32, device=device)
        self.all_reduce_sum_(one)
        torch.cuda.synchronize(device)
        return int(one.item())

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """rank-major concatenation of every rank's `t` (equal sizes), on the current stream."""
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("DirectComm.all_gather: contiguous device tensors only")
        out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
        st = torch.cuda.current_stream(t.device).cuda_stream
        _check(_load().ncclAllGather(t.data_ptr(), out.data_ptr(), t.numel(), _DT[t.dtype], self.comm, C.c_void_p(st)),
               "ncclAllGather")
        self.calls += 1
        return out

    def close(self):
        if self.comm:
            _load().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


_comms = {}


_agree_seq = 0


def _all_ranks_ok(ok: bool, group) -> bool:
    """True when EVERY rank of the default group reports `ok` (through the rendezvous store: no device work)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 or group is not None:
        return ok
    global _agree_seq
    store = dist.distributed_c10d._get_default_store()
    key = f"effq_rccl_ok_{_agree_seq}"
    _agree_seq += 1
    store.set(f"{key}_{dist.get_rank()}", b"1" if ok else b"0")
    return all(bytes(store.get(f"{key}_{r}")) == b"1" for r in range(world))


def get_comm(group=None) -> Optional[DirectComm]:
    """The communicator of `group` (created collectively on first use), or None when EFFQ_RCCL_DIRECT=0 - or when any rank
    failed to create it: then EVERY rank gets None and the collectives go through torch.distributed's own RCCL
    communicator (same library, two stream hops per call), with a warning on stderr."""
    if os.environ.get("EFFQ_RCCL_DIRECT", "1") == "0":
        return None
    key = id(group) if group is not None else 0
    if key not in _comms:
        comm, err = None, None
        try:
            comm = DirectComm(group)
        except Exception as e:           # noqa: BLE001 - whatever went wrong, the ranks must agree on the path they take
            err = e
        if not _all_ranks_ok(comm is not None, group):
            import sys
            print(f"[efficientq_amd.rccl] direct RCCL communicator unavailable ({err if err else 'another rank failed'}): "
                  "falling back to torch.distributed collectives", file=sys.stderr, flush=True)
            if comm is not None:
                comm.close()
            comm = None
        _comms[key] = comm
    return _comms[key]


def close_all():
    for c in _comms.values():
        if c is not None:
            c.close()
    _comms.clear()
