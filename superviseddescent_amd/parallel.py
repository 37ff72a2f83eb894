"""Data-parallel scaling of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

* **detect** shards by sample and needs no collective at all (rows are independent);
* **train** needs exactly one exchange per cascade level: every rank builds the Gram matrix ``A_r^T A_r`` and the
  right-hand side ``A_r^T b_r`` of ITS rows, the two are summed over ranks in one all-reduce, and every rank then
  solves the identical regularised system (no broadcast of the regressor needed).  ``MatrixNorm`` regularisation
  uses the norm of the GLOBAL Gram matrix and the GLOBAL row count, so the result equals single-process training
  on the concatenated rows up to summation order (reference: include/superviseddescent/regressors.hpp:126-148,
  199-234; the reference itself is single-process).

The C-ABI exposes the exchange as a callback (``sdm_set_allreduce``); this module supplies the torch one.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Tuple

import numpy as np


def shard_range(n_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of rows owned by ``rank`` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_rows, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class _DeviceSpan:
    """Zero-copy view of engine-owned HBM for torch (``__cuda_array_interface__``, also honoured on ROCm)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def make_torch_allreduce(device_index: int) -> Callable[[int, int, int], int]:
    """Callback for ``Context.set_allreduce``: sums ``count`` floats at ``ptr`` over all ranks in place.

    The engine hands over the HIP stream its tile-pack kernel was launched on (``stream``); the collective is issued
    with that stream as torch's current one, so it is ordered after the pack and before the unpack / Cholesky on ANY
    context stream -- the context's own non-blocking stream as well as one shared with torch."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", device_index)

    def allreduce(ptr: int, count: int, stream: int) -> int:
        t = torch.as_tensor(_DeviceSpan(ptr, count), device=dev)
        if stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:   # the legacy default stream
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return 0

    return allreduce


def make_torch_solve_collectives(device_index: int):
    """``(bcast, allgather)`` for ``Context.set_solve_sharding``: the two collectives of the sharded factorisation
    (include/sdm.h) through ``torch.distributed`` (nccl = RCCL), issued with the engine's stream as torch's current one."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", device_index)

    def on_stream(stream):
        return torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)) if stream else torch.cuda.stream(None)

    def bcast(ptr: int, count: int, root: int, stream: int) -> int:
        with on_stream(stream):
            dist.broadcast(torch.as_tensor(_DeviceSpan(ptr, count), device=dev), src=root)
        return 0

    def allgather(send: int, recv: int, count: int, stream: int) -> int:
        with on_stream(stream):
            dist.all_gather_into_tensor(torch.as_tensor(_DeviceSpan(recv, count * dist.get_world_size()), device=dev),
                                        torch.as_tensor(_DeviceSpan(send, count), device=dev))
        return 0

    return bcast, allgather


def make_torch_reduce_scatter(device_index: int):
    """Callback for ``Context.set_reduce_scatter``: ``world`` chunks of ``count`` floats at ``send``; the sum over the ranks of
    this rank's chunk arrives at ``recv`` (``torch.distributed.reduce_scatter_tensor``, nccl = RCCL), on the engine's stream."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", device_index)

    def reduce_scatter(send: int, recv: int, count: int, stream: int) -> int:
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)) if stream else torch.cuda.stream(None)
        with ctx:
            dist.reduce_scatter_tensor(torch.as_tensor(_DeviceSpan(recv, count), device=dev),
                                       torch.as_tensor(_DeviceSpan(send, count * dist.get_world_size()), device=dev),
                                       op=dist.ReduceOp.SUM)
        return 0

    return reduce_scatter


class RcclCommunicator:
    """An ``ncclComm_t`` of this process's own -- one rank per process / GPU -- for the library-native exchange: the engine then
    calls ``ncclAllReduce`` / ``ncclReduceScatter`` / ``ncclBroadcast`` / ``ncclAllGather`` itself, on its own HIP streams
    (``sdm_set_allreduce_rccl``, ``sdm_set_reduce_scatter_rccl``, ``sdm_set_solve_sharding_rccl``; include/sdm.h), with no Python
    between two kernels of a training level.  The communicator is created with ``ncclCommInitRank`` from a unique id that rank 0
    draws and ``torch.distributed`` (any backend) carries to the other ranks; RCCL itself is the copy torch already maps into
    the process (exactly one RCCL per process), else ROCm's.  The device of this rank must be current
    (``torch.cuda.set_device``) when the object is built.  The reference has no collective (superviseddescent.hpp:170-218 is
    single-process)."""

    class _UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    def __init__(self, rank: int, world_size: int):
        import torch
        import torch.distributed as dist
        self.rank, self.world_size = int(rank), int(world_size)
        self._rccl = self._load()
        uid = self._UniqueId()
        if self.rank == 0 and self._rccl.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        if self.world_size > 1:
            box = [bytes(uid.internal) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctypes.memmove(ctypes.byref(uid), box[0], 128)
        torch.cuda.current_stream().synchronize()
        self._rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._UniqueId, ctypes.c_int]
        comm = ctypes.c_void_p()
        rc = self._rccl.ncclCommInitRank(ctypes.byref(comm), self.world_size, uid, self.rank)
        if rc != 0 or not comm.value:
            raise RuntimeError("ncclCommInitRank failed with status %d" % rc)
        self.comm = comm
        self._fn = {n: ctypes.cast(getattr(self._rccl, n), ctypes.c_void_p)
                    for n in ("ncclAllReduce", "ncclReduceScatter", "ncclBroadcast", "ncclAllGather")}

    @staticmethod
    def _load():
        import importlib.util
        import os
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
            if os.path.exists(cand):
                return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        return ctypes.CDLL("/opt/rocm/lib/librccl.so", mode=ctypes.RTLD_GLOBAL)

    def install(self, ctx, shard_solve: bool = False, reduce_scatter: bool = False):
        """Registers the communicator with a ``Context``: the Gram / RHS exchange always; the sharded factorisation and the
        reduce-scatter form of the exchange on request (the latter only pays together with the former)."""
        ctx.set_allreduce_rccl(self.comm, self.world_size, self._fn["ncclAllReduce"])
        if shard_solve:
            ctx.set_solve_sharding_rccl(self.comm, self.rank, self.world_size, self._fn["ncclBroadcast"], self._fn["ncclAllGather"])
        else:
            ctx.set_solve_sharding_rccl(None)
        ctx.set_reduce_scatter_rccl(bool(shard_solve and reduce_scatter), self._fn["ncclReduceScatter"])

    @staticmethod
    def uninstall(ctx):
        ctx.set_reduce_scatter_rccl(False)
        ctx.set_solve_sharding_rccl(None)
        ctx.set_allreduce_rccl(None, 1)

    def destroy(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            self._rccl.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


def make_host_allreduce() -> Callable[[np.ndarray], None]:
    """All-reduce for host buffers (gloo): used by the CPU tests of the data-parallel logic."""
    import torch
    import torch.distributed as dist

    def allreduce(buf: np.ndarray) -> None:
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    return allreduce


def global_row_count(n_local: int) -> int:
    """Sum of the per-rank row counts (the N of ``MatrixNorm``'s lambda = param * ||G||_F / N)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return n_local
    t = torch.tensor([n_local], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
