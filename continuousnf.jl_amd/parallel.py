"""Batch-axis sharding.  Columns are independent in the RHS
(src/icnf.jl:330-349) and in post-processing (src/base_icnf.jl:174-187), so each rank
solves its own contiguous block of columns with no data-path collective; the only
exchange is one all-reduce (RCCL ncclSum over xGMI; gloo on CPU) of the five floats
(sum logpx, sum E, sum n, sum A, count) behind ``loss`` (src/icnf.jl:489)."""
from __future__ import annotations

import numpy as np


def shard_range(B: int, world_size: int, rank: int):
    """Contiguous column block of ``rank``: the first ``B % world_size`` ranks get one more."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    q, r = divmod(B, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allreduce_sums(sums, group=None):
    """All-reduce the 5-float vector.  ``sums`` is a torch tensor (CUDA with the nccl
    backend = RCCL, CPU with gloo) or array-like (moved to the backend's device)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sums
    if not isinstance(sums, torch.Tensor):
        sums = torch.as_tensor(np.asarray(sums, dtype=np.float32))
    backend = dist.get_backend(group)
    if backend == "nccl" and not sums.is_cuda:
        sums = sums.cuda()
    if backend == "gloo" and sums.is_cuda:
        sums = sums.cpu()
    sums = sums.clone()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


class RcclComm:
    """An RCCL communicator owned through the C ABI (cnf_comm_* in include/cnfhip.h): what a caller
    without torch.distributed (the Julia shim) uses for the 5-float all-reduce behind ``loss``
    (src/icnf.jl:489).  ``handle`` is an ``ncclComm_t``."""

    def __init__(self, world_size: int, rank: int, unique_id: bytes, device: int):
        from . import _lib
        import ctypes as C
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of RcclComm.unique_id()")
        l = _lib.lib()
        h = C.c_void_p()
        st = l.cnf_comm_init(C.byref(h), world_size, rank, unique_id, device)
        if st != _lib.OK:
            raise _lib.CNFError(st, l.cnf_comm_last_error().decode())
        self.handle, self.world_size, self.rank, self.device = h, world_size, rank, device

    @staticmethod
    def unique_id() -> bytes:
        """ncclGetUniqueId: call on ONE rank and hand the bytes to the others."""
        from . import _lib
        import ctypes as C
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        l = _lib.lib()
        st = l.cnf_comm_unique_id(buf)
        if st != _lib.OK:
            raise _lib.CNFError(st, l.cnf_comm_last_error().decode())
        return buf.raw

    @staticmethod
    def device_key(device: int) -> str:
        """cnf_comm_device_key: "<hostname>|<boot id>/<pci bus id>" of ``device``."""
        from . import _lib
        import ctypes as C
        buf = C.create_string_buffer(192)
        l = _lib.lib()
        st = l.cnf_comm_device_key(device, buf, len(buf))
        if st != _lib.OK:
            raise _lib.CNFError(st, l.cnf_comm_last_error().decode())
        return buf.value.decode()

    @classmethod
    def from_torch_group(cls, device: int, group=None):
        """Bootstrap over an initialised torch.distributed group (any backend): rank 0 draws the id,
        the group broadcasts the 128 bytes."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # No two ranks on one GPU: RCCL refuses that from inside ncclCommInitRank (cnf_comm_device_key, cnfhip.h); it is settled
        # here, where the ranks can still talk -- every rank sees every key and all of them fail together, before RCCL is entered.
        keys = [None] * world
        dist.all_gather_object(keys, cls.device_key(device), group=group)
        import os
        if len(set(keys)) != world and os.environ.get("CNF_COMM_ALLOW_SHARED_GPU") != "1":     # (the switch: to see RCCL's own answer)
            dup = sorted(k for k in set(keys) if keys.count(k) > 1)
            raise RuntimeError(f"RCCL needs one GPU per rank: ranks share {dup} (ranks by device: {keys})")
        uid, err = None, None
        if rank == 0:
            try:
                uid = cls.unique_id()
            except Exception as e:                      # noqa: BLE001 -- rank 0 must still reach the broadcast:
                err = e                                 # the other ranks are waiting in it
        box = [uid]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None:                              # every rank fails together
            raise RuntimeError(f"rank 0 could not draw an RCCL id: {err}" if rank == 0 else "rank 0 could not draw an RCCL id")
        return cls(world, rank, box[0], device)

    def size(self) -> int:
        from . import _lib
        import ctypes as C
        n = C.c_int()
        st = _lib.lib().cnf_comm_size(self.handle, C.byref(n))
        if st != _lib.OK:
            raise _lib.CNFError(st, _lib.lib().cnf_comm_last_error().decode())
        return n.value

    def allreduce_sums(self, icnf, sums):
        """cnf_loss_allreduce: in-place RCCL sum of the 5 device floats on torch's current stream."""
        from . import _lib
        import ctypes as C
        import torch
        if not (isinstance(sums, torch.Tensor) and sums.is_cuda and sums.dtype == torch.float32 and sums.numel() == 5
                and sums.is_contiguous()):
            raise ValueError("sums must be a contiguous float32 CUDA tensor of 5 elements")
        st = C.c_void_p(torch.cuda.current_stream(sums.device).cuda_stream)
        _lib.check(_lib.lib().cnf_loss_allreduce(icnf.handle(), self.handle, sums.data_ptr(), st), icnf.handle())
        return sums

    def allreduce(self, t):
        """cnf_comm_allreduce: in-place sum of a contiguous float32 CUDA tensor."""
        from . import _lib
        import ctypes as C
        import torch
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("need a contiguous float32 CUDA tensor")
        st = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        rc = _lib.lib().cnf_comm_allreduce(self.handle, t.data_ptr(), t.numel(), st)
        if rc != _lib.OK:
            raise _lib.CNFError(rc, _lib.lib().cnf_comm_last_error().decode())
        return t

    def lockstep(self, icnf, enable=True):
        """cnf_set_shard_comm: the adaptive controller of ``icnf`` reduces its three floats through this
        communicator on the solve's stream (no host callback)."""
        from . import _lib
        import weakref
        _lib.check(_lib.lib().cnf_set_shard_comm(icnf.handle(), self.handle if enable else None), icnf.handle())
        users = self.__dict__.setdefault("_lockstep_users", {})
        if enable:
            users[id(icnf)] = weakref.ref(icnf)
        else:
            users.pop(id(icnf), None)
        return icnf

    def close(self):
        if self.handle is not None:
            from . import _lib
            # handles still reducing through this communicator must not keep a dangling ncclComm_t
            for ref in list(self.__dict__.get("_lockstep_users", {}).values()):
                ic = ref()
                if ic is not None and ic._handle is not None:
                    _lib.lib().cnf_set_shard_comm(ic._handle, None)
            self.__dict__["_lockstep_users"] = {}
            _lib.lib().cnf_comm_destroy(self.handle)
            self.handle = None


def make_shard_reduce(group=None):
    """Callback for ``ICNF.set_shard_reduce``: sums the controller's three floats over the ranks of
    ``group`` (gloo: on the host buffer itself; nccl = RCCL: through a 3-float device tensor)."""
    import torch
    import torch.distributed as dist

    def reduce_(values):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        t = torch.from_numpy(values)             # shares the pinned host buffer
        if dist.get_backend(group) == "nccl":
            d = t.cuda()
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
            t.copy_(d)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return reduce_


def lockstep(icnf, group=None, enable=True):
    """Make every adaptive solve of ``icnf`` take the step sequence of the unsharded batch: the
    error norm (and the initial-dt norms) are summed over the ranks before the controller runs."""
    icnf.set_shard_reduce(make_shard_reduce(group) if enable else None)
    return icnf


def loss_from_global_sums(sums, mode_train: bool, lambdas):
    """Host formula of cnf_loss_from_sums, for callers that hold no handle (CPU tests)."""
    s = np.asarray(sums.detach().cpu() if hasattr(sums, "detach") else sums, dtype=np.float64)
    if not s[4] > 0:
        raise ValueError("empty batch")
    if mode_train:
        return float((-s[0] + lambdas[0] * s[1] + lambdas[1] * s[2] + lambdas[2] * s[3]) / s[4])
    return float(-s[0] / s[4])


def distributed_loss(icnf, mode, xs_local, ps, st=None, *, eps=None, group=None):
    """``loss`` over a batch sharded by columns: local inference on this rank's columns,
    then the 5-float all-reduce, then the mean."""
    from .base_icnf import _is_torch, inference, loss_from_sums, loss_sums
    if _is_torch(xs_local):
        _, _, local = inference(icnf, mode, xs_local, ps, st, eps=eps, with_sums=True)
    else:
        logpx, regs = inference(icnf, mode, xs_local, ps, st, eps=eps)
        local = loss_sums(icnf, logpx, regs)
    sums = allreduce_sums(local, group)
    return loss_from_sums(icnf, mode, sums)


def allreduce_mean_weighted(value: float, grad, count: int, group=None):
    """Combine per-shard (mean loss, mean gradient, column count) into the mean over ALL columns:
    one all-reduce of ``n_params + 2`` floats (RCCL ncclSum with the nccl backend).  ``grad`` is a torch
    tensor (CUDA or CPU) or a numpy array; the result has the same kind."""
    import torch
    import torch.distributed as dist
    is_np = not isinstance(grad, torch.Tensor)
    g = torch.as_tensor(np.asarray(grad, dtype=np.float32)) if is_np else grad
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value, grad
    backend = dist.get_backend(group)
    dev = g.device
    buf = torch.empty(g.numel() + 2, dtype=torch.float32, device=dev)
    buf[:-2] = g * float(count)
    buf[-2] = value * float(count)
    buf[-1] = float(count)
    if backend == "nccl" and not buf.is_cuda:
        buf = buf.cuda()
    if backend == "gloo" and buf.is_cuda:
        buf = buf.cpu()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    buf = buf.to(dev)
    total = float(buf[-1])
    out = buf[:-2] / total
    return float(buf[-2]) / total, (out.numpy() if is_np else out)


def distributed_loss_and_grad(icnf, mode, xs_local, *args, eps=None, group=None):
    """Data-parallel ``loss_and_grad``: every rank differentiates the loss of its own columns
    (``base_icnf.loss_and_grad``), then one all-reduce gives the loss and gradient of the whole batch --
    identical on all ranks, so identical optimiser steps keep the replicas in sync."""
    from .base_icnf import loss_and_grad
    val, grad = loss_and_grad(icnf, mode, xs_local, *args, eps=eps)
    B = xs_local.shape[1]
    return allreduce_mean_weighted(val, grad, B, group)
