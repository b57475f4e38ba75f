"""Sample-parallel collectives: the reference's only parallelism strategy (SURVEY.md sections 2.2, 8e).

Reference (under /root/reference/hippyflow/collectives):
  NullCollective                           collective.py:19-38
  MultipleSamePartitioningPDEsCollective   collective.py:43-159   (mpi4py Allreduce / Bcast)
  MultipleSerialPDEsCollective             collective.py:161-162
  CollectiveOperator                       collectiveOperator.py:14-55
  MatrixMultCollectiveOperator             collectiveOperator.py:58-97

Here the communicator is ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" on
CPU), one process per GPU.  A whole N x k block is reduced in ONE all-reduce on its device memory
(the reference sends k messages of length N through host copies, collective.py:98-111).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .multivector import MultiVector, Vector


class NullCollective:
    """No-overhead "parallel" reduction utilities on one process (collective.py:19-38)."""

    def bcast(self, v, root=0):
        return v

    def size(self):
        return 1

    def rank(self):
        return 0

    def allReduce(self, v, op):
        if op.lower() not in ["sum", "avg"]:
            err_msg = "Unknown operation *{0}* in NullCollective.allReduce".format(op)
            raise NotImplementedError(err_msg)
        return v


class _CudaArrayView:
    """Zero-copy hand-off of a block's HBM to torch (``__cuda_array_interface__`` v2)."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class TorchCollective:
    """Counterpart of MultipleSamePartitioningPDEsCollective over a torch.distributed process group.

    ``allReduce(v, op)``: ``op`` in {"sum", "avg"} (case-insensitive); ``v`` may be a float / int
    (returned), a numpy array (overwritten, returned), a device Vector or MultiVector (reduced in place
    in HBM), or any object with ``nvec``/``[i]``.  ``bcast(v, root)`` likewise.  Anything else raises
    NotImplementedError, as in the reference (collective.py:112-117,153-159)."""

    def __init__(self, group=None, is_serial_check=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.is_serial_check = is_serial_check
        if not dist.is_initialized():
            raise RuntimeError("TorchCollective: torch.distributed is not initialised")

    def size(self):
        return self.dist.get_world_size(self.group)

    def rank(self):
        return self.dist.get_rank(self.group)

    # -- helpers
    def _backend_device(self):
        import torch
        if self.dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def _allReduce_array(self, v, op):
        import torch
        err_msg = "Unknown operation *{0}* in TorchCollective.allReduce".format(op)
        if op not in ("sum", "avg"):
            raise NotImplementedError(err_msg)
        t = torch.from_numpy(np.ascontiguousarray(v)).to(self._backend_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        receive = t.cpu().numpy()
        if op == "sum":
            v[:] = receive
        else:
            v[:] = (1. / float(self.size())) * receive
        return v

    def _tensor_of(self, mv):
        """torch tensor aliasing the block's storage (including the zero padding rows).  Returns
        (tensor, stage): ``stage`` is None for the zero-copy view, else a torch-owned staging block."""
        import torch
        nelem = mv.leading_dimension() * mv.nvec()
        dev = torch.device("cuda", mv.ctx.device)
        try:
            t = torch.as_tensor(_CudaArrayView(mv.device_ptr(), nelem), device=dev)
            if t.data_ptr() == mv.device_ptr():
                return t, None
        except Exception:
            pass
        t = torch.empty(nelem, dtype=torch.float64, device=dev)
        h = C.c_void_p()
        L.call("hfmi_block_wrap", mv.ctx.handle, C.c_void_p(t.data_ptr()), mv.size(), mv.nvec(), mv.leading_dimension(), C.byref(h))
        stage = MultiVector(ctx=mv.ctx, _handle=h, _parent=t)
        stage.copy_from(mv)
        mv.ctx.synchronize()
        return t, stage

    def _on_block_stream(self, mv):
        """Context manager that makes libhfmi's stream torch's current stream: the collective is then ordered after the
        kernels that produced the block and before the ones that consume it by stream semantics alone (ProcessGroupNCCL
        makes its communication stream wait on the current stream and, for a blocking call, the current stream wait on
        the collective) -- no host synchronisation on the solve's critical path."""
        import torch
        sp = mv.ctx.get_stream()
        if not sp or self.dist.get_backend(self.group) != "nccl":
            return None
        return torch.cuda.stream(torch.cuda.ExternalStream(sp, device=torch.device("cuda", mv.ctx.device)))

    def _reduce_block(self, mv, op):
        import torch
        if op not in ("sum", "avg"):
            raise NotImplementedError("Unknown operation *{0}* in TorchCollective.allReduce".format(op))
        t, stage = self._tensor_of(mv)
        cm = self._on_block_stream(mv) if stage is None else None
        if cm is None:
            mv.ctx.synchronize()                   # libhfmi's stream -> host: the block is complete
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            if op == "avg":
                t.mul_(1.0 / float(self.size()))
            torch.cuda.current_stream(mv.ctx.device).synchronize()   # RCCL + scale done before libhfmi reads
            if stage is not None:
                mv.copy_from(stage)
                mv.ctx.synchronize()
            return mv
        with cm:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            if op == "avg":
                t.mul_(1.0 / float(self.size()))
        return mv

    def _bcast_block(self, mv, root):
        import torch
        t, stage = self._tensor_of(mv)
        cm = self._on_block_stream(mv) if stage is None else None
        if cm is None:
            mv.ctx.synchronize()
            self.dist.broadcast(t, src=root, group=self.group)
            torch.cuda.current_stream(mv.ctx.device).synchronize()
            if stage is not None:
                mv.copy_from(stage)
                mv.ctx.synchronize()
            return mv
        with cm:
            self.dist.broadcast(t, src=root, group=self.group)
        return mv

    def allReduce(self, v, op):
        op = op.lower()
        if type(v) in [float, np.float64]:
            v_array = np.array([v], dtype=np.float64)
            self._allReduce_array(v_array, op)
            return v_array[0]
        elif type(v) in [int, np.int32, np.int64]:
            v_array = np.array([v], dtype=np.float64)
            self._allReduce_array(v_array, op)
            return type(v)(v_array[0]) if op == "sum" else v_array[0]
        elif type(v) is np.ndarray:
            return self._allReduce_array(v, op)
        elif isinstance(v, MultiVector):
            return self._reduce_block(v, op)
        elif isinstance(v, Vector):
            self._reduce_block(v._mv, op)
            return v
        elif hasattr(v, "mpi_comm") and hasattr(v, "get_local"):
            v_array = v.get_local()
            self._allReduce_array(v_array, op)
            v.set_local(v_array)
            v.apply("")
            return v
        elif hasattr(v, 'nvec'):
            for i in range(v.nvec()):
                self.allReduce(v[i], op)
            return v
        else:
            msg = "TorchCollective.allReduce not implement for v of type {0}".format(type(v))
            raise NotImplementedError(msg)

    def bcast(self, v, root=0):
        import torch
        if type(v) in [float, np.float64, int, np.int32, np.int64]:
            t = torch.tensor([float(v)], dtype=torch.float64, device=self._backend_device())
            self.dist.broadcast(t, src=root, group=self.group)
            return type(v)(t.item())
        if type(v) is np.ndarray:
            t = torch.from_numpy(np.ascontiguousarray(v)).to(self._backend_device())
            self.dist.broadcast(t, src=root, group=self.group)
            v[...] = t.cpu().numpy()
            return v
        elif isinstance(v, MultiVector):
            return self._bcast_block(v, root)
        elif isinstance(v, Vector):
            self._bcast_block(v._mv, root)
            return v
        elif hasattr(v, "mpi_comm") and hasattr(v, "get_local"):
            v_local = v.get_local()
            self.bcast(v_local, root=root)
            v.set_local(v_local)
            v.apply("")
            return v
        elif hasattr(v, 'nvec'):
            for i in range(v.nvec()):
                self.bcast(v[i], root=root)
            return v
        else:
            msg = "TorchCollective.bcast not implement for v of type {0}".format(type(v))
            raise NotImplementedError(msg)


def MultipleSamePartitioningPDEsCollective(group=None, is_serial_check=False):
    return TorchCollective(group, is_serial_check=is_serial_check)


def MultipleSerialPDEsCollective(group=None):
    return TorchCollective(group, is_serial_check=True)


class CollectiveOperator:
    """Parallel version of a linear operator: apply the local operator, all-reduce the result
    (collectiveOperator.py:14-55)."""

    def __init__(self, local_op, collective, mpi_op='sum'):
        assert hasattr(local_op, 'mult')
        self.local_op = local_op
        self.collective = collective
        self.mpi_op = mpi_op

    def mult(self, x, y):
        self.local_op.mult(x, y)
        self.collective.allReduce(y, self.mpi_op)

    def transpmult(self, x, y):
        assert hasattr(self.local_op, 'transpmult')
        self.local_op.transpmult(x, y)
        self.collective.allReduce(y, self.mpi_op)

    def matMvMult(self, x, y):
        """Block fast path (one all-reduce for the whole block) when the local operator has one."""
        from .multivector import MatMvMult
        MatMvMult(self.local_op, x, y)
        self.collective.allReduce(y, self.mpi_op)

    def init_vector(self, x, dim):
        self.local_op.init_vector(x, dim)


class MatrixMultCollectiveOperator:
    """collectiveOperator.py:58-97."""

    def __init__(self, local_op, collective, mpi_op='sum'):
        assert hasattr(local_op, 'matMvMult')
        self.local_op = local_op
        self.collective = collective
        self.mpi_op = mpi_op

    def matMvMult(self, x, y):
        self.local_op.matMvMult(x, y)
        self.collective.allReduce(y, self.mpi_op)

    def matMvTranspmult(self, x, y):
        assert hasattr(self.local_op, 'matMvTranspmult')
        self.local_op.matMvTranspmult(x, y)
        self.collective.allReduce(y, self.mpi_op)

    def init_vector(self, x, dim=0):
        self.local_op.init_vector(x, dim)
