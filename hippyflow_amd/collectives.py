"""Sample-parallel collectives: the reference's only parallelism strategy (SURVEY.md sections 2.2, 8e).

Reference (under /root/reference/hippyflow/collectives):
  NullCollective                           collective.py:19-38
  MultipleSamePartitioningPDEsCollective   collective.py:43-159   (mpi4py Allreduce / Bcast)
  MultipleSerialPDEsCollective             collective.py:161-162
  CollectiveOperator                       collectiveOperator.py:14-55
  MatrixMultCollectiveOperator             collectiveOperator.py:58-97

Here the communicator is ``NativeCollective``: an ``hfmi_comm`` of libhfmi (include/hfmi.h), one process per
GPU, RCCL over xGMI underneath.  A whole N x k block is reduced by ONE all-reduce on its device memory, on the
context's stream (the reference sends k messages of length N through host copies, collective.py:98-111).  No torch
is imported on this path.  ``TorchCollective`` offers the same protocol over an existing ``torch.distributed``
process group (hosts that already run one; the gloo tests on CPU).

What a collective accepts (collective.py:74-159): a float / int (returned), a numpy array (overwritten,
returned), a device Vector or MultiVector (reduced in place in HBM), a dolfin-like vector
(``get_local`` / ``set_local`` / ``apply``), or any container with ``nvec()`` and ``[i]``.  ``op`` is 'sum' or 'avg',
any case.  Anything else raises NotImplementedError.
"""
import ctypes as C
import os
import tempfile

import numpy as np

from . import _lib as L
from .multivector import MultiVector, Vector

_REDUCE_CODES = {"sum": 0, "avg": 1}
_SCALAR_FLOATS = (float, np.floating)
_SCALAR_INTS = (int, np.integer)


def _reduce_code(op, who):
    code = _REDUCE_CODES.get(op.lower() if isinstance(op, str) else None)
    if code is None:
        raise NotImplementedError("%s: reduction %r is not available (use 'sum' or 'avg')" % (who, op))
    return code


class NullCollective:
    """The one-process communicator (collective.py:19-38): reductions and broadcasts hand their argument back."""

    def size(self):
        return 1

    def rank(self):
        return 0

    def allReduce(self, v, op):
        _reduce_code(op, "NullCollective.allReduce")
        return v

    def bcast(self, v, root=0):
        return v

    def barrier(self):
        pass


def _payload_kind(v):
    """Which of the payload shapes of collective.py:74-159 ``v`` is (bool is not a number here)."""
    if isinstance(v, bool):
        return None
    if isinstance(v, _SCALAR_FLOATS):
        return "float"
    if isinstance(v, _SCALAR_INTS):
        return "int"
    if isinstance(v, np.ndarray):
        return "array"
    if isinstance(v, MultiVector):
        return "block"
    if isinstance(v, Vector):
        return "vector"
    if hasattr(v, "get_local") and hasattr(v, "set_local"):
        return "dolfin"
    if hasattr(v, "nvec"):
        return "container"
    return None


class _Collective:
    """Payload dispatch shared by the communicators.  A subclass supplies four primitives:
    ``_reduce_f64(arr, code)`` and ``_bcast_raw(arr, root)`` on contiguous host arrays (in place),
    ``_reduce_block(mv, code)`` and ``_bcast_block(mv, root)`` on device blocks (in place)."""

    name = "collective"

    def _split_random(self):
        """hp.parRandom is seeded per process; here the process-wide generator's PRIVATE streams are keyed by the rank of the
        FIRST communicator the process builds (the world / sample-parallel one), so that the Monte-Carlo draws of the ranks
        differ while shared draws (probe blocks) stay identical (randomized._ParRandom).  Later collectives do not re-key."""
        from .randomized import parRandom
        parRandom.split(self.rank(), by_collective=True)

    # ---- reductions
    def allReduce(self, v, op):
        code = _reduce_code(op, self.name + ".allReduce")
        kind = _payload_kind(v)
        handler = getattr(self, "_reduce_" + kind, None) if kind else None
        if handler is None:
            raise NotImplementedError("%s.allReduce: no rule for a payload of type %s" % (self.name, type(v).__name__))
        return handler(v, code)

    def _reduce_float(self, v, code):
        box = np.array([v], dtype=np.float64)
        self._reduce_f64(box, code)
        return box[0]

    def _reduce_int(self, v, code):
        total = self._reduce_float(float(v), code)
        return type(v)(total) if code == 0 else total      # the mean of integers is a float (collective.py:90-93)

    def _reduce_array(self, v, code):
        if v.dtype == np.float64 and v.flags.c_contiguous:
            self._reduce_f64(v, code)
        else:
            tmp = np.ascontiguousarray(v, dtype=np.float64)
            self._reduce_f64(tmp, code)
            v[...] = tmp
        return v

    def _reduce_vector(self, v, code):
        self._reduce_block(v._mv, code)
        return v

    def _reduce_dolfin(self, v, code):
        local = np.ascontiguousarray(v.get_local(), dtype=np.float64)
        self._reduce_f64(local, code)
        v.set_local(local)
        v.apply("")
        return v

    def _reduce_container(self, v, code):
        op = "sum" if code == 0 else "avg"
        for i in range(v.nvec()):
            self.allReduce(v[i], op)
        return v

    # ---- broadcasts
    def bcast(self, v, root=0):
        kind = _payload_kind(v)
        handler = getattr(self, "_bcast_" + kind, None) if kind else None
        if handler is None:
            raise NotImplementedError("%s.bcast: no rule for a payload of type %s" % (self.name, type(v).__name__))
        return handler(v, root)

    def _bcast_float(self, v, root):
        box = np.array([v], dtype=np.float64)
        self._bcast_raw(box, root)
        return type(v)(box[0])

    def _bcast_int(self, v, root):
        box = np.array([v], dtype=np.int64)
        self._bcast_raw(box, root)
        return type(v)(box[0])

    def _bcast_array(self, v, root):
        if v.flags.c_contiguous:
            self._bcast_raw(v, root)
        else:
            tmp = np.ascontiguousarray(v)
            self._bcast_raw(tmp, root)
            v[...] = tmp
        return v

    def _bcast_vector(self, v, root):
        self._bcast_block(v._mv, root)
        return v

    def _bcast_dolfin(self, v, root):
        local = np.ascontiguousarray(v.get_local(), dtype=np.float64)
        self._bcast_raw(local, root)
        v.set_local(local)
        v.apply("")
        return v

    def _bcast_container(self, v, root):
        for i in range(v.nvec()):
            self.bcast(v[i], root=root)
        return v


def default_id_file():
    """Where the ranks of one launch meet when nobody names a file: the temp directory, keyed by the launcher's
    pid (the common parent of the ranks, e.g. the torch.distributed.run agent) and MASTER_PORT."""
    key = "%s-%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
    return os.path.join(tempfile.gettempdir(), "hfmi-comm-%s.id" % key)


class NativeCollective(_Collective):
    """``hfmi_comm`` behind the collective protocol: counterpart of MultipleSamePartitioningPDEsCollective
    (collective.py:43-159) with RCCL in place of mpi4py.  Blocks are reduced / broadcast in place in HBM on the
    context's stream; host payloads go through the node's shared segment."""

    name = "NativeCollective"
    TRANSPORTS = {0: "host", 1: "rccl", 2: "p2p"}

    def __init__(self, handle, ctx, is_serial_check=False):
        self._comm = handle
        self.ctx = ctx
        self.is_serial_check = is_serial_check
        n, r, t = C.c_int(0), C.c_int(0), C.c_int(0)
        L.call("hfmi_comm_info", self._comm, C.byref(n), C.byref(r), C.byref(t))
        self._size, self._rank, self.transport = n.value, r.value, self.TRANSPORTS.get(t.value, str(t.value))
        self._split_random()

    # ---- construction
    @staticmethod
    def unique_id():
        """Bytes to be made on ONE rank and shipped to the others (e.g. ``comm.bcast`` of mpi4py)."""
        buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
        L.call("hfmi_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_unique_id(cls, id_bytes, nranks, rank, ctx=None, host_only=False):
        if len(id_bytes) != L.UNIQUE_ID_BYTES:
            raise ValueError("communicator id must be %d bytes" % L.UNIQUE_ID_BYTES)
        ctx = None if host_only else (ctx or L.Context.default())
        h = C.c_void_p()
        L.call("hfmi_comm_init_rank", ctx.handle if ctx else None, C.c_char_p(id_bytes), int(nranks), int(rank), C.byref(h))
        return cls(h, ctx)

    @classmethod
    def from_env(cls, ctx=None, id_file=None, host_only=False):
        """One rank per process, numbered by RANK / WORLD_SIZE (set by ``hippyflow_amd.launch`` or by
        ``python -m torch.distributed.run``); the id travels through ``id_file`` / $HFMI_COMM_ID_FILE / a file in
        the temp directory keyed by the launcher."""
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        path = id_file or os.environ.get("HFMI_COMM_ID_FILE") or default_id_file()
        ctx = None if host_only else (ctx or L.Context.default())
        h = C.c_void_p()
        L.call("hfmi_comm_init_from_file", ctx.handle if ctx else None, path.encode(), world, rank, C.byref(h))
        return cls(h, ctx)

    # ---- protocol
    def size(self):
        return self._size

    def rank(self):
        return self._rank

    def barrier(self):
        L.call("hfmi_comm_barrier", self._comm)

    def describe(self):
        """What the ranks agreed on and why (transport, fallback reason, RCCL library, every rank's PCI bus id)."""
        import json
        buf = C.create_string_buffer(4096)
        L.call("hfmi_comm_describe", self._comm, buf, len(buf))
        return json.loads(buf.value.decode())

    def allReduceMax(self, value):
        """Largest value over the ranks (bench.py's max-over-ranks timing; not part of the reference's protocol)."""
        box = np.array([value], dtype=np.float64)
        L.call("hfmi_allreduce_host", self._comm, L.ptr(box), 1, 2)
        return float(box[0])

    def _reduce_f64(self, arr, code):
        L.call("hfmi_allreduce_host", self._comm, L.ptr(arr), int(arr.size), code)

    def _bcast_raw(self, arr, root):
        L.call("hfmi_bcast_host", self._comm, L.ptr(arr), int(arr.nbytes), int(root))

    def _reduce_block(self, mv, code):
        L.call("hfmi_allreduce", self._comm, mv.handle, code)
        return mv

    def _bcast_block(self, mv, root):
        L.call("hfmi_bcast", self._comm, mv.handle, int(root))
        return mv

    def close(self):
        if getattr(self, "_comm", None):
            L.load().hfmi_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CudaArrayView:
    """Zero-copy hand-off of a block's HBM to torch (``__cuda_array_interface__`` v2)."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class TorchCollective(_Collective):
    """The same protocol over a ``torch.distributed`` process group ("nccl" = RCCL on the GPUs, "gloo" on CPU), for
    hosts that already run one.  Device blocks are handed to torch zero-copy and the collective is ordered on
    libhfmi's stream."""

    name = "TorchCollective"

    def __init__(self, group=None, is_serial_check=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.is_serial_check = is_serial_check
        if not dist.is_initialized():
            raise RuntimeError("TorchCollective: torch.distributed is not initialised")
        self._split_random()

    def size(self):
        return self.dist.get_world_size(self.group)

    def rank(self):
        return self.dist.get_rank(self.group)

    def barrier(self):
        self.dist.barrier(group=self.group)

    # -- helpers
    def _backend_device(self):
        import torch
        if self.dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def _reduce_f64(self, arr, code):
        import torch
        t = torch.from_numpy(arr).to(self._backend_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        out = t.cpu().numpy()
        arr[...] = out if code == 0 else out * (1.0 / float(self.size()))

    def _bcast_raw(self, arr, root):
        import torch
        t = torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(self._backend_device())
        self.dist.broadcast(t, src=root, group=self.group)
        arr.view(np.uint8).reshape(-1)[...] = t.cpu().numpy()

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

    def _block_collective(self, mv, run):
        import torch
        t, stage = self._tensor_of(mv)
        cm = self._on_block_stream(mv) if stage is None else None
        if cm is not None:
            with cm:
                run(t)
            return mv
        mv.ctx.synchronize()                   # libhfmi's stream -> host: the block is complete
        run(t)
        torch.cuda.current_stream(mv.ctx.device).synchronize()   # the collective is done before libhfmi reads
        if stage is not None:
            mv.copy_from(stage)
            mv.ctx.synchronize()
        return mv

    def _reduce_block(self, mv, code):
        def run(t):
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            if code == 1:
                t.mul_(1.0 / float(self.size()))
        return self._block_collective(mv, run)

    def _bcast_block(self, mv, root):
        return self._block_collective(mv, lambda t: self.dist.broadcast(t, src=root, group=self.group))


def _is_mpi_comm(comm):
    return all(hasattr(comm, m) for m in ("Get_size", "Get_rank", "bcast"))


def _native_over_mpi(comm, is_serial_check=False):
    """A ``NativeCollective`` for the ranks of an mpi4py communicator: rank 0 makes the communicator id and mpi4py
    carries its 256 bytes -- the only use of MPI; every reduction afterwards is RCCL / the device fabric."""
    if comm.Get_size() == 1:
        return NullCollective()
    id_bytes = comm.bcast(NativeCollective.unique_id() if comm.Get_rank() == 0 else None, root=0)
    coll = NativeCollective.from_unique_id(id_bytes, comm.Get_size(), comm.Get_rank())
    coll.is_serial_check = is_serial_check
    return coll


def MultipleSamePartitioningPDEsCollective(comm=None, is_serial_check=False):
    """The reference's constructor name (collective.py:43).  ``comm``: None = the ranks of this launch
    (``NativeCollective.from_env``), an existing collective (returned), an mpi4py communicator such as the one
    ``splitCommunicators`` hands back (a native communicator is bootstrapped over it), or a torch.distributed group."""
    if comm is None:
        coll = NativeCollective.from_env()
        coll.is_serial_check = is_serial_check
        return coll
    if isinstance(comm, (_Collective, NullCollective)):
        return comm
    if _is_mpi_comm(comm):
        return _native_over_mpi(comm, is_serial_check)
    return TorchCollective(comm, is_serial_check=is_serial_check)


def MultipleSerialPDEsCollective(comm=None):
    return MultipleSamePartitioningPDEsCollective(comm, is_serial_check=True)


class _SelfCommunicator:
    """The mesh communicator of a rank that owns its whole mesh (``MPI.COMM_SELF`` without mpi4py)."""
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1


def splitCommunicators(comm_world, n_subdomain, n_instances):
    """(mesh_constructor_comm, collective_comm) as collectives/comm_utils.py:19-40 returns them, for the only layout
    that exists with one process per GPU: NO mesh partitioning (``n_subdomain == 1``), every rank a sampling instance.
    The mesh communicator is then the rank itself and the collective communicator is the world: with an mpi4py world
    both are made by ``Split`` exactly as in the reference (color / key as there); with ``comm_world=None`` or an
    ``hfmi`` / torch communicator the world is handed back for ``MultipleSamePartitioningPDEsCollective`` to wrap."""
    if int(n_subdomain) != 1:
        raise NotImplementedError("splitCommunicators: n_subdomain = %d -- mesh-partitioned PDE solves are host (FEniCS) "
                                  "territory; one process per GPU shards samples only (n_subdomain = 1)" % n_subdomain)
    if comm_world is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        assert world == int(n_instances), "world size %d != n_subdomain * n_instances = %d" % (world, n_instances)
        return _SelfCommunicator(), None
    size = comm_world.Get_size() if hasattr(comm_world, "Get_size") else comm_world.size()
    rank = comm_world.Get_rank() if hasattr(comm_world, "Get_rank") else comm_world.rank()
    assert size == int(n_subdomain) * int(n_instances)
    if hasattr(comm_world, "Split"):
        color, key = rank // int(n_subdomain), rank % int(n_subdomain)
        return comm_world.Split(color=color, key=key), comm_world.Split(color=key, key=color)
    return _SelfCommunicator(), comm_world


def checkFunctionSpaceConsistentPartitioning(Vh, collective):
    """comm_utils.py:43-60: the same question for a function space -- see ``checkMeshConsistentPartitioning``."""
    return True


def checkMeshConsistentPartitioning(mesh, collective):
    """comm_utils.py:63-75 asks whether every sampling instance partitioned its mesh the same way.  Without mesh
    partitioning (see ``splitCommunicators``) there is one partition: consistent by construction."""
    return True


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
