"""Host side of the reference's vector protocol.

The PDE objects of a hippyflow run -- observable, prior, their matrices and solvers -- stay FEniCS / hIPPYlib objects on
the host, and everything they accept or hand back is a ``dolfin.Vector`` (``init``, ``get_local``, ``set_local``,
``zero``, ``axpy``, ``inner``, ``*=`` ...) made compatible by the operator's own ``init_vector(x, dim)``.  The device
solve reaches them the way the reference's collective moves data (collectives/collective.py:98-107): ``get_local`` /
``set_local`` on vectors the host object itself shaped.

``new_host_vector(comm)`` makes such a vector: a real ``dolfin.Vector(comm)`` when dolfin can be imported (so that PETSc
matrices can shape it), else the numpy-backed ``HostVector`` below, which is also what observables written in numpy
receive.  ``HostMultiVector`` is the column list ``hp.MultiVector`` is on the host (``nvec``, ``[j]``, ``zero``),
for host operators that offer the block form ``matMvMult(X, Y)``.

Component indices of ``observable.generate_vector`` (hippylib's STATE / PARAMETER / ADJOINT, hippyflow's CONTROL,
modeling/observable.py:18): ``STATE, PARAMETER, ADJOINT, CONTROL = 0, 1, 2, 3``.
"""
import numpy as np

STATE, PARAMETER, ADJOINT, CONTROL = 0, 1, 2, 3

_factory = None


class _SelfComm:
    """Communicator of one process (what ``mpi_comm()`` of a serial vector returns)."""
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1


class HostVector:
    """numpy-backed vector with the ``dolfin.Vector`` members the reference's path uses."""

    def __init__(self, arg=None):
        self._a = np.zeros(0)
        self._comm = arg if (arg is not None and not isinstance(arg, HostVector)) else _SelfComm()
        if isinstance(arg, HostVector):             # dl.Vector(other): copy
            self._a = arg._a.copy()
            self._comm = arg._comm

    def init(self, n):
        self._a = np.zeros(int(n))

    def size(self):
        return int(self._a.shape[0])

    def local_size(self):
        return self.size()

    def mpi_comm(self):
        return self._comm

    def get_local(self):
        return self._a.copy()

    def set_local(self, values):
        values = np.asarray(values, dtype=np.float64).reshape(-1)
        if values.shape[0] != self._a.shape[0]:
            raise ValueError("set_local: expected %d entries, got %d" % (self._a.shape[0], values.shape[0]))
        self._a[...] = values

    def gather_on_zero(self):
        return self.get_local()

    def apply(self, mode=""):
        pass

    def zero(self):
        self._a[...] = 0.0

    def axpy(self, alpha, x):
        self._a += float(alpha) * x._a

    def inner(self, x):
        return float(self._a @ x._a)

    def norm(self, kind="l2"):
        if kind == "l2":
            return float(np.linalg.norm(self._a))
        if kind == "linf":
            return float(np.abs(self._a).max()) if self._a.size else 0.0
        raise NotImplementedError(kind)

    def copy(self):
        return HostVector(self)

    def __imul__(self, alpha):
        self._a *= float(alpha)
        return self

    def __mul__(self, other):
        out = HostVector(self)
        out._a *= other._a if isinstance(other, HostVector) else float(other)
        return out

    __rmul__ = __mul__

    def __len__(self):
        return self.size()


class HostMultiVector:
    """List of host vectors of one shape: ``hp.MultiVector(v, nvec)`` / copy constructor, on the host."""

    def __init__(self, v, nvec=None):
        if isinstance(v, HostMultiVector) and nvec is None:
            self._cols = [_copy_vector(c) for c in v._cols]
        else:
            self._cols = [_copy_vector(v) for _ in range(int(nvec))]
            self.zero()

    def nvec(self):
        return len(self._cols)

    def __len__(self):
        return len(self._cols)

    def __getitem__(self, j):
        return self._cols[j]

    def zero(self):
        for c in self._cols:
            c.zero()


def _copy_vector(v):
    if isinstance(v, HostVector):
        return HostVector(v)
    if hasattr(v, "copy"):
        return v.copy()
    return type(v)(v)


def set_host_vector_factory(factory):
    """Override how host vectors are made (``factory(comm) -> vector``); ``None`` restores the default."""
    global _factory
    _factory = factory


def new_host_vector(comm=None):
    """An uninitialised host vector for ``init_vector`` to shape."""
    if _factory is not None:
        return _factory(comm)
    try:
        import dolfin
        if hasattr(dolfin, "Vector") and hasattr(dolfin, "PETScMatrix"):          # a real FEniCS, not a stand-in module
            return dolfin.Vector(comm) if comm is not None else dolfin.Vector()
    except ImportError:
        pass
    return HostVector(comm)


def is_host_vector(v):
    """A dolfin-like host vector (anything with get_local / set_local that is not device-resident)."""
    return hasattr(v, "get_local") and hasattr(v, "set_local") and not hasattr(v, "_mv")


def shape_with(init_vector, dim, comm=None):
    """A new host vector shaped by ``init_vector(x, dim)``; operators whose ``init_vector`` takes the vector only
    (activeSubspaceProjector.py:144) are served too."""
    x = new_host_vector(comm)
    try:
        init_vector(x, dim)
    except TypeError:
        init_vector(x)
    return x


def find_init_vector(obj):
    """Where hp.Solver2Operator looks for the shape of a solver's vectors: the object, its ``operator()``, its
    ``get_operator()``."""
    if hasattr(obj, "init_vector"):
        return obj.init_vector
    for getter in ("operator", "get_operator"):
        if hasattr(obj, getter):
            try:
                op = getattr(obj, getter)()
            except Exception:          # noqa: BLE001 -- a solver without an operator set
                continue
            if hasattr(op, "init_vector"):
                return op.init_vector
    return None
