"""Device-resident counterparts of the vector / MultiVector protocol the reference's
hot path is written against (SURVEY.md section 8b "Block container").

hippylib's ``MultiVector`` (a C++ dolfin extension: ``nvec``, ``[j]``, ``zero``,
``dot_v``, ``dot_mv``, ``reduce``, ``orthogonalize``, ``Borthogonalize``, ``swap``)
and the dolfin ``Vector`` members hippyflow touches (``init``, ``get_local``,
``set_local``, ``axpy``, ``zero``, ``inner``, ``norm``, ``apply``,
``gather_on_zero``, ``mpi_comm``) are mirrored with the same names and argument
meaning; the storage is an ``hfmi_block`` in HBM (include/hfmi.h) and every
operation is a HIP kernel launched through the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class _NullComm:
    """Stand-in for ``mpi_comm()`` of a serial dolfin vector (mesh communicator of size 1)."""
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1


class MultiVector:
    """hp.MultiVector(vector_like, nvec) / hp.MultiVector(other) (copy)."""

    def __init__(self, v=None, nvec=None, ctx=None, _handle=None, _parent=None, _borrowed=False):
        self.ctx = ctx or getattr(v, "ctx", None) or L.Context.default()
        self._parent = _parent          # keeps the owning block alive for views
        self._borrowed = _borrowed      # handle owned by the C side (post-apply hooks): never destroyed here
        self._views = {}
        if _handle is not None:
            self.handle = _handle
        elif isinstance(v, MultiVector) and nvec is None:
            self.handle = C.c_void_p()
            L.call("hfmi_block_create", self.ctx.handle, v.size(), v.nvec(), C.byref(self.handle))
            L.call("hfmi_block_copy", self.handle, v.handle)
        else:
            n = v if isinstance(v, (int, np.integer)) else v.size()
            if n <= 0:
                raise ValueError("MultiVector: the template vector is not initialised (call init_vector first)")
            self.handle = C.c_void_p()
            L.call("hfmi_block_create", self.ctx.handle, int(n), int(nvec), C.byref(self.handle))
        N, k, ld, p = C.c_int64(), C.c_int(), C.c_int64(), C.c_void_p()
        L.call("hfmi_block_info", self.handle, C.byref(N), C.byref(k), C.byref(ld), C.byref(p))
        self._N, self._k, self._ld, self._ptr = N.value, k.value, ld.value, p.value

    # ---- construction helpers
    @classmethod
    def from_dense(cls, dense, ctx=None):
        """(N, nvec) array in the ``mv_to_dense`` layout -> block."""
        dense = L.as_f64(dense)
        mv = cls(int(dense.shape[0]), int(dense.shape[1]), ctx=ctx)
        L.call("hfmi_block_upload", mv.handle, L.ptr(dense), L.LAYOUT_DENSE)
        return mv

    @classmethod
    def from_vectors(cls, rows, ctx=None):
        """(nvec, N) array, one vector per row (u_data / q_data / stacked Jacobian rows) -> block."""
        rows = L.as_f64(rows)
        mv = cls(int(rows.shape[1]), int(rows.shape[0]), ctx=ctx)
        L.call("hfmi_block_upload", mv.handle, L.ptr(rows), L.LAYOUT_VECTORS)
        return mv

    def upload_async(self, rows, layout="vectors"):
        """Streaming ingest (PODProjector.py:343-357, activeSubspaceProjector.py:178-221: the reference fills its blocks
        sample by sample): copy a PINNED host array (``pinned_empty``) into this block -- normally a ``view`` of one
        sample's vectors -- on the context's ingest stream, without blocking.  Returns a ticket: ``ctx.ingest_wait(ticket)``
        before the pinned array is overwritten, ``ctx.ingest_fence()`` before compute that reads the block is enqueued."""
        if not isinstance(rows, np.ndarray) or rows.dtype != np.float64 or not rows.flags.c_contiguous:
            raise ValueError("upload_async: a C-contiguous float64 array from pinned_empty() is required")
        want = (self._k, self._N) if layout == "vectors" else (self._N, self._k)
        if tuple(rows.shape) != want:
            raise ValueError("upload_async: array of shape %s expected, got %s" % (want, tuple(rows.shape)))
        t = C.c_int64(-1)
        L.call("hfmi_block_upload_async", self.handle, L.ptr(rows), L.LAYOUT_VECTORS if layout == "vectors" else L.LAYOUT_DENSE,
               C.byref(t))
        return t.value

    def to_dense(self):
        out = np.empty((self._N, self._k), dtype=np.float64)
        L.call("hfmi_block_download", self.handle, L.ptr(out), L.LAYOUT_DENSE)
        return out

    def to_vectors(self):
        out = np.empty((self._k, self._N), dtype=np.float64)
        L.call("hfmi_block_download", self.handle, L.ptr(out), L.LAYOUT_VECTORS)
        return out

    # ---- protocol
    def nvec(self):
        return self._k

    def size(self):
        return self._N

    def __len__(self):
        return self._k

    def __getitem__(self, j):
        if not -self._k <= j < self._k:
            raise IndexError(j)
        j %= self._k
        if j not in self._views:
            h = C.c_void_p()
            L.call("hfmi_block_view", self.handle, j, 1, C.byref(h))
            self._views[j] = Vector(ctx=self.ctx, _mv=MultiVector(ctx=self.ctx, _handle=h, _parent=self))
        return self._views[j]

    def view(self, first, count):
        h = C.c_void_p()
        L.call("hfmi_block_view", self.handle, int(first), int(count), C.byref(h))
        return MultiVector(ctx=self.ctx, _handle=h, _parent=self)

    def zero(self):
        L.call("hfmi_block_zero", self.handle)

    def scale(self, alpha):
        L.call("hfmi_block_scale", self.handle, float(alpha))

    def axpy(self, alpha, X):
        L.call("hfmi_block_axpy", self.handle, float(alpha), X.handle)

    def copy_from(self, X):
        L.call("hfmi_block_copy", self.handle, X.handle)

    def swap(self, other):
        """MultiVector.swap: exchange storage with another block of the same shape."""
        if (self._N, self._k) != (other._N, other._k):
            raise ValueError("swap: shapes differ")
        for name in ("handle", "_parent", "_views", "_ptr", "_ld"):
            a, b = getattr(self, name), getattr(other, name)
            setattr(self, name, b)
            setattr(other, name, a)

    def norm(self, kind="l2"):
        if kind != "l2":
            raise NotImplementedError(kind)
        out = np.empty(self._k)
        L.call("hfmi_block_norms", self.handle, L.ptr(out))
        return out

    def dot_mv(self, other):
        """(nvec x other.nvec) matrix of inner products <self_i, other_j>."""
        out = np.empty((self._k, other.nvec()))
        L.call("hfmi_block_dot", self.handle, other.handle, L.ptr(out))
        return out

    def gram_eig(self, other, nvec, sort_by_abs=False):
        """``la.eigh(self^T other)`` with eigenvalues descending and the ``nvec`` leading eigenvectors: the Gram matrix is formed
        on the device and never leaves it (PODProjector.py:818-826: ``UtMU = u_data @ M @ u_data.T``, ``eigh``,
        ``U[:, :u_rank]``).  Both blocks hold the same number n <= 16384 of vectors; returns (d (n,), V (n, nvec))."""
        n, nvec = self._k, int(nvec)
        d, V = np.empty(n), np.empty((n, nvec))
        L.call("hfmi_block_gram_eig", self.handle, other.handle, 1 if sort_by_abs else 0, nvec, L.ptr(d), L.ptr(V))
        return d, V

    def dot_v(self, x):
        """Inner products of every vector with the vector x."""
        mv = x._mv if isinstance(x, Vector) else x
        out = np.empty((self._k, mv.nvec()))
        L.call("hfmi_block_dot", self.handle, mv.handle, L.ptr(out))
        return out[:, 0].copy()

    def reduce(self, y, alpha):
        """y += sum_i alpha_i self_i."""
        alpha = L.as_f64(np.asarray(alpha, dtype=np.float64).reshape(self._k, 1))
        L.call("hfmi_block_gemm_small", self.handle, L.ptr(alpha), 1.0, 1.0, y._mv.handle)

    def orthogonalize(self, method=L.QR_AUTO):
        """Thin QR in place (Q^T Q = I); returns R (nvec x nvec, upper triangular)."""
        R = np.zeros((self._k, self._k))
        passes = C.c_int(0)
        L.call("hfmi_borth_qr", self.handle, None, None, L.ptr(R), int(method), C.byref(passes))
        self.last_qr_passes = passes.value
        return R

    def Borthogonalize(self, B, method=L.QR_AUTO):
        """Thin QR in place in the B inner product (Q^T B Q = I); returns (BQ, R)."""
        from .operators import as_device_operator
        Bop = as_device_operator(B, self._N, self.ctx)
        BQ = MultiVector(self._N, self._k, ctx=self.ctx)
        R = np.zeros((self._k, self._k))
        passes = C.c_int(0)
        L.call("hfmi_borth_qr", self.handle, Bop._op, BQ.handle, L.ptr(R), int(method), C.byref(passes))
        self.last_qr_passes = passes.value
        return BQ, R

    def device_ptr(self):
        return self._ptr

    def leading_dimension(self):
        return self._ld

    def __del__(self):
        try:
            if getattr(self, "handle", None) and not getattr(self, "_borrowed", False):
                L.load().hfmi_block_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


class Vector:
    """The dolfin.Vector surface used by the reference's operators, backed by one block column."""

    def __init__(self, other=None, ctx=None, _mv=None):
        self.ctx = ctx or getattr(other, "ctx", None) or L.Context.default()
        self._mv = _mv
        if isinstance(other, Vector) and other._mv is not None:   # dl.Vector(y): copy
            self._mv = MultiVector(other._mv)

    def init(self, n):
        self._mv = MultiVector(int(n), 1, ctx=self.ctx)

    def size(self):
        return 0 if self._mv is None else self._mv.size()

    def local_size(self):
        return self.size()

    def mpi_comm(self):
        return _NullComm()

    def get_local(self):
        return self._mv.to_vectors()[0]

    def gather_on_zero(self):
        return self.get_local()

    def set_local(self, a):
        a = L.as_f64(np.asarray(a, dtype=np.float64).reshape(1, -1))
        if a.shape[1] != self.size():
            raise ValueError("set_local: expected %d entries, got %d" % (self.size(), a.shape[1]))
        L.call("hfmi_block_upload", self._mv.handle, L.ptr(a), L.LAYOUT_VECTORS)

    def apply(self, mode=""):
        pass

    def zero(self):
        self._mv.zero()

    def axpy(self, alpha, x):
        self._mv.axpy(alpha, x._mv)

    def inner(self, x):
        return float(self._mv.dot_mv(x._mv)[0, 0])

    def norm(self, kind="l2"):
        return float(self._mv.norm(kind)[0])

    def copy(self):
        return Vector(self)

    def __imul__(self, alpha):
        self._mv.scale(alpha)
        return self


def MatMvMult(A, X, Y):
    """hp.MatMvMult: ``A.matMvMult(X, Y)`` when the operator has it, else one ``mult`` per vector."""
    if X.nvec() != Y.nvec():
        raise AssertionError("x and y have non-matching number of vectors")
    if hasattr(A, "matMvMult"):
        A.matMvMult(X, Y)
    else:
        for j in range(X.nvec()):
            A.mult(X[j], Y[j])


def MatMvTranspmult(A, X, Y):
    if X.nvec() != Y.nvec():
        raise AssertionError("x and y have non-matching number of vectors")
    if hasattr(A, "matMvTranspmult"):
        A.matMvTranspmult(X, Y)
    else:
        for j in range(X.nvec()):
            A.transpmult(X[j], Y[j])


def MvDSmatMult(X, A_small, Y):
    """hp.MvDSmatMult: Y_j = sum_i X_i A[i, j]."""
    A_small = L.as_f64(A_small)
    if A_small.shape != (X.nvec(), Y.nvec()):
        raise AssertionError("MvDSmatMult: matrix shape %s does not match (%d, %d)" % (A_small.shape, X.nvec(), Y.nvec()))
    L.call("hfmi_block_gemm_small", X.handle, L.ptr(A_small), 1.0, 0.0, Y.handle)


def ingest_stream(items, count, vectors_per_item, N, ctx=None, nbuf=2):
    """Fill a block of ``count * vectors_per_item`` vectors from an iterable that PRODUCES its items on the host one at
    a time -- the reference's sampling loops (PODProjector.py:343-357: one observable per PDE solve;
    activeSubspaceProjector.py:178-221: one Jacobian per sample).  Item i (``(vectors_per_item, N)`` or ``(N,)``) is
    copied into one of ``nbuf`` pinned buffers and uploaded asynchronously while the producer works on item i + 1;
    nothing here waits for the GPU except when a pinned buffer is about to be reused.  Returns the block; compute
    enqueued afterwards sees all of it (``ingest_fence``)."""
    ctx = ctx or L.Context.default()
    block = MultiVector(int(N), int(count) * int(vectors_per_item), ctx=ctx)
    bufs = [L.pinned_empty((int(vectors_per_item), int(N))) for _ in range(nbuf)]
    tickets = [None] * nbuf
    got = 0
    for i, item in enumerate(items):
        if i >= count:
            raise ValueError("ingest_stream: the producer yielded more than %d items" % count)
        b = i % nbuf
        if tickets[b] is not None:
            ctx.ingest_wait(tickets[b])
        np.copyto(bufs[b], np.asarray(item, dtype=np.float64).reshape(vectors_per_item, N))
        tickets[b] = block.view(i * vectors_per_item, vectors_per_item).upload_async(bufs[b])
        got = i + 1
    if got != count:
        raise ValueError("ingest_stream: the producer yielded %d of %d items" % (got, count))
    for t in tickets:
        if t is not None:
            ctx.ingest_wait(t)             # the pinned buffers are freed when this function returns
    ctx.ingest_fence()
    return block


# ---- dense converters (hippyflow/utilities/mv_utilities.py:18-54): the layout contract at the host / device boundary.  A dense
# array is (N, nvec), C-ordered, as saved in ``*_decoder.npy`` / ``POD_projector.npy``.
def mv_to_dense_local(multivector):
    return multivector.to_dense()


def mv_to_dense(multivector):
    """One process per GPU and no mesh partition on the device: gather_on_zero is the identity."""
    return multivector.to_dense()


def dense_to_mv_local(dense_array, dl_vector=None):
    """(N, nvec) array -> MultiVector; ``dl_vector`` (a template vector) is accepted for signature
    compatibility and only supplies the context."""
    return MultiVector.from_dense(np.asarray(dense_array, dtype=np.float64), ctx=getattr(dl_vector, "ctx", None))
