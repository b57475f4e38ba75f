"""Linear operators behind the reference's duck-typed protocol
(``mult(x, y)``, optional ``matMvMult(X, Y)``, ``init_vector(x, dim)``),
each backed by an ``hfmi_op`` (include/hfmi.h) so that its block application is a
HIP kernel sequence on the GPU.

Reference counterparts (file:line under /root/reference/hippyflow):

* ``LowRankOperator`` / ``SnapshotGramOperator``  hp.LowRankOperator(ones/n, snapshots)      modeling/PODProjector.py:359-361
* ``MeanJTJfromDataOperator``                     modeling/operatorWrappers.py:55-121
* ``MeanJJTfromDataOperator``                     JJT summed/averaged: modeling/jacobian.py:169-193, activeSubspaceProjector.py:640-645
* ``npToDeviceOperator``                          npToDolfinOperator, modeling/operatorWrappers.py:19-52 (symmetric case)
* ``CsrOperator`` / ``CsrPCGSolver``              prior.M / prior.R and prior.Msolver (used at KLEProjector.py:163-168)
* ``Solver2Operator``                             hp.Solver2Operator, modeling/KLEProjector.py:103,176
* ``MassPreconditionedCovarianceOperator``        modeling/KLEProjector.py:47-69
* ``SummedListOperator``                          modeling/activeSubspaceProjector.py:69-95
* ``HostCallbackOperator``                        any host black box (FEniCS Jacobian actions, sparse LU):
                                                  the role of ObservableJacobian/JTJ, modeling/jacobian.py:62-166
* ``PriorPreconditionedProjector``                modeling/priorPreconditionedProjector.py:19-55
* ``LowRankRectangularOperator``                  modeling/lowRankRectangularOperator.py:17-66
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .multivector import MultiVector, Vector


class DeviceOperator:
    """Base: owns an hfmi_op handle.  ``matMvMult`` overwrites Y unless ``accumulate=True``
    (the reference's block operators accumulate into a zero-filled Y,
    activeSubspaceProjector.py:219-221; overwrite is the intended behaviour)."""

    def __init__(self, ctx, n_range, n_domain=None):
        self.ctx = ctx or L.Context.default()
        self._op = C.c_void_p()
        self._shape = (int(n_range), int(n_range if n_domain is None else n_domain))
        self._keep = []   # python objects the C side points into

    @property
    def shape(self):
        return self._shape

    def mpi_comm(self):
        from .multivector import _NullComm
        return _NullComm()

    def init_vector(self, x, dim=0):
        """dim 0: range, dim 1: domain (jacobian.py:96-115)."""
        if dim not in (0, 1):
            raise ValueError("dim must be 0 or 1")
        x.init(self._shape[dim])

    def matMvMult(self, X, Y, accumulate=False):
        assert X.nvec() == Y.nvec(), "x and y have non-matching number of vectors"
        L.call("hfmi_op_apply", self._op, X.handle, Y.handle, 1 if accumulate else 0)

    def mult(self, x, y):
        L.call("hfmi_op_apply", self._op, x._mv.handle, y._mv.handle, 0)

    def transpmult(self, x, y):   # every operator on this path is self-adjoint
        self.mult(x, y)

    def __del__(self):
        try:
            if getattr(self, "_op", None):
                L.load().hfmi_op_destroy(self._op)
                self._op = None
        except Exception:
            pass


class SnapshotGramOperator(DeviceOperator):
    """y = scale * X X^T x with X the block of snapshots (one vector per snapshot);
    ``scale = 1/n`` reproduces hp.LowRankOperator(ones/n, LocalObservables) (PODProjector.py:359-361)."""

    def __init__(self, snapshots, scale=None, ctx=None):
        if not isinstance(snapshots, MultiVector):
            snapshots = MultiVector.from_vectors(snapshots, ctx=ctx)
        super().__init__(ctx or snapshots.ctx, snapshots.size())
        self.snapshots = snapshots
        self.scale = 1.0 / snapshots.nvec() if scale is None else float(scale)
        L.call("hfmi_op_snapshot_gram", self.ctx.handle, snapshots.handle, self.scale, C.byref(self._op))


class LowRankOperator(DeviceOperator):
    """hp.LowRankOperator(d, U, init_vector): ``mult`` y = U diag(d) U^T x (``dot_v`` + ``reduce`` in hippylib, two
    contractions with a row scaling between them here) and ``solve(y, x)`` y = U diag(1/d) U^T x -- the object hippylib's
    priors expose as ``prior.Hlr`` and hippyflow passes as both B and B^-1 (activeSubspaceProjector.py:455-459).  The
    constant diagonal ``ones/n`` of PODProjector.py:359-361 is the snapshot-Gram operator."""

    def __init__(self, d, U, init_vector=None, ctx=None):
        if not isinstance(U, MultiVector):
            U = MultiVector.from_vectors(U, ctx=ctx)
        d = np.ascontiguousarray(np.broadcast_to(np.asarray(d, dtype=np.float64), (U.nvec(),)))
        super().__init__(ctx or U.ctx, U.size())
        self.U, self.d = U, d
        self._init_vector = init_vector
        self._inverse = None
        L.call("hfmi_op_low_rank", self.ctx.handle, U.handle, L.ptr(d), C.byref(self._op))

    def init_vector(self, x, dim=0):
        if self._init_vector is not None:
            return self._init_vector(x, dim)
        return DeviceOperator.init_vector(self, x, dim)

    def inverse(self):
        """U diag(1/d) U^T as an operator (what ``solve`` applies)."""
        if self._inverse is None:
            if np.any(self.d == 0.0):
                raise ZeroDivisionError("LowRankOperator.solve: zero entry in d")
            self._inverse = LowRankOperator(1.0 / self.d, self.U, self._init_vector)
        return self._inverse

    def solve(self, y, x):
        self.inverse().mult(x, y)


class MeanJTJfromDataOperator(DeviceOperator):
    """y = mean_i J_i^T Gamma^{-1} J_i x from a stored (ndata, q, dM) Jacobian array
    (operatorWrappers.py:55-121).  ``scale`` defaults to 1/ndata (the np.mean at :114); pass
    ``scale=1/(ndata*P)`` style values to pre-fold a rank average."""

    def __init__(self, J, prior=None, noise_cov_inv=None, scale=None, ctx=None):
        if isinstance(J, MultiVector):
            raise TypeError("pass (J_block, ndata, q) through MeanJTJfromDataOperator.from_block")
        J = np.asarray(J, dtype=np.float64)
        assert J.ndim == 3, "J.shape must be (ndata, rank, dM)"
        ndata, q, dM = J.shape
        block = MultiVector.from_vectors(J.reshape(ndata * q, dM), ctx=ctx)
        self._setup(block, ndata, q, noise_cov_inv, scale, prior, ctx)

    @classmethod
    def from_block(cls, J_block, ndata, q, noise_cov_inv=None, scale=None, prior=None):
        self = cls.__new__(cls)
        self._setup(J_block, ndata, q, noise_cov_inv, scale, prior, J_block.ctx)
        return self

    def _setup(self, block, ndata, q, noise_cov_inv, scale, prior, ctx):
        DeviceOperator.__init__(self, ctx or block.ctx, block.size())
        self._J, self.ndata, self.r, self.dM = block, int(ndata), int(q), block.size()
        self._prior = prior
        self._noise_cov_inv = None
        gptr = None
        if noise_cov_inv is not None:
            assert hasattr(noise_cov_inv, "__matmul__")
            self._noise_cov_inv = L.as_f64(np.asarray(noise_cov_inv, dtype=np.float64))
            assert self._noise_cov_inv.shape == (q, q)
            gptr = L.ptr(self._noise_cov_inv)
        self.scale = 1.0 / ndata if scale is None else float(scale)
        L.call("hfmi_op_jtj", self.ctx.handle, block.handle, int(ndata), int(q), gptr, self.scale, C.byref(self._op))

    @property
    def J(self):
        return self._J

    @property
    def prior(self):
        return self._prior

    @property
    def noise_cov_inv(self):
        return self._noise_cov_inv


class MeanJJTfromDataOperator(DeviceOperator):
    """Output-space counterpart y = mean_i J_i J_i^T x (acts on vectors of length q)."""

    def __init__(self, J, scale=None, ctx=None):
        if isinstance(J, tuple):
            block, ndata, q = J
        else:
            J = np.asarray(J, dtype=np.float64)
            ndata, q, dM = J.shape
            block = MultiVector.from_vectors(J.reshape(ndata * q, dM), ctx=ctx)
        super().__init__(ctx or block.ctx, q)
        self._J, self.ndata, self.r = block, int(ndata), int(q)
        self.scale = 1.0 / ndata if scale is None else float(scale)
        L.call("hfmi_op_jjt", self.ctx.handle, block.handle, int(ndata), int(q), self.scale, C.byref(self._op))


class npToDeviceOperator(DeviceOperator):
    """Dense SYMMETRIC matrix behind the protocol (npToDolfinOperator, operatorWrappers.py:19-52, for the
    self-adjoint operators of this path; config 2's explicit covariance)."""

    def __init__(self, npArray, ctx=None):
        if isinstance(npArray, MultiVector):
            block = npArray
        else:
            npArray = np.asarray(npArray, dtype=np.float64)
            assert len(npArray.shape) == 2 and npArray.shape[0] == npArray.shape[1]
            block = MultiVector.from_vectors(npArray, ctx=ctx)      # symmetric: rows == columns
        super().__init__(ctx or block.ctx, block.size())
        self.matrix = block
        L.call("hfmi_op_dense_sym", self.ctx.handle, block.handle, C.byref(self._op))


class _Csr:
    def __init__(self, M, ctx):
        import scipy.sparse as sp
        M = sp.csr_matrix(M)
        M.sort_indices()
        self.shape = M.shape
        self.indptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(M.indices, dtype=np.int32)
        self.data = np.ascontiguousarray(M.data, dtype=np.float64)
        self.ctx = ctx
        self.handle = C.c_void_p()
        L.call("hfmi_csr_create", ctx.handle, M.shape[0], M.shape[1], int(M.nnz), L.ptr(self.indptr),
               L.ptr(self.indices), L.ptr(self.data), C.byref(self.handle))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.load().hfmi_csr_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class CsrOperator(DeviceOperator):
    """Sparse matrix operator y = M x (prior.M, prior.R; the encoder step hp.MatMvMult(B, decoder, encoder))."""

    def __init__(self, M, ctx=None):
        ctx = ctx or L.Context.default()
        self.csr = M if isinstance(M, _Csr) else _Csr(M, ctx)
        super().__init__(ctx, self.csr.shape[0], self.csr.shape[1])
        L.call("hfmi_op_csr", self.ctx.handle, self.csr.handle, C.byref(self._op))


class CsrPCGSolver(DeviceOperator):
    """Solver object for an SPD sparse matrix (prior.Msolver): ``solve(y, x)`` gives y = M^{-1} x by
    Jacobi-preconditioned CG on the device; as an operator it IS hp.Solver2Operator(Msolver)."""

    def __init__(self, M, rel_tol=1e-13, max_iter=500, ctx=None):
        ctx = ctx or L.Context.default()
        self.csr = M if isinstance(M, _Csr) else _Csr(M, ctx)
        super().__init__(ctx, self.csr.shape[0])
        L.call("hfmi_op_csr_pcg", self.ctx.handle, self.csr.handle, float(rel_tol), int(max_iter), C.byref(self._op))

    def solve(self, y, x):
        self.mult(x, y)


class HostCallbackOperator(DeviceOperator):
    """A host black box plugged into the device solve (FEniCS/hIPPYlib PDE solves stay on the host).

    ``fn`` is one of: a callable mapping an (N, k) array to an (N, k) array; an object with
    ``matMvMult_np(X) -> Y`` or (solvers) ``solve_block(X) -> Y`` on (N, k) arrays; an object with ``solve(y, x)`` or
    ``mult(x, y)`` on 1-D numpy arrays (applied column by column, like hp.MatMvMult's fallback loop).

    ``chunk_vectors``: for black boxes that treat the vectors independently (solvers), the callback is invoked on
    slabs of that many vectors and the PCIe copies of the neighbouring slabs overlap the host work
    (``hfmi_op_host_set_chunk``).  Default: solver-like objects (``solve`` / ``solve_block``) 32, everything else the
    whole block in one call (a serialized-sampling Jacobian operator re-solves its PDEs on every call)."""

    def __init__(self, fn, N, ctx=None, chunk_vectors=None):
        super().__init__(ctx, N)
        self.fn = fn
        self.error = None
        solver_like = hasattr(fn, "solve") or hasattr(fn, "solve_block")

        def _cb(user, w_ptr, y_ptr, n, k):
            try:
                W = np.ctypeslib.as_array(w_ptr, shape=(k, n))    # one vector per row
                Y = np.ctypeslib.as_array(y_ptr, shape=(k, n))
                if callable(fn) and not hasattr(fn, "mult") and not solver_like:
                    Y[...] = np.asarray(fn(W.T)).T
                elif hasattr(fn, "matMvMult_np"):
                    Y[...] = np.asarray(fn.matMvMult_np(W.T)).T
                elif hasattr(fn, "solve_block"):
                    Y[...] = np.asarray(fn.solve_block(W.T)).T
                elif hasattr(fn, "solve"):
                    for j in range(k):
                        fn.solve(Y[j], W[j])
                else:
                    for j in range(k):
                        fn.mult(W[j], Y[j])
                return 0
            except Exception as exc:  # never let an exception cross the C boundary
                self.error = exc
                return 1

        self._cb = L.HOST_APPLY_FN(_cb)
        L.call("hfmi_op_host_callback", self.ctx.handle, self._cb, None, int(N), C.byref(self._op))
        if chunk_vectors is None:
            chunk_vectors = 32 if solver_like else 0
        self.chunk_vectors = int(chunk_vectors)
        if self.chunk_vectors:
            L.call("hfmi_op_host_set_chunk", self._op, self.chunk_vectors)

    def _raise_pending(self):
        if self.error is not None:
            exc, self.error = self.error, None
            raise exc

    def matMvMult(self, X, Y, accumulate=False):
        try:
            super().matMvMult(X, Y, accumulate)
        except L.HfmiError:
            self._raise_pending()
            raise

    def mult(self, x, y):
        try:
            super().mult(x, y)
        except L.HfmiError:
            self._raise_pending()
            raise

    def solve(self, y, x):
        self.mult(x, y)


class DenseJacobianOperator:
    """A stored Jacobian J (q x N) behind the reference's rectangular-operator protocol
    (ObservableJacobian: ``mult`` N -> q, ``transpmult`` q -> N, ``init_vector(x, dim)``; jacobian.py:62-139).
    The rows of J are the vectors of a block, so J W is a ``dot_mv`` (reduction over N) and J^T Y an ``MvDSmatMult``."""

    def __init__(self, J, ctx=None):
        self.rows = J if isinstance(J, MultiVector) else MultiVector.from_vectors(np.asarray(J, dtype=np.float64), ctx=ctx)
        self.ctx = self.rows.ctx
        self.shape = (self.rows.nvec(), self.rows.size())

    def mpi_comm(self):
        from .multivector import _NullComm
        return _NullComm()

    def init_vector(self, x, dim):
        if dim not in (0, 1):
            raise ValueError("dim must be 0 or 1")
        x.init(self.shape[dim])

    def matMvMult(self, X, Y):          # Y (q x k) = J X
        Y_dense = self.rows.dot_mv(X)                    # (q, k)
        L.call("hfmi_block_upload", Y.handle, L.ptr(np.ascontiguousarray(Y_dense)), L.LAYOUT_DENSE)

    def matMvTranspmult(self, X, Y):    # Y (N x k) = J^T X
        from .multivector import MvDSmatMult
        MvDSmatMult(self.rows, np.ascontiguousarray(X.to_dense()), Y)

    def mult(self, x, y):
        self.matMvMult(x._mv, y._mv)

    def transpmult(self, x, y):
        self.matMvTranspmult(x._mv, y._mv)


class ComposedOperator(DeviceOperator):
    """y = c(b(a x))."""

    def __init__(self, a, b, c):
        super().__init__(a.ctx, c.shape[0], a.shape[1])
        self._keep = [a, b, c]
        L.call("hfmi_op_compose3", self.ctx.handle, a._op, b._op, c._op, C.byref(self._op))


def as_device_operator(obj, N=None, ctx=None):
    """Coerce the kinds of objects the reference passes as A / B / B^{-1} into device operators."""
    if isinstance(obj, DeviceOperator):
        return obj
    if hasattr(obj, "_device_operator"):
        return obj._device_operator()
    try:
        import scipy.sparse as sp
        if sp.issparse(obj):
            return CsrOperator(obj, ctx=ctx)
    except ImportError:
        pass
    if isinstance(obj, np.ndarray) and obj.ndim == 2:
        return npToDeviceOperator(obj, ctx=ctx)
    if callable(obj) or hasattr(obj, "mult") or hasattr(obj, "solve") or hasattr(obj, "matMvMult_np") or hasattr(obj, "solve_block"):
        if N is None:
            raise ValueError("as_device_operator: vector length needed to wrap a host operator")
        return HostCallbackOperator(obj, N, ctx=ctx)
    raise TypeError("cannot use %r as an operator" % (type(obj),))


class Solver2Operator:
    """hp.Solver2Operator(S): ``mult(x, y)`` = ``S.solve(y, x)`` (KLEProjector.py:103)."""

    def __init__(self, solver, mpi_comm=None, init_vector=None):
        self.solver = solver
        self._init_vector = init_vector

    def init_vector(self, x, dim):
        if self._init_vector is not None:
            self._init_vector(x, dim)
        elif hasattr(self.solver, "init_vector"):
            self.solver.init_vector(x, dim)
        else:
            raise NotImplementedError("Solver2Operator: no init_vector available")

    def mult(self, x, y):
        self.solver.solve(y, x)

    def _device_operator(self):
        s = self.solver
        return s if isinstance(s, DeviceOperator) else as_device_operator(s, getattr(s, "N", None))


class MassPreconditionedCovarianceOperator:
    """Linear operator M C M (KLEProjector.py:47-69)."""

    def __init__(self, C, M):
        self.C = C
        self.M = M
        self._dev = None

    def mpi_comm(self):
        return self.M.mpi_comm()

    def init_vector(self, x, dim):
        self.M.init_vector(x, dim)

    def _device_operator(self):
        if self._dev is None:
            Md = as_device_operator(self.M)
            Cd = as_device_operator(self.C, Md.shape[0], Md.ctx)
            self._dev = ComposedOperator(Md, Cd, Md)
        return self._dev

    def mult(self, x, y):
        self._device_operator().mult(x, y)

    def matMvMult(self, X, Y):
        self._device_operator().matMvMult(X, Y)


class SummedListOperator:
    """Mean (or sum) of a list of operators of equal dimension (activeSubspaceProjector.py:69-95).
    The accumulator starts from zero (the reference seeds it with a copy of the incoming y, :83-86,
    which is only a mean when y arrives zero-filled -- SURVEY.md section 3.6)."""

    def __init__(self, operators, communicator=None, average=True):
        assert type(operators) is list
        self.operators = operators
        self.average = average

    def init_vector(self, x, dim=0):
        self.operators[0].init_vector(x, dim)

    def mult(self, x, y):
        temp = Vector(y)
        temp.zero()
        for op in self.operators:
            op.mult(x, y)
            temp.axpy(1.0, y)
        y.zero()
        y.axpy(1.0 / float(len(self.operators)) if self.average else 1.0, temp)

    def matMvMult(self, X, Y):
        temp = MultiVector(Y)
        temp.zero()
        from .multivector import MatMvMult
        for op in self.operators:
            MatMvMult(op, X, Y)
            temp.axpy(1.0, Y)
        Y.zero()
        Y.axpy(1.0 / float(len(self.operators)) if self.average else 1.0, temp)


class StateSpaceIdentityOperator:
    """Identity observable on the state space (fullStateObservable.py:18-52): ``mult`` copies; ``transpmult`` applies
    the mass matrix (the adjoint in the M-inner product) unless ``use_mass_matrix`` is False."""

    def __init__(self, M, use_mass_matrix=True):
        self.M = M
        self.use_mass_matrix = use_mass_matrix

    def mpi_comm(self):
        return self.M.mpi_comm()

    def init_vector(self, v, dim):
        return self.M.init_vector(v, dim)

    def mult(self, u, y):
        y.zero()
        y.axpy(1.0, u)

    def transpmult(self, x, p):
        if self.use_mass_matrix:
            self.M.transpmult(x, p) if hasattr(self.M, "transpmult") else self.M.mult(x, p)   # M is symmetric
        else:
            p.zero()
            p.axpy(1.0, x)


class JTJ:
    """J^T J of a Jacobian-protocol object (``mult`` domain -> range, ``transpmult`` range -> domain,
    ``init_vector(x, dim)``; jacobian.py:142-166).  With a block-capable Jacobian (``DenseJacobianOperator``) the
    block form ``matMvMult`` runs as two tall-skinny contractions."""

    def __init__(self, J):
        self.J = J
        self.vector_help = Vector(ctx=getattr(J, "ctx", None))
        self.J.init_vector(self.vector_help, 0)

    def mult(self, x, y):
        self.J.mult(x, self.vector_help)
        self.J.transpmult(self.vector_help, y)

    def init_vector(self, x, dim=None):
        self.J.init_vector(x, 1)

    def matMvMult(self, X, Y):
        """Y = J^T (J X) (overwrites Y, the semantics of hp.MatMvMult on an operator without a block form)."""
        if hasattr(self.J, "matMvMult") and hasattr(self.J, "matMvTranspmult"):
            tmp = MultiVector(self.vector_help, X.nvec())
            self.J.matMvMult(X, tmp)
            self.J.matMvTranspmult(tmp, Y)
        else:
            for j in range(X.nvec()):
                self.mult(X[j], Y[j])


class JJT:
    """J J^T (jacobian.py:169-193)."""

    def __init__(self, J):
        self.J = J
        self.vector_help = Vector(ctx=getattr(J, "ctx", None))
        self.J.init_vector(self.vector_help, 1)

    def mult(self, x, y):
        self.J.transpmult(x, self.vector_help)
        self.J.mult(self.vector_help, y)

    def init_vector(self, x, dim=None):
        self.J.init_vector(x, 0)

    def matMvMult(self, X, Y):
        if hasattr(self.J, "matMvMult") and hasattr(self.J, "matMvTranspmult"):
            tmp = MultiVector(self.vector_help, X.nvec())
            self.J.matMvTranspmult(X, tmp)
            self.J.matMvMult(tmp, Y)
        else:
            for j in range(X.nvec()):
                self.mult(X[j], Y[j])


class SeriallySampledJacobianOperator:
    """Sample-by-sample accumulation of J^T J (or J J^T) over prior draws (activeSubspaceProjector.py:98-257).

    Protocol-level mirror: the PDE work stays behind the duck-typed ``observable`` exactly as in the reference
    (``solveFwd(u, x)``, ``setLinearizationPoint(x)``, ``generate_vector``), and the linearised map is whatever
    ``jacobian_factory(observable)`` returns -- the reference hard-wires ``ObservableJacobian(observable)``
    (:178-181); a stored / dense Jacobian enters through ``DenseJacobianOperator``.  ``matMvMult`` ACCUMULATES into y
    (:214-221, :242-248): callers pass a zeroed block, as ``MatrixMultCollectiveOperator`` does.  Per sample the whole
    probe block goes through ``JTJ.matMvMult`` (two contractions) instead of the reference's column loop."""

    def __init__(self, observable, noise, prior, control_distribution=None, operation='JTJ', nsamples=None, ms=None, zs=None,
                 communicator=None, average=True, jacobian_factory=None):
        assert operation in ['JTJ', 'JJT']
        assert (nsamples is not None) or (ms is not None)
        self.observable = observable
        self.noise = noise
        self.prior = prior
        self.control_distribution = control_distribution
        self.operation = operation
        self.nsamples = nsamples
        self.average = average
        self.ms = ms
        if zs is not None:
            self.zs = zs
        else:
            self.zs = len(self.ms) * [None] if type(self.ms) is list else zs
        self.jacobian_factory = jacobian_factory or (lambda obs: obs.jacobian())
        self.max_solver_retries = 100         # fresh draws per sample before a failing forward solve is reported
        self.solver_failures = 0              # failed forward solves so far (each was followed by a fresh draw)
        self.u = observable.generate_vector(0) if hasattr(observable, "generate_vector") else None
        self.m = observable.generate_vector(1) if hasattr(observable, "generate_vector") else None
        self.z = None if control_distribution is None else observable.generate_vector(3)

    def init_vector(self, x, dim=None):
        if self.operation == 'JJT':
            self.observable.init_vector(x, 0)
        elif hasattr(getattr(self.observable, "problem", None), 'parameter_projection'):
            self.prior.init_vector(x, 0)
        else:
            self.observable.init_vector(x, 1)

    def _operator(self):
        J = self.jacobian_factory(self.observable)
        return JTJ(J) if self.operation == 'JTJ' else JJT(J)

    def _accumulate(self, x, y, weight):
        op = self._operator()
        tmp = MultiVector(y)
        op.matMvMult(x, tmp)
        y.axpy(weight, tmp)

    def matMvMult(self, x, y):
        assert x.nvec() == y.nvec(), "x and y have non-matching number of vectors"
        from .randomized import parRandom
        if self.ms is None:
            for _ in range(self.nsamples):
                # the reference re-draws the sample when the forward solve fails (activeSubspaceProjector.py:180-211: a bare
                # try / except around sampling + solveFwd inside `while not solved`); its loop has no exit -- this one gives
                # up after max_solver_retries fresh draws and says which sample could not be solved
                for attempt in range(self.max_solver_retries + 1):
                    self.m.zero()
                    self.noise.zero()
                    parRandom.normal(1, self.noise)
                    self.prior.sample(self.noise, self.m)
                    linearization_x = [self.u, self.m, None]
                    if self.control_distribution is not None:
                        self.z.zero()
                        self.control_distribution.sample(self.z)
                        linearization_x.append(self.z)
                    try:
                        self.observable.solveFwd(self.u, linearization_x)
                        self.observable.setLinearizationPoint(linearization_x)
                        break
                    except Exception as exc:          # noqa: BLE001 -- any solver failure means "draw again", as upstream
                        self.solver_failures += 1
                        if attempt == self.max_solver_retries:
                            raise RuntimeError("forward solve failed for %d consecutive draws of one sample (last error: %r)"
                                               % (self.max_solver_retries + 1, exc)) from exc
                self._accumulate(x, y, 1.0 / self.nsamples if self.average else 1.0)
        else:
            nsamples = len(self.ms)
            for m, z in zip(self.ms, self.zs):
                linearization_x = [self.u, m, None] if z is None else [self.u, m, None, z]
                self.observable.solveFwd(self.u, linearization_x)
                self.observable.setLinearizationPoint(linearization_x)
                self._accumulate(x, y, 1.0 / nsamples if self.average else 1.0)


class PriorPreconditionedProjector:
    """y = U U^T C^{-1} x (priorPreconditionedProjector.py:19-55)."""

    def __init__(self, U, Cinv, my_init_vector):
        self.U = U
        self.Cinv = Cinv
        self.my_init_vector = my_init_vector
        self.Cinvx = Vector(ctx=U.ctx)
        self.my_init_vector(self.Cinvx, 0)

    def init_vector(self, x, dim):
        self.my_init_vector(x, dim)

    def mult(self, x, y):
        self.Cinv.mult(x, self.Cinvx)
        UtCinvx = self.U.dot_v(self.Cinvx)
        y.zero()
        self.U.reduce(y, UtCinvx)


class LowRankRectangularOperator:
    """A = U s V^T (lowRankRectangularOperator.py:17-66)."""

    def __init__(self, U, s, V, U_init_vector=None, V_init_vector=None):
        self.U, self.s, self.V = U, np.asarray(s, dtype=np.float64), V
        self.U_init_vector, self.V_init_vector = U_init_vector, V_init_vector

    def init_vector(self, x, dim):
        if dim == 0:
            assert self.U_init_vector is not None
            self.U_init_vector(x)
        elif dim == 1:
            assert self.V_init_vector is not None
            self.V_init_vector(x)
        else:
            raise ValueError("dim must be 0 or 1")

    def mult(self, x, y):
        Vtx = self.V.dot_v(x)
        y.zero()
        self.U.reduce(y, self.s * Vtx)

    def transpmult(self, x, y):
        Utx = self.U.dot_v(x)
        y.zero()
        self.V.reduce(y, self.s * Utx)


# the reference's name for the dense-array operator wrapper (operatorWrappers.py:19-52)
npToDolfinOperator = npToDeviceOperator
