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
from . import hostvec as H
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
    """Solver object for an SPD sparse matrix (prior.Msolver): ``solve(y, x)`` gives y = M^{-1} x to a relative
    residual ``rel_tol`` per vector, on the device; as an operator it IS hp.Solver2Operator(Msolver).  The work is done by a
    Jacobi-preconditioned Chebyshev iteration (hfmi_cheb.hip) when the Jacobi-scaled spectrum is narrow -- mass matrices --
    and by Jacobi-preconditioned block CG otherwise; ``info()`` tells which."""

    def __init__(self, M, rel_tol=1e-13, max_iter=500, ctx=None):
        ctx = ctx or L.Context.default()
        self.csr = M if isinstance(M, _Csr) else _Csr(M, ctx)
        super().__init__(ctx, self.csr.shape[0])
        L.call("hfmi_op_csr_pcg", self.ctx.handle, self.csr.handle, float(rel_tol), int(max_iter), C.byref(self._op))

    def solve(self, y, x):
        self.mult(x, y)

    def info(self):
        """Of the last solve: {'iterations', 'method' ('chebyshev' | 'cg'), 'spectrum' (bracket of D^-1 M in use, or None)}."""
        it, method, lo, hi = C.c_int(0), C.c_int(0), C.c_double(0), C.c_double(0)
        L.call("hfmi_op_solver_info", self._op, C.byref(it), C.byref(method), C.byref(lo), C.byref(hi))
        return {"iterations": it.value, "method": "chebyshev" if method.value == 1 else "cg",
                "spectrum": (lo.value, hi.value) if hi.value > 0 else None}


class HostCallbackOperator(DeviceOperator):
    """A host black box plugged into the device solve (FEniCS/hIPPYlib PDE solves stay on the host).

    ``fn`` is, in this order of preference:

    * numpy fast paths -- a plain callable mapping an (N, k) array to an (N, k) array; an object with
      ``matMvMult_np(X) -> Y`` or (solvers) ``solve_block(X) -> Y`` on (N, k) arrays;
    * the REFERENCE'S OWN PROTOCOL -- an object whose vectors can be shaped (``init_vector(x, dim)`` on the object, on
      its ``operator()`` / ``get_operator()`` as hp.Solver2Operator looks it up, or passed as ``init_vector=``): the
      slab is moved column by column into host vectors made by ``hostvec.new_host_vector`` + that ``init_vector``
      through ``set_local`` and the result read back with ``get_local`` (how collectives/collective.py:98-107 moves
      data), calling ``fn.matMvMult(X, Y)`` with host column lists when the object has the block form (Y arrives
      zeroed: the reference's block operators accumulate, activeSubspaceProjector.py:214-221), else ``fn.mult(x, y)`` /
      ``fn.solve(y, x)`` per column -- so ``prior.R``, ``prior.Rsolver``, ``JTJ(ObservableJacobian(obs))``, PETSc
      matrices and Krylov solvers plug in as they are;
    * objects with ``solve(y, x)`` / ``mult(x, y)`` on 1-D numpy arrays and no ``init_vector`` (numpy test doubles).

    ``chunk_vectors``: for black boxes that treat the vectors independently (solvers), the callback is invoked on
    slabs of that many vectors and the PCIe copies of the neighbouring slabs overlap the host work
    (``hfmi_op_host_set_chunk``).  Default: solver-like objects (``solve`` / ``solve_block``) 32, everything else the
    whole block in one call (a serialized-sampling Jacobian operator re-solves its PDEs on every call)."""

    def __init__(self, fn, N=None, ctx=None, chunk_vectors=None, init_vector=None):
        self.fn = fn
        self.error = None
        solver_like = hasattr(fn, "solve") or hasattr(fn, "solve_block")
        shaper = init_vector
        if shaper is None and not (hasattr(fn, "matMvMult_np") or hasattr(fn, "solve_block")):
            shaper = H.find_init_vector(fn)
        plain_callable = callable(fn) and not hasattr(fn, "mult") and not hasattr(fn, "matMvMult") and not solver_like
        if plain_callable:
            self.mode = "array"
        elif hasattr(fn, "matMvMult_np"):
            self.mode = "block_np"
        elif hasattr(fn, "solve_block"):
            self.mode = "solve_np"
        elif shaper is not None:
            self.mode = "block" if hasattr(fn, "matMvMult") else ("solve" if hasattr(fn, "solve") else "mult")
        else:
            self.mode = "solve_1d" if hasattr(fn, "solve") else "mult_1d"
        self._shaper = shaper
        self._x = self._y = None           # host vectors, shaped on first use (domain, range)
        if N is None:
            if shaper is None:
                raise ValueError("HostCallbackOperator: the vector length is needed to wrap an operator without init_vector")
            N = H.shape_with(shaper, 0, self._comm()).size()
        super().__init__(ctx, N)

        def _cb(user, w_ptr, y_ptr, n, k):
            try:
                W = np.ctypeslib.as_array(w_ptr, shape=(k, n))    # one vector per row
                Y = np.ctypeslib.as_array(y_ptr, shape=(k, n))
                self._apply_rows(W, Y)
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

    def _comm(self):
        try:
            return self.fn.mpi_comm() if hasattr(self.fn, "mpi_comm") and callable(self.fn.mpi_comm) else None
        except Exception:          # noqa: BLE001
            return None

    def _host_pair(self):
        if self._x is None:
            comm = self._comm()
            # a solver maps the range of its operator back to the domain: both shapes coincide on this (square) path
            self._x = H.shape_with(self._shaper, 1, comm)
            self._y = H.shape_with(self._shaper, 0, comm)
        return self._x, self._y

    def _apply_rows(self, W, Y):
        fn, mode = self.fn, self.mode
        if mode == "array":
            Y[...] = np.asarray(fn(W.T)).T
        elif mode == "block_np":
            Y[...] = np.asarray(fn.matMvMult_np(W.T)).T
        elif mode == "solve_np":
            Y[...] = np.asarray(fn.solve_block(W.T)).T
        elif mode == "block":
            x, y = self._host_pair()
            X, Yh = H.HostMultiVector(x, W.shape[0]), H.HostMultiVector(y, W.shape[0])
            for j in range(W.shape[0]):
                X[j].set_local(W[j])
                X[j].apply("")
            fn.matMvMult(X, Yh)
            for j in range(W.shape[0]):
                Y[j] = Yh[j].get_local()
        elif mode in ("mult", "solve"):
            x, y = self._host_pair()
            for j in range(W.shape[0]):
                x.set_local(W[j])
                x.apply("")
                if mode == "mult":
                    fn.mult(x, y)
                else:
                    fn.solve(y, x)
                Y[j] = y.get_local()
        elif mode == "solve_1d":
            for j in range(W.shape[0]):
                fn.solve(Y[j], W[j])
        else:
            for j in range(W.shape[0]):
                fn.mult(W[j], Y[j])

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


def csr_from_matrix(M):
    """scipy CSR from the kinds of assembled matrices a prior carries: a scipy sparse matrix, a PETSc matrix
    (``getValuesCSR()``, the call the reference itself uses to export matrices, PODProjector.py:322-324), a dolfin
    matrix (``.mat()`` of its PETSc backend).  None when ``M`` is none of these."""
    import scipy.sparse as sp
    if sp.issparse(M):
        return sp.csr_matrix(M)
    for unwrap in (lambda m: m, lambda m: m.mat(), _dolfin_backend_mat):
        try:
            petsc = unwrap(M)
            indptr, indices, data = petsc.getValuesCSR()
        except Exception:          # noqa: BLE001 -- not that kind of matrix
            continue
        n_rows = len(indptr) - 1
        n_cols = petsc.getSize()[1] if hasattr(petsc, "getSize") else n_rows
        return sp.csr_matrix((np.asarray(data, dtype=np.float64), np.asarray(indices), np.asarray(indptr)), shape=(n_rows, n_cols))
    return None


def _dolfin_backend_mat(M):
    import dolfin
    return dolfin.as_backend_type(M).mat()


def as_device_operator(obj, N=None, ctx=None, init_vector=None):
    """Coerce the kinds of objects the reference passes as A / B / B^{-1} into device operators: device operators
    and adapters as they are, assembled matrices (scipy / PETSc / dolfin) as CSR in HBM, dense symmetric arrays, and
    every other operator or solver as a host callback -- through the reference's vector protocol when its vectors
    can be shaped (``init_vector`` on the object or passed in), on numpy slices otherwise."""
    if isinstance(obj, DeviceOperator):
        return obj
    if hasattr(obj, "_device_operator"):
        return obj._device_operator()
    if isinstance(obj, np.ndarray) and obj.ndim == 2:
        return npToDeviceOperator(obj, ctx=ctx)
    if hasattr(obj, "getValuesCSR") or hasattr(obj, "mat") or hasattr(obj, "tocsr"):
        csr = csr_from_matrix(obj)
        if csr is not None:
            return CsrOperator(csr, ctx=ctx)
    if callable(obj) or any(hasattr(obj, m) for m in ("mult", "solve", "matMvMult", "matMvMult_np", "solve_block")):
        return HostCallbackOperator(obj, N, ctx=ctx, init_vector=init_vector)
    raise TypeError("cannot use %r as an operator" % (type(obj),))


class Solver2Operator:
    """hp.Solver2Operator(S): ``mult(x, y)`` = ``S.solve(y, x)`` (KLEProjector.py:103)."""

    def __init__(self, solver, mpi_comm=None, init_vector=None):
        self.solver = solver
        self._mpi_comm = mpi_comm
        self._init_vector = init_vector or H.find_init_vector(solver)

    def mpi_comm(self):
        return self._mpi_comm

    def init_vector(self, x, dim):
        if self._init_vector is None:
            raise NotImplementedError("Solver2Operator: no init_vector available")
        self._init_vector(x, dim)

    def mult(self, x, y):
        self.solver.solve(y, x)

    def _device_operator(self):
        s = self.solver
        return s if isinstance(s, DeviceOperator) else as_device_operator(s, getattr(s, "N", None), init_vector=self._init_vector)


# ======================================================================================================================
# The reference's small adapter classes (same names, constructor arguments and duck types).  They are written around
# three helpers: a chain of maps through scratch vectors, a weighted sum of operator applications, and a retrying
# sample loop.  Every one of them works on device vectors / blocks and on host (dolfin-like) vectors alike: only
# ``zero``, ``axpy`` and the block members ``dot_v`` / ``reduce`` are used.
# ======================================================================================================================
def _scratch(like_init, dim, ctx=None):
    """A work vector shaped by an operator's ``init_vector``: in HBM when the owner lives on the device (it has a
    context), else a host vector of the kind the host operator expects."""
    if ctx is None:
        return H.shape_with(like_init, dim)
    v = Vector(ctx=ctx)
    like_init(v, dim)
    return v


def _run_chain(steps, x, y, scratch):
    """y = steps[-1](... steps[0](x)): each step is ``f(src, dst)``, intermediates live in ``scratch``."""
    src = x
    for f, dst in zip(steps[:-1], scratch):
        f(src, dst)
        src = dst
    steps[-1](src, y)


def _weighted_sum(applications, y, weight, make_like):
    """y = weight * sum of f(tmp) over ``applications`` (each fills ``tmp``); the accumulator starts from zero."""
    tmp, acc = make_like(y), make_like(y)
    acc.zero()
    for f in applications:
        f(tmp)
        acc.axpy(1.0, tmp)
    y.zero()
    y.axpy(weight, acc)


def _low_rank_apply(left, weights, right, x, y):
    """y = left diag(weights) right^T x."""
    coeff = right.dot_v(x)
    y.zero()
    left.reduce(y, coeff if weights is None else weights * coeff)


class MassPreconditionedCovarianceOperator:
    """M C M (KLEProjector.py:47-69).  On the device the three factors are composed into one operator; called with host
    vectors it runs the chain there."""

    def __init__(self, C, M):
        self.C, self.M = C, M
        self._dev = None
        self._host_scratch = None

    def mpi_comm(self):
        return self.M.mpi_comm()

    def init_vector(self, x, dim):
        self.M.init_vector(x, dim)

    def _device_operator(self):
        if self._dev is None:
            Md = as_device_operator(self.M)
            self._dev = ComposedOperator(Md, as_device_operator(self.C, Md.shape[0], Md.ctx), Md)
        return self._dev

    def mult(self, x, y):
        if H.is_host_vector(x):
            if self._host_scratch is None:
                self._host_scratch = [H.shape_with(self.M.init_vector, 0), H.shape_with(self.M.init_vector, 0)]
            _run_chain([self.M.mult, self.C.mult, self.M.mult], x, y, self._host_scratch)
        else:
            self._device_operator().mult(x, y)

    def matMvMult(self, X, Y):
        self._device_operator().matMvMult(X, Y)


class SummedListOperator:
    """Mean (``average``) or sum of operators of one shape (activeSubspaceProjector.py:69-95).  The accumulator starts
    from zero; the reference seeds it with the incoming y (:83-86), which is a mean only when y arrives zero-filled
    (SURVEY.md section 3.6)."""

    def __init__(self, operators, communicator=None, average=True):
        assert type(operators) is list
        self.operators = operators
        self.average = average

    def _weight(self):
        return 1.0 / float(len(self.operators)) if self.average else 1.0

    def init_vector(self, x, dim=0):
        self.operators[0].init_vector(x, dim)

    def mult(self, x, y):
        _weighted_sum([lambda t, op=op: op.mult(x, t) for op in self.operators], y, self._weight(), H._copy_vector if H.is_host_vector(y) else Vector)

    def matMvMult(self, X, Y):
        from .multivector import MatMvMult
        make = H.HostMultiVector if isinstance(Y, H.HostMultiVector) else MultiVector
        _weighted_sum([lambda T, op=op: MatMvMult(op, X, T) for op in self.operators], Y, self._weight(), make)


class StateSpaceIdentityOperator:
    """The observation operator of a full-state observable (fullStateObservable.py:18-52): the identity, whose adjoint
    in the M inner product is M itself unless ``use_mass_matrix`` is off."""

    def __init__(self, M, use_mass_matrix=True):
        self.M = M
        self.use_mass_matrix = use_mass_matrix

    def mpi_comm(self):
        return self.M.mpi_comm()

    def init_vector(self, v, dim):
        return self.M.init_vector(v, dim)

    @staticmethod
    def _copy(src, dst):
        dst.zero()
        dst.axpy(1.0, src)

    def mult(self, u, y):
        self._copy(u, y)

    def transpmult(self, x, p):
        if not self.use_mass_matrix:
            self._copy(x, p)
        else:
            getattr(self.M, "transpmult", self.M.mult)(x, p)          # M is symmetric


class Jacobian:
    """The interface a Jacobian offers (jacobian.py:19-60): ``init_vector(x, dim)`` with dim 0 = range and 1 = domain,
    ``mpi_comm()``, ``mult(x, y)``, ``transpmult(x, y)``.  A user-written Jacobian deriving from this plugs into ``JTJ`` / ``JJT``
    and the projectors' ``jacobian_factory`` like ``ObservableJacobian`` does."""

    def _missing(self, what):
        raise NotImplementedError("%s.%s: a Jacobian implements init_vector, mpi_comm, mult and transpmult"
                                  % (type(self).__name__, what))

    def init_vector(self, x, dim):
        self._missing("init_vector")

    def mpi_comm(self):
        self._missing("mpi_comm")

    def mult(self, x, y):
        self._missing("mult")

    def transpmult(self, x, y):
        self._missing("transpmult")


class ObservableJacobian(Jacobian):
    """Matrix-free Jacobian of the parameter-to-observable map at the observable's current linearisation point
    (jacobian.py:62-139), through the reference's own calls: ``mult`` is ``applyC -> solveFwdIncremental -> applyB`` and a
    sign, ``transpmult`` is ``applyBt -> solveAdjIncremental -> applyCt`` and a sign.  Everything stays on the host, in
    vectors the observable generated."""

    domain = H.PARAMETER                 # the variable the derivative is taken in
    domain_dim = 1                       # what observable.init_vector calls that space
    c_block = ("applyC", "applyCt")      # the observable's calls for d(residual)/d(domain variable) and its transpose

    def __init__(self, observable):
        self.observable = observable
        for name in self.c_block:
            assert hasattr(observable, name), "the observable must have attribute %s" % name
        self.ncalls = 0
        gen = observable.generate_vector
        self._rhs = {False: gen(H.STATE), True: gen(H.ADJOINT)}         # right-hand side of the incremental solve
        self._inc = {False: gen(H.STATE), True: gen(H.ADJOINT)}         # its solution
        q_like = H.new_host_vector(self.mpi_comm())
        observable.B.init_vector(q_like, 0)
        self.shape = (len(q_like.get_local()), len(gen(self.domain).get_local()))

    def mpi_comm(self):
        return self.observable.B.mpi_comm()

    def init_vector(self, x, dim):
        if dim not in (0, 1):
            raise ValueError("dim must be 0 (observable space) or 1 (the space of the derivative's variable)")
        self.observable.init_vector(x, 0 if dim == 0 else self.domain_dim)

    def _through(self, adjoint, into_rhs, solve, out_of_solution, x, y):
        into_rhs(x, self._rhs[adjoint])
        solve(self._inc[adjoint], self._rhs[adjoint])
        out_of_solution(self._inc[adjoint], y)
        y *= -1.0
        self.ncalls += 1

    def mult(self, x, y):
        obs = self.observable
        assert hasattr(obs, 'applyB'), 'LinearObservable must have attribute applyB'
        self._through(False, getattr(obs, self.c_block[0]), obs.solveFwdIncremental, obs.applyB, x, y)

    def transpmult(self, x, y):
        obs = self.observable
        assert hasattr(obs, 'applyBt'), 'LinearObservable must have attribute applyBt'
        self._through(True, obs.applyBt, obs.solveAdjIncremental, getattr(obs, self.c_block[1]), x, y)

    def rows(self, out=None):
        """The Jacobian as a dense (q, N) array, one adjoint solve per row (J^T e_i): how a sample's linearisation is
        materialised for the device when q is below the number of operator applications it would otherwise cost."""
        q, n = self.shape
        out = np.empty((q, n)) if out is None else out
        e, row = H.shape_with(self.init_vector, 0, self.mpi_comm()), H.shape_with(self.init_vector, 1, self.mpi_comm())
        unit = np.zeros(q)
        for i in range(q):
            unit[i] = 1.0
            e.set_local(unit)
            e.apply("")
            unit[i] = 0.0
            self.transpmult(e, row)
            out[i] = row.get_local()
        return out

    def dense(self):
        """The same array by whichever side is shorter: rows (adjoint solves) or, for a small domain such as a control
        variable, columns (one forward incremental solve each, J e_j)."""
        q, n = self.shape
        if q <= n:
            return self.rows()
        out = np.empty((q, n))
        e, col = H.shape_with(self.init_vector, 1, self.mpi_comm()), H.shape_with(self.init_vector, 0, self.mpi_comm())
        unit = np.zeros(n)
        for j in range(n):
            unit[j] = 1.0
            e.set_local(unit)
            e.apply("")
            unit[j] = 0.0
            self.mult(e, col)
            out[:, j] = col.get_local()
        return out


class ObservableControlJacobian(ObservableJacobian):
    """The same with respect to the CONTROL variable of a control problem (controlJacobian.py:21-95): ``applyCz`` /
    ``applyCzt`` in place of ``applyC`` / ``applyCt``, vectors of the control space (``observable.init_vector(x, 3)``)."""

    domain = H.CONTROL
    domain_dim = 3
    c_block = ("applyCz", "applyCzt")


class _NormalOperator:
    """J^T J or J J^T of a Jacobian-protocol object (``mult`` parameter -> observable, ``transpmult`` back,
    ``init_vector(x, dim)``); jacobian.py:142-193.  With a block-capable Jacobian (``DenseJacobianOperator``) the block
    form is two tall-skinny contractions."""

    inner_dim = None       # the space the intermediate vector lives in: 0 observable (JTJ), 1 parameter (JJT)

    def __init__(self, J):
        self.J = J
        self.vector_help = _scratch(J.init_vector, self.inner_dim, getattr(J, "ctx", None))

    def _steps(self, block=False):
        first, second = ("matMvMult", "matMvTranspmult") if block else ("mult", "transpmult")
        fs = (getattr(self.J, first), getattr(self.J, second))
        return fs if self.inner_dim == 0 else fs[::-1]

    def mult(self, x, y):
        _run_chain(list(self._steps()), x, y, [self.vector_help])

    def init_vector(self, x, dim=None):
        self.J.init_vector(x, 1 - self.inner_dim)

    def matMvMult(self, X, Y):
        """Y = op X (overwrites Y, the semantics of hp.MatMvMult on an operator without a block form)."""
        if hasattr(self.J, "matMvMult") and hasattr(self.J, "matMvTranspmult"):
            _run_chain(list(self._steps(block=True)), X, Y, [MultiVector(self.vector_help, X.nvec())])
        else:
            for j in range(X.nvec()):
                self.mult(X[j], Y[j])


class JTJ(_NormalOperator):
    """J^T J (jacobian.py:142-166)."""
    inner_dim = 0


class JJT(_NormalOperator):
    """J J^T (jacobian.py:169-193)."""
    inner_dim = 1


def default_jacobian_factory(observable):
    """What ``SeriallySampledJacobianOperator`` linearises with: the reference hard-wires ``ObservableJacobian``
    (activeSubspaceProjector.py:178-181); observables that hold their Jacobian in HBM offer ``jacobian()``."""
    if hasattr(observable, "jacobian"):
        return observable.jacobian()
    return ObservableJacobian(observable)


class SeriallySampledJacobianOperator:
    """Sample-by-sample accumulation of J^T J (or J J^T) over prior draws (activeSubspaceProjector.py:98-257).

    The PDE work stays behind the duck-typed ``observable`` exactly as in the reference: per sample ``noise`` is drawn,
    ``prior.sample(noise, m)``, the control (if any), ``solveFwd``, ``setLinearizationPoint``; then the linearised map
    ``jacobian_factory(observable)`` is applied to the whole probe block.  ``matMvMult`` ACCUMULATES into y (:214-221,
    :242-248): callers hand it a zeroed block, as ``MatrixMultCollectiveOperator`` + ``doublePass`` do.  A failing
    forward solve is answered with a fresh draw, as upstream (:180-211, whose ``while not solved`` has no exit) -- at most
    ``max_solver_retries`` times per sample here.

    Called with DEVICE blocks and a host observable, each sample's Jacobian is materialised row by row
    (``ObservableJacobian.rows``: q adjoint solves instead of the 2 k incremental solves of the column loop) into a pinned
    buffer, shipped to HBM on the ingest stream and contracted there while the host solves the next sample
    (``materialize`` = None: whenever q <= 2 k; True / False force it)."""

    def __init__(self, observable, noise, prior, control_distribution=None, operation='JTJ', nsamples=None, ms=None, zs=None,
                 communicator=None, average=True, jacobian_factory=None, materialize=None):
        assert operation in ['JTJ', 'JJT']
        assert (nsamples is not None) or (ms is not None)
        self.observable, self.noise, self.prior = observable, noise, prior
        self.control_distribution = control_distribution
        self.operation = operation
        self.nsamples = nsamples
        self.average = average
        self.ms = ms
        self.zs = zs if zs is not None else (len(ms) * [None] if type(ms) is list else None)
        self.jacobian_factory = jacobian_factory or default_jacobian_factory
        self.materialize = materialize
        self.max_solver_retries = 100         # fresh draws per sample before a failing forward solve is reported
        self.solver_failures = 0              # failed forward solves so far (each was followed by a fresh draw)
        self.samples_linearized = 0
        gen = getattr(observable, "generate_vector", None)
        projected = hasattr(getattr(observable, "problem", None), 'parameter_projection')
        self.u = gen(H.STATE) if gen else None
        if gen and projected:                 # the sample lives on the prior's space and is projected for the PDE (:131-137)
            self.m = H.shape_with(prior.init_vector, 0, communicator)
        else:
            self.m = gen(H.PARAMETER) if gen else None
        self.z = None if control_distribution is None else gen(H.CONTROL)

    def init_vector(self, x, dim=None):
        if self.operation == 'JJT':
            self.observable.init_vector(x, 0)
        elif hasattr(getattr(self.observable, "problem", None), 'parameter_projection'):
            self.prior.init_vector(x, 0)
        else:
            self.observable.init_vector(x, 1)

    # ---- one linearisation point
    def _linearize(self, m, z):
        point = [self.u, m, None] if z is None else [self.u, m, None, z]
        problem = getattr(self.observable, "problem", None)
        if hasattr(problem, 'parameter_projection') and m is self.m:
            point[1] = problem.parameter_projection(m)
        self.observable.solveFwd(self.u, point)
        self.observable.setLinearizationPoint(point)
        self.samples_linearized += 1

    def _draw_and_linearize(self):
        from .randomized import parRandom
        last = None
        for _ in range(self.max_solver_retries + 1):
            self.m.zero()
            self.noise.zero()
            parRandom.normal(1, self.noise)
            self.prior.sample(self.noise, self.m)
            if self.control_distribution is not None:
                self.z.zero()
                self.control_distribution.sample(self.z)
            try:
                return self._linearize(self.m, self.z)
            except Exception as exc:          # noqa: BLE001 -- any solver failure means "draw again", as upstream
                self.solver_failures += 1
                last = exc
        raise RuntimeError("forward solve failed for %d consecutive draws of one sample (last error: %r)"
                           % (self.max_solver_retries + 1, last)) from last

    def _points(self):
        """One linearisation per item: fresh draws, or the given ``ms`` (the reference's unit-test path, :223-248)."""
        if self.ms is None:
            for _ in range(self.nsamples):
                self._draw_and_linearize()
                yield
        else:
            for m, z in zip(self.ms, self.zs):
                self._linearize(m, z)
                yield

    # ---- the accumulation
    def matMvMult(self, x, y):
        assert x.nvec() == y.nvec(), "x and y have non-matching number of vectors"
        count = self.nsamples if self.ms is None else len(self.ms)
        weight = 1.0 / count if self.average else 1.0
        normal = JTJ if self.operation == 'JTJ' else JJT
        on_device = isinstance(x, MultiVector)
        J = None
        stager = None
        for _ in self._points():
            J = self.jacobian_factory(self.observable) if (J is None or not isinstance(J, ObservableJacobian)) else J
            if on_device and isinstance(J, ObservableJacobian):
                if stager is None:
                    stager = _JacobianStager(J, x.ctx, self.materialize, x.nvec())
                stager.accumulate(normal, x, y, weight)
                continue
            op = normal(J)
            if on_device:
                tmp = MultiVector(y)
                op.matMvMult(x, tmp)
                y.axpy(weight, tmp)
            else:                             # host column lists: the reference's loop (:214-221)
                tmp = H._copy_vector(y[0])
                for j in range(x.nvec()):
                    tmp.zero()
                    op.mult(x[j], tmp)
                    y[j].axpy(weight, tmp)
        if stager is not None:
            stager.finish()


class _JacobianStager:
    """Device accumulation of w J^T J X (or w J J^T X) for a HOST Jacobian.  Materialising route: the q rows of J go
    through one of two pinned buffers to a (q x N) block in HBM on the ingest stream, and the two contractions are enqueued
    on the solve's stream as a one-sample ``hfmi_op_jtj`` / ``hfmi_op_jjt`` accumulating into Y -- nothing waits for them:
    the host is already solving for the next sample's rows.  Column route (q > 2 k): the probe block is downloaded once
    and each column goes through ``mult`` / ``transpmult`` on host vectors."""

    def __init__(self, J, ctx, materialize, k):
        self.J, self.ctx = J, ctx
        q, n = J.shape
        self.q, self.n = q, n
        self.materialize = (q <= 2 * k) if materialize is None else bool(materialize)
        self._i = 0
        if self.materialize:
            self._pinned = [L.pinned_empty((q, n)) for _ in range(2)]
            self._blocks = [MultiVector(n, q, ctx=ctx) for _ in range(2)]
            self._tickets = [None, None]
            self._ops = [None, None]          # the operator reading block b: alive until its kernels have run
        else:
            self._x_host = None

    def accumulate(self, normal, X, Y, weight):
        if not self.materialize:
            return self._columns(normal, X, Y, weight)
        b = self._i % 2
        self._i += 1
        if self._tickets[b] is not None:
            self.ctx.ingest_wait(self._tickets[b])   # the pinned buffer has been read
            self.ctx.synchronize()                   # and the contraction that used this block two samples ago has run
        self.J.rows(self._pinned[b])                 # q adjoint solves on the host
        self._tickets[b] = self._blocks[b].upload_async(self._pinned[b])
        self.ctx.ingest_fence()
        if normal is JTJ:
            op = MeanJTJfromDataOperator.from_block(self._blocks[b], 1, self.q, scale=weight)
        else:
            op = MeanJJTfromDataOperator((self._blocks[b], 1, self.q), scale=weight)
        op.matMvMult(X, Y, accumulate=True)
        self._ops[b] = op

    def _columns(self, normal, X, Y, weight):
        op = normal(self.J)
        if self._x_host is None:
            self._x_host = X.to_vectors()
        dim_in = 1 if normal is JTJ else 0
        xin = H.shape_with(self.J.init_vector, dim_in, self.J.mpi_comm())
        yout = H.shape_with(self.J.init_vector, dim_in, self.J.mpi_comm())
        out = np.empty_like(self._x_host)
        for j in range(self._x_host.shape[0]):
            xin.set_local(self._x_host[j])
            xin.apply("")
            op.mult(xin, yout)
            out[j] = yout.get_local()
        Y.axpy(weight, MultiVector.from_vectors(out, ctx=self.ctx))

    def finish(self):
        if self.materialize:
            for t in self._tickets:
                if t is not None:
                    self.ctx.ingest_wait(t)
            self.ctx.synchronize()
            self._ops = [None, None]


class PriorPreconditionedProjector:
    """x -> U U^T C^{-1} x (priorPreconditionedProjector.py:19-55)."""

    def __init__(self, U, Cinv, my_init_vector):
        self.U, self.Cinv = U, Cinv
        self.my_init_vector = my_init_vector
        self.Cinvx = _scratch(my_init_vector, 0, getattr(U, "ctx", None))

    def init_vector(self, x, dim):
        self.my_init_vector(x, dim)

    def mult(self, x, y):
        self.Cinv.mult(x, self.Cinvx)
        _low_rank_apply(self.U, None, self.U, self.Cinvx, y)


class LowRankRectangularOperator:
    """U diag(s) V^T and its transpose (lowRankRectangularOperator.py:17-66)."""

    def __init__(self, U, s, V, U_init_vector=None, V_init_vector=None):
        self.U, self.s, self.V = U, np.asarray(s, dtype=np.float64), V
        self._shapers = (U_init_vector, V_init_vector)
        self.U_init_vector, self.V_init_vector = U_init_vector, V_init_vector

    def init_vector(self, x, dim):
        if dim not in (0, 1):
            raise ValueError("dim must be 0 or 1")
        assert self._shapers[dim] is not None
        self._shapers[dim](x)

    def mult(self, x, y):
        _low_rank_apply(self.U, self.s, self.V, x, y)

    def transpmult(self, x, y):
        _low_rank_apply(self.V, self.s, self.U, x, y)


# the reference's name for the dense-array operator wrapper (operatorWrappers.py:19-52)
npToDolfinOperator = npToDeviceOperator
