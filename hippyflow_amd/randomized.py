"""Randomized double-pass eigensolvers and the probe draw: the device counterparts of the hippylib
entry points hippyflow calls (hippylib is absent from /root/reference; call sites:
modeling/PODProjector.py:376, modeling/KLEProjector.py:163-164,177,
modeling/activeSubspaceProjector.py:449-463,556-577,654).

``doublePass(A, Omega, k, s=1)``             -> (d, U),   U^T U = I
``doublePassG(A, B, Binv, Omega, k, s=1)``   -> (d, U),   U^T B U = I,  A U ~ B U diag(d)
``parRandom.normal(sigma, out)``             -> N(0, sigma^2) fill of a Vector / MultiVector

Two execution routes, same arithmetic:
* fused: when A (and B, Binv) are device operators -- possibly wrapped in a CollectiveOperator --
  one call to ``hfmi_double_pass[_g]`` keeps every intermediate in HBM; with a ``NativeCollective`` the rank
  average is an RCCL all-reduce enqueued by the C solve itself (``hfmi_op_set_collective``), with any other
  collective object it is a post-apply hook that all-reduces the result block in place;
* generic: any object with the reference's ``mult`` / ``matMvMult`` protocol; the steps
  (MatMvMult, (B-)orthogonalize, dot_mv, small eigensolve, MvDSmatMult) are each a C-ABI call.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from . import hostvec as H
from .collectives import CollectiveOperator, MatrixMultCollectiveOperator, NativeCollective, NullCollective
from .multivector import MatMvMult, MultiVector, MvDSmatMult, Vector
from .operators import DeviceOperator, as_device_operator


class _ParRandom:
    """hp.parRandom stand-in: counter-based Philox4x32-10 + Box-Muller on the device (hfmi_randn_fill).

    hp.parRandom is seeded per process and only Omega, drawn on rank 0, travels (activeSubspaceProjector.py:433-443).
    Here the Philox key is (seed, namespace) and there are two families of streams:

    * SHARED draws (namespace 0, counter ``shared_stream``): every rank that makes the same sequence of calls gets the
      same numbers, so a probe block needs no broadcast.  Default for blocks (``MultiVector``).
    * PRIVATE draws (namespace rank + 1, counter ``stream``): distinct on every rank and disjoint from every shared
      draw -- the Monte-Carlo noise of the sampling loops.  Default for vectors, device or host.

    A host (dolfin-like) vector is filled through a device block of its length and ``set_local``."""

    def __init__(self, seed=1, rank=None):
        self.seed = int(seed)
        self.rank = L.launcher_rank() if rank is None else int(rank)
        self.stream = 0
        self.shared_stream = 0
        self._keyed = False

    def reseed(self, seed, stream=0):
        self.seed, self.stream, self.shared_stream = int(seed), int(stream), int(stream)

    def split(self, rank, by_collective=False):
        """Re-key the private draws for this rank of a sample-parallel run.  Called explicitly it always applies.  The FIRST
        collective a process constructs calls it with ``by_collective=True`` (the world / sample-parallel communicator, as in
        the reference, where hp.parRandom is seeded per MPI rank); collectives built later -- e.g. over a sub-group in which
        several processes hold the same group rank -- leave the key alone: re-keying there would collapse private streams
        onto one key and change the draws in the middle of a run."""
        if by_collective and self._keyed:
            return
        self.rank = int(rank)
        self._keyed = True          # an explicit split is final too: a collective constructed afterwards must not override it

    def key(self, shared):
        return (self.seed & 0xFFFFFFFF) | ((0 if shared else (self.rank + 1) & 0xFFFFFFFF) << 32)

    def normal(self, sigma, out, shared=None):
        host = H.is_host_vector(out)
        if shared is None:
            shared = isinstance(out, MultiVector)
        if shared:
            stream, self.shared_stream = self.shared_stream, self.shared_stream + 1
        else:
            stream, self.stream = self.stream, self.stream + 1
        if host:
            mv = MultiVector(int(len(out.get_local())), 1)
        else:
            mv = out._mv if isinstance(out, Vector) else out
        L.call("hfmi_randn_fill", mv.handle, C.c_uint64(self.key(shared)), C.c_uint32(stream & 0xFFFFFFFF), float(sigma))
        if host:
            out.set_local(mv.to_vectors()[0])
            out.apply("")

    def normal_perturb(self, sigma, out):
        if H.is_host_vector(out):
            tmp = H._copy_vector(out)
            self.normal(sigma, tmp)
            out.axpy(1.0, tmp)
            return
        tmp = MultiVector(out._mv if isinstance(out, Vector) else out)
        self.normal(sigma, Vector(ctx=tmp.ctx, _mv=tmp) if isinstance(out, Vector) else tmp)
        (out._mv if isinstance(out, Vector) else out).axpy(1.0, tmp)


parRandom = _ParRandom()


def sym_eig_small(T, sort_by_abs=False, ctx=None, method="dc", nvec=None):
    """np.linalg.eigh(T) + descending sort, on the device.  ``method="dc"``: Householder tridiagonalisation + divide
    and conquer (the algorithm family of the LAPACK routine behind np.linalg.eigh; up to 256 rows on one compute unit,
    up to 16384 over the whole GPU); ``"jacobi"``: one-workgroup cyclic Jacobi (high relative accuracy of small
    eigenvalues of graded positive definite matrices).  ``nvec``: return only the leading ``nvec`` eigenvectors
    (all eigenvalues still) -- ``la.eigh(G)[1][:, :u_rank]`` of PODProjector.py:821-826 without back-transforming
    and reading back the rest."""
    if method not in ("dc", "jacobi"):
        raise ValueError("sym_eig_small: method must be 'dc' or 'jacobi'")
    T = L.as_f64(T)
    k = T.shape[0]
    flags = (1 if sort_by_abs else 0) | (2 if method == "jacobi" else 0)
    handle = (ctx or L.Context.default()).handle
    if nvec is not None and int(nvec) < k:
        nvec = int(nvec)
        d, V = np.empty(k), np.empty((k, nvec))
        L.call("hfmi_sym_eig_leading", handle, L.ptr(T), k, flags, nvec, L.ptr(d), L.ptr(V))
        return d, V
    d, V = np.empty(k), np.empty((k, k))
    L.call("hfmi_sym_eig_small", handle, L.ptr(T), k, flags, L.ptr(d), L.ptr(V))
    return d, V


def _unwrap_collective(A):
    """(local device operator, collective, mpi_op) if A is a (possibly wrapped) device operator."""
    if isinstance(A, (CollectiveOperator, MatrixMultCollectiveOperator)):
        local = A.local_op
        dev = local if isinstance(local, DeviceOperator) else (local._device_operator() if hasattr(local, "_device_operator") else None)
        return dev, A.collective, A.mpi_op
    if isinstance(A, DeviceOperator):
        return A, None, None
    if hasattr(A, "_device_operator"):
        return A._device_operator(), None, None
    return None, None, None


class _PostApplyHook:
    """All-reduce of the operator's result block, called from inside the fused C solve."""

    def __init__(self, dev_op, collective, mpi_op):
        self.error = None

        def _cb(user, block_handle):
            try:
                mv = MultiVector(ctx=dev_op.ctx, _handle=C.c_void_p(block_handle), _borrowed=True)
                collective.allReduce(mv, mpi_op)
                return 0
            except Exception as exc:
                self.error = exc
                return 1

        self.cb = L.POST_APPLY_FN(_cb)
        self.dev_op = dev_op

    def __enter__(self):
        L.call("hfmi_op_set_post_apply", self.dev_op._op, self.cb, None)
        return self

    def __exit__(self, *exc):
        L.call("hfmi_op_set_post_apply", self.dev_op._op, L.POST_APPLY_FN(), None)
        return False


def _fused(A_dev, collective, mpi_op, B_dev, Binv_dev, Omega, k, s, sort_by_abs, use_mgs, literal_T=False):
    d = np.empty(k)
    U = MultiVector(Omega.size(), k, ctx=Omega.ctx)
    flags = (1 if sort_by_abs else 0) | (2 if use_mgs else 0) | (4 if literal_T else 0)

    def run():
        if B_dev is None:
            L.call("hfmi_double_pass", A_dev._op, Omega.handle, int(k), int(s), flags, L.ptr(d), U.handle)
        else:
            L.call("hfmi_double_pass_g", A_dev._op, B_dev._op, Binv_dev._op, Omega.handle, int(k), int(s), flags, L.ptr(d), U.handle)

    if isinstance(collective, NativeCollective):
        # the rank average runs inside the C solve on the context's stream (RCCL): no Python between the kernels
        code = {"sum": L.REDUCE_SUM, "avg": L.REDUCE_AVG}.get(str(mpi_op).lower())
        if code is None:
            raise NotImplementedError("reduction %r is not available (use 'sum' or 'avg')" % (mpi_op,))
        L.call("hfmi_op_set_collective", A_dev._op, collective._comm, code)
        try:
            run()
        finally:
            L.call("hfmi_op_set_collective", A_dev._op, None, 0)
    elif collective is not None and not isinstance(collective, NullCollective):
        hook = _PostApplyHook(A_dev, collective, mpi_op)
        with hook:
            try:
                run()
            except L.HfmiError:
                if hook.error is not None:
                    raise hook.error
                raise
    else:
        run()
    return d, U


def doublePass(A, Omega, k, s=1, check=False, sort_by_abs=False, use_mgs=False, fused=True, literal_T=False):
    """Randomized double pass for the dominant k eigenpairs of a Hermitian operator A.
    Omega: MultiVector with nvec >= k Gaussian probe vectors (not modified).
    ``literal_T=True`` forms T = (A Q)^T Q exactly as the reference does; by default the fused route forms the same
    matrix as scale (X Q)^T Gamma (X Q) for Gram-form operators (one fewer N x k product, k x k rank average)."""
    nvec = Omega.nvec()
    assert nvec >= k
    A_dev, coll, mpi_op = _unwrap_collective(A)
    if fused and A_dev is not None:
        return _fused(A_dev, coll, mpi_op, None, None, Omega, k, s, sort_by_abs, use_mgs, literal_T)
    Q = MultiVector(Omega)
    Y = MultiVector(Omega.size(), nvec, ctx=Omega.ctx)
    for i in range(s):
        if i:
            Y.zero()          # block operators of the reference accumulate into y (activeSubspaceProjector.py:214-221)
        MatMvMult(A, Q, Y)
        Q.swap(Y)
    Q.orthogonalize(L.QR_MGS if use_mgs else L.QR_AUTO)
    AQ = MultiVector(Omega.size(), nvec, ctx=Omega.ctx)
    MatMvMult(A, Q, AQ)
    T = AQ.dot_mv(Q)
    d, V = sym_eig_small(T, sort_by_abs, Omega.ctx)
    d = d[:k].copy()
    U = MultiVector(Omega.size(), k, ctx=Omega.ctx)
    MvDSmatMult(Q, np.ascontiguousarray(V[:, :k]), U)
    return d, U


def _as_solver_operator(Binv, N, ctx, B=None):
    """B^-1 as an operator.  An object that is both the operator and its own solver (``prior.Hlr``: ``mult`` applies B,
    ``solve`` applies B^-1; activeSubspaceProjector.py:455-459) contributes its ``inverse()``.  A host solver whose
    vectors cannot be shaped from the solver itself borrows ``B.init_vector`` (hippylib wraps it the same way)."""
    if isinstance(Binv, DeviceOperator) and hasattr(Binv, "inverse"):
        return Binv.inverse()
    shaper = None
    if not isinstance(Binv, DeviceOperator) and H.find_init_vector(Binv) is None and B is not None:
        if not isinstance(B, DeviceOperator):
            shaper = H.find_init_vector(B)
        else:
            shaper = getattr(B, "_shaper", None)          # a host operator already wrapped: the init_vector it was wrapped with
    return as_device_operator(Binv, N, ctx, init_vector=shaper)


def doublePassG(A, B, Binv, Omega, k, s=1, check=False, sort_by_abs=False, use_mgs=False, fused=True, literal_T=False):
    """Randomized double pass for A u = lambda B u (B SPD), U^T B U = I.
    ``Binv`` is a solver object (``solve(y, x)``) as in the reference, or an operator."""
    nvec = Omega.nvec()
    assert nvec >= k
    N = Omega.size()
    A_dev, coll, mpi_op = _unwrap_collective(A)
    if fused and A_dev is not None:
        B_dev = as_device_operator(B, N, Omega.ctx)
        Binv_dev = _as_solver_operator(Binv, N, Omega.ctx, B)
        return _fused(A_dev, coll, mpi_op, B_dev, Binv_dev, Omega, k, s, sort_by_abs, use_mgs, literal_T)
    # B^{-1}: a device operator / solver as is; a host solver object (solve(y, x) on numpy arrays, like the
    # PETSc solvers of the reference) is reached through a host-callback operator
    Binv_op = _as_solver_operator(Binv, N, Omega.ctx, B)
    Ybar = MultiVector(N, nvec, ctx=Omega.ctx)
    Q = MultiVector(Omega)
    for i in range(s):
        if i:
            Ybar.zero()
        MatMvMult(A, Q, Ybar)
        MatMvMult(Binv_op, Ybar, Q)
    Q.Borthogonalize(B, L.QR_MGS if use_mgs else L.QR_AUTO)
    AQ = MultiVector(N, nvec, ctx=Omega.ctx)
    MatMvMult(A, Q, AQ)
    T = AQ.dot_mv(Q)
    d, V = sym_eig_small(T, sort_by_abs, Omega.ctx)
    d = d[:k].copy()
    U = MultiVector(N, k, ctx=Omega.ctx)
    MvDSmatMult(Q, np.ascontiguousarray(V[:, :k]), U)
    return d, U


def svd_small(R, ctx=None):
    """np.linalg.svd(R) of a small square matrix on the device (one-sided Jacobi): R = U diag(s) V^T."""
    R = L.as_f64(R)
    k = R.shape[0]
    sv, U, V = np.empty(k), np.empty((k, k)), np.empty((k, k))
    L.call("hfmi_svd_small", (ctx or L.Context.default()).handle, L.ptr(R), k, L.ptr(sv), L.ptr(U), L.ptr(V))
    return U, sv, V


def accuracyEnhancedSVD(A, Omega, k, s=1, check=False):
    """hp.accuracyEnhancedSVD: randomized SVD A ~ U diag(d) V^T of a rectangular operator with ``mult`` /
    ``transpmult`` (or the block forms ``matMvMult`` / ``matMvTranspmult``) and ``init_vector(x, dim)``
    (call sites: activeSubspaceProjector.py:813-834,1026).  Omega lives in the domain.  Returns (U, d, V)."""
    from .multivector import MatMvTranspmult
    nvec = Omega.nvec()
    assert nvec >= k
    y_vec = Vector(ctx=Omega.ctx)
    A.init_vector(y_vec, 0)
    Z = MultiVector(Omega)
    Y = MultiVector(y_vec, nvec)
    MatMvMult(A, Omega, Y)
    for _ in range(s):
        MatMvTranspmult(A, Y, Z)
        MatMvMult(A, Z, Y)
    Q = MultiVector(Y)
    Q.orthogonalize()
    BT = MultiVector(Omega.size(), nvec, ctx=Omega.ctx)
    MatMvTranspmult(A, Q, BT)
    R = BT.orthogonalize()
    V_hat, d, U_hatT = svd_small(R, Omega.ctx)        # R = V_hat diag(d) U_hatT^T
    U = MultiVector(y_vec, k)
    MvDSmatMult(Q, np.ascontiguousarray(U_hatT[:, :k]), U)
    V = MultiVector(Omega.size(), k, ctx=Omega.ctx)
    MvDSmatMult(BT, np.ascontiguousarray(V_hat[:, :k]), V)
    return U, d[:k].copy(), V
