"""numpy fp64 restatement of the hippylib arithmetic on the hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED: hippylib
is a third-party dependency that is absent from /root/reference (located at run
time through ``$HIPPYLIB_PATH``: hippyflow/test/test_KLEProjector.py:23-24; CI
clones branch ``matmvmult``: .travis.yml:15).  What follows restates the
published algorithms and is anchored on how hippyflow drives them:

* ``hp.doublePass(A, Omega, k, s=1)``      -- PODProjector.py:376,
  KLEProjector.py:177, activeSubspaceProjector.py:461,568,577,654
* ``hp.doublePassG(A, B, Binv, Omega, k, s=1)`` -- KLEProjector.py:163-164,428,
  activeSubspaceProjector.py:449-450,455-456,556-557,562-563
* ``hp.MatMvMult`` dispatch (``matMvMult`` attribute, else column loop)
  -- collectiveOperator.py:73-80 relies on it
* ``MultiVector.orthogonalize / Borthogonalize`` -- KLEProjector.py:124
* ``hp.LowRankOperator`` -- PODProjector.py:359-361
* ``hp.Solver2Operator`` -- KLEProjector.py:103,176

Conventions: a "multivector" is a Fortran-ordered float64 array of shape (N, k)
(k contiguous columns of length N, the layout hippyflow's ``mv_to_dense``
produces: utilities/mv_utilities.py:31-41).  Operators follow the reference's
duck-typed protocol: ``mult(x, y)`` overwrites the 1-D array ``y``; an optional
``matMvMult(X, Y)`` ACCUMULATES into the block ``Y`` (that is what
activeSubspaceProjector.py:219-221 does), so callers hand it a zeroed block.
"""
import numpy as np

EPS = np.finfo(np.float64).eps


def new_block(N, k):
    return np.zeros((N, k), dtype=np.float64, order="F")


def as_block(X):
    return np.array(X, dtype=np.float64, order="F", copy=True)


# ----------------------------------------------------------------------------
# MatMvMult / MvDSmatMult
# ----------------------------------------------------------------------------
def mat_mv_mult(A, X, Y):
    """hp.MatMvMult: block fast path if the operator has ``matMvMult``, else
    one ``mult`` per column.  Y is zeroed first because fresh MultiVectors are
    zero-filled in the reference."""
    assert X.shape[1] == Y.shape[1]
    if hasattr(A, "matMvMult"):
        Y[...] = 0.0
        A.matMvMult(X, Y)
    else:
        for j in range(X.shape[1]):
            A.mult(X[:, j], Y[:, j])


def mv_ds_mat_mult(X, A_small, Y):
    """hp.MvDSmatMult: Y[:, j] = sum_i X[:, i] * A_small[i, j]."""
    assert X.shape[1] == A_small.shape[0] and Y.shape[1] == A_small.shape[1]
    Y[...] = 0.0
    for j in range(A_small.shape[1]):
        for i in range(A_small.shape[0]):
            Y[:, j] += A_small[i, j] * X[:, i]


# ----------------------------------------------------------------------------
# MultiVector.orthogonalize: MGS with re-orthogonalisation (Rutishauser/Gander)
# ----------------------------------------------------------------------------
def mgs_reortho(Q):
    """In-place thin QR of the columns of Q by modified Gram-Schmidt; a sweep is
    repeated while the column norm dropped by more than 10x but not to
    round-off (10*eps).  Returns the upper-triangular R (positive diagonal, or
    0 for a numerically dependent column, which is then zeroed)."""
    n = Q.shape[1]
    r = np.zeros((n, n))
    for k in range(n):
        t = np.sqrt(Q[:, k] @ Q[:, k])
        again = True
        while again:
            for i in range(k):
                s = Q[:, i] @ Q[:, k]
                r[i, k] += s
                Q[:, k] -= s * Q[:, i]
            tt = np.sqrt(Q[:, k] @ Q[:, k])
            if tt > t * 10.0 * EPS and tt < t / 10.0:
                again = True
                t = tt
            else:
                again = False
                if tt < 10.0 * EPS * t:
                    tt = 0.0
        r[k, k] = tt
        inv = 1.0 / tt if abs(tt * EPS) > 0.0 else 0.0
        Q[:, k] *= inv
    return r


def mgs_stable(Q, B):
    """MultiVector.Borthogonalize(B): B-inner-product MGS with
    re-orthogonalisation ("PreCholQR", Saibaba-Lee-Kitanidis Algorithm 2).
    In place on Q; returns (BQ, R) with Q^T B Q = I, Q R = Z."""
    N, n = Q.shape
    Bq = new_block(N, n)
    r = np.zeros((n, n))
    for k in range(n):
        B.mult(Q[:, k], Bq[:, k])
        t = np.sqrt(Bq[:, k] @ Q[:, k])
        again = True
        while again:
            for i in range(k):
                s = Bq[:, i] @ Q[:, k]
                r[i, k] += s
                Q[:, k] -= s * Q[:, i]
            B.mult(Q[:, k], Bq[:, k])
            tt = np.sqrt(Bq[:, k] @ Q[:, k])
            if tt > t * 10.0 * EPS and tt < t / 10.0:
                again = True
                t = tt
            else:
                again = False
                if tt < 10.0 * EPS * t:
                    tt = 0.0
        r[k, k] = tt
        inv = 1.0 / tt if abs(tt * EPS) > 0.0 else 0.0
        Q[:, k] *= inv
        Bq[:, k] *= inv
    return Bq, r


# ----------------------------------------------------------------------------
# Small operators
# ----------------------------------------------------------------------------
class LowRankOperator:
    """hp.LowRankOperator(d, U): y = U diag(d) U^T x via ``dot_v`` + ``reduce``
    (PODProjector.py:359-361 builds it with d = ones/n, U = snapshots)."""

    def __init__(self, d, U):
        self.d = np.asarray(d, dtype=np.float64)
        self.U = U

    def mult(self, x, y):
        g = self.U.T @ x          # MultiVector.dot_v
        y[...] = 0.0
        y += self.U @ (self.d * g)  # MultiVector.reduce


class Solver2Operator:
    """hp.Solver2Operator(S): mult(x, y) = S.solve(y, x) (KLEProjector.py:103)."""

    def __init__(self, solver):
        self.solver = solver

    def mult(self, x, y):
        self.solver.solve(y, x)


class DenseOperator:
    """Dense symmetric matrix behind the protocol (oracle-side convenience;
    mirrors npToDolfinOperator.mult, operatorWrappers.py:42-46)."""

    def __init__(self, A):
        self.A = np.asarray(A, dtype=np.float64)

    def mult(self, x, y):
        y[...] = self.A @ x


class SparseOperator:
    """scipy.sparse matrix behind the protocol (prior.M / prior.R stand-in)."""

    def __init__(self, M):
        self.M = M

    def mult(self, x, y):
        y[...] = self.M @ x


class SparseLUSolver:
    """``solve(y, x)`` object (prior.Msolver / prior.Rsolver stand-in)."""

    def __init__(self, M):
        import scipy.sparse.linalg as spla
        self.lu = spla.splu(M.tocsc())

    def solve(self, y, x):
        y[...] = self.lu.solve(np.ascontiguousarray(x))


# ----------------------------------------------------------------------------
# Randomized eigensolvers
# ----------------------------------------------------------------------------
def _sort_truncate(d, V, k, sort_by_abs):
    perm = (np.abs(d) if sort_by_abs else d).argsort()[::-1]
    return d[perm[:k]], V[:, perm[:k]]


def double_pass(A, Omega, k, s=1, sort_by_abs=False, return_parts=False):
    """hp.doublePass: dominant k eigenpairs of the Hermitian operator A.

    Q <- Omega; s times Q <- A Q; Q.orthogonalize(); T = (A Q)^T Q;
    eigh(T); sort descending; truncate to k; U = Q V.  Returns (d, U)."""
    N, nvec = Omega.shape
    assert nvec >= k
    Q = as_block(Omega)
    Y = new_block(N, nvec)
    for _ in range(s):
        mat_mv_mult(A, Q, Y)
        Q, Y = Y, Q
    R = mgs_reortho(Q)
    AQ = new_block(N, nvec)
    mat_mv_mult(A, Q, AQ)
    T = AQ.T @ Q                      # MultiVector.dot_mv
    d, V = np.linalg.eigh(T)
    d, V = _sort_truncate(d, V, k, sort_by_abs)
    U = new_block(N, k)
    mv_ds_mat_mult(Q, V, U)
    if return_parts:
        return d, U, dict(Q=Q, R=R, T=T, V=V)
    return d, U


def double_pass_g(A, B, Binv, Omega, k, s=1, sort_by_abs=False, return_parts=False):
    """hp.doublePassG: dominant k eigenpairs of A u = lambda B u, U^T B U = I.

    Q <- Omega; s times {Ybar <- A Q; Q <- B^{-1} Ybar}; Q.Borthogonalize(B);
    T = (A Q)^T Q; eigh; sort; U = Q V."""
    N, nvec = Omega.shape
    assert nvec >= k
    Ybar = new_block(N, nvec)
    Q = as_block(Omega)
    Binv_op = Solver2Operator(Binv)
    for _ in range(s):
        mat_mv_mult(A, Q, Ybar)
        mat_mv_mult(Binv_op, Ybar, Q)
    BQ, R = mgs_stable(Q, B)
    AQ = new_block(N, nvec)
    mat_mv_mult(A, Q, AQ)
    T = AQ.T @ Q
    d, V = np.linalg.eigh(T)
    d, V = _sort_truncate(d, V, k, sort_by_abs)
    U = new_block(N, k)
    mv_ds_mat_mult(Q, V, U)
    if return_parts:
        return d, U, dict(Q=Q, R=R, T=T, V=V, BQ=BQ)
    return d, U


# ----------------------------------------------------------------------------
# BLAS-3 twin ("best-effort CPU" of BASELINE.md section 3): same algorithm, block
# operator application and Householder QR instead of column loops.  Because a
# thin QR with positive diagonal R is unique, Q (hence T, d, span U) agree with
# the MGS version to round-off; used for full-size checks in bench.py.
# ----------------------------------------------------------------------------
def _qr_posdiag(Z):
    Q, R = np.linalg.qr(Z)
    sgn = np.sign(np.diag(R))
    sgn[sgn == 0] = 1.0
    return np.asfortranarray(Q * sgn), R * sgn[:, None]


def _borth_blas3(Z, apply_B):
    """B-orthonormalise the columns of Z by two rounds of Cholesky QR preceded
    by a Householder QR (so the Gram matrix is well conditioned)."""
    import scipy.linalg as sla
    Q, _ = _qr_posdiag(Z)
    for _ in range(2):
        G = Q.T @ apply_B(Q)
        L = np.linalg.cholesky(0.5 * (G + G.T))
        Q = np.asfortranarray(sla.solve_triangular(L, Q.T, lower=True).T)
    return Q


def double_pass_blas3(apply_A, Omega, k, s=1, apply_B=None, apply_Binv=None):
    """Block (BLAS-3) twin of double_pass / double_pass_g.  ``apply_*`` map an
    (N, m) array to an (N, m) array."""
    Q = np.asfortranarray(Omega, dtype=np.float64)
    for _ in range(s):
        Q = apply_A(Q)
        if apply_Binv is not None:
            Q = apply_Binv(Q)
    if apply_B is None:
        Q, _ = _qr_posdiag(Q)
    else:
        Q = _borth_blas3(Q, apply_B)
    AQ = apply_A(Q)
    T = AQ.T @ Q
    d, V = np.linalg.eigh(0.5 * (T + T.T))
    d, V = _sort_truncate(d, V, k, False)
    return d, np.asfortranarray(Q @ V)


# ----------------------------------------------------------------------------
# Comparison helpers (eigenvectors are defined up to sign / rotation inside
# clusters: compare subspaces, never entries -- SURVEY.md section 7 "hard parts")
# ----------------------------------------------------------------------------
def eig_rel_err(d, d_ref):
    d = np.asarray(d)
    d_ref = np.asarray(d_ref)
    return float(np.max(np.abs(d - d_ref) / np.abs(d_ref)))


def principal_angle(U, U_ref, apply_B=None):
    """Largest principal angle (radians) between span(U) and span(U_ref) in the
    B inner product (both assumed B-orthonormal), from the sine:
    sin(theta_max) = || (I - U U^T B) U_ref ||_B ."""
    BUref = U_ref if apply_B is None else apply_B(U_ref)
    P = U_ref - U @ (U.T @ BUref)
    BP = P if apply_B is None else apply_B(P)
    G = P.T @ BP
    lam = np.linalg.eigvalsh(0.5 * (G + G.T)).max()
    return float(np.arcsin(min(1.0, np.sqrt(max(lam, 0.0)))))


# ----------------------------------------------------------------------------
# hp.accuracyEnhancedSVD (hippylib randomizedSVD.py; call sites activeSubspaceProjector.py:813-834,1026,
# dataGenerator.py:182-193).  PARITY UNPINNED like the rest of this module.  A is a rectangular operator
# with mult (domain -> range) and transpmult; Omega has nvec >= k columns in the DOMAIN.
# ----------------------------------------------------------------------------
def accuracy_enhanced_svd(A_mult, A_transpmult, Omega, k, s=1):
    """Y = A Omega; s times {Z = A^T Y; Y = A Z}; Q = orth(Y); B^T = A^T Q = Q_B R; svd(R) = Vh diag(d) Uh;
    U = Q Uh^T[:, :k], V = Q_B Vh[:, :k].  ``A_mult`` / ``A_transpmult`` map 2-D arrays (block form)."""
    nvec = Omega.shape[1]
    assert nvec >= k
    Y = np.asfortranarray(A_mult(Omega))
    for _ in range(s):
        Z = A_transpmult(Y)
        Y = np.asfortranarray(A_mult(Z))
    Q = as_block(Y)
    mgs_reortho(Q)
    BT = as_block(A_transpmult(Q))
    R = mgs_reortho(BT)
    V_hat, d, U_hat = np.linalg.svd(R, full_matrices=False)
    return np.asfortranarray(Q @ U_hat.T[:, :k]), d[:k], np.asfortranarray(BT @ V_hat[:, :k])
