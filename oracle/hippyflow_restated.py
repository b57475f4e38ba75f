"""numpy fp64 restatement of the hot-path arithmetic that lives IN /root/reference.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Each function cites the
reference lines it follows.  These are pinned: ``tests/golden/make_goldens.py``
ran the reference's own code in the authoring container and stored its outputs
in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this module
against them.
"""
import numpy as np
import scipy.linalg as la
import scipy.sparse.linalg as spla


# ---------------------------------------------------------------------------
# operatorWrappers.py:95-114  MeanJTJfromDataOperator.mult (one vector), and the
# block form the device path implements (all Omega columns at once).
# ---------------------------------------------------------------------------
def mean_jtj_mult(J, x, noise_cov_inv=None):
    """y = mean_i J_i^T Gamma^{-1} J_i x ;  J has shape (ndata, r, dM)."""
    JX = np.einsum("ijk,k->ij", J, x)                 # :101-104 (tile + einsum)
    if noise_cov_inv is not None:
        JX = np.einsum("ij,kj->ki", noise_cov_inv, JX)  # :107-109
    JTJX = np.einsum("ijk,ij->ik", J, JX)             # :112
    return np.mean(JTJX, axis=0)                      # :114


def mean_jtj_block(J, W, noise_cov_inv=None):
    """Block form: Y = (1/ndata) sum_i J_i^T Gamma^{-1} (J_i W), W of shape (dM, k)."""
    ndata, r, dM = J.shape
    G = np.einsum("iod,dk->iok", J, W)                # (ndata, r, k)
    if noise_cov_inv is not None:
        G = np.einsum("op,ipk->iok", noise_cov_inv, G)
    Y = np.einsum("iod,iok->dk", J, G)
    return np.asfortranarray(Y / ndata)


def mean_jtj_block_blas3(J, W, noise_cov_inv=None):
    """Same as mean_jtj_block with the two contractions as threaded BLAS-3 matmuls over the stacked
    (ndata*r, dM) Jacobian -- the "best-effort CPU" form timed as bench.py's cpu_baseline."""
    ndata, r, dM = J.shape
    Js = J.reshape(ndata * r, dM)
    G = Js @ W                                        # (ndata*r, k)
    if noise_cov_inv is not None:
        G = (noise_cov_inv @ G.reshape(ndata, r, -1)).reshape(ndata * r, -1)
    return np.asfortranarray(Js.T @ G / ndata)


def mean_jjt_block(J, W):
    """Output-space counterpart (JJT, jacobian.py:169-193 averaged by
    SummedListOperator, activeSubspaceProjector.py:82-95):
    Y = (1/ndata) sum_i J_i (J_i^T W), W of shape (r, k)."""
    ndata = J.shape[0]
    G = np.einsum("iod,ok->idk", J, W)
    Y = np.einsum("iod,idk->ok", J, G)
    return np.asfortranarray(Y / ndata)


class MeanJTJOperator:
    """Protocol object around mean_jtj_*: ``mult`` (one column, as the
    reference) and ``matMvMult`` (block, accumulating like
    activeSubspaceProjector.py:214-221)."""

    def __init__(self, J, noise_cov_inv=None):
        self.J = J
        self.noise_cov_inv = noise_cov_inv

    def mult(self, x, y):
        y[...] = mean_jtj_mult(self.J, x, self.noise_cov_inv)

    def matMvMult(self, X, Y):
        Y += mean_jtj_block(self.J, X, self.noise_cov_inv)


# ---------------------------------------------------------------------------
# PODProjector.py:359-361: LowRankOperator(ones/n, snapshots) -> (1/n) X^T X
# with X the (n, N) snapshot matrix (rows = snapshots).
# ---------------------------------------------------------------------------
def snapshot_gram_block(X, W):
    """Y = (1/n) X^T (X W);  X (n, N), W (N, k)."""
    n = X.shape[0]
    return np.asfortranarray(X.T @ (X @ W) / n)


class SnapshotGramOperator:
    def __init__(self, X):
        self.X = X

    def mult(self, x, y):
        n = self.X.shape[0]
        g = self.X @ x                         # dot_v: n inner products
        y[...] = 0.0
        y += self.X.T @ (g / n)                # reduce: n axpys


# ---------------------------------------------------------------------------
# activeSubspaceProjector.py:82-95 SummedListOperator.mult (intended behaviour:
# mean/sum of the operators' actions; the accumulator quirk of :83-86 is not
# reproduced -- SURVEY.md section 3.6).
# ---------------------------------------------------------------------------
class SummedListOperator:
    def __init__(self, operators, average=True):
        self.operators = operators
        self.average = average

    def mult(self, x, y):
        acc = np.zeros_like(y)
        tmp = np.zeros_like(y)
        for op in self.operators:
            op.mult(x, tmp)
            acc += tmp
        y[...] = acc / len(self.operators) if self.average else acc


# ---------------------------------------------------------------------------
# collective.py:61-71 _allReduce_array and collectiveOperator.py:31-38,73-80.
# ``parts`` is the list of per-rank arrays.
# ---------------------------------------------------------------------------
def all_reduce(parts, op):
    op = op.lower()
    if op not in ("sum", "avg"):
        raise NotImplementedError(op)
    total = np.sum(parts, axis=0)
    return total / float(len(parts)) if op == "avg" else total


# ---------------------------------------------------------------------------
# KLEProjector.py:47-69 MassPreconditionedCovarianceOperator: y = M C M x
# ---------------------------------------------------------------------------
class MassPreconditionedCovarianceOperator:
    def __init__(self, C, M):
        self.C = C
        self.M = M

    def mult(self, x, y):
        Mx = np.zeros_like(x)
        CMx = np.zeros_like(x)
        self.M.mult(x, Mx)          # :67
        self.C.mult(Mx, CMx)        # :68
        self.M.mult(CMx, y)         # :69


# ---------------------------------------------------------------------------
# PODProjector.py:658-661 and :699-852  (deterministic mass-weighted POD)
# ---------------------------------------------------------------------------
def weighted_l2_norm_vector(x, W):
    Wx = W @ x
    return np.sqrt(np.einsum("ij,ij->j", Wx, x))


def pod_from_data(u_data, M_csr, u_rank, shifted=True, method="hep"):
    """Returns (d, phi, Mphi, u_shift) like
    PODProjectorFromData.construct_subspace."""
    n_data, dim_u = u_data.shape
    assert u_rank <= n_data
    if shifted:                                        # :732-735
        u_shift = np.mean(u_data, axis=0)
        u_data = u_data - u_shift
    else:                                              # :736-738
        u_shift = np.zeros(dim_u)
    X = u_data.T                                       # :740  (N, n)
    if method == "hep":                                # :812-833
        G = X.T @ (M_csr @ X)
        s, U = la.eigh(G)
        d = s[::-1][:u_rank] / n_data
        U = U[:, ::-1][:, :u_rank]
        phi = X @ U
        phi = phi / weighted_l2_norm_vector(phi, M_csr)
        Mphi = M_csr @ phi
    elif method == "ghep":                             # :743-773
        MX = M_csr @ X
        H = spla.LinearOperator(matvec=lambda v: MX @ (MX.T @ v) / n_data,
                                shape=(dim_u, dim_u), dtype=np.float64)
        d, phi = spla.eigsh(H, M=M_csr, k=u_rank)
        d = d[::-1][:u_rank]
        phi = phi[:, ::-1][:, :u_rank]
        Mphi = M_csr @ phi
    elif method == "inverse_ghep":                     # :775-810
        lu = spla.splu(M_csr.tocsc())
        Minv = spla.LinearOperator(shape=M_csr.shape, matvec=lu.solve, dtype=np.float64)
        Mop = spla.aslinearoperator(M_csr)
        H = spla.LinearOperator(matvec=lambda v: X @ (X.T @ v) / n_data,
                                shape=(dim_u, dim_u), dtype=np.float64)
        d, Mphi = spla.eigsh(H, k=u_rank, M=Minv, Minv=Mop)
        d = d[::-1]
        Mphi = Mphi[:, ::-1]
        phi = Minv @ Mphi
    else:
        raise ValueError("Unavailable method")
    return d, phi, Mphi, u_shift


# ---------------------------------------------------------------------------
# utilities/mv_utilities.py:31-54 layout contract: list of k columns <-> (N, k)
# C-ordered dense array.
# ---------------------------------------------------------------------------
def mv_to_dense(block):
    out = np.zeros(block.shape)
    for i in range(block.shape[1]):
        out[:, i] = block[:, i]
    return out
