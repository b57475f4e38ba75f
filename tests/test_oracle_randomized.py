"""The hippylib restatement (parity UNPINNED against hippylib itself -- it is
absent) checked against (a) the invariants the reference's tests assert at that
boundary, with the reference's tolerances, and (b) independent dense solvers."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

from oracle import hippyflow_restated as hf_o
from oracle import hippylib_restated as hp_o


def _decaying_snapshots(n, N, rate, rng):
    U0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    W0, _ = np.linalg.qr(rng.standard_normal((N, n)))
    return (U0 * np.exp(-rate * np.arange(n))) @ W0.T


def _fem_matrices(N):
    h = 1.0 / (N - 1)
    main = np.full(N, 4 * h / 6)
    main[[0, -1]] = 2 * h / 6
    M = sp.diags([np.full(N - 1, h / 6), main, np.full(N - 1, h / 6)], [-1, 0, 1], format="csr")
    kd = np.full(N, 2 / h)
    kd[[0, -1]] = 1 / h
    K = sp.diags([np.full(N - 1, -1 / h), kd, np.full(N - 1, -1 / h)], [-1, 0, 1], format="csr")
    return M, K


def test_mgs_reortho_is_a_qr():
    rng = np.random.default_rng(0)
    Z = np.asfortranarray(rng.standard_normal((300, 20)) @ np.diag(np.logspace(0, -9, 20)))
    Q = Z.copy(order="F")
    R = hp_o.mgs_reortho(Q)
    assert np.allclose(np.triu(R), R) and np.all(np.diag(R) > 0)
    assert np.linalg.norm(Q.T @ Q - np.eye(20)) < 1e-13
    assert np.linalg.norm(Q @ R - Z) / np.linalg.norm(Z) < 1e-14
    # thin QR with positive diagonal is unique -> equals Householder QR
    Qh, Rh = hp_o._qr_posdiag(Z)
    assert np.linalg.norm(Q - Qh) < 1e-6


def test_mgs_reortho_rank_deficient_column_is_zeroed():
    rng = np.random.default_rng(1)
    Z = np.asfortranarray(rng.standard_normal((50, 4)))
    Z[:, 2] = Z[:, 0] - 2 * Z[:, 1]
    Q = Z.copy(order="F")
    R = hp_o.mgs_reortho(Q)
    assert R[2, 2] == 0.0 and np.all(Q[:, 2] == 0.0)
    assert abs(np.linalg.norm(Q[:, 3]) - 1.0) < 1e-14


def test_mgs_stable_b_orthonormal():
    rng = np.random.default_rng(2)
    M, _ = _fem_matrices(200)
    Z = np.asfortranarray(rng.standard_normal((200, 15)))
    Q = Z.copy(order="F")
    BQ, R = hp_o.mgs_stable(Q, hp_o.SparseOperator(M))
    assert np.linalg.norm(Q.T @ (M @ Q) - np.eye(15)) < 1e-13
    assert np.linalg.norm(BQ - M @ Q) / np.linalg.norm(BQ) < 1e-14
    assert np.linalg.norm(Q @ R - Z) / np.linalg.norm(Z) < 1e-13


def test_double_pass_vs_exact_eigenvalues():
    """SURVEY section 8c cross-check (1): the operator PODProjector hands to doublePass
    is X^T X / n whose exact eigenvalues the reference's 'hep' path computes."""
    rng = np.random.default_rng(0)
    n, N, r, p = 64, 512, 12, 10
    X = _decaying_snapshots(n, N, 0.35, rng)
    Omega = np.asfortranarray(np.random.default_rng(1).standard_normal((N, r + p)))
    d, U, parts = hp_o.double_pass(hf_o.SnapshotGramOperator(X), Omega, r, s=1, return_parts=True)
    exact = np.linalg.eigvalsh(X @ X.T / n)[::-1][:r]
    assert hp_o.eig_rel_err(d, exact) < 1e-5          # randomization error
    assert np.linalg.norm(U.T @ U - np.eye(r)) / np.sqrt(r) < 1e-10   # test_KLEProjector.py:183-196 tolerance
    AU = hf_o.snapshot_gram_block(X, U)
    assert np.linalg.norm(AU - U * d) / np.linalg.norm(AU) < 1e-4      # test_KLEProjector.py:198-217 tolerance
    # BLAS-3 twin agrees to round-off on eigenvalues and subspace
    d3, U3 = hp_o.double_pass_blas3(lambda W: hf_o.snapshot_gram_block(X, W), Omega, r)
    assert hp_o.eig_rel_err(d3, d) < 1e-9
    assert hp_o.principal_angle(U3[:, :6], U[:, :6]) < 1e-6


def test_column_loop_equals_block_path():
    """The reference's batched (per-column mult) and serialized (matMvMult)
    formulations give identical eigenvalues for identical Omega and samples:
    ||d_batch - d_serial||_2 < 1e-12 (test_derivativeSubspace.py:92-102)."""
    rng = np.random.default_rng(3)
    J = rng.standard_normal((6, 10, 120)) * np.exp(-0.2 * np.arange(10))[None, :, None]
    Omega = np.asfortranarray(rng.standard_normal((120, 9)))
    op = hf_o.MeanJTJOperator(J)

    class ColumnOnly:
        def mult(self, x, y):
            op.mult(x, y)

    d_block, _ = hp_o.double_pass(op, Omega, 6)
    d_col, _ = hp_o.double_pass(ColumnOnly(), Omega, 6)
    assert np.linalg.norm(d_block - d_col) < 1e-12


def test_double_pass_g_mass_kle_invariants():
    """KLE 'mass' mode, the invariants and tolerances of
    test_KLEProjector.py:91-129: V^T M V = I (1e-10), residual 1e-4; plus an
    independent dense generalized eigensolve."""
    rng = np.random.default_rng(4)
    N, r, p = 300, 20, 10
    M, K = _fem_matrices(N)
    A = (1.0 * M + 0.05 * K).toarray()
    Minv_lumped = np.diag(1.0 / np.asarray(M.sum(axis=1)).ravel())
    R = A @ Minv_lumped @ A                       # BiLaplacian-shaped precision
    C = np.linalg.inv(R)
    KLE = hf_o.MassPreconditionedCovarianceOperator(hp_o.DenseOperator(C), hp_o.SparseOperator(M))
    Omega = np.asfortranarray(rng.standard_normal((N, r + p)))
    d, V = hp_o.double_pass_g(KLE, hp_o.SparseOperator(M), hp_o.SparseLUSolver(M), Omega, r, s=1)
    assert np.linalg.norm(V.T @ (M @ V) - np.eye(r)) / np.sqrt(r) < 1e-10
    MCMV = M @ (C @ (M @ V))
    assert np.linalg.norm(MCMV - (M @ V) * d) / np.linalg.norm(MCMV) < 1e-4
    w = sla.eigh(M.toarray() @ C @ M.toarray(), M.toarray(), eigvals_only=True)[::-1][:r]
    assert hp_o.eig_rel_err(d[:10], w[:10]) < 1e-4   # randomization error (algebraic spectral decay)
    d3, V3 = hp_o.double_pass_blas3(lambda W: M @ (C @ (M @ W)), Omega, r, apply_B=lambda W: M @ W,
                                    apply_Binv=lambda W: np.asfortranarray(sla.solve(M.toarray(), W)))
    assert hp_o.eig_rel_err(d3, d) < 1e-9
    assert hp_o.principal_angle(V3[:, :8], V[:, :8], lambda W: M @ W) < 1e-6


@pytest.mark.parametrize("sort_by_abs", [False, True])
def test_sort_order_psd(sort_by_abs):
    rng = np.random.default_rng(5)
    X = _decaying_snapshots(20, 80, 0.3, rng)
    Omega = np.asfortranarray(rng.standard_normal((80, 12)))
    d, _ = hp_o.double_pass(hf_o.SnapshotGramOperator(X), Omega, 8, sort_by_abs=sort_by_abs)
    assert np.all(np.diff(d) <= 0)
