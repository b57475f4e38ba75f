"""The oracle's hippylib restatement (parity unpinned: hippylib is absent from /root/reference) against fixtures made
by INDEPENDENT dense solvers (tests/golden/make_independent_goldens.py: scipy.linalg.eigh / eigh(A, B), LAPACK).  With
s = 3 power iterations the randomisation error is below 1e-10, so these pin the restated doublePass / doublePassG /
MGS at 1e-9 from outside; the -m gpu twins of these tests (tests/test_gpu_configs_r2.py) hold the HIP path to the same
numbers."""
import os

import numpy as np
import scipy.sparse as sp

from oracle import hippylib_restated as hp_o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g():
    return np.load(os.path.join(ROOT, "tests", "golden", "independent_eig.npz"))


def _B(g):
    return sp.diags([g["ghep_B_off"], g["ghep_B_diag"], g["ghep_B_off"]], [-1, 0, 1], format="csr")


def test_double_pass_against_exact_dense_eigh():
    g = _g()
    r, s = int(g["hep_r"]), int(g["hep_s"])
    d, U = hp_o.double_pass(hp_o.DenseOperator(g["hep_A"]), np.asfortranarray(g["hep_Omega"]), r, s=s)
    np.testing.assert_allclose(d, g["hep_d_exact"], rtol=1e-9)
    assert hp_o.principal_angle(U, np.asfortranarray(g["hep_U_exact"])) < 1e-7
    d3, U3 = hp_o.double_pass_blas3(lambda W: np.asfortranarray(g["hep_A"] @ W), np.asfortranarray(g["hep_Omega"]), r, s=s)
    np.testing.assert_allclose(d3, g["hep_d_exact"], rtol=1e-9)


def test_double_pass_g_against_exact_dense_generalized_eigh():
    g = _g()
    r, s = int(g["hep_r"]), int(g["hep_s"])
    B = _B(g)
    d, U = hp_o.double_pass_g(hp_o.DenseOperator(g["hep_A"]), hp_o.SparseOperator(B), hp_o.SparseLUSolver(B),
                              np.asfortranarray(g["hep_Omega"]), r, s=s)
    np.testing.assert_allclose(d, g["ghep_d_exact"], rtol=1e-9)
    assert np.abs(U.T @ (B @ U) - np.eye(r)).max() < 1e-12
    assert hp_o.principal_angle(U, np.asfortranarray(g["ghep_U_exact"]), lambda W: B @ W) < 1e-7


def test_mgs_r_factor_against_householder_qr():
    """MultiVector.orthogonalize's R against LAPACK's Householder QR with the sign fixed (thin QR with a positive
    diagonal is unique)."""
    rng = np.random.default_rng(5)
    Z = rng.standard_normal((500, 24)) @ np.diag(np.exp(-0.3 * np.arange(24)))
    Q = hp_o.as_block(Z)
    R = hp_o.mgs_reortho(Q)
    Qh, Rh = np.linalg.qr(Z)
    sgn = np.sign(np.diag(Rh))
    np.testing.assert_allclose(R, Rh * sgn[:, None], rtol=1e-9, atol=1e-12 * np.abs(Rh).max())
    np.testing.assert_allclose(Q, Qh * sgn, atol=1e-9)


def test_matern_miniature_oracle_against_exact():
    """SURVEY 8d config 2 in miniature: the oracle's doublePassG on M C M (Matern-3/2) against the exact generalized
    eigenvalues.  One pass (s = 1, the reference's setting) only bounds the leading modes of this slowly decaying
    spectrum; the tolerances state that."""
    from hippyflow_amd import workloads
    g = _g()
    nx, ny, N = int(g["matern_nx"]), int(g["matern_ny"]), int(g["matern_N"])
    C = workloads.matern32_host(N, nx, ny, float(g["matern_sigma"]), float(g["matern_ell"]))
    np.testing.assert_allclose(C[:5, :5], g["matern_C_corner"], rtol=1e-14)
    assert abs(C.sum() - float(g["matern_C_checksum"])) < 1e-8 * abs(float(g["matern_C_checksum"]))
    M = workloads.grid_mass_matrix(nx, ny)[:N, :N].tocsr()
    import scipy.sparse.linalg as spla
    lu = spla.splu(M.tocsc())
    Omega = np.asfortranarray(np.random.default_rng(2).standard_normal((N, 30)))
    d, U = hp_o.double_pass_blas3(lambda W: np.asfortranarray(M @ (C @ (M @ W))), Omega, 20, s=1, apply_B=lambda W: M @ W,
                                  apply_Binv=lambda W: np.asfortranarray(lu.solve(np.ascontiguousarray(W))))
    exact = g["matern_d_exact"]
    assert np.all(d <= exact[:20] * (1 + 1e-12))                     # Ritz values never exceed the exact ones
    np.testing.assert_allclose(d[:5], exact[:5], rtol=0.1)          # observed 4.5e-2: lambda_j ~ j^-2.5 decays slowly
    np.testing.assert_allclose(d[:12], exact[:12], rtol=0.25)
