"""GPU parity tests added in round 2: the HIP path against fixtures from independent dense solvers, SURVEY.md section 8d's
inputs as specified (config 4 with its noise term and the bi-Laplacian prior, config 2's Matern covariance), the
reference's default prior-preconditioned AS solve at full size, the deterministic POD beyond 256 snapshots and the
general-diagonal low-rank operator."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

hf = pytest.importorskip("hippyflow_amd")
from oracle import hippyflow_restated as hf_o   # noqa: E402
from oracle import hippylib_restated as hp_o    # noqa: E402
from oracle import philox as philox_o           # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    if hf.device_count() < 1:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return hf.Context.default()


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _g():
    return np.load(os.path.join(ROOT, "tests", "golden", "independent_eig.npz"))


# ------------------------------------------------------------------ independent dense solvers (VERDICT r1 task 7)
@pytest.mark.parametrize("route", ["fused", "generic", "mgs"])
def test_double_pass_against_exact_dense_eigh(ctx, route):
    g = _g()
    r, s = int(g["hep_r"]), int(g["hep_s"])
    A = hf.npToDeviceOperator(g["hep_A"])
    Omega = hf.MultiVector.from_dense(g["hep_Omega"])
    d, U = hf.doublePass(A, Omega, r, s=s, fused=(route != "generic"), use_mgs=(route == "mgs"))
    np.testing.assert_allclose(d, g["hep_d_exact"], rtol=1e-9)                      # exact eigenvalues (LAPACK)
    assert hp_o.principal_angle(np.asfortranarray(U.to_dense()), np.asfortranarray(g["hep_U_exact"])) < 1e-7
    d_o, _ = hp_o.double_pass(hp_o.DenseOperator(g["hep_A"]), np.asfortranarray(g["hep_Omega"]), r, s=s)
    np.testing.assert_allclose(d, d_o, rtol=1e-9)                                   # and the oracle on the same Omega


@pytest.mark.parametrize("route", ["fused", "generic"])
def test_double_pass_g_against_exact_dense_generalized_eigh(ctx, route):
    g = _g()
    r, s = int(g["hep_r"]), int(g["hep_s"])
    B = sp.diags([g["ghep_B_off"], g["ghep_B_diag"], g["ghep_B_off"]], [-1, 0, 1], format="csr")
    A = hf.npToDeviceOperator(g["hep_A"])
    Omega = hf.MultiVector.from_dense(g["hep_Omega"])
    d, U = hf.doublePassG(A, hf.CsrOperator(B), hf.CsrPCGSolver(B, rel_tol=1e-14), Omega, r, s=s, fused=(route == "fused"))
    np.testing.assert_allclose(d, g["ghep_d_exact"], rtol=1e-9)
    Ud = U.to_dense()
    assert np.abs(Ud.T @ (B @ Ud) - np.eye(r)).max() < 1e-10
    assert hp_o.principal_angle(np.asfortranarray(Ud), np.asfortranarray(g["ghep_U_exact"]), lambda W: B @ W) < 1e-7


def test_orthogonalize_r_factor_against_householder_qr(ctx):
    rng = np.random.default_rng(5)
    Z = rng.standard_normal((500, 24)) @ np.diag(np.exp(-0.3 * np.arange(24)))
    Qh, Rh = np.linalg.qr(Z)
    sgn = np.sign(np.diag(Rh))
    for method in (hf._lib.QR_CHOL, hf._lib.QR_MGS):
        Q = hf.MultiVector.from_dense(Z)
        R = Q.orthogonalize(method)
        np.testing.assert_allclose(R, Rh * sgn[:, None], rtol=1e-9, atol=1e-12 * np.abs(Rh).max())
        np.testing.assert_allclose(Q.to_dense(), Qh * sgn, atol=1e-9)


# ------------------------------------------------------------------ SURVEY 8d inputs as specified
def test_matern_miniature_kle(ctx):
    """Config 2 in miniature (4000 nodes of a 64 x 63 grid): the device fill of the Matern-3/2 covariance against the
    host formula, the mass-orthogonal KLE against the oracle on the same Omega (1e-9) and against the exact generalized
    eigenvalues of the fixture (randomisation error of one pass: stated tolerance)."""
    from hippyflow_amd import workloads
    import scipy.sparse.linalg as spla
    g = _g()
    nx, ny, N = int(g["matern_nx"]), int(g["matern_ny"]), int(g["matern_N"])
    wl = workloads.kle_matern_workload(nx, ny, N=N, sigma=float(g["matern_sigma"]), ell=float(g["matern_ell"]))
    C_host = workloads.matern32_host(N, nx, ny, float(g["matern_sigma"]), float(g["matern_ell"]))
    Cd = wl.C.to_vectors()
    assert np.abs(Cd - C_host).max() < 1e-14
    np.testing.assert_allclose(Cd[:5, :5], g["matern_C_corner"], rtol=1e-14)
    Omega_h = np.asfortranarray(np.random.default_rng(2).standard_normal((N, 30)))
    A = hf.MassPreconditionedCovarianceOperator(wl.C_operator, wl.M_operator)
    d, U = hf.doublePassG(A, wl.M_operator, hf.CsrPCGSolver(wl.M_operator.csr), hf.MultiVector.from_dense(Omega_h), 20, s=1)
    M = wl.M
    lu = spla.splu(M.tocsc())
    d_o, U_o = hp_o.double_pass_blas3(lambda W: np.asfortranarray(M @ (C_host @ (M @ W))), Omega_h, 20, s=1, apply_B=lambda W: M @ W,
                                      apply_Binv=lambda W: np.asfortranarray(lu.solve(np.ascontiguousarray(W))))
    np.testing.assert_allclose(d, d_o, rtol=1e-9)
    exact = g["matern_d_exact"]
    assert np.all(d <= exact[:20] * (1 + 1e-10))
    np.testing.assert_allclose(d[:5], exact[:5], rtol=0.1)
    Ud = U.to_dense()
    assert np.abs(Ud.T @ (M @ Ud) - np.eye(20)).max() < 1e-10


def test_config4_noise_term_reduced_size_dense_host_check(ctx):
    """J_i = A_i P^T + 0.01 E_i: the device-generated Jacobians against a host regeneration of the same Philox counters
    (oracle/philox.py, pinned by the Random123 vectors), and the double pass against the oracle on the dense host J."""
    from hippyflow_amd import workloads
    N, ns, q, r, p = 2001, 6, 5, 4, 3
    wl = workloads.as_workload(N, ns, q=q, latent=q, rate=0.3, seed=4, first_sample=10, ns_total=32, noise=0.01)
    P = wl.P.to_dense()
    J_host = np.empty((ns, q, N))
    for i in range(ns):
        E = philox_o.randn_block(N, q, 4, stream=workloads.NOISE_STREAM0 + 10 + i)          # (N, q)
        J_host[i] = wl.A[i] @ P.T + 0.01 * E.T
    Jd = wl.J.to_vectors().reshape(ns, q, N)
    assert np.abs(Jd - J_host).max() < 1e-12
    assert np.abs(Jd - np.einsum("ioc,tc->iot", wl.A, P)).max() > 1e-3                       # the noise is really there
    hf.parRandom.reseed(9)
    Omega = hf.MultiVector(N, r + p)
    hf.parRandom.normal(1.0, Omega)
    Oh = np.asfortranarray(Omega.to_dense())
    d, U = hf.doublePass(wl.operator, Omega, r, s=1)
    d_o, U_o = hp_o.double_pass(hf_o.MeanJTJOperator(J_host), Oh, r, s=1)
    np.testing.assert_allclose(d, d_o, rtol=1e-9)
    assert hp_o.principal_angle(np.asfortranarray(U.to_dense()), U_o) < 1e-6


def test_full_size_config4_prior_preconditioned_shard(ctx):
    """The reference's DEFAULT active-subspace solve (construct_input_subspace(prior_preconditioned=True):
    doublePassG(A, prior.R, prior.Rsolver, ...), activeSubspaceProjector.py:447-453) on the per-GPU share of config 4:
    64 samples of 100 x 2e5 with the noise term, R = A M_l^-1 A as CSR on the device, R^-1 a host sparse-LU callback fed
    through pinned slabs.  Oracle: the same algorithm on the host with the Jacobians copied back and applied densely."""
    from hippyflow_amd import workloads
    nx, ny, q, ns, r, k = 500, 400, 100, 64, 64, 74
    N = nx * ny
    wl = workloads.as_workload(N, ns, q=q, latent=q, rate=0.06, seed=4, first_sample=0, ns_total=512, noise=0.01)
    prior = workloads.BiLaplacianPrior(nx, ny, delta=1.0, gamma=0.1)
    B = hf.CsrOperator(prior.R)
    Binv = hf.HostCallbackOperator(prior.Rsolver, N)
    assert Binv.chunk_vectors == 32
    hf.parRandom.reseed(1)
    Omega = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, Omega)
    ctx.profile_begin()
    d, U = hf.doublePassG(wl.operator, B, Binv, Omega, r, s=1)
    ctx.profile_end()
    ph = ctx.profile_phases()
    assert prior.Rsolver.vectors == k and prior.Rsolver.calls == 3                  # 74 vectors in slabs of 32
    assert ph["host_function"] > 0 and ph["apply_Binv"] >= ph["host_function"] and ph["apply_A"] > 0 and ph["orthogonalize"] > 0
    Ud = U.to_dense()
    RU = prior.R @ Ud
    defect = np.abs(Ud.T @ RU - np.eye(r)).max()
    # encoder = R decoder on the device
    enc = hf.MultiVector(N, r)
    hf.MatMvMult(B, U, enc)
    # R u cancels heavily for the smooth eigenvectors (||R|| ||u|| / ||R u|| ~ 1e4): compare in the scale of |R| |u|
    assert np.linalg.norm(enc.to_dense() - RU) < 1e-14 * np.linalg.norm(abs(prior.R) @ np.abs(Ud))
    Jh = wl.J.to_vectors()                                                          # (6400, 2e5): 10.2 GB on the host
    Oh = np.asfortranarray(Omega.to_dense())
    d_o, U_o = hp_o.double_pass_blas3(lambda W: np.asfortranarray(Jh.T @ (Jh @ W) / ns), Oh, r, s=1, apply_B=lambda W: prior.R @ W,
                                      apply_Binv=lambda W: np.asfortranarray(prior.Rsolver.solve_block(W)))
    assert hp_o.eig_rel_err(d, d_o) < 1e-6, "north-star tolerance"
    assert hp_o.eig_rel_err(d, d_o) < 1e-8          # observed 9e-10; two HOST variants of the algorithm differ by 6e-11 at N = 5e4
    # R-orthonormality: cond(R) ~ 3e10 here, and eps * sqrt(cond) * k bounds what fp64 gives ANY implementation of the
    # algorithm -- the oracle's own U has the same defect (1.5e-10 ... 2.4e-10 already at N = 5e4, cond 1.7e9), so the
    # bar is "no worse than the oracle", not the 1e-10 the reference asserts for the well-conditioned B = M
    defect_o = np.abs(U_o.T @ (prior.R @ U_o) - np.eye(r)).max()
    assert defect < 1e-8 and defect < 4.0 * defect_o + 1e-10, (defect, defect_o)
    assert hp_o.principal_angle(np.asfortranarray(Ud[:, :32]), U_o[:, :32], lambda W: prior.R @ W) < 1e-6
    # the whole-block callback (no slabs) gives the same result
    Binv1 = hf.HostCallbackOperator(prior.Rsolver, N, chunk_vectors=0)
    d1, U1 = hf.doublePassG(wl.operator, B, Binv1, Omega, r, s=1)
    np.testing.assert_allclose(d1, d, rtol=1e-8)      # SuperLU rounds a 74-vector solve and three slab solves differently; cond(R) amplifies it (observed 1.7e-9)


def test_batched_equals_serialized_without_a_prior_solve_reference_tolerance(ctx):
    """The reference's own assertion, ||d_batched - d_serialized||_2 < 1e-12 (test_derivativeSubspace.py:92-102), with no
    prior solve in the loop (prior_preconditioned=False): the two routes then differ only by summation order."""
    rng = np.random.default_rng(9)
    ns, q, N = 16, 30, 1800
    P, _ = np.linalg.qr(rng.standard_normal((N, q)))
    J = np.einsum("ioc,tc->iot", rng.standard_normal((ns, q, q)) * np.exp(-0.15 * np.arange(q))[None, None, :], P) / np.sqrt(q)

    class Obs:
        def jacobian_data(self, n):
            return J[:n]

        def input_dimension(self):
            return N

        def output_dimension(self):
            return q

        def jtj_host_operator(self):
            return hf_o.MeanJTJOperator(J)

        def jjt_host_operator(self):
            class JJT:
                def matMvMult_np(self, W):
                    return hf_o.mean_jjt_block(J, W)
            return JJT()

    res = {}
    for serialized in (False, True):
        params = hf.ActiveSubspaceParameterList()
        params["rank"], params["oversampling"], params["samples_per_process"] = 20, 8, ns
        params["serialized_sampling"], params["verbose"], params["save_and_plot"] = serialized, False, False
        hf.parRandom.reseed(123)
        asp = hf.ActiveSubspaceProjector(Obs(), None, parameters=params)
        d, _, _ = asp.construct_input_subspace(prior_preconditioned=False)
        res[serialized] = d
    assert res[False][0] < 10.0 and res[False][0] > 0.1                               # O(1) eigenvalues: absolute == relative scale
    assert np.linalg.norm(res[False] - res[True]) < 1e-12


# ------------------------------------------------------------------ lifted limits (VERDICT r1 task 8)
@pytest.mark.parametrize("n", [257, 300, 640])
def test_sym_eig_beyond_one_workgroup(ctx, n):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 50)) * np.exp(-0.01 * np.arange(n + 50))[None, :]
    G = X @ X.T                                                   # Gram matrix: positive definite
    d, V = hf.sym_eig_small(G)
    w = np.linalg.eigvalsh(G)[::-1]
    np.testing.assert_allclose(d, w, rtol=1e-10, atol=1e-12 * w[0])
    assert np.abs(V.T @ V - np.eye(n)).max() < 2e-12
    assert np.abs(G @ V - V * d).max() < 1e-11 * w[0]
    S = rng.standard_normal((n, n))
    S = 0.5 * (S + S.T)                                                              # indefinite
    d2, V2 = hf.sym_eig_small(S)
    w2 = np.linalg.eigvalsh(S)[::-1]
    np.testing.assert_allclose(d2, w2, atol=1e-11 * np.abs(w2).max())
    assert np.abs(S @ V2 - V2 * d2).max() < 1e-11 * np.abs(w2).max()


def test_pod_from_data_with_more_than_256_snapshots(ctx):
    """PODProjectorFromData.construct_subspace(method='hep') with 400 snapshots (PODProjector.py:812-833: any n) against
    the oracle's restatement, which tests/test_oracle_golden.py pins to the reference's own outputs."""
    from hippyflow_amd import workloads
    rng = np.random.default_rng(3)
    n, nx, ny, r = 400, 40, 30, 25
    N = nx * ny
    M = workloads.grid_mass_matrix(nx, ny)
    W0, _ = np.linalg.qr(rng.standard_normal((N, 60)))
    u_data = (rng.standard_normal((n, 60)) * np.exp(-0.15 * np.arange(60))) @ W0.T + 0.5
    pod = hf.PODProjectorFromData(M_output=M)
    for shifted in (True, False):
        d, phi, Mphi, shift = pod.construct_subspace(u_data.copy(), r, shifted=shifted, method="hep")
        d_o, phi_o, Mphi_o, shift_o = hf_o.pod_from_data(u_data.copy(), M, r, shifted=shifted, method="hep")
        np.testing.assert_allclose(d, d_o, rtol=1e-9)
        np.testing.assert_allclose(shift, shift_o, atol=1e-14)
        sg = np.sign(np.sum(phi * (M @ phi_o), axis=0))
        np.testing.assert_allclose(phi * sg, phi_o, atol=1e-7)
        np.testing.assert_allclose(Mphi * sg, Mphi_o, atol=1e-7)
        assert np.abs(phi.T @ Mphi - np.eye(r)).max() < 1e-9
    # the two Lanczos formulations of the reference (ghep :743-773, inverse_ghep :775-810) reduce to the same Gram problem
    d_h, phi_h, _, _ = pod.construct_subspace(u_data.copy(), r, shifted=True, method="hep")
    for method in ("ghep", "inverse_ghep"):
        d_m, phi_m, Mphi_m, _ = pod.construct_subspace(u_data.copy(), r, shifted=True, method=method)
        d_o, phi_o, _, _ = hf_o.pod_from_data(u_data.copy(), M, r, shifted=True, method=method)     # ARPACK on the host
        np.testing.assert_allclose(d_m, d_o, rtol=1e-7)
        np.testing.assert_array_equal(d_m, d_h)
        sg = np.sign(np.sum(phi_m * (M @ phi_o), axis=0))
        np.testing.assert_allclose(phi_m * sg, phi_o, atol=1e-6)


def test_low_rank_operator_with_a_general_diagonal(ctx):
    """hp.LowRankOperator(d, U): mult = U diag(d) U^T, solve = U diag(1/d) U^T; as B and B^-1 of doublePassG (prior.Hlr,
    activeSubspaceProjector.py:455-459) and inside PriorPreconditionedProjector (priorPreconditionedProjector.py:48-55)."""
    rng = np.random.default_rng(12)
    N, m = 1203, 1203
    Uq, _ = np.linalg.qr(rng.standard_normal((N, m)))
    dvec = np.exp(rng.uniform(-1.0, 1.0, m))
    Hlr = hf.LowRankOperator(dvec, Uq.T.copy())                   # full rank here so that B is SPD
    x = rng.standard_normal(N)
    xv, yv = hf.Vector(), hf.Vector()
    Hlr.init_vector(xv, 1)
    Hlr.init_vector(yv, 0)
    xv.set_local(x)
    Hlr.mult(xv, yv)
    np.testing.assert_allclose(yv.get_local(), Uq @ (dvec * (Uq.T @ x)), rtol=1e-12, atol=1e-13)
    Hlr.solve(yv, xv)
    np.testing.assert_allclose(yv.get_local(), Uq @ ((Uq.T @ x) / dvec), rtol=1e-12, atol=1e-13)
    assert hf.LowRankOperator(np.ones(3) / 3, rng.standard_normal((3, 50))).d.shape == (3,)
    # GHEP with B = B^-1-capable low-rank operator
    J = rng.standard_normal((5, 7, N)) * 0.3
    A = hf.MeanJTJfromDataOperator(J)
    Omega = np.asfortranarray(rng.standard_normal((N, 9)))
    d, U = hf.doublePassG(A, Hlr, Hlr, hf.MultiVector.from_dense(Omega), 6, s=1)
    Bd = (Uq * dvec) @ Uq.T

    class Solve:
        def solve(self, y, x):
            y[...] = Uq @ ((Uq.T @ x) / dvec)
    d_o, U_o = hp_o.double_pass_g(hf_o.MeanJTJOperator(J), hp_o.DenseOperator(Bd), Solve(), Omega, 6, s=1)
    np.testing.assert_allclose(d, d_o, rtol=1e-9)
    Ud = U.to_dense()
    assert np.abs(Ud.T @ Bd @ Ud - np.eye(6)).max() < 1e-10


def test_pod_from_data_8300_snapshots_is_the_exact_gram_route(ctx):
    """The reference's deterministic POD takes any number of snapshots (PODProjector.py:812-833); up to 16384 the device solves
    the n x n Gram problem exactly (whole-GPU eigensolver) -- no warning, no randomization: rank-30 snapshots against the oracle."""
    import warnings
    from hippyflow_amd import workloads
    rng = np.random.default_rng(8)
    n, nx, ny, r = 8300, 30, 20, 18
    N = nx * ny
    M = workloads.grid_mass_matrix(nx, ny)
    W0, _ = np.linalg.qr(rng.standard_normal((N, 30)))
    u_data = (rng.standard_normal((n, 30)) * np.exp(-0.3 * np.arange(30))) @ W0.T + 0.25
    pod = hf.PODProjectorFromData(M_output=M)
    pod.prefer_state_dimension = False                   # the n x n form (the default here would be the 600 x 600 one)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        d, phi, Mphi, shift = pod.construct_subspace(u_data.copy(), r, shifted=True, method="hep")
    d_o, phi_o, Mphi_o, shift_o = hf_o.pod_from_data(u_data.copy(), M, r, shifted=True, method="hep")
    np.testing.assert_allclose(d, d_o, rtol=1e-8)
    np.testing.assert_allclose(shift, shift_o, atol=1e-14)
    sg = np.sign(np.sum(phi * (M @ phi_o), axis=0))
    np.testing.assert_allclose(phi * sg, phi_o, atol=1e-6)
    np.testing.assert_allclose(Mphi * sg, Mphi_o, atol=1e-6)
    assert np.abs(phi.T @ Mphi - np.eye(r)).max() < 1e-9


def _slow_decay_snapshots(n, nx, ny, K, seed):
    from hippyflow_amd import workloads
    rng = np.random.default_rng(seed)
    N = nx * ny
    M = workloads.grid_mass_matrix(nx, ny)
    W0, _ = np.linalg.qr(rng.standard_normal((N, K)))
    return (rng.standard_normal((n, K)) * 0.95 ** np.arange(K)) @ W0.T, M


def test_pod_from_data_beyond_16384_snapshots_is_exact_in_the_state_dimension(ctx):
    """More snapshots than the n x n eigensolver takes, a state dimension it does take (the POD of an output over a large training set,
    dataGenerator.py:278-279): the N-dimensional route -- M = B B^T from the device eigendecomposition of the mass matrix, S = B^T (X^T X) B
    by two device products, one more device eigensolve -- gives the reference's eigenpairs EXACTLY (no warning, nothing randomized):
    16500 snapshots of a slowly decaying spectrum against the dense generalized eigenproblem of M H M / n and M."""
    import warnings
    import scipy.linalg as sla
    n, r = 16500, 12
    u_data, M = _slow_decay_snapshots(n, 30, 20, 200, 17)
    pod = hf.PODProjectorFromData(M_output=M)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        d, phi, Mphi, shift = pod.construct_subspace(u_data.copy(), r, shifted=True, method="hep")
    Md = M.toarray()
    Xc = u_data - u_data.mean(axis=0)
    lam, vec = sla.eigh(Md @ (Xc.T @ Xc / n) @ Md, Md)
    lam, vec = lam[::-1], vec[:, ::-1]
    np.testing.assert_allclose(d, lam[:r], rtol=1e-10)
    cos = np.abs(np.einsum("ij,ij->j", phi, Md @ vec[:, :r]))
    np.testing.assert_allclose(cos[:r - 2], 1.0, atol=1e-8)
    assert np.abs(phi.T @ Mphi - np.eye(r)).max() < 1e-10 and np.abs(Md @ phi - Mphi).max() < 1e-10 * np.abs(Mphi).max()
    np.testing.assert_allclose(shift, u_data.mean(axis=0), atol=1e-13)


def test_pod_from_data_randomized_fallback_tolerance(ctx):
    """Both the number of snapshots and the state dimension beyond the exact solver: the modes come from the randomized double pass on the
    N-dimensional generalized problem, k = r + 40 probe columns, 3 passes, WITH A STATED TOLERANCE: relative eigenvalue error of mode i
    <= (lambda_{k+1} / lambda_i)^5 (PODProjectorFromData._randomized).  Checked on a SLOWLY decaying spectrum (0.95^i per singular value)
    with the exact routes switched off through the class limit (a genuine case needs a 16385 x 16385 dense reference solve)."""
    import scipy.linalg as sla
    n, r = 3000, 12
    u_data, M = _slow_decay_snapshots(n, 30, 20, 200, 18)
    pod = hf.PODProjectorFromData(M_output=M)
    pod.EXACT_MAX_SNAPSHOTS = 256                        # instance attribute: this projector only
    with pytest.warns(UserWarning, match="randomized double pass"):
        d, phi, Mphi, shift = pod.construct_subspace(u_data.copy(), r, shifted=False, method="hep")
    Md = M.toarray()
    lam, vec = sla.eigh(Md @ (u_data.T @ u_data / n) @ Md, Md)
    lam, vec = lam[::-1], vec[:, ::-1]
    k = r + pod.RANDOMIZED_OVERSAMPLING
    bound = (lam[k] / lam[:r]) ** (2 * pod.RANDOMIZED_PASSES - 1)
    rel = np.abs(d - lam[:r]) / lam[:r]
    assert np.all(rel <= np.maximum(bound, 1e-10)), (rel, bound)
    assert rel.max() < 1e-6                                                   # lambda_{k+1} / lambda_r = 0.95^80 = 0.0165: bound 1.2e-9
    cos = np.abs(np.einsum("ij,ij->j", phi, Md @ vec[:, :r]))
    np.testing.assert_allclose(cos[:r - 2], 1.0, atol=1e-6)
    assert np.abs(phi.T @ Mphi - np.eye(r)).max() < 1e-9


# ------------------------------------------------------------------ error behaviour of the round-2 entry points
def test_round2_entry_points_reject_bad_arguments(ctx):
    import ctypes as C
    L = hf._lib
    # sym_eig beyond 16384 (the range check comes before anything is read: a small buffer stands in for the 2 GB matrix)
    with pytest.raises(hf.HfmiError) as e:
        small = np.eye(4)
        L.call("hfmi_sym_eig_small", hf.Context.default().handle, L.ptr(small), 16385, 0, L.ptr(np.empty(4)), None)
    assert "out of range" in str(e.value)
    # host-callback slab size: negative, and on an operator that is not a host callback
    cb = hf.HostCallbackOperator(lambda W: W, 64)
    with pytest.raises(hf.HfmiError):
        L.call("hfmi_op_host_set_chunk", cb._op, -1)
    csr = hf.CsrOperator(sp.identity(64, format="csr"))
    with pytest.raises(hf.HfmiError):
        L.call("hfmi_op_host_set_chunk", csr._op, 4)
    # a failing host callback surfaces as the original Python exception, also through the slab pipeline
    class Boom:
        def solve_block(self, X):
            raise ValueError("boom in slab")
    bad = hf.HostCallbackOperator(Boom(), 64, chunk_vectors=2)
    X, Y = hf.MultiVector(64, 5), hf.MultiVector(64, 5)
    with pytest.raises(ValueError, match="boom in slab"):
        bad.matMvMult(X, Y)
    # the slab pipeline gives the same block as one call (identity black box scaled by 3)
    tri = hf.HostCallbackOperator(type("S", (), {"solve_block": staticmethod(lambda X: 3.0 * X)})(), 64, chunk_vectors=2)
    hf.parRandom.normal(1.0, X)
    tri.matMvMult(X, Y)
    np.testing.assert_array_equal(Y.to_dense(), 3.0 * X.to_dense())
    tri.matMvMult(X, Y, accumulate=True)
    np.testing.assert_allclose(Y.to_dense(), 6.0 * X.to_dense(), rtol=1e-15)
    # communicator: reductions other than sum / avg cannot be attached to an operator; blocks of another context are refused
    coll = hf.NativeCollective.from_unique_id(hf.NativeCollective.unique_id(), 1, 0)
    with pytest.raises(hf.HfmiError):
        L.call("hfmi_op_set_collective", csr._op, coll._comm, 2)
    with pytest.raises(hf.HfmiError):
        L.call("hfmi_bcast", coll._comm, X.handle, 3)                 # root outside the communicator
    with pytest.raises(hf.HfmiError):
        L.call("hfmi_allreduce", coll._comm, X.handle, 7)             # unknown reduction
    coll.close()
    # low-rank operator needs its diagonal
    with pytest.raises(hf.HfmiError):
        L.call("hfmi_op_low_rank", ctx.handle, X.handle, None, C.byref(C.c_void_p()))
    with pytest.raises(ZeroDivisionError):
        hf.LowRankOperator(np.array([1.0, 0.0, 2.0, 1.0, 1.0]), X).inverse()
    # Matern fill: grid too small for the block
    Cb = hf.MultiVector(50, 50)
    with pytest.raises(hf.HfmiError):
        L.call("hfmi_block_fill_matern32", Cb.handle, 5, 5, 1.0, 0.1)
