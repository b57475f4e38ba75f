"""Seeded random-shape sweeps of the contraction kernels, the orthogonalisation and the fused double pass against numpy / the CPU
oracle (promoted from scripts/fuzz_gemm.py, fuzz_qr.py, fuzz_solve.py: VERDICT r4 item 7).  Shapes include vector lengths that
are not multiples of 32, k in {1, 3, 13, 75, 139, 255}, fewer than 16 vectors, rank-deficient inputs and power iterations."""
import numpy as np
import pytest

import hippyflow_amd as hf
from oracle import hippylib_restated as hp_o

pytestmark = pytest.mark.gpu

EDGE_K = [1, 3, 13, 75, 139, 255]


@pytest.fixture(scope="module")
def ctx():
    if hf.device_count() < 1:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return hf.Context.default()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_contractions_against_numpy(ctx, seed):
    """dot_mv (tn / skinny kernels) and MvDSmatMult (nn / LDS-resident nn) on 30 random shapes per seed."""
    rng = np.random.default_rng(1000 + seed)
    for it in range(30):
        N = int(rng.choice([1, 31, 32, 33, 100, 1000, 4225, 9999, 20000, 50001]))
        m = int(rng.choice([1, 2, 7, 15, 16, 17, 30, 64, 74, 84, 100, 129, 138, 160, 200, 255, 256, 300, 700, 2000]))
        k = int(rng.choice(EDGE_K)) if rng.random() < 0.35 else int(rng.integers(1, 270))
        if N * (m + k) > 4e7:
            N = max(1, int(4e7 // (m + k)))
        A, B = rng.standard_normal((N, m)), rng.standard_normal((N, k))
        Am, Bm = hf.MultiVector.from_dense(A), hf.MultiVector.from_dense(B)
        got, ref = Am.dot_mv(Bm), A.T @ B
        sc = np.linalg.norm(A, axis=0)[:, None] * np.linalg.norm(B, axis=0)[None, :] + 1e-300
        e1 = np.max(np.abs(got - ref) / sc)
        S = rng.standard_normal((m, k))
        Y = hf.MultiVector(N, k)
        hf.MvDSmatMult(Am, S, Y)
        ref2 = A @ S
        e2 = np.max(np.abs(Y.to_dense() - ref2)) / (np.max(np.abs(ref2)) + 1e-300)
        assert np.isfinite(e1 + e2) and e1 < 1e-12 and e2 < 1e-12, (seed, it, N, m, k, e1, e2)


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_orthogonalize_against_unique_thin_qr(ctx, seed):
    rng = np.random.default_rng(2000 + seed)
    for it in range(15):
        k = int(rng.choice(EDGE_K)) if rng.random() < 0.4 else int(rng.integers(1, 257))
        N = int(rng.choice([k, k + 1, 300, 1000, 4225, 20000])) if rng.random() < 0.8 else int(rng.integers(k, 5000))
        N = max(N, k)
        cond = float(10.0 ** rng.uniform(0, 8))
        Z = rng.standard_normal((N, k)) @ np.diag(np.logspace(0, -np.log10(cond), k)) @ np.linalg.qr(rng.standard_normal((k, k)))[0]
        Q = hf.MultiVector.from_dense(Z)
        R = Q.orthogonalize()
        Qd = Q.to_dense()
        o = np.linalg.norm(Qd.T @ Qd - np.eye(k)) / np.sqrt(k)
        rec = np.linalg.norm(Qd @ R - Z) / np.linalg.norm(Z)
        assert o < 1e-12 and rec < 1e-12, (seed, it, N, k, cond, o, rec)
        assert np.allclose(np.tril(R, -1), 0) and np.all(np.diag(R) > 0)


@pytest.mark.parametrize("seed", range(5))
def test_fuzz_double_pass_against_the_oracle(ctx, seed):
    """20 random (N, n, k, r, s) per seed: fused double pass over a snapshot operator vs the oracle's BLAS-3 restatement on the
    same Omega.  Snapshot sets of rank min(n, 80) < k happen (rank-deficient operator: trailing Ritz values are round-off);
    eigenvalues are compared relative to max(lambda, 1e-7 lambda_0) -- both sides use tridiagonalisation-based eigensolvers of
    ABSOLUTE accuracy eps ||T||."""
    rng = np.random.default_rng(3000 + seed)
    for it in range(20):
        N = int(rng.choice([300, 1000, 4225, 10000, 30011]))
        n = int(rng.choice([5, 12, 40, 100, 256, 600]))
        kmax = min(N // 2, 256)
        k = int(rng.choice([kk for kk in EDGE_K if kk <= kmax])) if rng.random() < 0.35 else int(rng.integers(1, min(kmax, 200) + 1))
        r = int(rng.integers(1, k + 1))
        s = int(rng.choice([1, 1, 2]))
        rate = float(rng.choice([0.02, 0.1, 0.3]))
        latent = min(n, 80)
        U0 = np.linalg.qr(rng.standard_normal((N, latent)))[0]
        X = (rng.standard_normal((n, latent)) * np.exp(-rate * np.arange(latent))) @ U0.T      # n snapshots of length N
        op = hf.SnapshotGramOperator(X)
        Om = rng.standard_normal((N, k))
        d, U = hf.doublePass(op, hf.MultiVector.from_dense(Om), r, s=s)
        d_ref, U_ref = hp_o.double_pass_blas3(lambda W: np.asfortranarray(X.T @ (X @ W) / n), np.asfortranarray(Om), r, s=s)
        # what the probe block still carries after s applications: components below (lambda / lambda_0)^s ~ 1e-10 are round-off in
        # A^s Omega on both sides (two power iterations square the spectrum), so only eigenvalues above that are compared
        big = d_ref > (1e-10 if s == 1 else 1e-5) * d_ref[0]
        e = np.max(np.abs(d[big] - d_ref[big]) / np.maximum(d_ref[big], 1e-7 * d_ref[0])) if big.any() else 0.0
        Ud = U.to_dense()
        o = np.linalg.norm(Ud[:, big].T @ Ud[:, big] - np.eye(int(big.sum())))
        assert e < 1e-8 and o < 1e-9, (seed, it, N, n, k, r, s, rate, e, o)


@pytest.mark.parametrize("rate,s", [(0.05, 1), (0.3, 1), (0.6, 2)])
def test_first_qr_pass_on_trust_equals_the_checked_path(ctx, rate, s):
    """Round 5: the Gram-form solve no longer waits for the status words of its FIRST Cholesky-QR pass (no host round trip in the
    middle of a solve); they are read at the end with the second pass's, and a shifted / failed first pass sends the solve to
    the checked path.  Either way the result is the checked path's, bit for bit -- also when the block handed to the QR is so
    ill-conditioned (fast decay, two power iterations) that the first factorisation needs its shift."""
    from hippyflow_amd import _lib as L
    rng = np.random.default_rng(int(rate * 100) + s)
    N, n, k, r = 20011, 300, 60, 40
    U0 = np.linalg.qr(rng.standard_normal((N, 100)))[0]
    X = (rng.standard_normal((n, 100)) * np.exp(-rate * np.arange(100))) @ U0.T
    op = hf.SnapshotGramOperator(X)
    Om = hf.MultiVector.from_dense(rng.standard_normal((N, k)))
    res = {}
    try:
        for trust in (0, 1):
            L.call("hfmi_tuning_set", b"qr_trust_first", trust)
            d, U = hf.doublePass(op, Om, r, s=s)
            res[trust] = (d, U.to_dense())
    finally:
        L.call("hfmi_tuning_set", b"qr_trust_first", 1)
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_fuzz_whole_gpu_eigensolver_sizes(ctx):
    """20 random sizes n in [257, 3000] of the whole-GPU eigensolver (hfmi_eig_blocked.hip: panels above 2048 rows, one launch per column
    below, divide and conquer, block reflectors), primes and n = 1 mod 64 included, three kinds of spectra, against numpy.linalg.eigh:
    eigenvalues, orthonormality, residual."""
    rng = np.random.default_rng(6006)
    sizes = [257, 263, 449, 513, 577, 641, 1009, 1025, 1153, 1601, 2049, 2111, 2113, 2999] + [int(v) for v in rng.integers(257, 3001, 6)]
    for it, n in enumerate(sizes):
        kind = it % 3
        if kind == 0:                              # Gram matrix of decaying snapshots (the POD's matrix)
            X = rng.standard_normal((n, n // 3 + 5)) * np.exp(-0.02 * np.arange(n // 3 + 5))[None, :]
            T = X @ X.T
        elif kind == 1:                            # indefinite
            S = rng.standard_normal((n, n))
            T = S + S.T
        else:                                      # clustered: eight-fold eigenvalues
            Qm = np.linalg.qr(rng.standard_normal((n, n)))[0]
            T = (Qm * np.repeat(np.arange(1.0, 2.0 + n // 8), 8)[:n]) @ Qm.T
        T = 0.5 * (T + T.T)
        nv = int(rng.choice([n, 17, 130]))
        d, V = hf.sym_eig_small(T, nvec=nv)
        w = np.linalg.eigvalsh(T)[::-1]
        sc = np.abs(w).max()
        assert np.abs(d - w).max() <= 4e-12 * sc, (n, kind, np.abs(d - w).max() / sc)
        assert np.abs(V.T @ V - np.eye(nv)).max() <= 5e-12, (n, kind)
        assert np.abs(T @ V - V * d[:nv]).max() <= 4e-12 * sc, (n, kind)
