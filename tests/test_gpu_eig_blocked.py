"""hfmi_sym_eig_small for 256 < n <= 16384 (hfmi_eig_blocked.hip: panel tridiagonalisation on the MFMA, divide and conquer over
the whole GPU, block-reflector back-transformation) against numpy.linalg.eigh -- what the reference calls at
PODProjector.py:821 (la.eigh(G)).  Bar: eigenvalues to 1e-12 ||T||, ||V^T V - I|| <= 1e-12,
residual <= 1e-12 ||T|| (max-abs entries; n eps grows to 9e-13 at n = 4096, so the largest sizes get 4e-12, 8e-12 beyond 4096)."""
import numpy as np
import pytest

import hippyflow_amd as hf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    if hf.device_count() < 1:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return hf.Context.default()


def _spectrum_matrix(n, lam, rng):
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Qm * lam) @ Qm.T


def _check(T, tol):
    n = T.shape[0]
    d, V = hf.sym_eig_small(T)
    w = np.linalg.eigvalsh(0.5 * (T + T.T))[::-1]
    scale = max(np.abs(w).max(), 1e-300)
    assert np.all(np.diff(d) <= 0.0)
    assert np.abs(d - w).max() <= tol * scale
    assert np.abs(V.T @ V - np.eye(n)).max() <= tol
    assert np.abs(T @ V - V * d).max() <= tol * scale
    return d, V


@pytest.mark.parametrize("n", [257, 300, 511, 512, 640, 1000, 1024, 2048])
def test_sym_eig_blocked_random_and_gram(ctx, n):
    rng = np.random.default_rng(n)
    tol = 1e-12 if n <= 1024 else 4e-12
    S = rng.standard_normal((n, n))
    _check(S + S.T, tol)                                                          # indefinite
    X = rng.standard_normal((n, n + 50)) * np.exp(-0.01 * np.arange(n + 50))[None, :]
    _check(X @ X.T, tol)                                                          # Gram matrix, positive definite


@pytest.mark.parametrize("n", [384, 1500])
@pytest.mark.parametrize("kind", ["graded", "clustered", "rank-deficient", "two-clusters", "diagonal", "tridiagonal",
                                  "block-diagonal", "identity", "zero", "tiny-scale", "huge-scale"])
def test_sym_eig_blocked_hard_spectra(ctx, n, kind):
    rng = np.random.default_rng(hash(kind) % 1000 + n)
    tol = 1e-12 if n <= 1024 else 4e-12
    if kind == "graded":
        T = _spectrum_matrix(n, np.exp(-0.05 * np.arange(n)), rng)
    elif kind == "clustered":
        T = _spectrum_matrix(n, np.repeat(np.arange(1.0, 1.0 + (n + 15) // 16), 16)[:n], rng)
    elif kind == "rank-deficient":
        X = rng.standard_normal((n, n // 5))
        T = X @ X.T
    elif kind == "two-clusters":
        lam = np.where(np.arange(n) < n // 2, 1.0, -1.0) + 1e-13 * rng.standard_normal(n)
        T = _spectrum_matrix(n, lam, rng)
    elif kind == "diagonal":
        T = np.diag(rng.standard_normal(n))
    elif kind == "tridiagonal":
        T = np.diag(rng.standard_normal(n)) + np.diag(rng.standard_normal(n - 1), 1)
        T = T + np.triu(T, 1).T
    elif kind == "block-diagonal":
        T = np.zeros((n, n))
        for a in range(0, n, 37):
            b = min(n, a + 37)
            B = rng.standard_normal((b - a, b - a))
            T[a:b, a:b] = B + B.T
    elif kind == "identity":
        T = np.eye(n)
    elif kind == "zero":
        T = np.zeros((n, n))
    elif kind == "tiny-scale":
        S = rng.standard_normal((n, n))
        T = (S + S.T) * 1e-150
    else:
        S = rng.standard_normal((n, n))
        T = (S + S.T) * 1e150
    _check(T, tol)


def test_sym_eig_blocked_4096_and_sort_by_abs(ctx):
    n = 4096
    rng = np.random.default_rng(7)
    X = rng.standard_normal((n, 600)) * np.exp(-0.01 * np.arange(600))[None, :]
    G = X @ X.T                                                                    # rank 600: the POD case (few snapshots span)
    d, V = _check(G, 4e-12)
    S = rng.standard_normal((n, n))
    S = S + S.T
    d2, V2 = hf.sym_eig_small(S, sort_by_abs=True)
    assert np.all(np.diff(np.abs(d2)) <= 0.0)
    w = np.linalg.eigvalsh(S)
    w = w[np.argsort(-np.abs(w), kind="stable")]
    assert np.abs(d2 - w).max() <= 4e-12 * np.abs(w).max()
    assert np.abs(S @ V2 - V2 * d2).max() <= 4e-12 * np.abs(w).max()


def _hard_matrix(n, kind, rng):
    if kind == "clustered":
        return _spectrum_matrix(n, np.repeat(np.arange(1.0, 1.0 + (n + 15) // 16), 16)[:n], rng)
    if kind == "rank-deficient":
        X = rng.standard_normal((n, n // 5))
        return X @ X.T
    if kind == "two-clusters":
        return _spectrum_matrix(n, np.where(np.arange(n) < n // 2, 1.0, -1.0) + 1e-13 * rng.standard_normal(n), rng)
    if kind == "tridiagonal":
        T = np.diag(rng.standard_normal(n)) + np.diag(rng.standard_normal(n - 1), 1)
        return T + np.triu(T, 1).T
    if kind == "identity":
        return np.eye(n)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["clustered", "rank-deficient", "two-clusters", "tridiagonal", "identity"])
def test_sym_eig_blocked_beyond_4096_hard_spectra(ctx, kind):
    """4096 < n <= 8192: the top merge's deflation keeps its index lists in global memory (k_dcl_deflate<true>) and k_tri_b holds twice
    the column share per thread -- the spectra that exercise the rotations and the mass deflation, just past the switch."""
    n = 4200
    T = _hard_matrix(n, kind, np.random.default_rng(len(kind) + n))
    _check(T, 8e-12)


@pytest.mark.parametrize("n", [4097, 6000])
def test_sym_eig_blocked_beyond_4096_random_and_gram(ctx, n):
    rng = np.random.default_rng(n)
    S = rng.standard_normal((n, n))
    _check(S + S.T, 8e-12)                                                        # indefinite, nothing deflates
    X = rng.standard_normal((n, 700)) * np.exp(-0.01 * np.arange(700))[None, :]
    _check(X @ X.T, 8e-12)                                                        # the POD case: rank 700


def test_sym_eig_blocked_8192_gram_and_leading(ctx):
    """The largest size: Gram matrix of 8192 decaying snapshots formed on the device (hfmi_block_gram_eig, 100 leading eigenvectors)
    against numpy on the same blocks, and the full solve of an indefinite matrix."""
    n, N, nvec = 8192, 3000, 100
    rng = np.random.default_rng(8192)
    X = rng.standard_normal((N, 900)) @ (rng.standard_normal((900, n)) * np.exp(-0.01 * np.arange(900))[:, None])
    Xm = hf.MultiVector.from_dense(X)
    d, V = Xm.gram_eig(Xm, nvec)
    G = X.T @ X
    w = np.linalg.eigvalsh(G)[::-1]
    assert V.shape == (n, nvec)
    assert np.abs(d - w).max() <= 8e-12 * w[0]
    assert np.abs(G @ V - V * d[:nvec]).max() <= 8e-12 * w[0]
    assert np.abs(V.T @ V - np.eye(nvec)).max() <= 1e-12
    S = rng.standard_normal((n, n))
    _check(S + S.T, 1.6e-11)


def test_sym_eig_blocked_is_bit_reproducible(ctx):
    rng = np.random.default_rng(11)
    n = 777
    X = rng.standard_normal((n, 200))
    G = X @ X.T + 1e-3 * np.eye(n)
    d1, V1 = hf.sym_eig_small(G)
    d2, V2 = hf.sym_eig_small(G)
    assert np.array_equal(d1, d2) and np.array_equal(V1, V2)


@pytest.mark.parametrize("k", [30, 74, 128, 138, 180, 200, 256])
@pytest.mark.parametrize("kind", ["diagonal", "tridiagonal", "block-diagonal", "arrow-then-diagonal"])
def test_sym_eig_small_on_already_reduced_inputs(ctx, k, kind):
    """Inputs whose columns are (partly) reduced already: the tridiagonalisation kernel of the one-workgroup path then takes its
    tau == 0 branch, where no barrier of the arithmetic path separates the waves (advisor r4: a race between the flush of
    reflector j and the forming of reflector j + 1; fixed with a barrier).  PRE (k <= 160) and LDS-row (k > 160) instances."""
    rng = np.random.default_rng(k)
    if kind == "diagonal":
        T = np.diag(rng.standard_normal(k))
    elif kind == "tridiagonal":
        T = np.diag(rng.standard_normal(k)) + np.diag(rng.standard_normal(k - 1), 1)
        T = T + np.triu(T, 1).T
    elif kind == "block-diagonal":
        T = np.zeros((k, k))
        for a in range(0, k, 7):
            b = min(k, a + 7)
            B = rng.standard_normal((b - a, b - a))
            T[a:b, a:b] = B + B.T
    else:
        T = np.diag(rng.standard_normal(k))
        m = k // 3
        B = rng.standard_normal((m, m))
        T[:m, :m] = B + B.T                       # a dense leading block, the rest decoupled
    for _ in range(3):
        d, V = hf.sym_eig_small(T, method="dc")
        w = np.linalg.eigvalsh(T)[::-1]
        scale = np.abs(w).max()
        assert np.abs(d - w).max() <= 1e-13 * scale * k
        assert np.abs(V.T @ V - np.eye(k)).max() <= 1e-12
        assert np.abs(T @ V - V * d).max() <= 1e-13 * scale * k


@pytest.mark.parametrize("n,nvec", [(100, 7), (256, 256), (300, 1), (700, 64), (1300, 200), (2048, 128)])
def test_sym_eig_leading_eigenvectors_only(ctx, n, nvec):
    """hfmi_sym_eig_leading: all eigenvalues, the nvec leading eigenvectors -- what PODProjectorFromData uses of la.eigh(G)
    (PODProjector.py:821-826); beyond 256 rows the back-transformation runs over nvec columns.  Same bits as the full call."""
    rng = np.random.default_rng(n + nvec)
    X = rng.standard_normal((n, 80)) * np.exp(-0.05 * np.arange(80))[None, :]
    G = X @ X.T
    d_full, V_full = hf.sym_eig_small(G)
    d, V = hf.sym_eig_small(G, nvec=nvec)
    assert V.shape == (n, min(nvec, n))
    np.testing.assert_array_equal(d, d_full)
    np.testing.assert_allclose(V, V_full[:, :nvec], atol=1e-13)
    lead = min(nvec, 60)                                       # (beyond the rank of G the eigenvectors of the zero cluster are arbitrary)
    assert np.abs(G @ V[:, :lead] - V[:, :lead] * d[:lead]).max() <= 1e-12 * d[0]


@pytest.mark.parametrize("n,N,nvec", [(40, 3000, 10), (256, 2000, 256), (300, 5000, 20), (1000, 2500, 128)])
def test_gram_eig_of_two_blocks_stays_on_the_device(ctx, n, N, nvec):
    """hfmi_block_gram_eig = la.eigh(X^T (M X))[:, :u_rank] of PODProjectorFromData (PODProjector.py:818-826) without the n x n
    matrix crossing PCIe: against numpy on the same blocks."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((N, min(n, 60))) @ (rng.standard_normal((min(n, 60), n)) * np.exp(-0.1 * np.arange(min(n, 60)))[:, None])
    w = rng.uniform(0.5, 2.0, N)                                # a diagonal "mass matrix"
    Xm, MXm = hf.MultiVector.from_dense(X), hf.MultiVector.from_dense(X * w[:, None])
    d, V = Xm.gram_eig(MXm, nvec)
    G = X.T @ (X * w[:, None])
    wr = np.linalg.eigvalsh(G)[::-1]
    assert V.shape == (n, nvec)
    assert np.abs(d - wr).max() <= 1e-12 * wr[0]
    lead = min(nvec, 40)
    assert np.abs(G @ V[:, :lead] - V[:, :lead] * d[:lead]).max() <= 2e-12 * wr[0]
    assert np.abs(V[:, :lead].T @ V[:, :lead] - np.eye(lead)).max() <= 1e-12
    d2, V2 = hf.sym_eig_small(Xm.dot_mv(MXm), nvec=nvec)       # the two-call form: same eigenvalues, same leading subspace
    np.testing.assert_allclose(d2, d, atol=1e-13 * wr[0])
    np.testing.assert_allclose(np.abs(np.sum(V2[:, :lead] * V[:, :lead], axis=0)), 1.0, atol=1e-9)


@pytest.mark.parametrize("n", [40, 300, 1100])
@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_sym_eig_refuses_non_finite_input(ctx, n, bad):
    """numpy.linalg.eigh raises on NaN / inf input; so does the device solver (HFMI_ERR_NUMERIC), on both sides of n = 256, through
    the host-matrix entry and through the on-device Gram form -- and the context stays usable."""
    rng = np.random.default_rng(n)
    T = rng.standard_normal((n, n))
    T = T + T.T
    Tb = T.copy()
    Tb[n // 3, n // 2] = bad
    with pytest.raises(hf.HfmiError, match="non-finite"):
        hf.sym_eig_small(Tb)
    with pytest.raises(hf.HfmiError, match="non-finite"):
        hf.sym_eig_small(Tb, nvec=5)
    if n > 256:
        X = rng.standard_normal((2000, n))
        X[17, 5] = bad
        Xm = hf.MultiVector.from_dense(X)
        with pytest.raises(hf.HfmiError, match="non-finite"):
            Xm.gram_eig(Xm, 5)
    _check(T, 2e-12)


@pytest.mark.parametrize("shifted", [True, False])
@pytest.mark.parametrize("method", ["hep", "ghep", "inverse_ghep"])
def test_pod_from_data_320_snapshots_matches_the_reference(ctx, golden_dir, method, shifted):
    """PODProjectorFromData.construct_subspace on 320 snapshots -- the n x n Gram problem (la.eigh(G), PODProjector.py:812-833) now goes
    through hfmi_block_gram_eig and the whole-GPU eigensolver -- against the outputs of the REFERENCE's own construct_subspace on the same
    (integer-valued, hence bit-identical) snapshot matrix: tests/golden/pod_from_data_320.npz, made by tests/golden/make_pod_large_golden.py."""
    import os
    import scipy.sparse as sp
    g = np.load(os.path.join(golden_dir, "pod_from_data_320.npz"))
    N, r = int(g["N"]), int(g["r"])
    M = sp.csr_matrix((g["M_data"], g["M_indices"], g["M_indptr"]), shape=(N, N))
    u_data = g["u_int16"].astype(np.float64) * float(g["scale"])
    d, phi, Mphi, shift = hf.PODProjectorFromData(None, M).construct_subspace(u_data.copy(), r, shifted=shifted, method=method)
    tag = "%s_%d" % (method, int(shifted))
    np.testing.assert_allclose(shift, g["shift_" + tag], atol=1e-14)
    np.testing.assert_allclose(d, g["d_" + tag], rtol=1e-8)
    cos = np.abs(np.einsum("ij,ij->j", phi[:, :6], M @ g["phi_" + tag][:, :6]))
    np.testing.assert_allclose(cos, 1.0, atol=1e-8)
    eye = np.eye(r)
    assert np.linalg.norm(eye - phi.T @ Mphi) / np.linalg.norm(eye) < 1e-8        # test_PODProjector.py:154-168
    assert np.linalg.norm(M @ phi - Mphi) / np.linalg.norm(Mphi) < 1e-8           # :170-174


@pytest.mark.parametrize("state_dimension_form", [False, True])
@pytest.mark.parametrize("shifted", [True, False])
def test_pod_from_data_8300_snapshots_matches_the_reference(ctx, golden_dir, shifted, state_dimension_form):
    """MORE THAN 8192 snapshots with a slowly decaying spectrum (0.95^k per singular value, rank 150 + noise floor): the reference's
    exact la.eigh(G) (PODProjector.py:812-833; dataGenerator.py:278-279 hands it the whole training set) against the device's exact
    n x n route (hfmi_block_gram_eig up to 16384) -- tests/golden/pod_from_data_8300.npz holds the REFERENCE's outputs, the snapshot
    matrix is rebuilt from the stored seed with integer arithmetic (tests/golden/make_pod_huge_golden.py: snapshots()).  No warning:
    nothing is randomized here."""
    import os
    import sys
    import warnings
    import scipy.sparse as sp
    sys.path.insert(0, golden_dir)
    from make_pod_huge_golden import snapshots
    g = np.load(os.path.join(golden_dir, "pod_from_data_8300.npz"))
    N, r = int(g["N"]), int(g["r"])
    M = sp.csr_matrix((g["M_data"], g["M_indices"], g["M_indptr"]), shape=(N, N))
    u_data = snapshots(int(g["seed"]), int(g["n"]), N, int(g["K"]))
    pod = hf.PODProjectorFromData(None, M)
    # both exact forms of the pencil against the reference's outputs: the n x n Gram problem through the whole-GPU eigensolver
    # (8300 > 8192: the size round 5 could not take), and the 600 x 600 state-dimension form the projector picks by default here
    pod.prefer_state_dimension = state_dimension_form
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        d, phi, Mphi, shift = pod.construct_subspace(u_data.copy(), r, shifted=shifted, method="hep")
    tag = "hep_%d" % int(shifted)
    np.testing.assert_allclose(shift, g["shift_" + tag], atol=1e-13)
    np.testing.assert_allclose(d, g["d_" + tag], rtol=1e-8)
    cos = np.abs(np.einsum("ij,ij->j", phi[:, :r - 2], M @ g["phi_" + tag][:, :r - 2]))
    np.testing.assert_allclose(cos, 1.0, atol=1e-8)
    eye = np.eye(r)
    assert np.linalg.norm(eye - phi.T @ Mphi) / np.linalg.norm(eye) < 1e-8


@pytest.mark.parametrize("n", [8193, 9000, 12500, 16384])
def test_sym_eig_blocked_beyond_8192(ctx, n):
    """8192 < n <= 16384: the first columns take full-column products (v of k_tri_b is up to 128 KB of LDS), the top merge of the divide
    and conquer keeps everything in global memory beyond 9984 poles (k_dcl_deflate<2>).  A matrix with a KNOWN spectrum (a diagonal
    conjugated by three Householder reflectors: O(n^2) to build, no host eigh): every eigenvalue, and residual + orthonormality of the
    64 leading eigenvectors."""
    rng = np.random.default_rng(n)
    lam = np.sort(np.concatenate([np.exp(-0.002 * np.arange(n - 40)), np.repeat([2.0, 3.0], 10), -np.linspace(0.1, 1.0, 20)]))[::-1]
    T = np.diag(lam)
    for _ in range(3):
        u = rng.standard_normal(n)
        u /= np.linalg.norm(u)
        T -= 2.0 * np.outer(u, u @ T)
        T -= 2.0 * np.outer(T @ u, u)
    T = 0.5 * (T + T.T)
    d, V = hf.sym_eig_small(T, nvec=64)
    assert np.abs(d - lam).max() <= 2e-11 * np.abs(lam).max()
    assert np.abs(V.T @ V - np.eye(64)).max() <= 1e-11
    assert np.abs(T @ V - V * d[:64]).max() <= 2e-11 * np.abs(lam).max()


_KNOB_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, ".")
import hippyflow_amd as hf
n = int(sys.argv[1])
rng = np.random.default_rng(n)
X = rng.standard_normal((n, 300)) * np.exp(-0.02 * np.arange(300))[None, :]
G = X @ X.T + 1e-6 * np.eye(n)
d, V = hf.sym_eig_small(G, nvec=64)
w = np.linalg.eigvalsh(G)[::-1]
assert np.abs(d - w).max() <= 8e-12 * w[0], np.abs(d - w).max() / w[0]
assert np.abs(G @ V - V * d[:64]).max() <= 8e-12 * w[0]
assert np.abs(V.T @ V - np.eye(64)).max() <= 1e-12
np.save(sys.argv[2], d)
"""


_KNOB_DEFAULT = {}


def _knob_child(n, extra, out):
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _KNOB_SCRIPT, str(n), str(out)], env=dict(os.environ, **extra), capture_output=True, text=True,
                       timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("env", [{"HFMI_EIG_SYM_MIN": "0"}, {"HFMI_EIG_SYM_MIN": "1024"}, {"HFMI_EIG_LEAF": "64"}, {"HFMI_EIG_TRI_UNR": "4"}, {"HFMI_EIG_UNB_MAX": "0"}, {"HFMI_EIG_UNB_MAX": "700"}, {"HFMI_XFER_PLAIN": "1"}, {"HFMI_EIG_GEMM": "0"}, {"HFMI_EIG_WY": "256"}, {"HFMI_EIG_WY": "512"}, {"HFMI_XFER_D2H_ENGINE": "1"}, {"HFMI_EIG_FULL_UPDATE": "1"}, {"HFMI_EIG_NO_LD_PAD": "1"},
                                 {"HFMI_EIG_LARGE": "jacobi"}])
def test_sym_eig_blocked_ab_knobs_give_the_same_spectrum(ctx, tmp_path, env):
    """The A/B switches of the whole-GPU solver (read once per process, hence a child interpreter each): full-column products only
    / lower-triangle products from 1024 rows on, 64-row leaves, four loads in flight, the Jacobi of rounds 2-4 -- every route passes the
    same bars and agrees with the default route's eigenvalues to rounding."""
    n = 1500 if "HFMI_EIG_LARGE" in env else 4300
    if n not in _KNOB_DEFAULT:
        _KNOB_DEFAULT[n] = _knob_child(n, {}, tmp_path / "default.npy")
    d = _knob_child(n, env, tmp_path / "knob.npy")
    assert np.abs(d - _KNOB_DEFAULT[n]).max() <= 1e-12 * d[0]


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(1536, 2048, 200), (1111, 1793, 77), (256, 8192, 1030), (2050, 2050, 128), (300, 200, 64)])
def test_dgemm_of_the_eigensolver_matches_numpy(ctx, ta, tb, M, N, K):
    """The general fp64 MFMA product behind the trailing updates, the merges and the block reflectors (software-pipelined 128 x 128 /
    64 x 128 tiles where they fill the chip, the 64 x 64 kernel otherwise; odd sizes take the 8-byte load path and the edge guards)
    against numpy, to a few ulps of the accumulated magnitude."""
    rng = np.random.default_rng(M + 3 * N + 7 * K + ta + 2 * tb)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    Cm, ms = ctx.bench_dgemm(A, B, ta=ta, tb=tb, reps=1)
    ref = (A.T if ta else A) @ (B.T if tb else B)
    assert np.abs(Cm - ref).max() <= 1e-13 * K * max(1.0, np.abs(ref).max())

