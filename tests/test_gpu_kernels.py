"""GPU parity tests, kernel level: every libhfmi primitive against numpy / the oracle on seeded inputs.
All calls go through the C ABI (ctypes).  Run on the MI355X box: ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

hf = pytest.importorskip("hippyflow_amd")


@pytest.fixture(scope="module")
def ctx():
    if hf.device_count() < 1:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return hf.Context.default()


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


# ------------------------------------------------------------------ layouts
@pytest.mark.parametrize("N,k", [(1, 1), (5, 3), (33, 2), (4225, 30), (1000, 257)])
def test_upload_download_roundtrip(ctx, N, k):
    rng = np.random.default_rng(N + k)
    dense = rng.standard_normal((N, k))
    mv = hf.MultiVector.from_dense(dense)
    assert mv.nvec() == k and mv.size() == N and mv.leading_dimension() % 32 == 0
    np.testing.assert_array_equal(mv.to_dense(), dense)
    np.testing.assert_array_equal(mv.to_vectors(), dense.T)
    mv2 = hf.MultiVector.from_vectors(dense.T.copy())
    np.testing.assert_array_equal(mv2.to_dense(), dense)
    # views and the vector protocol
    j = k // 2
    np.testing.assert_array_equal(mv[j].get_local(), dense[:, j])
    v = hf.Vector()
    v.init(N)
    v.set_local(np.arange(N, dtype=float))
    mv[j].axpy(2.0, v)
    np.testing.assert_allclose(mv.to_dense()[:, j], dense[:, j] + 2.0 * np.arange(N), rtol=1e-15)
    assert abs(mv[j].inner(v) - (dense[:, j] + 2.0 * np.arange(N)) @ np.arange(N)) <= 1e-12 * max(1.0, N ** 3)
    cp = hf.MultiVector(mv)
    cp.scale(0.5)
    np.testing.assert_allclose(cp.to_dense(), 0.5 * mv.to_dense(), rtol=1e-15)
    np.testing.assert_allclose(mv.norm(), np.linalg.norm(mv.to_dense(), axis=0), rtol=1e-13)
    cp.zero()
    assert not cp.to_dense().any()


def test_swap_and_copy_semantics(ctx):
    a = hf.MultiVector.from_dense(np.ones((10, 2)))
    b = hf.MultiVector.from_dense(2 * np.ones((10, 2)))
    a.swap(b)
    assert a.to_dense()[0, 0] == 2.0 and b.to_dense()[0, 0] == 1.0


# ------------------------------------------------------------------ Philox / probe draw (a1)
def test_philox_stream_bit_exact_and_normals(ctx):
    import ctypes as C
    from hippyflow_amd import _lib as L
    from oracle import philox
    N, k, seed, stream = 1001, 7, 0x1234567890ABCDEF, 3
    mv = hf.MultiVector(N, k)
    raw = np.empty((k, (N + 3) // 4, 4), dtype=np.uint32)
    L.call("hfmi_philox_raw", mv.handle, C.c_uint64(seed), C.c_uint32(stream), L.ptr(raw))
    np.testing.assert_array_equal(raw, philox.raw_block(N, k, seed, stream))       # integer stream: bit-exact
    L.call("hfmi_randn_fill", mv.handle, C.c_uint64(seed), C.c_uint32(stream), 1.5)
    ref = philox.randn_block(N, k, seed, stream, sigma=1.5)
    # the device log / sqrt / sincos are range-specific (hfmi_randn_math.h; bounds in test_randn_math_twin.py) and differ from libm
    # by a few ulp; |z| <= 6.76 sigma, so 1e-13 absolute is ~50 ulp
    np.testing.assert_allclose(mv.to_dense(), ref, rtol=0, atol=1e-13)


def test_parRandom_is_reproducible_across_contexts(ctx):
    hf.parRandom.reseed(42)
    a = hf.MultiVector(3001, 5)
    hf.parRandom.normal(1.0, a)
    hf.parRandom.reseed(42)
    b = hf.MultiVector(3001, 5)
    hf.parRandom.normal(1.0, b)
    np.testing.assert_array_equal(a.to_dense(), b.to_dense())
    c = hf.MultiVector(3001, 5)
    hf.parRandom.normal(1.0, c)          # next stream: independent draw
    assert abs(np.corrcoef(a.to_dense()[:, 0], c.to_dense()[:, 0])[0, 1]) < 0.1
    Z = a.to_dense()
    assert abs(Z.mean()) < 0.05 and abs(Z.std() - 1) < 0.05


# ------------------------------------------------------------------ tall-skinny contractions
TN_SHAPES = [(1, 1, 1), (3, 2, 7), (9, 13, 100), (16, 16, 32), (17, 33, 1000), (30, 30, 4225), (138, 138, 4225),
             (256, 30, 4225), (64, 74, 20011), (300, 260, 1000), (500, 84, 3000), (2048, 138, 20000), (700, 5, 50000),
             # both operands skinny (tsgemm_ss): every tiles-per-wave class, ragged tiles, more slices than stages
             (8, 138, 70001), (138, 8, 70001), (160, 128, 5000), (144, 144, 33), (150, 130, 9999), (97, 160, 64),
             (33, 17, 300000), (160, 1, 2049), (112, 110, 12345),
             # column remainders of the last tile handled as 1 / 2 / 3 groups on the 4x4x4 MFMA (and 13: a full tile again)
             (500, 74, 3000), (400, 36, 5000), (300, 100, 4000), (1000, 9, 4000), (1000, 12, 3000), (333, 45, 2000),
             (3000, 4, 2000), (700, 170, 3000)]


@pytest.mark.parametrize("m,k,N", TN_SHAPES)
def test_block_dot_matches_numpy(ctx, m, k, N):
    """dot_mv = A^T B (tsgemm_tn): asymmetric random operands catch transposed fragment maps."""
    rng = np.random.default_rng(m * 1000 + k)
    A = rng.standard_normal((N, m)) * np.logspace(0, -3, m)[None, :]
    B = rng.standard_normal((N, k)) + 0.1
    got = hf.MultiVector.from_dense(A).dot_mv(hf.MultiVector.from_dense(B))
    ref = A.T @ B
    scale = np.linalg.norm(A, axis=0)[:, None] * np.linalg.norm(B, axis=0)[None, :]
    assert np.max(np.abs(got - ref) / scale) < 1e-13


@pytest.mark.parametrize("m,k,N", [(300, 74, 5000), (1000, 84, 3000), (500, 138, 4000), (383, 30, 2000), (700, 5, 9000)])
def test_tsgemm_tn_single_split_writes_the_result_directly(ctx, m, k, N):
    """One split, nothing to scale: tsgemm_tn stores the result block itself (bounds-checked, no partial + reduce)."""
    import ctypes as C
    from hippyflow_amd import _lib as L
    rng = np.random.default_rng(m + k)
    A = rng.standard_normal((N, m))
    B = rng.standard_normal((N, k))
    Am, Bm = hf.MultiVector.from_dense(A), hf.MultiVector.from_dense(B)
    got = np.full((m, k), np.nan)
    L.call("hfmi_tuning_set", b"ss", 0)          # keep skinny shapes on the tn kernel
    try:
        L.call("hfmi_bench_tsgemm_tn", Am.handle, Bm.handle, 1, 0, L.ptr(got), None)
    finally:
        L.call("hfmi_tuning_set", b"ss", 1)
    ref = A.T @ B
    scale = np.linalg.norm(A, axis=0)[:, None] * np.linalg.norm(B, axis=0)[None, :]
    assert np.max(np.abs(got - ref) / scale) < 1e-13


@pytest.mark.parametrize("k,N", [(5, 1000), (74, 30011), (138, 50000), (160, 4225)])
def test_block_gram_same_operand(ctx, k, N):
    """Q^T Q with both arguments the SAME block (staged once through LDS) equals the two-operand product."""
    rng = np.random.default_rng(k)
    Q = rng.standard_normal((N, k)) * np.logspace(0, -2, k)[None, :]
    mv = hf.MultiVector.from_dense(Q)
    got = mv.dot_mv(mv)
    ref = Q.T @ Q
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    assert np.max(np.abs(got - ref) / scale) < 1e-13
    np.testing.assert_array_equal(got, got.T)     # same summation order for (i,j) and (j,i)
    other = hf.MultiVector.from_dense(Q)
    assert np.max(np.abs(mv.dot_mv(other) - got) / scale) < 1e-13


def test_block_dot_is_deterministic(ctx):
    rng = np.random.default_rng(0)
    A = hf.MultiVector.from_dense(rng.standard_normal((50000, 40)))
    g1, g2 = A.dot_mv(A), A.dot_mv(A)
    np.testing.assert_array_equal(g1, g2)         # fixed split-reduction order


NN_SHAPES = [(1, 1, 1), (7, 3, 2), (100, 9, 13), (4225, 30, 20), (4225, 256, 30), (20011, 74, 64), (1000, 300, 260),
             (50000, 138, 128), (3000, 2048, 138), (999, 5, 1),
             # column remainders on the 4x4x4 MFMA: r = 74 (3 groups), 84 (1), 36 (1), 100 (1), 9 (3), 12 (3), 45 (full tile)
             (20011, 200, 74), (5000, 300, 84), (7000, 100, 36), (3000, 64, 100), (4000, 50, 9), (4000, 50, 12), (2000, 40, 45),
             (30000, 700, 138), (2500, 64, 170)]


@pytest.mark.parametrize("N,m,r", NN_SHAPES)
def test_gemm_small_matches_numpy(ctx, N, m, r):
    """MvDSmatMult / reduce = A S (tsgemm_nn), with alpha/beta."""
    rng = np.random.default_rng(N + m + r)
    A = rng.standard_normal((N, m))
    S = rng.standard_normal((m, r)) * np.logspace(0, -2, r)[None, :]
    Y0 = rng.standard_normal((N, r))
    Amv, Y = hf.MultiVector.from_dense(A), hf.MultiVector.from_dense(Y0)
    hf.MvDSmatMult(Amv, S, Y)
    ref = A @ S
    assert rel(Y.to_dense(), ref) < 1e-13
    # accumulate form through the C ABI: Y = 0.5 A S - 2 Y
    import ctypes as C
    from hippyflow_amd import _lib as L
    Y2 = hf.MultiVector.from_dense(Y0)
    L.call("hfmi_block_gemm_small", Amv.handle, L.ptr(np.ascontiguousarray(S)), 0.5, -2.0, Y2.handle)
    assert rel(Y2.to_dense(), 0.5 * ref - 2.0 * Y0) < 1e-13
    # padding rows of the result stay zero (later reductions run over them unmasked)
    if r <= 256:
        assert abs(Y.dot_mv(Y) - ref.T @ ref).max() <= 1e-11 * max(1.0, np.abs(ref.T @ ref).max())


def test_reduce_and_dot_v(ctx):
    rng = np.random.default_rng(5)
    U = rng.standard_normal((777, 6))
    x = rng.standard_normal(777)
    Umv = hf.MultiVector.from_dense(U)
    xv = hf.Vector()
    xv.init(777)
    xv.set_local(x)
    np.testing.assert_allclose(Umv.dot_v(xv), U.T @ x, rtol=1e-12, atol=1e-12)
    y = hf.Vector()
    y.init(777)
    y.set_local(np.ones(777))
    alpha = rng.standard_normal(6)
    Umv.reduce(y, alpha)
    np.testing.assert_allclose(y.get_local(), 1.0 + U @ alpha, rtol=1e-12, atol=1e-12)


def test_shape_mismatch_raises(ctx):
    a, b = hf.MultiVector(10, 2), hf.MultiVector(11, 2)
    with pytest.raises(hf.HfmiError):
        a.dot_mv(b)
    with pytest.raises(AssertionError):
        hf.MatMvMult(hf.SnapshotGramOperator(np.ones((3, 10))), a, hf.MultiVector(10, 3))


# ------------------------------------------------------------------ sparse
def _fem(N):
    h = 1.0 / (N - 1)
    main = np.full(N, 4 * h / 6)
    main[[0, -1]] = 2 * h / 6
    M = sp.diags([np.full(N - 1, h / 6), main, np.full(N - 1, h / 6)], [-1, 0, 1], format="csr")
    kd = np.full(N, 2 / h)
    kd[[0, -1]] = 1 / h
    K = sp.diags([np.full(N - 1, -1 / h), kd, np.full(N - 1, -1 / h)], [-1, 0, 1], format="csr")
    return M, K


@pytest.mark.parametrize("N,k", [(50, 1), (1000, 9), (4225, 30)])
def test_csr_spmm_and_pcg(ctx, N, k):
    rng = np.random.default_rng(N)
    M, K = _fem(N)
    A = (M + 0.01 * K).tocsr()
    X = rng.standard_normal((N, k))
    op = hf.CsrOperator(A)
    Xmv, Y = hf.MultiVector.from_dense(X), hf.MultiVector(N, k)
    op.matMvMult(Xmv, Y)
    assert rel(Y.to_dense(), A @ X) < 1e-14
    op.matMvMult(Xmv, Y, accumulate=True)
    assert rel(Y.to_dense(), 2 * (A @ X)) < 1e-14
    solver = hf.CsrPCGSolver(M, rel_tol=1e-13)
    Z = hf.MultiVector(N, k)
    solver.matMvMult(Xmv, Z)
    import scipy.sparse.linalg as spla
    ref = spla.splu(M.tocsc()).solve(X)
    assert rel(Z.to_dense(), ref) < 1e-10


@pytest.mark.parametrize("k", [1, 7, 84, 130, 300])
def test_sparse_solver_chebyshev_and_cg_routes(ctx, k):
    """M^-1 (prior.Msolver, KLEProjector.py:163-164).  A mass matrix has a narrow Jacobi-scaled spectrum ([1/2, 2] for P1
    elements): the Chebyshev iteration (hfmi_cheb.hip) serves it -- wave-per-row kernel for even k <= 128, thread-per-entry
    kernel otherwise -- with the step count its bracket predicts; a stiffness-dominated matrix (condition number ~1e4) is left
    to the block CG.  Both reach the residual they were asked for."""
    import scipy.sparse.linalg as spla
    N = 6000
    rng = np.random.default_rng(k)
    M, K = _fem(N)
    X = rng.standard_normal((N, k)) * np.exp(rng.standard_normal(k))[None, :]        # columns of very different size
    X[:, k // 2] = 0.0                                                               # and a zero right-hand side
    Xmv, Z = hf.MultiVector.from_dense(X), hf.MultiVector(N, k)
    solver = hf.CsrPCGSolver(M, rel_tol=1e-13)
    solver.matMvMult(Xmv, Z)
    info = solver.info()
    lo, hi = info["spectrum"]
    assert info["method"] == "chebyshev" and 20 <= info["iterations"] <= 40
    ev = np.linalg.eigvalsh((M.toarray() / M.diagonal()[:, None] ** 0.5) / M.diagonal()[None, :] ** 0.5)
    assert lo <= ev[0] and ev[-1] <= hi * (1 + 1e-9) and lo > 0.8 * ev[0] and hi < 1.1 * ev[-1]      # a bracket, and a tight one
    res = M @ Z.to_dense() - X
    assert np.all(np.linalg.norm(res, axis=0) <= 2e-13 * np.linalg.norm(X, axis=0) + 1e-300)
    assert not Z.to_dense()[:, k // 2].any()
    assert rel(Z.to_dense(), spla.splu(M.tocsc()).solve(X)) < 1e-11
    stiff = (M + 1e-4 * K).tocsr()
    solver2 = hf.CsrPCGSolver(stiff, rel_tol=1e-12, max_iter=4000)
    solver2.matMvMult(Xmv, Z)
    assert solver2.info()["method"] == "cg" and solver2.info()["spectrum"] is None
    res = stiff @ Z.to_dense() - X
    assert np.all(np.linalg.norm(res, axis=0) <= 1e-10 * np.linalg.norm(X, axis=0) + 1e-300)


@pytest.mark.parametrize("k", [1, 13, 84])
def test_csr_irregular_rows_use_the_csr_kernel(ctx, k):
    """A matrix with one dense row and empty rows has no ELL image (padding > 1.5x): SpMM and PCG must take the
    CSR kernels and agree with scipy; an SPD arrow matrix keeps PCG meaningful."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    N = 3001
    rng = np.random.default_rng(k)
    R = sp.random(N, N, density=2e-3, random_state=3, format="lil")
    R[7, :] = rng.standard_normal(N)            # one full row
    R[11, :] = 0.0                              # an empty row
    R = R.tocsr()
    X = rng.standard_normal((N, k))
    Xmv, Y = hf.MultiVector.from_dense(X), hf.MultiVector(N, k)
    hf.CsrOperator(R).matMvMult(Xmv, Y)
    scale = np.abs(R) @ np.abs(X) + 1e-300
    assert np.max(np.abs(Y.to_dense() - R @ X) / scale) < 1e-13
    arrow = sp.diags(np.linspace(2.0, 3.0, N), format="lil")
    arrow[0, 1:] = 1e-2
    arrow[1:, 0] = 1e-2
    arrow = arrow.tocsr()
    Z = hf.MultiVector(N, k)
    hf.CsrPCGSolver(arrow, rel_tol=1e-13).matMvMult(Xmv, Z)
    assert rel(Z.to_dense(), spla.splu(arrow.tocsc()).solve(X)) < 1e-10


@pytest.mark.parametrize("m,k,N", [(100003, 9, 2048), (51277, 20, 1024), (70000, 74, 512)])
def test_tn_hybrid_plan_whole_rounds_plus_split_tail(ctx, m, k, N):
    """tsgemm_tn with more row blocks than CUs: the blocks that make whole rounds are split coarsely (or not at all and written
    straight to the result), the ragged rest finely, into its own partial buffer.  Against the uniform split (tuning
    "tn_hybrid" = 0) and numpy, through the row-major result of block_dot and through the scaled, column-major one of a
    snapshot-Gram apply."""
    from hippyflow_amd import _lib as L
    A = hf.MultiVector(N, m)
    B = hf.MultiVector(N, k)
    hf.parRandom.reseed(m)
    hf.parRandom.normal(1.0, A)
    hf.parRandom.normal(1.0, B)
    Ad, Bd = A.to_dense(), B.to_dense()
    ref = Ad.T @ Bd
    out, app = {}, {}
    for hyb in (0, 1):
        L.call("hfmi_tuning_set", b"tn_hybrid", hyb)
        try:
            out[hyb] = A.dot_mv(B)
            if k >= 20:      # Y = X^T (X W) / m with the m "snapshots" as the vectors of A: the first half is the scaled tn launch
                op = hf.SnapshotGramOperator(A)
                Y = hf.MultiVector(N, k)
                op.matMvMult(B, Y)
                app[hyb] = Y.to_dense()
        finally:
            L.call("hfmi_tuning_set", b"tn_hybrid", 1)
    scale = np.abs(ref).max()
    assert np.abs(out[1] - ref).max() < 1e-13 * scale * np.sqrt(N)
    assert np.abs(out[1] - out[0]).max() < 1e-13 * scale * np.sqrt(N)
    if app:
        refY = Ad @ (Ad.T @ Bd) / m
        assert rel(app[1], refY) < 1e-12 and rel(app[1], app[0]) < 1e-13


def test_resident_nn_product_is_independent_of_the_tile_height(ctx):
    """Q R^-1 with the small matrix resident in LDS: the tile height is chosen by the number of rounds the persistent
    workgroups need (N = 2e5, k = 74: two 16-row tiles per wave instead of three); the choice moves rows between waves and
    must not change a single bit, including in place."""
    from hippyflow_amd import _lib as L
    import ctypes as C
    rng = np.random.default_rng(3)
    for N, m, r in [(200000, 74, 74), (70001, 40, 33)]:
        A = hf.MultiVector(N, m)
        hf.parRandom.normal(1.0, A)
        S = rng.standard_normal((m, r))
        outs = []
        for tt in (1, 2, 0):
            L.call("hfmi_tuning_set", b"nn_res_tt", tt)
            try:
                Y = hf.MultiVector(N, r)
                ms = C.c_double(0)
                L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 1, C.byref(ms))
                outs.append(Y.to_dense())
            finally:
                L.call("hfmi_tuning_set", b"nn_res_tt", 0)
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
        ref = A.to_dense() @ S
        assert np.abs(outs[0] - ref).max() < 1e-12 * np.abs(ref).max() * m


@pytest.mark.parametrize("N,k", [(4096, 17), (20000, 74), (50001, 138), (8192, 160), (30000, 33)])
def test_triangular_skip_of_q_rinv_is_bit_identical(ctx, N, k):
    """Q <- Q R^-1 of the Cholesky-QR: R^-1 is upper triangular, the resident-S kernel skips the column tiles that are structurally
    zero at each reduction step (tuning key nn_upper).  Skipped products are with exact zeros, so Q and R must not change by a bit
    -- with the blocked MFMA Cholesky and with the column-at-a-time kernels (both must leave the strict lower triangle of R^-1
    as zeros), for a graded block that needs more than one pass."""
    from hippyflow_amd import _lib as L
    rng = np.random.default_rng(N + k)
    Z = rng.standard_normal((N, k)) * np.exp(-0.08 * np.arange(k))
    for chol in (0, 1):
        got = {}
        for upper in (0, 1):
            L.call("hfmi_tuning_set", b"chol", chol)
            L.call("hfmi_tuning_set", b"nn_upper", upper)
            try:
                Q = hf.MultiVector.from_dense(Z)
                R = Q.orthogonalize()
                got[upper] = (R, Q.to_dense(), Q.last_qr_passes)
            finally:
                L.call("hfmi_tuning_set", b"chol", 0)
                L.call("hfmi_tuning_set", b"nn_upper", 1)
        assert got[0][2] == got[1][2]
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]), "chol=%d" % chol
        Qd = got[1][1]
        assert np.abs(Qd.T @ Qd - np.eye(k)).max() < 1e-14 * max(k, 8) and rel(Qd @ got[1][0], Z) < 1e-12


# ------------------------------------------------------------------ QR (a7)
@pytest.mark.parametrize("N,k,cond", [(300, 20, 1e0), (4225, 30, 1e3), (4225, 30, 1e9), (20000, 138, 1e5), (1000, 200, 1e2)])
def test_orthogonalize_matches_reference_mgs(ctx, N, k, cond):
    from oracle import hippylib_restated as hp_o
    rng = np.random.default_rng(k)
    Z = rng.standard_normal((N, k)) @ np.diag(np.logspace(0, -np.log10(cond), k)) @ np.linalg.qr(rng.standard_normal((k, k)))[0]
    Zf = np.asfortranarray(Z)
    Q = hf.MultiVector.from_dense(Z)
    R = Q.orthogonalize()
    Qd = Q.to_dense()
    assert np.linalg.norm(Qd.T @ Qd - np.eye(k)) / np.sqrt(k) < 1e-13, "orthonormality"
    assert np.allclose(np.tril(R, -1), 0) and np.all(np.diag(R) > 0)
    assert rel(Qd @ R, Z) < 1e-12, "QR = Z"
    if N * k <= 4225 * 30:
        Qo = Zf.copy(order="F")
        Ro = hp_o.mgs_reortho(Qo)
        # thin QR with positive diagonal is unique: the Cholesky-QR Q equals the reference's MGS Q
        assert np.abs(Qd - Qo).max() < 1e-8 * cond ** 0.5 + 1e-12
        assert rel(R, Ro) < 1e-9
    if cond >= 1e9:
        assert Q.last_qr_passes >= 3      # shifted / repeated passes were needed


def test_orthogonalize_mgs_method_and_rank_deficiency(ctx):
    from hippyflow_amd import _lib as L
    from oracle import hippylib_restated as hp_o
    rng = np.random.default_rng(7)
    Z = rng.standard_normal((500, 6))
    Z[:, 3] = Z[:, 0] - 2 * Z[:, 1]                     # numerically dependent column
    Qo = np.asfortranarray(Z.copy())
    Ro = hp_o.mgs_reortho(Qo)
    for method in (L.QR_MGS, L.QR_AUTO):               # AUTO: Cholesky breaks down -> falls back to MGS
        Q = hf.MultiVector.from_dense(Z)
        R = Q.orthogonalize(method)
        Qd = Q.to_dense()
        assert R[3, 3] == 0.0 and not Qd[:, 3].any(), "method %d" % method    # the reference zeroes the dependent column
        np.testing.assert_allclose(np.abs(Qd), np.abs(Qo), atol=1e-10)
        np.testing.assert_allclose(R, Ro, atol=1e-10)


@pytest.mark.parametrize("method", ["chol", "mgs"])
def test_Borthogonalize(ctx, method):
    from hippyflow_amd import _lib as L
    from oracle import hippylib_restated as hp_o
    rng = np.random.default_rng(11)
    N, k = 2000, 25
    M, K = _fem(N)
    B = (M + 1e-3 * K).tocsr()
    Z = rng.standard_normal((N, k)) @ np.diag(np.logspace(0, -4, k))
    Q = hf.MultiVector.from_dense(Z)
    BQ, R = Q.Borthogonalize(B, L.QR_CHOL if method == "chol" else L.QR_MGS)
    Qd = Q.to_dense()
    assert np.linalg.norm(Qd.T @ (B @ Qd) - np.eye(k)) / np.sqrt(k) < 1e-12
    assert rel(BQ.to_dense(), B @ Qd) < 1e-12
    assert rel(Qd @ R, Z) < 1e-11
    Qo = np.asfortranarray(Z.copy())
    BQo, Ro = hp_o.mgs_stable(Qo, hp_o.SparseOperator(B))
    assert np.abs(Qd - Qo).max() < 1e-9 and rel(R, Ro) < 1e-9


@pytest.mark.parametrize("k", [1, 2, 15, 16, 17, 31, 33, 74, 80, 81, 112, 113, 138, 144, 145, 192, 193, 200, 255, 256])
def test_blocked_cholesky_against_lapack_and_the_column_kernel(ctx, k):
    """hfmi_chol.hip (16 x 16 blocks on the MFMA) at every block-count boundary: thin QR of a graded block vs numpy's
    Householder QR, and the same call on the column-at-a-time kernels (tuning "chol" = 1)."""
    from hippyflow_amd import _lib as L
    rng = np.random.default_rng(1000 + k)
    N = 4 * k + 64
    Z = rng.standard_normal((N, k)) * np.exp(-0.03 * np.arange(k))
    out = {}
    for which in (0, 1):
        L.call("hfmi_tuning_set", b"chol", which)
        try:
            Q = hf.MultiVector.from_dense(Z)
            out[which] = (Q.orthogonalize(), Q.to_dense())
        finally:
            L.call("hfmi_tuning_set", b"chol", 0)
    R, Qd = out[0]
    Qn, Rn = np.linalg.qr(Z)
    sgn = np.sign(np.diag(Rn))
    Rn, Qn = Rn * sgn[:, None], Qn * sgn
    assert np.abs(Qd.T @ Qd - np.eye(k)).max() < 1e-14 * max(k, 8)
    assert not np.tril(R, -1).any() and np.all(np.diag(R) > 0)
    assert rel(R, Rn) < 1e-12 and np.abs(Qd - Qn).max() < 1e-11
    assert rel(R, out[1][0]) < 1e-13 and np.abs(Qd - out[1][1]).max() < 1e-12


def test_blocked_cholesky_shifted_retry_and_rank_deficiency(ctx):
    """cond(Z) = 1e9: the first Gram matrix is numerically singular, the factorisation breaks down inside a diagonal block,
    the kernel restarts with the diagonal shift and the later passes repair the basis; an exactly dependent column ends on
    the Gram-Schmidt route exactly as with the column kernel."""
    rng = np.random.default_rng(5)
    N, k = 6000, 100
    Z = rng.standard_normal((N, k)) @ np.diag(np.logspace(0, -9, k)) @ np.linalg.qr(rng.standard_normal((k, k)))[0]
    Q = hf.MultiVector.from_dense(Z)
    R = Q.orthogonalize()
    Qd = Q.to_dense()
    assert Q.last_qr_passes >= 3
    assert np.abs(Qd.T @ Qd - np.eye(k)).max() < 1e-13 and rel(Qd @ R, Z) < 1e-12
    Z2 = rng.standard_normal((800, 40))
    Z2[:, 33] = Z2[:, 2] + Z2[:, 17]
    Q2 = hf.MultiVector.from_dense(Z2)
    R2 = Q2.orthogonalize()
    assert R2[33, 33] == 0.0 and not Q2.to_dense()[:, 33].any()


# ------------------------------------------------------------------ small eigensolve (a8)
@pytest.mark.parametrize("method", ["dc", "jacobi"])
@pytest.mark.parametrize("k", [1, 2, 3, 5, 16, 17, 30, 31, 64, 65, 74, 80, 81, 84, 96, 97, 128, 129, 137, 138, 144, 145, 148, 192, 193, 200, 256])
def test_sym_eig_small_matches_eigh(ctx, k, method):
    rng = np.random.default_rng(k)
    Qm = np.linalg.qr(rng.standard_normal((k, k)))[0]
    lam = np.exp(-0.25 * np.arange(k)) * np.where(np.arange(k) % 7 == 3, -1.0, 1.0)   # decaying, some negative
    T = (Qm * lam) @ Qm.T
    T = 0.5 * (T + T.T)
    d, V = hf.sym_eig_small(T, method=method)
    w = np.linalg.eigvalsh(T)[::-1]
    assert np.max(np.abs(d - w)) < 1e-14 * k * np.abs(w).max()
    assert np.all(np.diff(d) <= 0)
    assert np.linalg.norm(V.T @ V - np.eye(k)) < 1e-13 * k
    assert np.linalg.norm(T @ V - V * d) < 1e-13 * k * np.abs(w).max()
    d_abs, _ = hf.sym_eig_small(T, sort_by_abs=True, method=method)
    assert np.all(np.diff(np.abs(d_abs)) <= 0)


def _hard_spectra():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((74, 74))
    yield "random", A + A.T
    X = rng.standard_normal((60, 20))
    yield "rank-deficient", X @ X.T
    yield "identity", np.eye(50)
    yield "zero", np.zeros((20, 20))
    Q = np.linalg.qr(rng.standard_normal((100, 100)))[0]
    lam = np.concatenate([np.ones(40), np.ones(30) * (1 + 1e-10), np.linspace(0, 1, 30)])
    T = (Q * lam) @ Q.T
    yield "clusters", 0.5 * (T + T.T)
    n = 41
    yield "wilkinson", np.diag(np.abs(np.arange(n) - 20.0)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    n = 128
    yield "laplace", 2 * np.eye(n) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1)
    J = rng.standard_normal((6400, 74))
    yield "wishart", J.T @ J
    J = rng.standard_normal((2048, 138)) * np.exp(-0.05 * np.arange(138))
    yield "decay138", J.T @ J
    yield "huge", 1e150 * (A + A.T)
    yield "tiny", 1e-150 * (A + A.T)
    yield "diagonal", np.diag(np.arange(70.0) - 30.0)
    B = rng.standard_normal((250, 250))
    yield "random250", B + B.T


@pytest.mark.parametrize("name,T", list(_hard_spectra()), ids=[n for n, _ in _hard_spectra()])
def test_sym_eig_small_divide_and_conquer_hard_spectra(ctx, name, T):
    """The cases tests/test_dc_twin.py runs through the numpy twin, through the kernels: deflation by small components
    and by close poles, all-deflated merges, clusters, exact zeros, scaling."""
    k = T.shape[0]
    d, V = hf.sym_eig_small(T, method="dc")
    w = np.linalg.eigvalsh(T)[::-1]
    nrm = max(np.abs(w).max(), 1e-300)
    assert np.all(np.diff(d) <= 0)
    assert np.max(np.abs(d - w)) < 1e-14 * k * nrm
    assert np.linalg.norm(V.T @ V - np.eye(k)) < 1e-13 * k
    assert np.linalg.norm(T @ V - V * d) < 1e-13 * k * nrm


def test_sym_eig_small_dc_equals_numpy_twin(ctx):
    """Same tree, same deflation rule, same secular iteration: kernel and twin agree far below the error of either."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    import dc_eig_twin as tw
    rng = np.random.default_rng(11)
    J = rng.standard_normal((500, 74))
    T = J.T @ J
    d, V = hf.sym_eig_small(T, method="dc")
    dt, Vt = tw.eigh_dc(T)
    assert np.max(np.abs(d - dt)) < 5e-15 * np.abs(dt).max()
    sgn = np.sign(np.sum(V * Vt, axis=0))
    assert np.abs(V * sgn - Vt).max() < 1e-11


def test_sym_eig_small_graded_relative_accuracy(ctx):
    """Jacobi (method="jacobi") resolves tiny eigenvalues of a graded PSD matrix to high relative accuracy; divide and
    conquer, like the LAPACK routine behind the reference's np.linalg.eigh, to eps ||T|| absolutely."""
    rng = np.random.default_rng(3)
    k = 40
    G = rng.standard_normal((200, k)) @ np.diag(np.logspace(0, -6, k))
    T = G.T @ G
    w = np.linalg.svd(G, compute_uv=False) ** 2
    d, _ = hf.sym_eig_small(T, method="jacobi")
    assert np.max(np.abs(d - w) / w) < 1e-9
    d, _ = hf.sym_eig_small(T, method="dc")
    assert np.max(np.abs(d - w)) < 1e-14 * k * w[0]
    ref = np.linalg.eigvalsh(T)[::-1]
    assert np.max(np.abs(d - w)) < 20 * max(np.max(np.abs(ref - w)), 1e-16 * w[0])    # no worse than LAPACK's eigh in kind


@pytest.mark.parametrize("k", [1, 2, 7, 30, 75, 138, 200])
def test_svd_small_matches_numpy(ctx, k):
    rng = np.random.default_rng(k)
    R = np.triu(rng.standard_normal((k, k))) @ np.diag(np.logspace(0, -10, k))      # graded, like the QR factor of a decaying block
    U, sv, V = hf.svd_small(R)
    ref = np.linalg.svd(R, compute_uv=False)
    assert np.all(np.diff(sv) <= 0)
    np.testing.assert_allclose(sv, ref, rtol=1e-9, atol=1e-15 * ref[0])               # relative accuracy of small singular values
    assert np.linalg.norm(U.T @ U - np.eye(k)) < 1e-12 * k
    assert np.linalg.norm(V.T @ V - np.eye(k)) < 1e-12 * k
    assert np.linalg.norm((U * sv) @ V.T - R) < 1e-13 * k * np.linalg.norm(R)


# ------------------------------------------------------------------ streaming ingest (f3 / verdict r2 item 7)
def test_async_upload_from_pinned_memory_and_ingest_stream(ctx):
    """hfmi_block_upload_async / hfmi_ingest_wait / hfmi_ingest_fence: sample-by-sample filling of a block through pinned
    double buffers (PODProjector.py:343-357, activeSubspaceProjector.py:178-221) gives exactly the synchronous upload."""
    rng = np.random.default_rng(5)
    ns, q, N = 7, 5, 4099
    J = rng.standard_normal((ns, q, N))
    ref = hf.MultiVector.from_vectors(J.reshape(ns * q, N))
    blk = hf.ingest_stream((J[i] for i in range(ns)), ns, q, N)
    np.testing.assert_array_equal(blk.to_vectors(), ref.to_vectors())
    # compute enqueued after the fence sees the data: Gram of the ingested block == Gram of the reference block
    np.testing.assert_array_equal(blk.dot_mv(blk), ref.dot_mv(ref))
    # dense layout, explicit tickets, more uploads in flight than the ring of tickets is long
    D = rng.standard_normal((N, 3))
    pin = hf.pinned_empty((N, 3))
    pin[...] = D
    dst = hf.MultiVector(N, 3 * 12)
    tickets = [dst.view(3 * i, 3).upload_async(pin, layout="dense") for i in range(12)]
    for t in tickets:
        hf.Context.default().ingest_wait(t)
    hf.Context.default().ingest_fence()
    got = dst.to_dense()
    for i in range(12):
        np.testing.assert_array_equal(got[:, 3 * i:3 * i + 3], D)
    with pytest.raises(ValueError):
        dst.view(0, 3).upload_async(np.zeros((N, 3)))            # (N, nvec) is the dense layout, not "vectors"
    with pytest.raises(hf.HfmiError):
        hf.Context.default().ingest_wait(10 ** 6)


def test_projectors_take_streamed_samples(ctx):
    """An observable that produces its Jacobians one at a time (jacobian_stream) gives the same active subspace as the
    same Jacobians handed over as one array."""
    rng = np.random.default_rng(8)
    ns, q, N = 6, 4, 1500
    J = rng.standard_normal((ns, q, N)) * np.exp(-0.002 * np.arange(N))

    class Batch:
        def jacobian_data(self, n):
            return J[:n]

    class Stream:
        def jacobian_shape(self):
            return q, N

        def jacobian_stream(self, n):
            for i in range(n):
                yield J[i]

    out = []
    for obs in (Batch(), Stream()):
        pars = hf.ActiveSubspaceParameterList()
        pars['samples_per_process'] = ns
        pars['rank'] = 5
        pars['oversampling'] = 3
        pars['save_and_plot'] = False
        pars['verbose'] = False
        hf.parRandom.reseed(21)
        prj = hf.ActiveSubspaceProjector(obs, None, parameters=pars)
        d, V, _ = prj.construct_input_subspace(prior_preconditioned=False)
        out.append((d, V.to_dense() if hasattr(V, "to_dense") else np.asarray(V)))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])


def test_in_job_roofline_denominators_are_plausible(ctx):
    """The micro-benchmarks bench.py quotes its fractions against (denominators measured in the same job): the MFMA loop on
    constant operands, the same loop on Gaussian operands rotated through the registers (round 6: the ceiling the contractions can
    reach on the solve's data), the copy and the read-only stream -- each inside the range a working MI355X can produce, and ordered
    the way the power limit orders them (full-mantissa operands never faster than constants by more than noise)."""
    p = ctx.bench_peaks()
    r = ctx.bench_random_peaks()
    h = ctx.bench_hbm_read()
    assert 40.0 < p["mfma_f64_tflops"] < 82.0 and 2000.0 < p["hbm_copy_gbs"] < 8000.0
    assert 40.0 < r["mfma_f64_tflops_random_operands"] < 82.0 and 30.0 < r["mfma_f64_tflops_random_operands_while_streaming"] < 82.0
    assert r["mfma_f64_tflops_random_operands"] <= 1.05 * p["mfma_f64_tflops"]
    assert 2500.0 < h["hbm_read_gbs"] < 8000.0
