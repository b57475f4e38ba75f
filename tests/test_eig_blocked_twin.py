"""The numpy twin of the whole-GPU eigensolver (tests/helpers/eig_blocked_twin.py) against numpy.linalg.eigh: the panel
recurrences with the lazily finalised W column, the leaf tearing and the block-reflector back-transformation are right
before a kernel runs.  CPU only."""
import numpy as np
import pytest

from tests.helpers import dc_eig_twin as dct
from tests.helpers import eig_blocked_twin as bt


def _cases(n, rng):
    S = rng.standard_normal((n, n))
    yield "indefinite", S + S.T
    X = rng.standard_normal((n, max(2, n // 4)))
    yield "rank-deficient", X @ X.T
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    yield "graded", (Qm * np.exp(-0.2 * np.arange(n))) @ Qm.T
    lam = np.repeat(np.arange(1.0, 1.0 + (n + 7) // 8), 8)[:n]
    yield "clustered", (Qm * lam) @ Qm.T
    yield "diagonal", np.diag(rng.standard_normal(n))
    off = rng.standard_normal(n - 1)
    yield "tridiagonal", np.diag(rng.standard_normal(n)) + np.diag(off, 1) + np.diag(off, -1)


@pytest.mark.parametrize("n,nb,leaf", [(45, 8, 12), (97, 16, 20), (130, 32, 40)])
def test_blocked_twin_matches_eigh(n, nb, leaf):
    rng = np.random.default_rng(n)
    for name, T in _cases(n, rng):
        w, V = bt.eigh_blocked(T, nb=nb, leaf_max=leaf)
        wr = np.linalg.eigvalsh(T)[::-1]
        scale = max(np.abs(wr).max(), 1e-300)
        assert np.abs(w - wr).max() <= 1e-13 * n * scale, name
        assert np.abs(V.T @ V - np.eye(n)).max() <= 1e-13 * n, name
        assert np.abs(T @ V - V * w).max() <= 1e-13 * n * scale, name


def test_blocked_tridiagonalisation_is_the_unblocked_factorisation():
    rng = np.random.default_rng(5)
    n = 70
    S = rng.standard_normal((n, n))
    S = S + S.T
    d, e, Vh, tau = bt.tridiagonalize_blocked(S, 16)
    d0, e0, V0, tau0 = dct.tridiagonalize(S)
    assert np.abs(d - d0).max() < 1e-11 and np.abs(e - e0).max() < 1e-11
    assert np.abs(Vh - V0).max() < 1e-11 and np.abs(tau - tau0).max() < 1e-11
    # the block reflectors reproduce the product of the single ones
    Z = rng.standard_normal((n, 9))
    assert np.abs(bt.back_transform_blocked(Vh, tau, Z, 16) - dct.back_transform(V0, tau0, Z)).max() < 1e-11


@pytest.mark.parametrize("n,nb,unb", [(70, 16, 70), (70, 16, 30), (131, 32, 64), (90, 8, 1)])
def test_unblocked_tail_with_the_delayed_update_is_the_same_factorisation(n, nb, unb):
    """k_tri_u's recurrence (tests/helpers/eig_blocked_twin.py unblocked_tail): alone (unb >= n), behind some panels, and for a
    tail of one panel's worth: d, e, reflectors and tau of LAPACK's factorisation, for every kind of matrix of the list above."""
    rng = np.random.default_rng(n + unb)
    for name, S in _cases(n, rng):
        d, e, Vh, tau = bt.tridiagonalize_blocked(S, nb, unb_max=unb)
        d0, e0, V0, tau0 = dct.tridiagonalize(S)
        scale = max(np.abs(S).max(), 1e-300)
        # the reflectors of a (numerically) rank-deficient matrix are not unique: compare what they produce
        Q = dct.back_transform(Vh, tau, np.eye(n))
        Tm = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-12 * n, name
        assert np.abs(Q @ Tm @ Q.T - 0.5 * (S + S.T)).max() < 1e-12 * n * scale, name
        if name == "indefinite":
            # (step n - 2 leaves the unit vector e_{n-1} with tau = 0 in column n - 2 of the reflector matrix: the identity)
            assert np.abs(d - d0).max() < 1e-10 and np.abs(e - e0).max() < 1e-10 and np.abs(tau - tau0).max() < 1e-10
            assert np.abs(Vh[:, :n - 2] - V0[:, :n - 2]).max() < 1e-10 and tau[n - 2] == 0.0


@pytest.mark.parametrize("n,j", [(500, 0), (500, 126), (500, 127), (513, 200), (700, 383), (640, 511)])
def test_lower_triangle_products_cover_every_row_once_per_slot(n, j):
    """The tile / slot scheme of k_tri_bs + k_tri_a<true>: half the matrix is read, every row finds one partial value in each of its nb
    slots, and their sum is A v on the rows below j (v vanishes on rows <= j; rows and columns <= j inside the first block hold
    older data: they must not leak into the result)."""
    rng = np.random.default_rng(n + j)
    A = rng.standard_normal((n, n))
    A = A + A.T
    v = rng.standard_normal(n)
    v[:j + 1] = 0.0
    y, reads = bt.lower_triangle_products(A, v, j)
    ref = A @ v
    ref[:j + 1] = 0.0
    np.testing.assert_allclose(y, ref, atol=1e-11)
    m = -(-n // 128) * 128 - ((j + 1) // 128) * 128
    assert reads == 128 * 128 * (m // 128) * (m // 128 + 1) // 2
    assert reads <= 0.5 * m * m + 128 * m                                  # half the block + its diagonal tiles
