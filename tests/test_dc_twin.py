"""CPU suite: the numpy twin of the device Rayleigh-Ritz eigensolver (tests/helpers/dc_eig_twin.py -- Householder
tridiagonalisation, divide and conquer down to 1 x 1 leaves, back-transformation; the kernels of
hippyflow_amd/csrc/hfmi_eig_dc.hip follow it step by step) against numpy.linalg.eigh, the call the reference makes
inside hippylib's doublePass / doublePassG and at PODProjector.py:821."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import dc_eig_twin as tw  # noqa: E402


def _check(T, tol=60.0):
    n = T.shape[0]
    d, V = tw.eigh_dc(T)
    w = np.linalg.eigvalsh(T)[::-1]
    nrm = max(np.abs(w).max(), 1e-300)
    assert np.all(np.diff(d) <= 0)
    assert np.max(np.abs(d - w)) <= tol * tw.EPS * nrm
    assert np.linalg.norm(V.T @ V - np.eye(n)) <= tol * tw.EPS * n
    assert np.linalg.norm(T @ V - V * d) <= tol * tw.EPS * n * nrm


def test_tree_in_closed_form_is_a_partition():
    for n in range(1, 257):
        levels = 0
        while (1 << levels) < n:
            levels += 1
        for L in range(levels + 1):
            bounds = [(i * n) >> L for i in range((1 << L) + 1)]
            assert bounds[0] == 0 and bounds[-1] == n and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
            for i in range(1 << L):
                lo, hi = bounds[i], bounds[i + 1]
                mid = ((2 * i + 1) * n) >> (L + 1)
                if hi - lo >= 2:
                    assert lo < mid < hi
                # thread t finds its node as ((t + 1) 2^L - 1) // n
                for t in range(lo, hi):
                    assert (((t + 1) << L) - 1) // n == i
        assert all((((i + 1) * n) >> levels) - ((i * n) >> levels) <= 1 for i in range(1 << levels))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 30, 31, 74, 138, 200])
def test_twin_matches_eigh_decaying_indefinite(n):
    rng = np.random.default_rng(n)
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    lam = np.exp(-0.25 * np.arange(n)) * np.where(np.arange(n) % 7 == 3, -1.0, 1.0)
    T = (Q * lam) @ Q.T
    _check(0.5 * (T + T.T))


def test_twin_hard_spectra():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((74, 74))
    _check(A + A.T)
    X = rng.standard_normal((60, 20))
    _check(X @ X.T)                                     # rank deficient: 40 zero eigenvalues
    _check(np.eye(50))
    _check(np.zeros((20, 20)))
    Q = np.linalg.qr(rng.standard_normal((100, 100)))[0]
    lam = np.concatenate([np.ones(40), np.ones(30) * (1 + 1e-10), np.linspace(0, 1, 30)])
    T = (Q * lam) @ Q.T
    _check(0.5 * (T + T.T))                             # clusters
    n = 41
    _check(np.diag(np.abs(np.arange(n) - 20.0)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1))   # Wilkinson
    n = 128
    _check(2 * np.eye(n) - np.diag(np.ones(n - 1), 1) - np.diag(np.ones(n - 1), -1))
    J = rng.standard_normal((6400, 74))
    _check(J.T @ J)                                     # config-4-like Wishart spectrum
    J = rng.standard_normal((2048, 138)) * np.exp(-0.05 * np.arange(138))
    _check(J.T @ J)                                     # config-3-like decay
    _check(1e150 * (A + A.T))                           # scaling
    _check(1e-150 * (A + A.T))


def test_twin_sort_by_abs():
    rng = np.random.default_rng(5)
    A = rng.standard_normal((33, 33))
    d, V = tw.eigh_dc(A + A.T, sort_by_abs=True)
    assert np.all(np.diff(np.abs(d)) <= 0)
    assert np.linalg.norm((A + A.T) @ V - V * d) < 1e-12
