"""CPU checks of the recurrences behind the blocked Cholesky kernel (tests/helpers/chol_blocked_twin.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
from chol_blocked_twin import chol_blocked, diag_block  # noqa: E402


@pytest.mark.parametrize("k", [1, 5, 16, 17, 31, 32, 33, 74, 138, 145, 200, 256])
def test_blocked_factor_and_inverse_match_lapack(k):
    rng = np.random.default_rng(k)
    y = rng.standard_normal((3 * k + 5, k)) * np.exp(-0.04 * np.arange(k))
    g = y.T @ y
    r, ri, piv = chol_blocked(g)
    rl = np.linalg.cholesky(g).T
    assert np.abs(r - rl).max() <= 1e-13 * np.abs(rl).max()
    assert np.abs(r @ ri - np.eye(k)).max() < 1e-12
    assert not np.tril(r, -1).any() and not np.tril(ri, -1).any()
    assert np.allclose(piv, np.diag(rl) ** 2, rtol=1e-10)


def test_diagonal_block_elimination_gives_the_transposed_inverse():
    rng = np.random.default_rng(3)
    b = rng.standard_normal((40, 16))
    a = b.T @ b
    r, z, piv = diag_block(a)
    assert np.abs(r.T @ r - a).max() < 1e-12 * np.abs(a).max()
    assert np.abs(z @ r.T - np.eye(16)).max() < 1e-12
    assert np.all(piv > 0)
