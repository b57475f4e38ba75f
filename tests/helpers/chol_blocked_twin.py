"""numpy twin of hippyflow_amd/csrc/hfmi_chol.hip: blocked Cholesky G = R^T R with the inverse carried as the augmented
identity, 16 x 16 blocks, block (J, I) of the matrix and block (I, J) of the augmented part sharing one slot.  Test
infrastructure only (the CPU suite checks the recurrences here; the GPU suite checks the kernel against LAPACK)."""
import numpy as np

NB = 16


def _p(x, y):
    """the only product the kernel has: X^T Y of two blocks in accumulator layout"""
    return x.T @ y


def diag_block(a):
    """Gaussian elimination on [A | I] without pivoting: returns R (upper), Z = R^-T (lower) and the pivots"""
    v = a.copy()
    w = np.eye(NB)
    piv = np.zeros(NB)
    for j in range(NB):
        piv[j] = v[j, j]
        inv = 1.0 / piv[j]
        tv, tw = v[j] * inv, w[j] * inv
        for i in range(j + 1, NB):
            s = v[j, i]                       # U[j][i]: by symmetry the multiplier of row i is s / pivot
            v[i] -= s * tv
            w[i] -= s * tw
    rs = 1.0 / np.sqrt(piv)
    return np.triu(v * rs[:, None]), np.tril(w * rs[:, None]), piv


def chol_blocked(g):
    """returns (R, R^-1, pivots) of the symmetrised g"""
    k = g.shape[0]
    nb = (k + NB - 1) // NB
    n = nb * NB
    a = np.eye(n)
    a[:k, :k] = 0.5 * (g + g.T)
    slot = {(i, j): a[NB * i:NB * i + NB, NB * j:NB * j + NB].copy() for j in range(nb) for i in range(j + 1)}
    r_out, r_inv, pivots = np.zeros((n, n)), np.zeros((n, n)), np.zeros(n)
    for K in range(nb):
        sl = slice(NB * K, NB * K + NB)
        r, z, piv = diag_block(slot[(K, K)])
        pivots[sl] = piv
        x = z.T
        panel = {K: z}
        r_out[sl, sl], r_inv[sl, sl] = r, x
        for (i, j), blk in slot.items():
            if i == K and j > K:              # A_KJ -> R_KJ; the slot then carries W_JK
                panel[j] = _p(x, blk)
                r_out[sl, NB * j:NB * j + NB] = panel[j]
                slot[(i, j)] = np.zeros((NB, NB))
            elif j == K and i < K:            # W_KI -> Z_KI = (R^-1)_IK^T
                panel[i] = _p(x, blk)
                r_inv[NB * i:NB * i + NB, sl] = panel[i].T
        for (i, j), blk in slot.items():
            if j > K:
                ia, ib = (i, j) if i > K else (j, i)
                slot[(i, j)] = blk - _p(panel[ia], panel[ib])
    return r_out[:k, :k], r_inv[:k, :k], pivots[:k]
