"""CPU twin of the whole-GPU symmetric eigensolver for 256 < n <= 8192 (hippyflow_amd/csrc/hfmi_eig_blocked.hip):
panel Householder tridiagonalisation (LAPACK dlatrd's recurrences, with the last column of W finalised lazily so that a
column costs two launches), divide and conquer with leaves of at most 256 solved independently and the upper merges on the
whole GPU, blocked (compact WY) back-transformation.

Test infrastructure (never imported by the product).  It states in numpy the steps and the order of operations of the
kernels so that the algebra -- in particular the lazily finalised W column, alpha from the identity
v^T w' = tau (v^T y - 2 x1.x2), and the tearing of leaf boundaries -- is pinned against numpy.linalg.eigh on the CPU
(tests/test_eig_blocked_twin.py) before a kernel runs.  The reference's counterpart: la.eigh(G) of PODProjector.py:812-833.
"""
import numpy as np

from . import dc_eig_twin as dct

EPS = np.finfo(np.float64).eps


def reflector_scalars(alpha0, xn2):
    """(beta, tau, scl) of the reflector that maps (alpha0, x) to (beta, 0): v = (1, x * scl)."""
    if xn2 <= 1e-280:
        return alpha0, 0.0, 0.0
    nrm = np.sqrt(alpha0 * alpha0 + xn2)
    beta = -np.copysign(nrm, alpha0)
    tau = (beta - alpha0) / beta
    scl = 1.0 / (alpha0 - beta)
    return beta, tau, scl


def unblocked_tail(A, j0, d, e, Vh, tau):
    """Twin of k_tri_u (one launch per column): from column j0 on -- where A is fully updated -- LAPACK's unblocked recurrence
    with the rank-2 update of reflector j - 1 DELAYED into step j: every step first turns y = A v of the last step into
    w = tau y + alpha v (alpha = -tau^2/2 v.y), forms column j of the reduced matrix from the not-yet-updated one, builds
    reflector j from it, and then, column by column of the full symmetric trailing block, applies the pending update and takes
    the product with the new v.  The last two steps only collect d and e of the final 2 x 2 block (tau = 0 there)."""
    n = A.shape[0]
    y = v = None
    for j in range(j0, n):
        if j > j0:
            t = tau[j - 1]
            alpha = -0.5 * t * t * float(v[j:] @ y[j:])
            w = np.zeros(n)
            w[j:] = t * y[j:] + alpha * v[j:]
            c = A[j:, j] - v[j:] * w[j] - w[j:] * v[j]          # v[j] = 1
        else:
            w = None
            c = A[j:, j].copy()
        d[j] = c[0]
        vn = np.zeros(n)
        if j + 1 < n:
            beta, tj, scl = reflector_scalars(c[1], float(c[2:] @ c[2:]))
            vn[j + 1] = 1.0
            vn[j + 2:] = c[2:] * scl
            e[j] = beta
            tau[j] = tj
            Vh[:, j] = vn
        yn = np.zeros(n)
        for q in range(j + 1, n):                               # a wave per column
            if w is not None:
                A[j + 1:, q] -= v[j + 1:] * w[q] + w[j + 1:] * v[q]
            yn[q] = float(A[j + 1:, q] @ vn[j + 1:])
        y, v = yn, vn


def tridiagonalize_blocked(T, nb=64, unb_max=0):
    """Symmetric T -> (d, e, Vh, tau): the same factorisation as dc_eig_twin.tridiagonalize, computed panel by panel.
    Inside a panel the trailing block is NOT updated (dlatrd): column j is corrected with the panel's V and W, the
    product with the trailing block is corrected likewise; one symmetric rank-2nb update per panel.  Once the trailing block
    has at most ``unb_max`` rows at a panel boundary the rest is left to unblocked_tail (one launch per column)."""
    A = np.array(T, dtype=np.float64)
    A = 0.5 * (A + A.T)
    n = A.shape[0]
    Vh = np.zeros((n, n))
    tau = np.zeros(n)
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 0))
    p0 = 0
    while p0 < n - 2:
        if n - p0 <= unb_max:
            unblocked_tail(A, p0, d, e, Vh, tau)
            return d, e, Vh, tau
        ncols = min(nb, n - 2 - p0)
        W = np.zeros((n, ncols))
        V = Vh[:, p0:p0 + ncols]                                  # a window: columns fill in as the panel advances
        y = x1 = x2 = None
        vy = 0.0

        def finalize(jj_done, j_done):
            """column jj_done of W from the products of step j_done (kernel A's first half)."""
            v = Vh[:, j_done]
            t = tau[j_done]
            k = jj_done
            r = slice(j_done + 1, n)
            yc = y[r] - V[r, :k] @ x2[:k] - W[r, :k] @ x1[:k]
            vtw = t * (vy - 2.0 * float(x1[:k] @ x2[:k]))         # v^T w' without a second reduction
            alpha = -0.5 * t * vtw
            W[r, k] = t * yc + alpha * v[r]

        for jj in range(ncols):
            j = p0 + jj
            # ---- kernel A: finalise the previous W column, then column j of the reduced matrix
            if jj > 0:
                finalize(jj - 1, j - 1)
            col = A[j:, j] - V[j:, :jj] @ W[j, :jj] - W[j:, :jj] @ V[j, :jj]
            d[j] = col[0]
            xn2 = float(col[2:] @ col[2:])
            # ---- kernel B: reflector, products with the trailing block and the panel
            beta, tj, scl = reflector_scalars(col[1], xn2)
            v = np.zeros(n)
            v[j + 1] = 1.0
            v[j + 2:] = col[2:] * scl
            e[j] = beta
            tau[j] = tj
            Vh[:, j] = v
            y = np.zeros(n)
            y[j + 1:] = A[j + 1:, j + 1:] @ v[j + 1:]
            x1 = V[:, :jj].T @ v
            x2 = W[:, :jj].T @ v
            vy = float(v @ y)
        finalize(ncols - 1, p0 + ncols - 1)
        t0 = p0 + ncols
        Vt, Wt = V[t0:, :], W[t0:, :]
        A[t0:, t0:] -= Vt @ Wt.T + Wt @ Vt.T
        p0 = t0
    if n >= 2:
        d[n - 2] = A[n - 2, n - 2]
        e[n - 2] = A[n - 1, n - 2]
    d[n - 1] = A[n - 1, n - 1]
    return d, e, Vh, tau


def lower_triangle_products(A, v, j, ts=128):
    """Twin of k_tri_bs + the slot sums of k_tri_a<true>: y = A v for a symmetric A and a v that vanishes on rows <= j, touching only the tiles
    (I, J), I >= J, of the trailing block counted from rs2 = (j + 1) rounded down to ``ts`` -- a tile below the diagonal serves
    y_I += A_IJ v_J and y_J += A_IJ^T v_I; tile (I, J) leaves the first product in slot J (rows of I), the second in slot I
    (rows of J), so that every row receives exactly nb partial values, one per slot, added in a fixed order.  Returns (y with
    rows <= j set to 0, number of matrix entries read)."""
    n = A.shape[0]
    ld = -(-n // ts) * ts
    rs2 = ((j + 1) // ts) * ts
    nb = (ld - rs2) // ts
    Ap = np.zeros((ld, ld))
    Ap[:n, :n] = A
    vp = np.zeros(ld)
    vp[:n] = v
    part = np.full((nb, ld), np.nan)                    # every (slot, row >= rs2) must be written exactly once
    reads = 0
    for I in range(nb):
        for J in range(I + 1):
            r0, c0 = rs2 + ts * I, rs2 + ts * J
            tile = Ap[r0:r0 + ts, c0:c0 + ts]
            reads += tile.size
            assert np.all(np.isnan(part[J, r0:r0 + ts]))
            part[J, r0:r0 + ts] = tile @ vp[c0:c0 + ts]
            if I != J:
                assert np.all(np.isnan(part[I, c0:c0 + ts]))
                part[I, c0:c0 + ts] = tile.T @ vp[r0:r0 + ts]
    assert not np.isnan(part[:, rs2:]).any()
    y = np.zeros(ld)
    for k in range(nb):
        y[rs2:] += part[k, rs2:]
    y[:j + 1] = 0.0
    return y[:n], reads


def wy_factor(V, tau):
    """Upper triangular Tf with H_0 H_1 ... H_{b-1} = I - V Tf V^T (LAPACK dlarft, forward, columnwise), from the Gram
    matrix of the panel."""
    b = V.shape[1]
    G = V.T @ V
    Tf = np.zeros((b, b))
    for i in range(b):
        Tf[i, i] = tau[i]
        if i:
            Tf[:i, i] = -tau[i] * (Tf[:i, :i] @ G[:i, i])
    return Tf


def wy_factor_merged(V, tau, nb):
    """The triangular factor of a block of several panels, assembled as the kernels do it: dlarft on the diagonal nb x nb
    blocks, then neighbours merged pairwise, [V_a V_b] -> [[T_a, -T_a (V_a^T V_b) T_b], [0, T_b]], width nb -> 2 nb -> ..."""
    b = V.shape[1]
    assert b % nb == 0 and ((b // nb) & (b // nb - 1)) == 0
    G = V.T @ V
    Tf = np.zeros((b, b))
    for a in range(0, b, nb):
        Tf[a:a + nb, a:a + nb] = wy_factor(V[:, a:a + nb], tau[a:a + nb])
    w = nb
    while w < b:
        for a in range(0, b, 2 * w):
            Ta, Tb = Tf[a:a + w, a:a + w], Tf[a + w:a + 2 * w, a + w:a + 2 * w]
            Tf[a:a + w, a + w:a + 2 * w] = -Ta @ (G[a:a + w, a + w:a + 2 * w] @ Tb)
        w *= 2
    return Tf


def back_transform_blocked(Vh, tau, Z, nb=64, panels_per_block=4):
    """(H_0 H_1 ... H_{n-3}) Z, block reflector by block reflector from the last: Z <- Z - (V Tf) (V^T Z); a block is
    ``panels_per_block`` panels of nb columns (columns beyond n - 3 are zero reflectors with tau = 0)."""
    n = Z.shape[0]
    Z = Z.copy()
    wb = nb * panels_per_block
    npad = -(-n // wb) * wb
    Vp = np.zeros((n, npad))
    Vp[:, :n] = Vh
    tp = np.zeros(npad)
    tp[:n] = tau
    for p0 in reversed(range(0, npad, wb)):
        if p0 >= n - 2:
            continue
        V = Vp[:, p0:p0 + wb]
        Y = V @ wy_factor_merged(V, tp[p0:p0 + wb], nb)
        r = slice(p0 + 1, n)
        Z[r, :] -= Y[r, :] @ (V[r, :].T @ Z[r, :])
    return Z


def leaf_level(n, leaf_max=256):
    L = 0
    while ((n + (1 << L) - 1) >> L) > leaf_max:
        L += 1
    return L


def dc_tridiagonal_large(d, e, leaf_max=256, stats=None):
    """Eigen-decomposition of tridiag(d, e): the couplings of tree levels 0 .. Lf-1 are torn up front, the 2^Lf leaves
    (<= leaf_max rows each) are solved on their own (by the one-workgroup divide and conquer), the upper merges follow
    level by level.  Returns (lam, Z), unsorted."""
    n = len(d)
    D = np.array(d, dtype=np.float64)
    e = np.array(e, dtype=np.float64)
    Lf = leaf_level(n, leaf_max)
    Q = np.zeros((n, n))
    for i in range(1 << Lf):
        lo, hi = (i * n) >> Lf, ((i + 1) * n) >> Lf
        dl = D[lo:hi].copy()
        if lo > 0:
            dl[0] -= abs(e[lo - 1])
        if hi < n:
            dl[-1] -= abs(e[hi - 1])
        lam, Zl = dct.dc_tridiagonal(dl, e[lo:hi - 1])
        D[lo:hi] = lam
        Q[lo:hi, lo:hi] = Zl
    for L in range(Lf - 1, -1, -1):
        for i in range(1 << L):
            lo, hi = (i * n) >> L, ((i + 1) * n) >> L
            mid = ((2 * i + 1) * n) >> (L + 1)
            dct.merge(D, Q, lo, mid, hi, e[mid - 1], stats)
    return D, Q


def eigh_blocked(T, sort_by_abs=False, nb=64, leaf_max=256):
    """np.linalg.eigh(T) with eigenvalues DEscending (or by |d|): the interface of hfmi_sym_eig_small for n > 256."""
    T = np.asarray(T, dtype=np.float64)
    amax = float(np.max(np.abs(T)))
    sexp = int(np.floor(np.log2(amax))) if amax > 0.0 and np.isfinite(amax) else 0
    d, e, Vh, tau = tridiagonalize_blocked(np.ldexp(T, -sexp), nb)
    lam, Z = dc_tridiagonal_large(d, e, leaf_max)
    lam = np.ldexp(lam, sexp)
    W = back_transform_blocked(Vh, tau, Z, nb)
    key = -np.abs(lam) if sort_by_abs else -lam
    order = np.argsort(key, kind="stable")
    return lam[order], W[:, order]
