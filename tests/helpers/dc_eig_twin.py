"""CPU twin of the device Rayleigh-Ritz eigensolver (hippyflow_amd/csrc/hfmi_eig_dc.hip): Householder
tridiagonalisation + Cuppen divide and conquer down to 1 x 1 leaves + back-transformation.

Test infrastructure (never imported by the product): it states, in numpy, exactly the steps and the order of
operations the kernels use -- the tearing of every coupling up front, the level-by-level merges, the deflation scan
in sorted order, the secular solver in the variable shifted to the nearer pole, the Gu-Eisenstat (Loewner)
re-computation of the rank-one vector -- so that a logic error shows up on the CPU (tests/test_dc_twin.py pins it
against numpy.linalg.eigh) before a kernel is run.  The reference's counterpart is the np.linalg.eigh call inside
hippylib's doublePass / doublePassG (LAPACK dsyevd: the same algorithm family) and la.eigh of PODProjector.py:821.
"""
import numpy as np

EPS = np.finfo(np.float64).eps


# ------------------------------------------------------------------ Householder tridiagonalisation
def tridiagonalize(T):
    """Symmetric T -> (d, e, V, tau): T = H (tridiag(d, e)) H^T with H = H_0 H_1 ... H_{n-3},
    H_j = I - tau_j v_j v_j^T, v_j zero above row j+1 and v_j[j+1] = 1."""
    A = np.array(T, dtype=np.float64)
    A = 0.5 * (A + A.T)
    n = A.shape[0]
    V = np.zeros((n, n))
    tau = np.zeros(n)
    for j in range(n - 2):
        x = A[j + 1:, j].copy()
        alpha = x[0]
        xnorm2 = float(x[1:] @ x[1:])
        if xnorm2 == 0.0:
            continue                               # nothing to annihilate: H_j = I
        beta = -np.copysign(np.sqrt(alpha * alpha + xnorm2), alpha)
        tau[j] = (beta - alpha) / beta
        v = x / (alpha - beta)
        v[0] = 1.0
        V[j + 1:, j] = v
        # two-sided update of the trailing block: A <- H A H
        p = tau[j] * (A[j + 1:, j + 1:] @ v)
        w = p - (0.5 * tau[j] * (p @ v)) * v
        A[j + 1:, j + 1:] -= np.outer(v, w) + np.outer(w, v)
        A[j + 1, j] = A[j, j + 1] = beta
        A[j + 2:, j] = 0.0
        A[j, j + 2:] = 0.0
    d = np.diag(A).copy()
    e = np.diag(A, -1).copy() if n > 1 else np.zeros(0)
    return d, e, V, tau


def back_transform(V, tau, Z):
    """H Z: the reflectors applied in reverse order to the columns of Z."""
    n = Z.shape[0]
    Z = Z.copy()
    for j in range(n - 3, -1, -1):
        if tau[j] == 0.0:
            continue
        v = V[:, j]
        Z -= np.outer(tau[j] * v, v @ Z)
    return Z


# ------------------------------------------------------------------ secular equation
def secular_root(i, dl, z2, rho, max_iter=100, stats=None):
    """Root number i (0-based, ascending) of 1 + rho sum_j z2_j / (dl_j - lam) = 0 for strictly increasing dl and
    z2 > 0, rho > 0.  Returns (origin index, tau): lam = dl[origin] + tau, with tau computed in the shifted variable
    so that dl_j - lam = (dl_j - dl[origin]) - tau carries no cancellation."""
    K = len(dl)
    if K == 1:
        return 0, rho * z2[0]
    last = i == K - 1
    if last:
        org = K - 1
        delta = dl - dl[org]
        lo, hi = 0.0, rho * float(np.sum(z2))       # f(0+) = -inf, f(hi) >= 0
    else:
        gap = dl[i + 1] - dl[i]
        delta = dl - dl[i]
        fmid = 1.0 + rho * float(np.sum(z2 / (delta - 0.5 * gap)))
        if fmid >= 0.0:                             # root in the lower half: origin = left pole
            org, lo, hi = i, 0.0, 0.5 * gap
        else:
            org, lo, hi = i + 1, -0.5 * gap, 0.0
            delta = dl - dl[i + 1]
    i0 = K - 2 if last else i                       # psi = poles 0..i0, phi = poles i0+1..
    i1 = i0 + 1

    def evaluate(tau):
        t = z2 / (delta - tau)
        psi = rho * float(np.sum(t[:i1]))
        phi = rho * float(np.sum(t[i1:]))
        dt = t / (delta - tau)
        dpsi = rho * float(np.sum(dt[:i1]))
        dphi = rho * float(np.sum(dt[i1:]))
        err = 8.0 * (abs(psi) + abs(phi)) + 1.0 + abs(tau) * (dpsi + dphi)
        return 1.0 + psi + phi, dpsi, dphi, err

    tau = 0.5 * (lo + hi)                           # first evaluation at the middle of the bracket
    for it in range(max_iter):
        f, dpsi, dphi, err = evaluate(tau)
        if stats is not None:
            stats[0] += 1
        if abs(f) <= EPS * err:
            break
        if f < 0.0:
            lo = tau
        else:
            hi = tau
        if hi - lo <= 2.0 * EPS * max(abs(lo), abs(hi)):
            tau = 0.5 * (lo + hi)
            break
        # "middle way" (Li 1994; dlaed4): psi ~ s + S / (delta_i0 - x), phi ~ r + R / (delta_i1 - x) matched in value and
        # slope at tau; the increment eta solves  c eta^2 - a eta + b = 0
        D0, D1 = delta[i0] - tau, delta[i1] - tau
        dw = dpsi + dphi
        a = (D0 + D1) * f - D0 * D1 * dw
        b = D0 * D1 * f
        c = f - D0 * dpsi - D1 * dphi
        cands = []
        disc = a * a - 4.0 * b * c
        if np.isfinite(disc) and disc >= 0.0:
            sq = np.sqrt(disc)
            q = 0.5 * (a + np.copysign(sq, a))      # stable pair of roots: q / c and b / q
            if c != 0.0:
                cands.append(q / c)
            if q != 0.0:
                cands.append(b / q)
        cands.append(-f / dw)                       # Newton: always towards the root (f is increasing)
        new = None
        for eta in cands[:2] if len(cands) > 1 else cands:
            x = tau + eta
            if np.isfinite(x) and lo < x < hi and (new is None or abs(x - tau) < abs(new - tau)):
                new = x
        if new is None:
            x = tau + cands[-1]
            if np.isfinite(x) and lo < x < hi:
                new = x
        if new is None:                             # safeguard: bisection, geometric where the bracket spans decades
            if lo > 0.0 and hi > 4.0 * lo:
                new = np.sqrt(lo * hi)
            elif hi < 0.0 and lo < 4.0 * hi:
                new = -np.sqrt(lo * hi)
            elif lo == 0.0:
                new = hi / 16.0
            elif hi == 0.0:
                new = lo / 16.0
            else:
                new = 0.5 * (lo + hi)
        tau = new
    return org, tau


# ------------------------------------------------------------------ one merge
def merge(D, Q, lo, mid, hi, beta, stats=None):
    """Merge the eigen-decompositions of the two children [lo, mid) and [mid, hi) (eigenvalues D[lo:hi], eigenvectors
    Q[lo:hi, lo:hi] block diagonal) across the coupling beta = e[mid - 1], in place."""
    nn = hi - lo
    rho = abs(beta)
    sgn = 1.0 if beta >= 0 else -1.0
    d = D[lo:hi].copy()
    z = np.concatenate([Q[mid - 1, lo:mid], sgn * Q[mid, mid:hi]])
    # normalise: z has norm sqrt(2)
    z = z / np.sqrt(2.0)
    rho = 2.0 * rho
    Qn = Q[lo:hi, lo:hi]
    dmax, zmax = np.max(np.abs(d)), np.max(np.abs(z))
    tol = 8.0 * EPS * max(dmax, zmax)
    order = np.lexsort((np.arange(nn), d))          # ascending d, ties by index
    if rho * zmax <= tol:
        return                                       # nothing couples: D and Q stand
    # deflation scan in sorted order
    keep = []                                        # local column indices that stay in the secular problem
    pj = -1
    rots = []
    for j in order:
        if rho * abs(z[j]) <= tol:
            continue                                 # deflated as is
        if pj < 0:
            pj = j
            continue
        s_, c_ = z[pj], z[j]
        tau = np.hypot(c_, s_)
        t = d[j] - d[pj]
        c_, s_ = c_ / tau, -s_ / tau
        if abs(t * c_ * s_) <= tol:
            # rotate columns (pj, j): z[pj] becomes 0 (deflated), z[j] = tau
            z[j], z[pj] = tau, 0.0
            rots.append((pj, j, c_, s_))
            dpj, dj = d[pj], d[j]
            d[pj] = dpj * c_ * c_ + dj * s_ * s_
            d[j] = dpj * s_ * s_ + dj * c_ * c_
            pj = j
        else:
            keep.append(pj)
            pj = j
    keep.append(pj)
    for (a_, b_, c_, s_) in rots:                    # in scan order, each over all rows
        qa, qb = Qn[:, a_].copy(), Qn[:, b_].copy()
        Qn[:, a_] = c_ * qa + s_ * qb
        Qn[:, b_] = -s_ * qa + c_ * qb
    keep = np.array(keep, dtype=int)
    K = len(keep)
    if stats is not None:
        stats.append((nn, K, len(rots)))
    # the kept poles stay strictly increasing: a rotated pair's new value lies between the two old ones
    dl, w = d[keep], z[keep]
    assert np.all(np.diff(dl) > 0.0)
    z2 = w * w
    org = np.zeros(K, dtype=int)
    tau = np.zeros(K)
    for i in range(K):
        org[i], tau[i] = secular_root(i, dl, z2, rho)
    # differences dl_i - lam_j, computed through the shifted variable
    diff = (dl[:, None] - dl[org][None, :]) - tau[None, :]
    # Gu-Eisenstat: z_hat_i^2 = prod_j (lam_j - dl_i) / prod_{j != i} (dl_j - dl_i)
    zhat = np.empty(K)
    for i in range(K):
        num = -diff[i, :]                            # lam_j - dl_i
        den = dl - dl[i]
        den[i] = 1.0
        p = num[i]
        for j in range(K):
            if j != i:
                p *= num[j] / den[j]
        zhat[i] = np.copysign(np.sqrt(abs(p)), w[i])
    S = zhat[:, None] / diff
    S /= np.linalg.norm(S, axis=0)[None, :]
    Qk = Qn[:, keep] @ S
    Qn[:, keep] = Qk
    d[keep] = dl[org] + tau
    D[lo:hi] = d


def pair_2x2(D, Q, lo, e):
    """[[a, e], [e, c]] in closed form (the arithmetic of LAPACK's dlaev2); a, c = the pair's diagonal with its own coupling
    given back (the tearing took |e| off both)."""
    a, c, b = D[lo] + abs(e), D[lo + 1] + abs(e), e
    sm, df = a + c, a - c
    adf, tb = abs(df), b + b
    ab = abs(tb)
    acmx, acmn = (a, c) if abs(a) > abs(c) else (c, a)
    if adf > ab:
        rt = adf * np.sqrt(1.0 + (ab / adf) ** 2)
    elif adf < ab:
        rt = ab * np.sqrt(1.0 + (adf / ab) ** 2)
    else:
        rt = ab * np.sqrt(2.0)
    if sm < 0.0:
        rt1, sgn1 = 0.5 * (sm - rt), -1
        rt2 = (acmx / rt1) * acmn - (b / rt1) * b
    elif sm > 0.0:
        rt1, sgn1 = 0.5 * (sm + rt), 1
        rt2 = (acmx / rt1) * acmn - (b / rt1) * b
    else:
        rt1, rt2, sgn1 = 0.5 * rt, -0.5 * rt, 1
    if df >= 0.0:
        cs, sgn2 = df + rt, 1
    else:
        cs, sgn2 = df - rt, -1
    if abs(cs) > ab:
        ct = -tb / cs
        sn1 = 1.0 / np.sqrt(1.0 + ct * ct)
        cs1 = ct * sn1
    elif ab == 0.0:
        cs1, sn1 = 1.0, 0.0
    else:
        tn = -cs / tb
        cs1 = 1.0 / np.sqrt(1.0 + tn * tn)
        sn1 = tn * cs1
    if sgn1 == sgn2:
        cs1, sn1 = -sn1, cs1
    D[lo], D[lo + 1] = rt1, rt2
    Q[lo:lo + 2, lo:lo + 2] = [[cs1, -sn1], [sn1, cs1]]


def dc_tridiagonal(d, e, stats=None):
    """Eigen-decomposition of tridiag(d, e) by divide and conquer down to 1 x 1 leaves.  Returns (lam, Z) unsorted."""
    n = len(d)
    D = np.array(d, dtype=np.float64)
    e = np.array(e, dtype=np.float64)
    # tearing, all couplings at once: every boundary is the split point of exactly one tree node
    for i in range(n - 1):
        D[i] -= abs(e[i])
        D[i + 1] -= abs(e[i])
    Q = np.eye(n)
    # the tree in closed form (what the kernel evaluates per thread): node i of level L covers
    # [i n // 2^L, (i + 1) n // 2^L), split at (2 i + 1) n // 2^(L + 1); merges run bottom up
    levels = 0
    while (1 << levels) < n:
        levels += 1
    for L in range(levels - 1, -1, -1):
        for i in range(1 << L):
            lo, hi = (i * n) >> L, ((i + 1) * n) >> L
            if hi - lo >= 2:
                mid = ((2 * i + 1) * n) >> (L + 1)
                assert lo < mid < hi
                if L == levels - 1:
                    assert hi - lo == 2              # the deepest level only pairs up 1 x 1 leaves: closed form
                    pair_2x2(D, Q, lo, e[lo])
                else:
                    merge(D, Q, lo, mid, hi, e[mid - 1], stats)
    return D, Q


def eigh_dc(T, sort_by_abs=False, stats=None):
    """np.linalg.eigh(T) with eigenvalues DEscending (or by |d|): the interface of hfmi_sym_eig_small."""
    T = np.asarray(T, dtype=np.float64)
    n = T.shape[0]
    if n == 1:
        return T[0, :1].copy(), np.ones((1, 1))
    # power-of-two scaling to max |entry| in [1, 2) (exact both ways), as the kernel does while loading
    amax = float(np.max(np.abs(T)))
    sexp = int(np.floor(np.log2(amax))) if amax > 0.0 and np.isfinite(amax) else 0
    d, e, V, tau = tridiagonalize(np.ldexp(T, -sexp))
    lam, Z = dc_tridiagonal(d, e, stats)
    lam = np.ldexp(lam, sexp)
    W = back_transform(V, tau, Z)
    key = -np.abs(lam) if sort_by_abs else -lam
    order = np.argsort(key, kind="stable")
    return lam[order], W[:, order]
