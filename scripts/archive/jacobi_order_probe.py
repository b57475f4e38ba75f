"""Does sorting the diagonal of T = Q^T A Q before the Jacobi sweeps save sweeps?  Run with HFMI_DEBUG_TIMING=1:
the library prints the sweep count of every eigensolve."""
import sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
rng = np.random.default_rng(0)
for k, decay in ((74, 0.1), (138, 0.1), (138, 0.05), (84, 0.08), (138, 0.02)):
    N = 700
    U0 = np.linalg.qr(rng.standard_normal((N, N)))[0]
    A = (U0 * np.exp(-decay * np.arange(N))) @ U0.T
    Q = np.linalg.qr(A @ rng.standard_normal((N, k)))[0]
    T = Q.T @ A @ Q
    T = 0.5 * (T + T.T)
    p = np.argsort(-np.diag(T))
    print("k=%d decay=%g: unsorted, descending diagonal, ascending diagonal" % (k, decay), file=sys.stderr, flush=True)
    for M in (T, T[np.ix_(p, p)], T[np.ix_(p[::-1], p[::-1])]):
        d, V = hf.sym_eig_small(M)
        assert np.allclose(d, np.linalg.eigvalsh(M)[::-1], rtol=1e-9, atol=1e-14)
