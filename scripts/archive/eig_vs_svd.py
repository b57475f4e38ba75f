"""One-workgroup two-sided Jacobi eigensolve of T against one-sided Jacobi SVD of its Cholesky factor (timing only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hippyflow_amd as hf
ctx = hf.Context.default()
rng = np.random.default_rng(0)
for k in (74, 84, 138):
    # a Rayleigh-quotient-like matrix: decaying spectrum with a flat noise floor (config 4 with its noise term)
    Q, _ = np.linalg.qr(rng.standard_normal((k, k)))
    lam = np.maximum(100.0 * np.exp(-0.12 * np.arange(k)), 0.05 + 0.04 * rng.random(k))
    T = (Q * lam) @ Q.T
    R = np.linalg.cholesky(T).T
    for name, fn in (("eig", lambda: hf.sym_eig_small(T)), ("svd(R)", lambda: hf.svd_small(R))):
        fn(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): out = fn()
        ctx.synchronize()
        print(k, name, "%.3f ms per call (includes host round trips)" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
    d, V = hf.sym_eig_small(T); U, s, W = hf.svd_small(R)
    print("   eig err", np.abs(np.sort(d) - np.sort(lam)).max(), " svd err", np.abs(np.sort(s**2) - np.sort(lam)).max())
