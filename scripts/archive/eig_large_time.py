import sys, time
import numpy as np
sys.path.insert(0, ".")
import hippyflow_amd as hf
from hippyflow_amd import randomized as R
rng = np.random.default_rng(0)
for n in (300, 512, 640, 1024, 2048):
    Z = rng.standard_normal((n + 50, n)) * np.exp(-0.01 * np.arange(n))
    T = Z.T @ Z
    t0 = time.perf_counter()
    d, V = hf.sym_eig_small(T)
    t = time.perf_counter() - t0
    w = np.linalg.eigvalsh(T)[::-1]
    t1 = time.perf_counter(); np.linalg.eigh(T); t2 = time.perf_counter() - t1
    print("n=%4d  device %.3f s   numpy eigh (all host threads) %.3f s   rel err %.1e  orth %.1e" % (n, t, t2, np.abs(d - w).max() / w[0], np.abs(V.T @ V - np.eye(n)).max()), flush=True)
