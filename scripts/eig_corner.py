"""The Rayleigh-Ritz eigensolve beyond the BASELINE sizes (160 <= k <= 256, where the tridiagonalisation keeps its last rows in
LDS): wall time of hfmi_sym_eig_small per call; per-kernel times come from the rocprofv3 kernel trace of this script.

    python scripts/eig_corner.py [k ...]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hippyflow_amd as hf  # noqa: E402


def main():
    ks = [int(a) for a in sys.argv[1:]] or [160, 192, 224, 256]
    ctx = hf.Context.default()
    for k in ks:
        rng = np.random.default_rng(k)
        J = rng.standard_normal((4 * k, k)) * np.exp(-0.02 * np.arange(k))
        T = J.T @ J
        w = np.linalg.eigvalsh(T)[::-1]
        d, V = hf.sym_eig_small(T, method="dc")
        err = np.max(np.abs(d - w)) / w[0]
        ctx.synchronize()
        times = []
        for _ in range(20):
            t0 = time.perf_counter()
            hf.sym_eig_small(T, method="dc")
            times.append(time.perf_counter() - t0)
        print("k=%3d dc: median %.3f ms, min %.3f ms (eig err %.1e)" % (k, 1e3 * np.median(times), 1e3 * min(times), err), flush=True)


if __name__ == "__main__":
    main()
