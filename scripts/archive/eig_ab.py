"""A/B of the two Rayleigh-Ritz eigensolvers through the C ABI: correctness against numpy and the wall time of
hfmi_sym_eig_small (upload + kernels + read-back; the kernels alone are in the rocprofv3 kernel trace of this script).

    python scripts/eig_ab.py [k ...]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hippyflow_amd as hf  # noqa: E402


def main():
    ks = [int(a) for a in sys.argv[1:]] or [30, 74, 84, 138, 148, 192, 200, 256]
    ctx = hf.Context.default()
    for k in ks:
        rng = np.random.default_rng(k)
        J = rng.standard_normal((4 * k, k)) * np.exp(-0.02 * np.arange(k))
        T = J.T @ J
        w = np.linalg.eigvalsh(T)[::-1]
        t0 = time.perf_counter()
        for _ in range(20):
            np.linalg.eigh(T)
        t_np = (time.perf_counter() - t0) / 20
        line = "k=%3d numpy.eigh %.3f ms |" % (k, 1e3 * t_np)
        for method in ("dc", "jacobi"):
            d, V = hf.sym_eig_small(T, method=method)
            err = np.max(np.abs(d - w)) / w[0]
            orth = np.linalg.norm(V.T @ V - np.eye(k))
            res = np.linalg.norm(T @ V - V * d) / w[0]
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                hf.sym_eig_small(T, method=method)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / 20
            line += " %s: %.3f ms (eig %.1e orth %.1e res %.1e) |" % (method, 1e3 * dt, err, orth, res)
        print(line, flush=True)


if __name__ == "__main__":
    main()
