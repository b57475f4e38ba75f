"""The one-workgroup kernels at and beyond the old 139-column limit: a thin QR (Cholesky + triangular inverse) and the
Rayleigh-Ritz eigensolve for k = 138, 148, 200, 256; the per-kernel times are read from the rocprofv3 kernel trace of this
script (scripts/small_k_trace.sh).      python scripts/small_k_point.py [k ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hippyflow_amd as hf  # noqa: E402


def main():
    ks = [int(a) for a in sys.argv[1:]] or [138, 148, 200, 256]
    ctx = hf.Context.default()
    for k in ks:
        N = 20000
        Z = hf.MultiVector(N, k)
        hf.parRandom.reseed(k)
        hf.parRandom.normal(1.0, Z)
        Zd = Z.to_dense() * np.exp(-0.02 * np.arange(k))
        for rep in range(5):
            Q = hf.MultiVector.from_dense(Zd)
            ctx.synchronize()
            t0 = time.perf_counter()
            Q.orthogonalize()
            ctx.synchronize()
            t_qr = time.perf_counter() - t0
        Qd = Q.to_dense()
        defect = np.abs(Qd.T @ Qd - np.eye(k)).max()
        T = Zd.T @ Zd
        for rep in range(5):
            t0 = time.perf_counter()
            d, V = hf.sym_eig_small(T)
            t_eig = time.perf_counter() - t0
        w = np.linalg.eigvalsh(T)[::-1]
        print("k=%3d  orthogonalize %.3f ms (defect %.1e)  sym_eig_small %.3f ms (eig err %.1e)"
              % (k, 1e3 * t_qr, defect, 1e3 * t_eig, np.max(np.abs(d - w)) / w[0]), flush=True)


if __name__ == "__main__":
    main()
