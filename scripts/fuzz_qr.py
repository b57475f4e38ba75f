"""Random-size check of orthogonalize() (Cholesky-QR on the device) against the thin QR with positive diagonal (unique)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = (0.0, 0.0)
for it in range(ncase):
    k = int(rng.integers(1, 257))
    N = int(rng.choice([k, k + 1, 300, 1000, 4225, 20000])) if rng.random() < 0.8 else int(rng.integers(k, 5000))
    N = max(N, k)
    cond = float(10.0 ** rng.uniform(0, 8))
    Z = rng.standard_normal((N, k)) @ np.diag(np.logspace(0, -np.log10(cond), k)) @ np.linalg.qr(rng.standard_normal((k, k)))[0]
    Q = hf.MultiVector.from_dense(Z)
    R = Q.orthogonalize()
    Qd = Q.to_dense()
    o = np.linalg.norm(Qd.T @ Qd - np.eye(k)) / np.sqrt(k)
    rec = np.linalg.norm(Qd @ R - Z) / np.linalg.norm(Z)
    tri = np.allclose(np.tril(R, -1), 0) and np.all(np.diag(R) > 0)
    worst = (max(worst[0], o), max(worst[1], rec))
    if o > 1e-12 or rec > 1e-12 or not tri:
        print("FAIL", (N, k, cond), o, rec, tri, "passes", Q.last_qr_passes); sys.exit(1)
print("qr fuzz ok: %d cases, worst orthonormality %.2e, worst reconstruction %.2e" % (ncase, worst[0], worst[1]))
