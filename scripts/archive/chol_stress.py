"""Stress of the blocked Cholesky kernel: random sizes 1..256, random conditioning, thin QR against numpy; repeated runs of the
same input must agree to the last bit (a race between the waves of the one workgroup would show as a flip).
    python scripts/chol_stress.py [cases]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import hippyflow_amd as hf  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(12345)
worst = 0.0
for it in range(cases):
    k = int(rng.integers(1, 257))
    N = 2 * k + int(rng.integers(16, 400))
    decay = float(rng.uniform(0.0, 8.0 / k))
    Z = rng.standard_normal((N, k)) * np.exp(-decay * np.arange(k))
    first = None
    for rep in range(3):
        Q = hf.MultiVector.from_dense(Z)
        R = Q.orthogonalize()
        Qd = Q.to_dense()
        if first is None:
            first = (R.copy(), Qd.copy())
        else:
            assert np.array_equal(R, first[0]) and np.array_equal(Qd, first[1]), "run-to-run difference at k=%d (case %d)" % (k, it)
    Qn, Rn = np.linalg.qr(Z)
    sg = np.sign(np.diag(Rn))
    Rn = Rn * sg[:, None]
    e1 = np.abs(Qd.T @ Qd - np.eye(k)).max()
    e2 = np.abs(R - Rn).max() / np.abs(Rn).max()
    worst = max(worst, e1, e2 / 1e2)
    assert e1 < 1e-13 and e2 < 1e-10, (k, N, decay, e1, e2)
print("%d cases, three runs each bit-identical; worst orthogonality / scaled R error %.2e" % (cases, worst))
