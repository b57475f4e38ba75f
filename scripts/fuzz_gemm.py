"""Random-shape check of the three contraction kernels against numpy (dot_mv = tn / ss, MvDSmatMult = nn / nn_res)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150
worst = 0.0
for it in range(ncase):
    N = int(rng.choice([1, 31, 32, 33, 100, 1000, 4225, 9999, 20000, 50001]))
    m = int(rng.choice([1, 2, 7, 15, 16, 17, 30, 64, 74, 84, 100, 129, 138, 160, 200, 255, 256, 300, 700, 2000]))
    k = int(rng.integers(1, 270)) if rng.random() < 0.7 else int(rng.choice([4, 8, 12, 13, 16, 20, 36, 74, 84, 100, 138, 256, 257]))
    if N * (m + k) > 6e7: N = max(1, int(6e7 // (m + k)))
    A = rng.standard_normal((N, m)); B = rng.standard_normal((N, k))
    Am, Bm = hf.MultiVector.from_dense(A), hf.MultiVector.from_dense(B)
    got = Am.dot_mv(Bm)
    ref = A.T @ B
    sc = np.linalg.norm(A, axis=0)[:, None] * np.linalg.norm(B, axis=0)[None, :] + 1e-300
    e1 = np.max(np.abs(got - ref) / sc)
    S = rng.standard_normal((m, k))
    Y = hf.MultiVector(N, k)
    hf.MvDSmatMult(Am, S, Y)
    ref2 = A @ S
    e2 = np.max(np.abs(Y.to_dense() - ref2)) / (np.max(np.abs(ref2)) + 1e-300)
    worst = max(worst, e1, e2)
    if e1 > 1e-12 or e2 > 1e-12 or not np.isfinite(e1 + e2):
        print("FAIL", (N, m, k), e1, e2); sys.exit(1)
print("fuzz ok: %d cases, worst error %.2e" % (ncase, worst))
