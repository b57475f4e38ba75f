"""tsgemm_tn: the uniform split of the reduction axis against the hybrid plan (whole rounds of row blocks coarsely split, the
blocks beyond them finely split).  Shapes: config 4 (m = 51200), config 2 (m = 1e5), config 4's 2-rank share, config 3 and the
shard (no whole round: the plans coincide).  Correctness of both against numpy on a reduced size."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L

# correctness on a shape that takes the hybrid plan with a ragged last block
rng = np.random.default_rng(0)
for (m, k, N) in ((100003, 9, 2048), (51200 + 77, 20, 1024)):
    A = hf.MultiVector(N, m); B = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, A); hf.parRandom.normal(1.0, B)
    res = []
    for hyb in (0, 1):
        L.call("hfmi_tuning_set", b"tn_hybrid", hyb)
        res.append(A.dot_mv(B))
    L.call("hfmi_tuning_set", b"tn_hybrid", 1)
    ref = A.to_dense().T @ B.to_dense()
    print("check", (m, k, N), "uniform %.2e  hybrid %.2e  (max abs err / max abs)" % tuple(np.abs(r - ref).max() / np.abs(ref).max() for r in res), flush=True)
    del A, B
for name, m, k, N in (("as full", 51200, 74, 200000), ("kle", 100000, 84, 100000), ("as 2 ranks", 25600, 74, 200000), ("pod", 2048, 138, 500000),
                      ("as shard", 6400, 74, 200000)):
    A = hf.MultiVector(N, m); B = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, A); hf.parRandom.normal(1.0, B)
    out = []
    for hyb in (0, 1, 0, 1):
        L.call("hfmi_tuning_set", b"tn_hybrid", hyb)
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 4, None, C.byref(ms))
        out.append("hybrid %d: %.3f ms %.1f TF" % (hyb, ms.value, 2.0 * N * m * k / ms.value / 1e9))
    L.call("hfmi_tuning_set", b"tn_hybrid", 1)
    print(name, (m, k, N), " | ".join(out), flush=True)
    del A, B
