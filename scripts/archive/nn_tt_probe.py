"""tsgemm_nn: wave-tile height A/B (nn_tt = 0 auto, 1 tallest, 2, 3) on the shapes of the three workloads."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
for name, m, r, N in (("pod nn", 2048, 138, 500000), ("pod QR nn", 138, 138, 500000), ("as nn", 12800, 74, 200000), ("as shard nn", 6400, 74, 200000),
                      ("as QR nn", 74, 74, 200000), ("kle QR nn", 84, 84, 100000), ("k=110", 512, 110, 263169)):
    A = hf.MultiVector(N, m); hf.parRandom.normal(1.0, A)
    Y = hf.MultiVector(N, r)
    S = np.random.default_rng(0).standard_normal((m, r))
    res = {}
    for rep in range(3):
        for hyb in (1, 0):
            L.call("hfmi_tuning_set", b"nn_hybrid", hyb)
            for tt in (0, 1, 2):
                L.call("hfmi_tuning_set", b"nn_tt", tt)
                ms = C.c_double(0)
                L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 3, C.byref(ms))
                res.setdefault((hyb, tt), []).append(ms.value)
    L.call("hfmi_tuning_set", b"nn_tt", 0); L.call("hfmi_tuning_set", b"nn_hybrid", 1)
    fl = 2.0 * N * m * r
    print(name, (m, r, N), "  ".join("h%d/tt%d: %.3f ms %.1f TF" % (h, tt, np.median(v), fl / np.median(v) / 1e9) for (h, tt), v in res.items()), flush=True)
    del A, Y
