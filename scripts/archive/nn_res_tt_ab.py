"""A/B of the tile height of the LDS-resident nn product: the table's height (nn_res_tt=1) against one 16-row tile less per wave
(nn_res_tt=2) and the automatic choice by round count (0).  Includes the shapes of the QR / Rayleigh-Ritz steps of configs 2-4."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, m, r, N in [("pod Q R^-1", 138, 138, 500000), ("pod U=QV", 138, 128, 500000), ("as Q R^-1", 74, 74, 200000),
                      ("as U=QV", 74, 64, 200000), ("kle Q R^-1", 84, 84, 100000), ("kle U=QV", 84, 64, 100000), ("k=160", 150, 160, 100000)]:
    A = hf.MultiVector(N, m); Y = hf.MultiVector(N, r)
    hf.parRandom.normal(1.0, A)
    S = np.random.default_rng(0).standard_normal((m, r))
    res = {0: [], 1: [], 2: []}
    for it in range(rounds):
        for v in res:
            L.call("hfmi_tuning_set", b"nn_res_tt", v)
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 5, C.byref(ms))
            res[v].append(ms.value)
    L.call("hfmi_tuning_set", b"nn_res_tt", 0)
    fl = 2.0 * N * m * r
    print("%-12s %s " % (name, (m, r, N)) + "  ".join("tt=%d: %.4f ms %.1f TF" % (v, np.median(t), fl / np.median(t) / 1e9) for v, t in res.items()))
    del A, Y
