"""Kernel point N = 1e6, k = 138: tsgemm_ss against tsgemm_tn on the same skinny shapes (knob "ss")."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
for N, m, k in ((1000000, 48, 138), (1000000, 64, 138), (1000000, 96, 138), (1000000, 138, 138), (1000000, 160, 128), (500000, 138, 138)):
    A = hf.MultiVector(N, m); hf.parRandom.normal(1.0, A)
    B = hf.MultiVector(N, k); hf.parRandom.normal(1.0, B)
    res = {}
    for v in (1, 0):
        L.call("hfmi_tuning_set", b"ss", v)
        ts = []
        for rep in range(4):
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 8, None, C.byref(ms))
            ts.append(ms.value)
        res[v] = np.median(ts)
    L.call("hfmi_tuning_set", b"ss", 1)
    fl = 2.0 * N * m * k
    print("%3d x %3d N=%d: ss %.4f ms %.1f TF | tn %.4f ms %.1f TF" % (m, k, N, res[1], fl / res[1] / 1e9, res[0], fl / res[0] / 1e9))
    del A, B
