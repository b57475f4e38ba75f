"""tsgemm_ss timings on the kernel point (N = 1e6, k = 138, n snapshots) and the Gram shapes; run under HFMI_LIB=<other build>
for A/B."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
out = []
for N, m, k, same in ((1000000, 8, 138, 0), (1000000, 32, 138, 0), (1000000, 48, 138, 0), (1000000, 64, 138, 0), (1000000, 96, 138, 0),
                      (1000000, 138, 138, 0), (500000, 138, 138, 1), (200000, 74, 74, 1), (100000, 84, 84, 1), (1000000, 150, 130, 0)):
    A = hf.MultiVector(N, m); hf.parRandom.normal(1.0, A)
    B = A if same else hf.MultiVector(N, k)
    if not same: hf.parRandom.normal(1.0, B)
    ts = []
    for rep in range(5):
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 10, None, C.byref(ms))
        ts.append(ms.value)
    t = np.median(ts)
    fl = (N * k * (k + 1.0)) if same else 2.0 * N * m * k
    out.append("%dx%d%s: %.4f ms %.1f TF" % (m, k, "s" if same else "", t, fl / t / 1e9))
    del A, B
print(" | ".join(out))
