"""tsgemm_ss timings on the kernel point (N = 1e6, k = 138, n snapshots) and the Gram shapes; run under HFMI_LIB=<other build>
for A/B."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
import numpy.testing as npt
out = []
for N, m, k, same in ((1000000, 8, 138, 0), (1000000, 32, 138, 0), (1000000, 138, 64, 0), (200000, 74, 74, 0), (100000, 84, 84, 0), (1000000, 48, 138, 0), (1000000, 64, 138, 0), (1000000, 96, 138, 0),
                      (1000000, 138, 138, 0), (500000, 138, 138, 1), (200000, 74, 74, 1), (100000, 84, 84, 1), (1000000, 150, 130, 0)):
    A = hf.MultiVector(N, m); hf.parRandom.normal(1.0, A)
    B = A if same else hf.MultiVector(N, k)
    if not same: hf.parRandom.normal(1.0, B)
    res, Cs = [], []
    for blocked in (0, 2, 1):
        L.call("hfmi_tuning_set", b"ss_blocked", blocked)
        ts = []
        Ch = np.zeros((m, k))
        for rep in range(5):
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 10, L.ptr(Ch), C.byref(ms))
            ts.append(ms.value)
        res.append(np.median(ts)); Cs.append(Ch)
    err = max(np.abs(Cs[0] - Cs[1]).max(), np.abs(Cs[0] - Cs[2]).max()) / np.abs(Cs[0]).max()
    fl = (N * k * (k + 1.0)) if same else 2.0 * N * m * k
    out.append("%dx%d%s: round-robin %.4f ms %.1f TF | blocked %.4f ms %.1f TF | blocked + pipelined stages %.4f ms %.1f TF | diff %.1e"
               % (m, k, "s" if same else "", res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9, res[2], fl / res[2] / 1e9, err))
    del A, B
L.call("hfmi_tuning_set", b"ss_blocked", 1)
print("\n".join(out))
