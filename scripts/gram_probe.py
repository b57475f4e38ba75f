"""The symmetric Gram product Q^T Q of the orthogonalisation (k_tsgemm_ss, one operand staged once, 45 of 81 tiles) at config 3's and
config 4's shapes with the library named by HFMI_LIB: median of seven 10-launch batches.  A/B partner of the -DSS_PROBE=2 timing probe
(the second fragment of every MFMA pair not read from LDS): what a blocked tile assignment could gain at most."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import hippyflow_amd as hf  # noqa: E402
from hippyflow_amd import _lib as L  # noqa: E402

hf.Context.default()
tag = os.path.basename(os.environ.get("HFMI_LIB", "libhfmi.so"))
for N, k in ((500000, 138), (200000, 74), (100000, 84)):
    Q = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, Q)
    reps = []
    for _ in range(7):
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_tn", Q.handle, Q.handle, 0, 10, None, C.byref(ms))
        reps.append(ms.value)
    t = float(np.median(reps)) * 1e-3
    fl, by = float(N) * k * (k + 1), 8.0 * N * k
    print("%-18s N=%-7d k=%-4d %.4f ms  %.1f TF (%.3f of 78.6)  %.2f TB/s" % (tag, N, k, t * 1e3, fl / t / 1e12, fl / t / 78.6e12, by / t / 1e12), flush=True)
    del Q
