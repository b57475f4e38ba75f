"""Ridge shapes of the blocked skinny kernel (G = X^T Omega, N = 1e6, k = 138, n = 48 / 64 / 96 snapshots) with the library named by
HFMI_LIB: median of seven 10-launch batches per shape.  Used for the round-5 A/B of the timing probes (-DSS_PROBE=1 / 3,
scripts/build_variant.sh) against the product build; per-kernel counters come from rocprofv3 --pmc passes over this script."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import hippyflow_amd as hf  # noqa: E402
from hippyflow_amd import _lib as L  # noqa: E402

hf.Context.default()
N, k = 1000000, 138
tag = os.path.basename(os.environ.get("HFMI_LIB", "libhfmi.so"))
W = hf.MultiVector(N, k)
hf.parRandom.normal(1.0, W)
for n in (48, 64, 96, 138):
    X = hf.MultiVector(N, n)
    hf.parRandom.normal(1.0, X)
    reps = []
    for _ in range(7):
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_tn", W.handle, X.handle, 0, 10, None, C.byref(ms))     # (W^T X)^T: the orientation the dispatcher picks
        reps.append(ms.value)
    t = float(np.median(reps)) * 1e-3
    by, fl = 8.0 * (N * n + N * k + n * k), 2.0 * N * n * k
    print("%-18s n=%-4d %.4f ms  hbm %.3f of 8 TB/s  mfma %.3f of 78.6 TF" % (tag, n, t * 1e3, by / t / 8e12, fl / t / 78.6e12), flush=True)
    del X
