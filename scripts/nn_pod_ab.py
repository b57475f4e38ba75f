"""A/B of the tsgemm_nn knobs on config 3's shape (m = 2048 snapshots, r = 138 columns, N = 5e5 rows)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
m, r, N = 2048, 138, 500000
A = hf.MultiVector(N, m); Y = hf.MultiVector(N, r)
hf.parRandom.normal(1.0, A)
S = np.random.default_rng(0).standard_normal((m, r))
def t(reps=5):
    out = []
    for _ in range(reps):
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 3, C.byref(ms))
        out.append(ms.value)
    return float(np.median(out))
fl = 2.0 * N * m * r
for rem4 in (1, 0):
    for waves in (8, 4):
        for tt in (0, 1, 2, 3):
            L.call("hfmi_tuning_set", b"rem4", rem4); L.call("hfmi_tuning_set", b"nn_waves", waves); L.call("hfmi_tuning_set", b"nn_tt", tt)
            ms = t()
            print("rem4 %d waves %d nn_tt %d: %.3f ms %.1f TF" % (rem4, waves, tt, ms, fl / ms / 1e9), flush=True)
