"""tsgemm_nn on fixed shapes with the library named by HFMI_LIB (A/B of builds: run once per library, interleaved by the caller).
    HFMI_LIB=hippyflow_amd/build/libhfmi_p0.so python scripts/nn_lib_ab.py [rounds]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
shapes = [("pod", 2048, 138, 500000), ("as", 12800, 74, 200000), ("kle-like", 8192, 84, 200000), ("k=100", 8192, 100, 200000),
          ("k=128", 4096, 128, 400000), ("k=240", 2048, 240, 300000)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tag = os.path.basename(os.environ.get("HFMI_LIB", "libhfmi.so"))
for name, m, r, N in shapes:
    A = hf.MultiVector(N, m); Y = hf.MultiVector(N, r)
    hf.parRandom.normal(1.0, A)
    S = np.random.default_rng(0).standard_normal((m, r))
    ts = []
    for it in range(rounds):
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 3, C.byref(ms))
        ts.append(ms.value)
    ref = A.to_dense()[:4096] @ S if it == rounds - 1 else None
    err = np.abs(Y.to_dense()[:4096] - ref).max() / np.abs(ref).max()
    fl = 2.0 * N * m * r
    print("%-16s %-9s %-22s %.3f ms %.1f TF  (rel err of the first 4096 rows %.1e)" % (tag, name, (m, r, N), np.median(ts), fl / np.median(ts) / 1e9, err), flush=True)
    del A, Y
