"""Interleaved A/B timing of the MFMA kernel variants (waves 4|8, rem4 0|1) in ONE process on fixed shapes."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
ctx = hf.Context.default()
shapes = [("as  tn", 12800, 74, 200000), ("as  tn", 25600, 74, 200000), ("kle tn", 25000, 84, 100000), ("pod tn", 2048, 138, 500000), ("as shard tn", 6400, 74, 200000),
          ("k=36 tn", 8192, 36, 200000), ("k=100 tn", 8192, 100, 200000), ("k=168 tn", 4096, 168, 200000)]
variants = [(8, 0), (8, 1)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, m, k, N in shapes:
    A = hf.MultiVector(N, m); B = hf.MultiVector(N, k)
    rnd = hf.parRandom; rnd.normal(1.0, A); rnd.normal(1.0, B)
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            L.call("hfmi_tuning_set", b"waves", v[0]); L.call("hfmi_tuning_set", b"rem4", v[1])
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 3, None, C.byref(ms))
            res[v].append(ms.value)
    fl = 2.0 * N * m * k
    print(name, (m, k, N), "  ".join("w%d/rem4=%d: %.3f ms %.1f TF" % (v[0], v[1], np.median(t), fl / np.median(t) / 1e9) for v, t in res.items()))
    # nn: Y = A S with the same A (m vectors) -> r = k columns
    S = np.random.default_rng(0).standard_normal((m, k))
    Y = hf.MultiVector(N, k)
    resn = {w: [] for w in (0, 1)}
    for r in range(rounds):
        for w in (0, 1):
            L.call("hfmi_tuning_set", b"rem4", w)
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 3, C.byref(ms))
            resn[w].append(ms.value)
    print(name.replace("tn", "nn"), "  ".join("rem4=%d: %.3f ms %.1f TF" % (w, np.median(t), fl / np.median(t) / 1e9) for w, t in resn.items()))
    del A, B, Y
