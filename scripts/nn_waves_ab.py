"""In-process A/B of tsgemm_nn with one (nn_waves=4) or two (nn_waves=8) waves per SIMD on fixed shapes."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
shapes = [("pod", 2048, 138, 500000), ("as", 12800, 74, 200000), ("kle-like", 8192, 84, 200000), ("k=100", 8192, 100, 200000), ("k=128", 4096, 128, 400000)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, m, r, N in shapes:
    A = hf.MultiVector(N, m); Y = hf.MultiVector(N, r)
    hf.parRandom.normal(1.0, A)
    S = np.random.default_rng(0).standard_normal((m, r))
    res = {w: [] for w in (4, 8)}
    for it in range(rounds):
        for w in (4, 8):
            L.call("hfmi_tuning_set", b"nn_waves", w)
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 3, C.byref(ms))
            res[w].append(ms.value)
    L.call("hfmi_tuning_set", b"nn_waves", 0)
    fl = 2.0 * N * m * r
    print(name, (m, r, N), "  ".join("waves %d: %.3f ms %.1f TF" % (w, np.median(t), fl / np.median(t) / 1e9) for w, t in res.items()))
    del A, Y
