"""tsgemm_tn: time vs the number of reduction-axis slices for the per-GPU shard of config 4 (m = 6400) and config 3."""
import ctypes as C, sys
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
for name, m, k, N in (("as shard", 6400, 74, 200000), ("pod", 2048, 138, 500000), ("as full/4", 12800, 74, 200000)):
    A = hf.MultiVector(N, m); B = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, A); hf.parRandom.normal(1.0, B)
    out = []
    for ns in (0, 15, 30, 0, 15, 30, 45, 0, 60, 15):
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, ns, 4, None, C.byref(ms))
        out.append("%d: %.3f" % (ns, ms.value))
    print(name, (m, k, N), " | ".join(out), flush=True)
    del A, B
