"""tsgemm_tn: wave tile height (16-row tiles per wave) vs time for the shard / full shapes of config 4 and config 3.
The automatic choice is the tallest tile the accumulator budget allows; a shorter one can divide m without padding rows
(m = 6400: 17 blocks of 384 rows = 6528, or 25 blocks of 256 rows exactly)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
for name, m, k, N in (("as shard", 6400, 74, 200000), ("as 2 ranks", 25600, 74, 200000), ("pod", 2048, 138, 500000), ("kle", 25000, 84, 100000)):
    A = hf.MultiVector(N, m); B = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, A); hf.parRandom.normal(1.0, B)
    out = []
    for mt in (0, 1, 2, 3, 0, 2, 3):
        L.call("hfmi_tuning_set", b"tn_mt", mt)
        ms = C.c_double(0)
        L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 6, None, C.byref(ms))
        out.append("mt %d: %.3f ms %.1f TF" % (mt, ms.value, 2.0 * N * m * k / ms.value / 1e9))
    L.call("hfmi_tuning_set", b"tn_mt", 0)
    print(name, (m, k, N), " | ".join(out), flush=True)
    del A, B
