"""Where does tsgemm_tn's matrix-pipe idle time go?  probe 1: streamed operand re-reads one address (cache hits),
probe 2: no stage loads / LDS stores / barriers, 3: both.  (Round 2 also measured 4 = staging kept, barrier dropped and 8 = LDS stores before the last iteration, profiles/archive/r02e_tn_probe.txt; the two extra branches made the <2,9> instance spill 40 bytes per lane, so they were taken out again.)  Results are garbage in probe modes; timing only."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
for name, m, k, N in [("as", 12800, 74, 200000), ("kle", 25000, 84, 100000), ("pod", 2048, 138, 500000)]:
    A = hf.MultiVector(N, m); B = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, A); hf.parRandom.normal(1.0, B)
    out = []
    for rep in range(3):
        row = []
        for probe in (0, 1, 2, 3):
            L.call("hfmi_tuning_set", b"probe", probe)
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 3, None, C.byref(ms))
            row.append(ms.value)
        out.append(row)
    L.call("hfmi_tuning_set", b"probe", 0)
    med = np.median(np.array(out), axis=0)
    print(name, (m, k, N), "  ".join("probe%d: %.3f ms %.1f TF" % (p, t, 2.0 * N * m * k / t / 1e9) for p, t in zip((0, 1, 2, 3), med)))
    del A, B
