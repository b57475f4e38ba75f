"""Gram-type contractions on the double-pass path (Q^T Q, Q^T (A Q)) at the BASELINE shapes: tsgemm_ss vs tsgemm_tn."""
import ctypes as C, json, sys
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
rows = []
for N, k in ((200000, 74), (1000000, 138), (263169, 110), (16641, 40)):
    Q = hf.MultiVector(N, k); hf.parRandom.normal(1.0, Q)
    Y = hf.MultiVector(N, k); hf.parRandom.normal(1.0, Y)
    for name, A, B in (("Q^T Q", Q, Q), ("Q^T Y", Q, Y)):
        r = {"N": N, "k": k, "product": name}
        for ss in (1, 0):
            L.call("hfmi_tuning_set", b"ss", ss)
            ms = C.c_double(0)
            L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 20, None, C.byref(ms))
            r["ss_ms" if ss else "tn_ms"] = ms.value
        L.call("hfmi_tuning_set", b"ss", 1)
        ops = 1 if A is B else 2
        r["ss_GBs"] = 8.0 * N * k * ops / (r["ss_ms"] * 1e-3) / 1e9
        r["ss_TFLOPs"] = 2.0 * N * k * k / (r["ss_ms"] * 1e-3) / 1e12
        rows.append(r); print(r, flush=True)
json.dump(rows, open("gpurun_out/gram_point.json", "w"), indent=1)
