"""A/B harness for tsgemm_ss knobs: run as  HFMI_LIB=<variant.so> python scripts/ss_ab.py <percu>"""
import ctypes as C, sys
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
percu = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L.call("hfmi_tuning_set", b"ss_percu", percu)
out = []
for N, m, k, same in ((1000000, 8, 138, 0), (1000000, 32, 138, 0), (1000000, 64, 138, 0), (1000000, 138, 138, 0),
                      (1000000, 138, 138, 1), (200000, 74, 74, 0), (200000, 74, 74, 1), (263169, 110, 110, 0)):
    A = hf.MultiVector(N, m); hf.parRandom.normal(1.0, A)
    B = A if same else hf.MultiVector(N, k)
    if not same: hf.parRandom.normal(1.0, B)
    ms = C.c_double(0)
    L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 20, None, C.byref(ms))
    ctx = hf.Context.default(); ctx.profile_begin()
    for _ in range(10): A.dot_mv(B)
    rec = ctx.profile_end()
    kms = sum(r["ms"] for r in rec) / max(1, sum(r["launches"] for r in rec))
    out.append("%dx%d%s N=%d: %.4f (kernel %.4f)" % (m, k, "s" if same else "", N, ms.value, kms))
    del A, B
print("percu=%d | " % percu + " | ".join(out))
