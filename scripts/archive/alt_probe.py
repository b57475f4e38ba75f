"""Why do the big contractions take 5 % longer inside a solve than in the kernel A/B harnesses?  Shard-size shapes of
config 4 (m = 6400, k = 74, N = 2e5), per-launch HIP events from the library's profiler.  Finding (r01i): neither the
neighbouring kernel nor the data matters (workload Jacobians = N(0,1) entries, alternating = same-kind); what matters is
whether the launches are back to back.  hfmi_bench_tsgemm_nn uploads its small matrix (a host synchronisation, ~0.1 ms of
idle GPU) before every launch and then runs 3.11-3.17 ms; asynchronous back-to-back launches -- the situation inside a
solve -- run 3.31-3.33 ms: the kernels are power-limited and an idle gap buys clock."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
ctx = hf.Context.default()
N, m, k = 200000, 6400, 74
B = hf.MultiVector(N, k); Y = hf.MultiVector(N, k)
hf.parRandom.normal(1.0, B)
if len(sys.argv) > 1 and sys.argv[1] == "workload":      # the synthetic Jacobians of the bench instead of N(0,1) entries
    from hippyflow_amd import workloads
    A = workloads.as_workload(N, 64, q=100, latent=100, rate=0.06, seed=4, first_sample=0, ns_total=512).J
else:
    A = hf.MultiVector(N, m)
    hf.parRandom.normal(1.0, A)
S = np.random.default_rng(0).standard_normal((m, k))
ms = C.c_double(0)
def tn(): L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 0, None, None)
def nn(): L.call("hfmi_bench_tsgemm_nn", A.handle, L.ptr(S), Y.handle, 0, None)
for name, seq in (("nn only", [nn] * 8), ("tn only", [tn] * 8), ("tn,nn alternating", [tn, nn] * 8), ("tn,tn,nn", [tn, tn, nn] * 5)):
    for f in seq[:3]: f()
    ctx.synchronize()
    ctx.profile_begin()
    for f in seq: f()
    rec = ctx.profile_end()
    print(name, "  ".join("%s m=%d: %.3f ms x%d" % (r["kernel"][9:], r["m"], r["ms"] / r["launches"], r["launches"]) for r in rec))
