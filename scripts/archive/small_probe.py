import sys, time, numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
ctx = hf.Context.default()
rng = np.random.default_rng(0)
for k in (74, 138):
    Qm = np.linalg.qr(rng.standard_normal((k, k)))[0]
    T = (Qm * np.exp(-0.1*np.arange(k))) @ Qm.T
    hf.sym_eig_small(T)
    t0 = time.perf_counter()
    for _ in range(5): hf.sym_eig_small(T)
    t1 = time.perf_counter()
    Z = rng.standard_normal((20000, k))
    Q = hf.MultiVector.from_dense(Z); Q.orthogonalize()
    ts = []
    for _ in range(3):
        Q = hf.MultiVector.from_dense(Z); ctx.synchronize()
        a = time.perf_counter(); Q.orthogonalize(0); ctx.synchronize(); ts.append(time.perf_counter()-a)
    print("k=%d eig %.3f ms  qr(N=20000) %.3f ms passes %d" % (k, (t1-t0)/5*1e3, min(ts)*1e3, Q.last_qr_passes))
