"""Accuracy and time of the small symmetric eigensolver (run twice: HFMI_JACOBI_DB=0 / 1)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
ctx = hf.Context.default()
rng = np.random.default_rng(0)
for k in (2, 3, 7, 30, 74, 75, 84, 110, 137, 138):
    Qm = np.linalg.qr(rng.standard_normal((k, k)))[0]
    T = (Qm * np.exp(-0.1 * np.arange(k))) @ Qm.T
    T = 0.5 * (T + T.T)
    d, V = hf.sym_eig_small(T)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): hf.sym_eig_small(T)
    ctx.synchronize()
    t1 = time.perf_counter()
    dref = np.sort(np.linalg.eigvalsh(T))[::-1]
    res = np.linalg.norm(T @ V - V * d) / np.linalg.norm(T)
    orth = np.linalg.norm(V.T @ V - np.eye(k))
    print("DB=%s k=%3d  %.3f ms/call  eig err %.2e  residual %.2e  orth %.2e" % (os.environ.get("HFMI_JACOBI_DB", "1"), k, (t1 - t0) / 10 * 1e3, np.max(np.abs(d - dref) / np.abs(dref)), res, orth))
