"""North-star kernel point: G = X^T Omega (snapshot^T x probe block) at N = 1e6, k = r + p = 138, for several snapshot
counts n -- reports achieved algorithmic GB/s vs the 8 TB/s HBM roof AND TFLOP/s vs the 78.6 TFLOP/s fp64 MFMA roof
and names the binding one."""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, '.')
import hippyflow_amd as hf
from hippyflow_amd import _lib as L
ctx = hf.Context.default()
N, k = 1000000, 138
W = hf.MultiVector(N, k); hf.parRandom.normal(1.0, W)
rows = []
for n in (8, 16, 32, 48, 64, 96, 138, 512, 2048):
    X = hf.MultiVector(N, n); hf.parRandom.normal(1.0, X)
    best = None
    for ss in (1, 0):          # 1: tsgemm_ss where applicable (the default), 0: force tsgemm_tn
        L.call("hfmi_tuning_set", b"ss", ss)
        for orient in ("X^T W", "(W^T X)^T"):
            A, B = (X, W) if orient == "X^T W" else (W, X)
            reps = []
            for _ in range(5):           # median of five 10-launch averages (a single batch right after an idle gap reads 5-13 % slow)
                ms = C.c_double(0)
                L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 10, None, C.byref(ms))
                reps.append(ms.value)
            t_med = float(np.median(reps))
            print("   n=%d ss=%d %s: %.4f ms" % (n, ss, orient, t_med), flush=True)
            if best is None or t_med < best[1]:
                best = (orient + (" [ss]" if ss and n <= 160 else " [tn]"), t_med)
    L.call("hfmi_tuning_set", b"ss", 1)
    t = best[1] * 1e-3
    by, fl = 8.0 * (N * n + N * k + n * k), 2.0 * N * n * k
    ai = fl / by
    rows.append({"n": n, "orientation": best[0], "ms": best[1], "algorithmic_GBs": by / t / 1e9, "hbm_frac_of_8TBs": by / t / 8e12,
                 "TFLOPs": fl / t / 1e12, "mfma_frac_of_78.6": fl / t / 78.6e12, "AI_flop_per_byte": ai, "binding_roof": "mfma" if ai > 9.8 else "hbm"})
    print(rows[-1], flush=True)
    del X
import os
if not os.environ.get("ROCPROFILER_CONFIGURED") and "rocprof" not in os.environ.get("LD_PRELOAD", ""):
    json.dump(rows, open("gpurun_out/kernel_point.json", "w"), indent=1)   # never from a profiled (slowed-down) run
