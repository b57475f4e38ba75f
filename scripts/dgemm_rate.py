"""TFLOP/s of the eigensolver's general fp64 MFMA product (k_dgemm_p / k_dgemm, hfmi_eig_blocked.hip) on the shapes the solver runs:
python scripts/dgemm_rate.py [--old]   (--old: HFMI_EIG_GEMM=0, the 64 x 64 kernel of round 5 everywhere)."""
import os
import sys

if "--old" in sys.argv:
    os.environ["HFMI_EIG_GEMM"] = "0"
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import hippyflow_amd as hf  # noqa: E402

ctx = hf.Context.default()
rng = np.random.default_rng(0)
shapes = [("NN 4096^3 (Q S of a top merge)", 4096, 4096, 4096, False, False),
          ("NT 4096 x 4096 x 128 (rank-2k trailing update)", 4096, 4096, 128, False, True),
          ("NT 8192 x 8192 x 128", 8192, 8192, 128, False, True),
          ("TN 256 x 8192 x 8192 (V^T Z of a block reflector)", 256, 8192, 8192, True, False),
          ("NN 8192 x 8192 x 256 (Z -= Y W)", 8192, 8192, 256, False, False),
          ("TN 256 x 4096 x 4096", 256, 4096, 4096, True, False),
          ("NN 4096 x 4096 x 256", 4096, 4096, 256, False, False),
          ("NN 2048 x 2048 x 256", 2048, 2048, 256, False, False),
          ("TN 256 x 256 x 8192 (panel Gram)", 256, 256, 8192, True, False)]
for name, M, N, K, ta, tb in shapes:
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    _, ms = ctx.bench_dgemm(A, B, ta=ta, tb=tb, reps=5, want_c=False)
    print("%-52s %8.3f ms  %6.1f TFLOP/s" % (name, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12), flush=True)
