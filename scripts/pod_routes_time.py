"""The deterministic POD (PODProjectorFromData.construct_subspace, 'hep') in its two exact forms -- the reference's n x n Gram problem and
the N x N state-dimension form -- on shapes where the state dimension is the small one (the POD of an output over a training set,
dataGenerator.py:278-279): wall time of each and the agreement of their eigenvalues.   python scripts/pod_routes_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf  # noqa: E402
from hippyflow_amd import workloads  # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

threadpool_limits(limits=8)
rng = np.random.default_rng(1)
for n, nx, ny, r in ((2000, 10, 10, 20), (8300, 30, 20, 20), (16000, 20, 20, 50), (16384, 64, 32, 50)):
    N = nx * ny
    M = workloads.grid_mass_matrix(nx, ny)
    K = min(N, 150)
    W0, _ = np.linalg.qr(rng.standard_normal((N, K)))
    u = (rng.standard_normal((n, K)) * 0.95 ** np.arange(K)) @ W0.T + 0.1
    res = {}
    for form in (False, True):
        pod = hf.PODProjectorFromData(M_output=M)
        pod.prefer_state_dimension = form
        pod.construct_subspace(u.copy(), r, shifted=True, method="hep")
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            d, phi, Mphi, shift = pod.construct_subspace(u.copy(), r, shifted=True, method="hep")
            ts.append(time.perf_counter() - t0)
        res[form] = (min(ts), d)
    print("n = %5d snapshots, N = %4d, r = %2d:  n x n form %8.1f ms   N x N form %7.1f ms   max rel. eigenvalue difference %.1e" % (
        n, N, r, 1e3 * res[False][0], 1e3 * res[True][0], np.abs(res[False][1] - res[True][1]).max() / res[False][1][0]), flush=True)
