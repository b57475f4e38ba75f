"""Feasibility estimate ONLY (no kernel): what a level-scheduled device triangular solve would have to do to replace the host
sparse-LU callback of the prior-preconditioned active-subspace step (activeSubspaceProjector.py:447-450: doublePassG with
Rsolver; here R^-1 = A^-1 M_l A^-1, A = M + 0.1 K on the config-4 grid, SuperLU under its default COLAMD ordering).
For L and U of splu(A): nnz, and the number of LEVELS of the dependency graph (row i of a triangular factor can be eliminated one
level after the last row it depends on) -- a level-scheduled SpTRSM runs one device-wide step per level.  Usage:
python scripts/sptrsv_feasibility.py [nx ny] ; prints one JSON object (also used by bench.py --prior as `sptrsv_feasibility`)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def levels_lower(Lcsr):
    """number of dependency levels of a lower-triangular CSR matrix (diagonal included or not)"""
    n = Lcsr.shape[0]
    ip, ix = Lcsr.indptr, Lcsr.indices
    lev = np.zeros(n, dtype=np.int32)
    for i in range(n):
        c = ix[ip[i]:ip[i + 1]]
        c = c[c < i]
        lev[i] = (lev[c].max() + 1) if c.size else 0
    return int(lev.max()) + 1, lev


def estimate(A, nrhs=74, us_per_level=5.0, hbm_tbs=6.3):
    import scipy.sparse.linalg as spla
    t0 = time.perf_counter()
    lu = spla.splu(A.tocsc())
    t_fac = time.perf_counter() - t0
    L, U = lu.L.tocsr(), lu.U.tocsr()
    nl, _ = levels_lower(L)
    # U x = y is solved from the last row up: reverse the order -> a lower-triangular pattern
    n = U.shape[0]
    Ur = U[::-1, ::-1].tocsr()
    nu, _ = levels_lower(Ur)
    nnz = int(L.nnz + U.nnz)
    # one A^-1 = one L sweep + one U sweep; R^-1 = two A^-1; doublePassG applies R^-1 twice per step
    t_levels = 2 * 2 * (nl + nu) * us_per_level * 1e-6                             # 2 applications x 2 A^-1 each x (nl + nu) levels
    bytes_per_solve = 12.0 * nnz + 16.0 * n * nrhs                                 # factors (value + index) + the block in and out
    t_bytes = 2 * 2 * bytes_per_solve / (hbm_tbs * 1e12)
    return {"N": int(n), "nnz_A": int(A.nnz), "nnz_L": int(L.nnz), "nnz_U": int(U.nnz), "fill_ratio": nnz / float(A.nnz),
            "levels_L": nl, "levels_U": nu, "ordering": "SuperLU default (COLAMD)", "host_factor_seconds": t_fac,
            "assumed_us_per_level": us_per_level, "rhs": nrhs,
            "level_scheduled_ms_per_step_latency_part": 1e3 * t_levels,
            "level_scheduled_ms_per_step_byte_part_at_%.1f_TBs" % hbm_tbs: 1e3 * t_bytes,
            "note": "per doublePassG step: 2 applications of R^-1 = 4 solves with A = 4 (L sweep + U sweep); a level costs one "
                    "device-wide step whatever its width"}


if __name__ == "__main__":
    from hippyflow_amd import workloads as W
    nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500, 400)
    A = (W.grid_mass_matrix(nx, ny) + 0.1 * W.grid_stiffness_matrix(nx, ny)).tocsr()
    print(json.dumps(estimate(A), indent=1))
