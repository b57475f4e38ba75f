#!/usr/bin/env python3
"""Generate tests/golden/pod_from_data_8300.npz by running the REFERENCE's own ``PODProjectorFromData.construct_subspace``
(modeling/PODProjector.py:699-852, method 'hep': la.eigh of the n x n matrix at :821) on MORE THAN 8192 snapshots with a SLOWLY
decaying spectrum -- the case dataGenerator.py:278-279 produces when it hands the whole training set over, and the one in which a
randomized solve with a few extra probe columns is NOT the reference's exact result.  The device path serves it with the exact
whole-GPU eigensolver (n <= 16384).

Run ONLY in the authoring container (needs /root/reference; about two minutes per la.eigh(8300 x 8300) on 8 cores):

    python tests/golden/make_pod_huge_golden.py

The 8300 x 600 snapshot matrix is NOT stored: it is rebuilt from a seed by ``snapshots()`` below with integer arithmetic only
(numpy's integers() stream and an int64 matrix product: exact on every machine), scaled by a power of two.  Stored: the seed,
the sizes, the mass matrix and the reference's outputs (d, phi, shift)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def snapshots(seed, n, N, K):
    """n x N snapshot matrix of rank K + noise floor, singular values decaying like 0.95^k: integers times 2^-10."""
    rng = np.random.default_rng(seed)
    A = rng.integers(-2, 3, size=(n, K), dtype=np.int64)
    B = rng.integers(-2, 3, size=(K, N), dtype=np.int64)
    w = np.rint(2000.0 * 0.95 ** np.arange(K)).astype(np.int64)          # (exact: a table of K integers)
    U = (A * w[None, :]) @ B + rng.integers(-2, 3, size=(n, N), dtype=np.int64) + 37
    return U.astype(np.float64) / 1024.0


def main():
    sys.path.insert(0, HERE)
    import make_goldens as mg
    mg._install_standins()
    sys.path.insert(0, mg.REF)
    import hippyflow as hf
    seed, n, N, K, r = 20261006, 8300, 600, 150, 12
    u_data = snapshots(seed, n, N, K)
    M = mg.mass_matrix_1d(N)
    pod = object.__new__(hf.PODProjectorFromData)   # ctor needs dolfin function spaces
    pod.M_csr = M
    out = dict(seed=seed, n=n, N=N, K=K, r=r, M_data=M.data, M_indices=M.indices, M_indptr=M.indptr)
    for shifted in (True, False):
        d, phi, Mphi, shift = pod.construct_subspace(u_data.copy(), r, shifted=shifted, method="hep", verify=False)
        tag = "hep_%d" % int(shifted)
        out["d_" + tag], out["phi_" + tag], out["shift_" + tag] = d, phi, shift
        print(tag, d, flush=True)
    np.savez_compressed(os.path.join(HERE, "pod_from_data_8300.npz"), **out)
    print("wrote pod_from_data_8300.npz")


if __name__ == "__main__":
    main()
