#!/usr/bin/env python3
"""Generate tests/golden/pod_from_data_320.npz by running the REFERENCE's own ``PODProjectorFromData.construct_subspace``
(modeling/PODProjector.py:699-852) on a snapshot set with MORE THAN 256 snapshots: the regime in which the device serves
``la.eigh(G)`` (:821) with the whole-GPU eigensolver of round 5 (hippyflow_amd/csrc/hfmi_eig_blocked.hip) instead of the
one-workgroup kernels -- so that path is pinned against the reference itself, not only against numpy.

Run ONLY in the authoring container (needs /root/reference); the stand-in modules of make_goldens.py let ``import hippyflow``
succeed in this process:

    python tests/golden/make_pod_large_golden.py

Only inputs and outputs are stored.  The snapshot matrix is integer-valued (int16: exact in fp64 on every machine)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402


def main():
    mg._install_standins()
    sys.path.insert(0, mg.REF)
    import hippyflow as hf
    rng = np.random.default_rng(20261003)
    n, N, r = 320, 600, 8
    U0, _ = np.linalg.qr(rng.standard_normal((n, 60)))
    W0, _ = np.linalg.qr(rng.standard_normal((N, 60)))
    smooth = (U0 * np.exp(-0.25 * np.arange(60))) @ W0.T + 0.2 * np.cos(np.linspace(0, 2, N))[None, :]
    u_int = np.rint(4096.0 * smooth + 3.0 * rng.standard_normal((n, N))).astype(np.int16)      # low rank + decay + a noise floor
    u_data = u_int.astype(np.float64) / 4096.0                                                 # exact: integers times a power of two
    M = mg.mass_matrix_1d(N)
    pod = object.__new__(hf.PODProjectorFromData)   # ctor needs dolfin function spaces
    pod.M_csr = M
    out = dict(u_int16=u_int, scale=1.0 / 4096.0, M_data=M.data, M_indices=M.indices, M_indptr=M.indptr, N=N, n=n, r=r)
    for shifted in (True, False):
        for method in ("hep", "ghep", "inverse_ghep"):
            d, phi, Mphi, shift = pod.construct_subspace(u_data.copy(), r, shifted=shifted, method=method, verify=False)
            tag = "%s_%d" % (method, int(shifted))
            out["d_" + tag], out["phi_" + tag], out["shift_" + tag] = d, phi, shift
    np.savez_compressed(os.path.join(HERE, "pod_from_data_320.npz"), **out)
    print("wrote pod_from_data_320.npz:", {k: getattr(v, "shape", v) for k, v in out.items() if k.startswith("d_")}, out["d_hep_1"])


if __name__ == "__main__":
    main()
