"""CPU oracle for the hippyflow model-based-projector hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``hippyflow_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

Two modules:

* ``hippylib_restated``  -- numpy fp64 restatement of the third-party
  ``hippylib`` arithmetic the path calls (``doublePass``, ``doublePassG``,
  ``MultiVector.orthogonalize/Borthogonalize``, ``MatMvMult``, ``MvDSmatMult``,
  ``LowRankOperator``, ``Solver2Operator``).  hippylib is NOT vendored under
  /root/reference and cannot be installed here, so for these
  functions **parity is unpinned**: they restate the published algorithm
  (Saibaba, Lee, Kitanidis, "Randomized algorithms for generalized Hermitian
  eigenvalue problems with application to computing Karhunen-Loeve expansion",
  NLAA 2016, Algorithms 2 and 5/6; hippylib 3.x ``randomizedEigensolver.py`` /
  ``multivector.py``, branch ``matmvmult`` per /root/reference/.travis.yml:15)
  and are anchored on the reference's own call sites and test invariants.
* ``hippyflow_restated`` -- numpy restatement of the arithmetic that IS in
  /root/reference (deterministic POD, mean J^T J action, collective averaging,
  mass-preconditioned covariance).  These are pinned against the reference
  itself, executed in the authoring container by
  ``tests/golden/make_goldens.py``; the resulting vectors live in
  ``tests/golden/*.npz``.
"""
