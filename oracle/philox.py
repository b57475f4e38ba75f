"""Philox4x32-10 counter-based generator + Box-Muller normals, numpy restatement.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Role on the path: the probe block Omega.  The reference draws it with
``hp.parRandom.normal(1., Omega)`` on collective-rank 0 and broadcasts it
(activeSubspaceProjector.py:433-443,536-551; PODProjector.py:365-374;
KLEProjector.py:151-160).  hippylib's stream (a per-rank seeded mt19937 inside a
dolfin C++ extension) cannot be reproduced outside hippylib, so there is no
reference bit pattern to match; the device generator is instead a counter-based
Philox (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) so
that every GPU regenerates the identical Omega from (seed, stream) with no
broadcast.  This file is the checker for that generator:

* the 32-bit integer stream is compared bit-exactly, and is itself pinned by
  the Random123 known-answer vectors in ``tests/test_oracle_philox.py``;
* the normals are compared to a few ulp (device log/sincos differ from libm).

Element map (shared with ``hippyflow_amd/csrc/hfmi_randn_math.h``): for column j and row group g
(rows 4g .. 4g+3):  ctr = (g & 0xffffffff, g >> 32, j, stream), key = (seed &
0xffffffff, seed >> 32);  x = philox4x32_10(ctr, key);  u(x) = (x + 0.5) * 2^-32
(the 32-bit uniform lattice of hipRAND / cuRAND's Philox normal draws: four
normals per generator call, |z| <= 6.76);
rows 4g, 4g+1 = sqrt(-2 ln u(x0)) * (cos, sin)(2 pi u(x1)),
rows 4g+2, 4g+3 = sqrt(-2 ln u(x2)) * (cos, sin)(2 pi u(x3)).
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint64(0x9E3779B9)
_W1 = np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Inputs broadcastable arrays of values
    < 2^32 (any integer dtype); returns four uint32 arrays."""
    c0, c1, c2, c3, k0, k1 = [np.asarray(v).astype(np.uint64) & _MASK
                              for v in (c0, c1, c2, c3, k0, k1)]
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(c0, c1, c2, c3, k0, k1)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def raw_block(N, k, seed, stream=0):
    """uint32 array (k, ngroups, 4), ngroups = ceil(N / 4): the integer stream behind randn_block."""
    ngroups = (N + 3) // 4
    g = np.arange(ngroups, dtype=np.uint64)[None, :]
    j = np.arange(k, dtype=np.uint64)[:, None]
    seed = int(seed)
    x = philox4x32_10(g & _MASK, g >> _S32, j, np.uint64(stream & 0xFFFFFFFF),
                      np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF))
    return np.stack(x, axis=-1)


def randn_block(N, k, seed, stream=0, sigma=1.0):
    """(N, k) Fortran-ordered block of i.i.d. N(0, sigma^2)."""
    x = raw_block(N, k, seed, stream).astype(np.float64)
    u = (x + 0.5) * 2.0 ** -32
    z = np.empty((k, 4 * u.shape[1]))
    for half in (0, 1):
        r = np.sqrt(-2.0 * np.log(u[..., 2 * half]))
        ang = 2.0 * np.pi * u[..., 2 * half + 1]
        z[:, 2 * half::4] = r * np.cos(ang)
        z[:, 2 * half + 1::4] = r * np.sin(ang)
    return np.asfortranarray(sigma * z[:, :N].T)
