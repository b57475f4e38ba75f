"""CPU suite: error bounds of the range-specific fp64 functions behind the probe draw (a1), on their numpy twin
(tests/helpers/randn_math_twin.py follows hippyflow_amd/csrc/hfmi_randn_math.h step by step), and agreement of the twin
with the element map of oracle/philox.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import randn_math_twin as tw  # noqa: E402
from oracle import philox  # noqa: E402


def _lattice(n=400000, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64)
    x[:2000] = np.arange(2000)                        # both ends of the lattice
    x[2000:4000] = 2 ** 32 - 1 - np.arange(2000)
    return x.astype(np.float64)


def test_log_sqrt_bounds():
    u = (_lattice() + 0.5) * 2.0 ** -32
    t = tw.neg2log_scaled(u)
    ref = -2.0 * np.log(u)
    # the residual step squares the seed error: (2^-24)^2 relative on the device, (float32 division)^2 ~ 8e-15 here
    assert np.max(np.abs(t - ref) / ref) < 2e-14
    r = tw.sqrt_pos(t)
    assert np.max(np.abs(r - np.sqrt(ref)) / np.sqrt(ref)) < 1e-14
    assert np.max(np.abs(r - np.sqrt(ref))) < 1e-14
    assert r.min() > 0 and r.max() < 6.77
    assert t.min() >= 2.0 ** -32          # the square root never sees 0 on this lattice


def test_rotation_bounds():
    x = _lattice(seed=1)
    u = (x + 0.5) * 2.0 ** -32
    z0, z1 = tw.rotate_turn(x, np.ones_like(u))
    assert np.max(np.abs(z0 - np.cos(2 * np.pi * u))) < 2e-15
    assert np.max(np.abs(z1 - np.sin(2 * np.pi * u))) < 2e-15
    assert np.max(np.abs(z0 * z0 + z1 * z1 - 1.0)) < 3e-15


def test_twin_follows_the_oracle_map():
    N, k, seed, stream = 1003, 5, 0xABCDEF0123456789, 9
    raw = philox.raw_block(N, k, seed, stream)                     # (k, ngroups, 4)
    z = tw.box_muller4(raw, sigma=1.5).reshape(k, -1)[:, :N].T
    ref = philox.randn_block(N, k, seed, stream, sigma=1.5)
    np.testing.assert_allclose(z, ref, rtol=0, atol=3e-14)
