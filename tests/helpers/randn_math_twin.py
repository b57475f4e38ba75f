"""numpy twin of hippyflow_amd/csrc/hfmi_randn_math.h (the range-specific fp64 functions of the Box-Muller kernel):
the same reductions, the same coefficients and operation order; the hardware seeds v_rcp_f64 / v_rsq_f64 (2^-24) are
stood in for by float32 arithmetic.  numpy has no fused multiply-add, so every ``a * b + c`` here rounds twice where
the device rounds once: the bounds checked in tests/test_randn_math_twin.py hold for both."""
import numpy as np

f32, f64 = np.float32, np.float64
LOG_C = [float.fromhex(h) for h in ("0x1.5555555555558p-2", "0x1.9999999995219p-3", "0x1.2492492e0644ep-3", "0x1.c71c62cbfea3fp-4",
                                   "0x1.7462bba41c635p-4", "0x1.39fd807189bb6p-4", "0x1.2b62ca3da2e33p-4")]
SIN_C = [float.fromhex(h) for h in ("0x1.921fb54442d18p+0", "-0x1.4abbce625be41p-1", "0x1.466bc677587e9p-4", "-0x1.32d2cce2e4892p-8",
                                   "0x1.50782fd8d055dp-13", "-0x1.e300707084987p-19", "0x1.e3f30e54e0ae1p-25")]
COS_C = [float.fromhex(h) for h in ("0x1.0000000000000p+0", "-0x1.3bd3cc9be458bp+0", "0x1.03c1f081b075cp-2", "-0x1.55d3c7dbfa5f9p-6",
                                   "0x1.e1f4fb5d356afp-11", "-0x1.a6c9c0485eb30p-16", "0x1.f3db44f6585e0p-22")]
SQRT_HALF = float.fromhex("0x1.6a09e667f3bcdp-1")
LN2X2 = float.fromhex("0x1.62e42fefa39efp+0")


def _horner(c, z):
    acc = np.full_like(z, c[-1])
    for a in c[-2::-1]:
        acc = acc * z + a
    return acc


def neg2log_scaled(u, sigma=1.0):
    m, e = np.frexp(u)
    low = m < SQRT_HALF
    sc = np.where(low, 2.0, 1.0)
    e = (e - low).astype(f64)
    num, den = m * sc - 1.0, m * sc + 1.0
    x = (f32(1) / den.astype(f32)).astype(f64)
    s = num * x
    s = s + (num - s * den) * x
    z = s * s
    at = s + (s * z) * _horner(LOG_C, z)
    return at * (-4.0 * sigma * sigma) + e * (-LN2X2 * sigma * sigma)


def sqrt_pos(t):
    with np.errstate(over="ignore", divide="ignore"):
        y32 = f32(1) / np.sqrt(t.astype(f32))
    y = np.where(np.isfinite(y32) & (y32 > 0), y32.astype(f64), 1.0 / np.sqrt(t) * (1 + 2.0 ** -25))
    hy = 0.5 * y
    g = t * y
    g = g + (t - g * g) * hy
    return g + (t - g * g) * hy


def rotate_turn(x, rad):
    """x: the 32-bit integer behind the angle u = (x + 1/2) 2^-32 (uint64 / float array of integer values)."""
    x = np.asarray(x).astype(np.uint64)
    odd = ((x >> np.uint64(31)) ^ (x >> np.uint64(30))) & np.uint64(1)              # rint(2u) == 1
    xs = (x ^ (odd << np.uint64(31))).astype(np.uint32).view(np.int32).astype(f64)   # 2^31 (2u - rint(2u)) - 1/2
    f = xs * 2.0 ** -31 + 2.0 ** -32
    y = f * f
    S = f * _horner(SIN_C, y)
    Cc = _horner(COS_C, y)
    c2 = Cc * Cc - S * S
    sc = S * Cc
    sign = np.where(odd == 1, -1.0, 1.0)
    return (sign * rad) * c2, (2.0 * sign * rad) * sc


def box_muller4(x, sigma=1.0):
    """x: uint32 array (..., 4) -> float64 array (..., 4)."""
    x = x.astype(f64)
    out = np.empty(x.shape)
    for h in (0, 1):
        u1 = (x[..., 2 * h] + 0.5) * 2.0 ** -32
        out[..., 2 * h], out[..., 2 * h + 1] = rotate_turn(x[..., 2 * h + 1], sqrt_pos(neg2log_scaled(u1, sigma)))
    return out
