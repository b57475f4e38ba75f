"""How many Jacobi sweeps does the Rayleigh-Ritz eigensolve take on the config-4 shard with and without the noise term?
HFMI_DEBUG_TIMING=1 python scripts/jacobi_sweeps.py   (the library prints cycles and sweeps per small kernel on stderr)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf
from hippyflow_amd import workloads
for noise in (0.0, 0.01):
    wl = workloads.as_workload(200000, 16, q=100, latent=100, rate=0.06, seed=4, noise=noise)
    hf.parRandom.reseed(1)
    Om = hf.MultiVector(200000, 74); hf.parRandom.normal(1.0, Om)
    print("noise", noise, file=sys.stderr, flush=True)
    d, U = hf.doublePass(wl.operator, Om, 64)
    print("   d[0], d[-1] =", d[0], d[-1], file=sys.stderr, flush=True)
