"""What are the 15-80 ms stalls of an eigensolve?  Evidence that they are the container's CPU quota, not the GPU:
the cgroup of these boxes grants 16 CPUs (cpu.max = 1600000 100000) on a host that reports 256; a process that makes more threads
runnable than that is THROTTLED by the scheduler for the rest of the 100 ms period -- every thread, the one that feeds the GPU's queue
included, and the tridiagonalisation is 2-16 thousand dependent launches that the host must keep issuing.
Experiment: 30 calls of hfmi_sym_eig_small at n = 2048 with a host matmul (what a caller's checks, or a benchmark's test-matrix
construction, do) between calls, (a) numpy's BLAS pool at its default size (all hardware threads), (b) limited to the CPU quota.
Printed per mode: min / median / max of the call, and the cgroup's own counters over the loop (cpu.stat: nr_throttled, throttled_usec).
    python scripts/eig_stall_diagnosis.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hippyflow_amd as hf  # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402


def cpu_stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            out[k] = int(v)
    except OSError:
        pass
    return out


def quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, round(float(q) / float(p)))
    except (OSError, ValueError):
        return None


n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(0)
X = rng.standard_normal((n, n + 50)) * np.exp(-0.01 * np.arange(n + 50))[None, :]
with threadpool_limits(limits=quota() or 8):
    G = X @ X.T
B = rng.standard_normal((1500, 1500))
print("hardware threads %d, cgroup CPU quota %s, cpu.max = %s" % (os.cpu_count(), quota(), open("/sys/fs/cgroup/cpu.max").read().strip()
                                                                    if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"), flush=True)
hf.sym_eig_small(G)
for label, limit in (("BLAS pool = all hardware threads (numpy's default)", os.cpu_count()), ("BLAS pool = CPU quota", quota() or 8),
                     ("BLAS pool = all hardware threads, again", os.cpu_count()), ("BLAS pool = CPU quota, again", quota() or 8)):
    time.sleep(0.3)
    s0 = cpu_stat()
    ts = []
    with threadpool_limits(limits=limit):
        for _ in range(30):
            _ = B @ B                      # host work between calls
            t0 = time.perf_counter()
            d, V = hf.sym_eig_small(G)
            ts.append(time.perf_counter() - t0)
    s1 = cpu_stat()
    ts = np.array(ts) * 1e3
    print("%-52s min %.2f  median %.2f  max %.2f ms  max/min %.2f | calls > 1.15 min: %d | cgroup: nr_throttled +%d, throttled +%.1f ms" % (
        label, ts.min(), np.median(ts), ts.max(), ts.max() / ts.min(), int((ts > 1.15 * ts.min()).sum()),
        s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3), flush=True)
