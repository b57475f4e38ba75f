"""Wall time of hfmi_sym_eig_small beyond one workgroup (256 < n <= 8192: hfmi_eig_blocked.hip) next to numpy.linalg.eigh on
the box's host, with the phase split the library prints under HFMI_EIG_LARGE_TIMING=1 (stderr).  Usage:
python scripts/eig_large_time.py [n ...] [--no-host] [--low-rank] [--reps=N]; per-kernel times come from the rocprofv3 kernel trace
of this script.  --reps=N: N back-to-back timed calls per size (default 3); the line reports min, median and max and the max/min ratio."""
import resource
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import hippyflow_amd as hf  # noqa: E402

def _steal():
    """steal ticks of all CPUs (/proc/stat, 8th value of the first line): time the hypervisor ran something else"""
    try:
        return int(open("/proc/stat").readline().split()[8])
    except (OSError, IndexError, ValueError):
        return 0


def _cpu_budget():
    """what the process may use, next to what the machine reports: affinity mask, cgroup quota"""
    out = {"os.cpu_count": __import__("os").cpu_count(), "affinity": len(__import__("os").sched_getaffinity(0))}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            out[path] = open(path).read().strip()
        except OSError:
            pass
    return out


blas_threads = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--blas-threads=")]
if blas_threads:      # the host BLAS pool (numpy's matmul that builds the test matrix) limited for the WHOLE script: its workers keep
    from threadpoolctl import threadpool_limits      # spinning for a while after a product and compete with the thread that feeds the GPU
    threadpool_limits(limits=blas_threads[0])
if "--diag" in sys.argv:
    print("cpu budget:", _cpu_budget(), "blas threads:", blas_threads or "default", flush=True)
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 1024, 2048, 4096]
host = "--no-host" not in sys.argv
rng = np.random.default_rng(0)
reps = max([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--reps=")] or [3])
lowrank = "--low-rank" in sys.argv          # Gram matrix of n snapshots that span n / 8 dimensions: most poles deflate in the merges
for n in sizes:
    if lowrank:
        X = rng.standard_normal((n, max(2, n // 8))) * np.exp(-0.05 * np.arange(max(2, n // 8)))[None, :]
    else:
        X = rng.standard_normal((n, n + 50)) * np.exp(-0.01 * np.arange(n + 50))[None, :]
    G = X @ X.T
    d, V = hf.sym_eig_small(G)                  # warm-up: workspace allocation
    ts = []
    diag = []
    for _ in range(reps):
        ru0, st0, c0 = resource.getrusage(resource.RUSAGE_SELF), _steal(), time.process_time()
        t0 = time.perf_counter()
        d, V = hf.sym_eig_small(G)
        ts.append(time.perf_counter() - t0)
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        diag.append((ts[-1], time.process_time() - c0, ru1.ru_nivcsw - ru0.ru_nivcsw, ru1.ru_nvcsw - ru0.ru_nvcsw,
                     ru1.ru_minflt - ru0.ru_minflt, ru1.ru_majflt - ru0.ru_majflt, _steal() - st0))
    if "--diag" in sys.argv:      # host-side picture of the slowest and of a typical call: is a stall the GPU's or the host's?
        med = sorted(diag)[len(diag) // 2]
        for label, row in (("median call", med), ("slowest call", max(diag))):
            print("   %-12s wall %.2f ms, process cpu %.2f ms, ctx switches invol / vol %d / %d, page faults minor / major %d / %d, "
                  "steal ticks (whole box) %d" % ((label, 1e3 * row[0], 1e3 * row[1]) + tuple(row[2:])), flush=True)
    line = ("low-rank " if lowrank else "") + "n=%d  sym_eig_small %.2f ms (min of %d; median %.2f, max %.2f, max/min %.3f)" % (
        n, 1e3 * min(ts), reps, 1e3 * float(np.median(ts)), 1e3 * max(ts), max(ts) / min(ts))
    if host:
        t0 = time.perf_counter()
        w, _ = np.linalg.eigh(G)
        th = time.perf_counter() - t0
        w = w[::-1]
        line += "  numpy.linalg.eigh %.1f ms  | eig err %.1e  orth %.1e  resid %.1e" % (
            1e3 * th, np.abs(d - w).max() / w[0], np.abs(V.T @ V - np.eye(n)).max(), np.abs(G @ V - V * d).max() / w[0])
    print(line, flush=True)
