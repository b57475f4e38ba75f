"""One process per GPU without an external launcher (the reference starts its ranks with ``mpirun -n P python driver.py``).

``spawn_ranks`` is called by a parent that has NOT touched the GPU (it imports no HIP-using module and makes no
device query): it starts P fresh child interpreters, one per device, gives each its RANK / LOCAL_RANK / WORLD_SIZE
and the path of the file through which rank 0 publishes the communicator id (``NativeCollective.from_env``), waits
for all of them and returns the worst exit code.  Children are ordinary child processes -- nothing is re-exec'ed.

    python -m hippyflow_amd.launch -n 8 driver.py --flag ...
"""
import os
import signal
import subprocess
import sys
import tempfile
import time


def launched():
    """True inside a rank started by ``spawn_ranks`` or by ``python -m torch.distributed.run``."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def _tail(path, limit=1500):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - limit))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def spawn_ranks(argv, nranks, env=None, timeout=None, report=None):
    """Run ``python argv...`` as ``nranks`` ranks; returns the largest exit code (124 on time-out).

    With ``report`` (a dict) the ranks' output is CAPTURED instead of inherited: rank 0's stdout and every rank's stderr
    (+ the other ranks' stdout) go to temporary files, and the dict receives ``codes`` (exit code per rank, negative =
    signal, None never happens), ``stopped`` (ranks this parent stopped after a peer failed or time ran out: their codes
    say nothing), ``first_failed``, ``timed_out``, ``seconds``, ``stdout_rank0`` and ``stderr_tail`` (rank -> last 1500
    bytes).  The caller decides what reaches its own stdout -- bench.py prints either rank 0's one JSON line or one JSON
    error line, never both and never a hang: a rank that dies takes its peers down within one poll interval."""
    fd, id_file = tempfile.mkstemp(prefix="hfmi-comm-", suffix=".id")
    os.close(fd)
    os.unlink(id_file)                       # rank 0 creates it (write + rename); the name is what is reserved
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(nranks), HFMI_COMM_ID_FILE=id_file, HFMI_LAUNCHER="hippyflow_amd.launch")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL and the p2p transport need here
    capture_dir = tempfile.mkdtemp(prefix="hfmi-ranks-") if report is not None else None
    procs, files = [], []
    t_start = time.time()
    for r in range(nranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        if capture_dir is None:
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
            continue
        err = open(os.path.join(capture_dir, "stderr.%d" % r), "wb")
        out = open(os.path.join(capture_dir, "stdout.0"), "wb") if r == 0 else err
        files.extend([err] if r else [err, out])
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=out, stderr=err))
    deadline = None if timeout is None else time.time() + timeout
    codes = [None] * nranks
    stopped = set()                          # ranks stopped HERE after a peer failed: their exit codes say nothing
    first_failed, timed_out_flag = None, False
    try:
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            failed = [i for i, c in enumerate(codes) if c not in (None, 0)]
            timed_out = deadline is not None and time.time() > deadline
            if failed or timed_out:
                # one rank died (its peers would wait for it until the communicator's time-out) or time is up:
                # stop exactly the processes started here
                first_failed = failed[0] if failed else None
                timed_out_flag = bool(timed_out and not failed)
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        stopped.add(i)
                        p.send_signal(signal.SIGTERM)
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        try:
                            codes[i] = p.wait(timeout=10)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[i] = p.wait()
                break
            time.sleep(0.02)
    finally:
        if os.path.exists(id_file):
            os.unlink(id_file)
        for f in files:
            f.close()
    worst = 0
    for i, c in enumerate(codes):
        if i in stopped or c is None:
            continue
        worst = max(worst, 128 - c if c < 0 else c)
    if timed_out_flag:
        worst = 124
    if report is not None:
        report.update(codes=list(codes), stopped=sorted(stopped), first_failed=first_failed, timed_out=timed_out_flag,
                      seconds=time.time() - t_start,
                      stdout_rank0=_tail(os.path.join(capture_dir, "stdout.0"), 1 << 22),
                      stderr_tail={str(r): _tail(os.path.join(capture_dir, "stderr.%d" % r)) for r in range(nranks)})
        import shutil
        shutil.rmtree(capture_dir, ignore_errors=True)
    return worst


def main(args=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m hippyflow_amd.launch", description=__doc__.splitlines()[0])
    ap.add_argument("-n", "--nranks", type=int, required=True)
    ap.add_argument("--timeout", type=float, default=None, help="seconds before the ranks are stopped")
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    ns = ap.parse_args(args)
    return spawn_ranks([ns.script] + ns.script_args, ns.nranks, timeout=ns.timeout)


if __name__ == "__main__":
    sys.exit(main())
