"""Determinism soak of the p2p transport (ranks sharing the one GPU of the box): the `soak` mode of tests/helpers/gpu_comm_worker2.py
(200 all-reduces of the same rank-specific inputs, two block sizes, sum / avg) repeated over both synchronisation modes and rank
counts; prints, per run, whether every repetition gave the bits of the first one and the first mismatches if not.
    python scripts/p2p_soak.py [rounds [stream,host [4,2,8]]]"""
import json
import os
import sys
import tempfile

sys.path.insert(0, ".")
from hippyflow_amd.launch import spawn_ranks  # noqa: E402

W = os.path.join("tests", "helpers", "gpu_comm_worker2.py")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
syncs = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("stream", "host")
worlds = tuple(int(w) for w in sys.argv[3].split(",")) if len(sys.argv) > 3 else (4, 2, 8)
for rnd in range(rounds):
    for sync in syncs:
        for world in worlds:
            d = tempfile.mkdtemp()
            env = dict(os.environ, HFMI_COMM_TIMEOUT_S="120", HFMI_P2P_SYNC=sync)
            report = {}
            rc = spawn_ranks([W, d, "soak"], world, env=env, timeout=900, report=report)
            rs = []
            for r in range(world):
                try:
                    rs.append(json.load(open(os.path.join(d, "soak_rank%d.json" % r))))
                except OSError:
                    rs.append(None)
            ok = [None if r is None else r["same"] for r in rs]
            dig = [r["digests"] for r in rs if r is not None]
            retries = [None if r is None else r.get("probe_retries_total") for r in rs]
            print("round %d %-6s world %d rc %d same %s digests equal across ranks: %s  probe retries %s" % (rnd, sync, world, rc, ok, all(x == dig[0] for x in dig) if dig else None, retries), flush=True)
            for r in rs:
                if r is not None and r["mismatches"]:
                    print("   rank", r["rank"], len(r["mismatches"]), r["mismatches"][:3], flush=True)
            if rc != 0:
                for k, tail in report.get("stderr_tail", {}).items():
                    last = [ln for ln in tail.splitlines() if "Error" in ln or "error" in ln][-2:]
                    print("   rank", k, "stderr:", last, flush=True)
