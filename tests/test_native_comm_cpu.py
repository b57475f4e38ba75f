"""CPU tests (no GPU) of the native communicator's host side: hippyflow_amd.launch starts the ranks as fresh child
processes, rank 0 publishes the communicator id through a file, the ranks meet in a shared-memory segment and run
host-payload collectives with the payload rules of the reference (collective.py:74-159).  The device transports
(RCCL, p2p) are covered by the -m gpu tests."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "comm_worker.py")


@pytest.mark.parametrize("world", [2, 3])
def test_native_collective_host_payloads(tmp_path, world):
    from hippyflow_amd.launch import spawn_ranks
    env = dict(os.environ, HFMI_COMM_TIMEOUT_S="60")
    assert spawn_ranks([WORKER, str(tmp_path), "payloads"], world, env=env, timeout=120) == 0
    rs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    tri = world * (world + 1) // 2
    for rank, r in enumerate(rs):
        assert int(r["size"]) == world and int(r["rank"]) == rank and str(r["transport"]) == "host"
        assert float(r["sum_float"]) == tri and float(r["avg_float"]) == tri / world
        assert int(r["sum_int"]) == tri and bool(r["sum_int_is_int"]) and float(r["avg_int"]) == tri / world
        assert float(r["big_avg_err"]) < 1e-9 and bool(r["big_inplace"])
        np.testing.assert_array_equal(r["strided"], np.full((6, 2), float(tri)))
        assert bool(r["bcast_ok"])
        np.testing.assert_array_equal(r["bcast_ints"], np.array([0, 3], dtype=np.uint64))
        assert float(r["bcast_scalar"]) == 0.5 and int(r["bcast_int_scalar"]) == world - 1 + 5
        assert float(r["max"]) == 10.0 * (world - 1)
        assert bool(r["bad_op_max"]) and bool(r["bad_op_None"]) and bool(r["bad_type"])
        np.testing.assert_array_equal(r["d_par"], rs[0]["d_par"])          # identical bits on every rank
    np.testing.assert_allclose(rs[0]["d_par"], rs[0]["d_ser"], rtol=1e-12)  # P ranks averaged == one rank over all samples


def test_missing_peer_times_out_instead_of_hanging(tmp_path):
    from hippyflow_amd.launch import spawn_ranks
    env = dict(os.environ, HFMI_COMM_TIMEOUT_S="2")
    t0 = time.time()
    code = spawn_ranks([WORKER, str(tmp_path), "missing_peer"], 2, env=env, timeout=60)
    assert code != 0 and time.time() - t0 < 30


def test_launcher_cli_and_failure_propagation(tmp_path):
    script = tmp_path / "w.py"
    script.write_text("import os, sys\nr = int(os.environ['RANK'])\nopen(sys.argv[1] + '/r%d' % r, 'w').write(os.environ['WORLD_SIZE'] + ' ' + os.environ['LOCAL_RANK'])\n"
                      "sys.exit(3 if (len(sys.argv) > 2 and r == 1) else 0)\n")
    cmd = [sys.executable, "-m", "hippyflow_amd.launch", "-n", "3", str(script), str(tmp_path)]
    assert subprocess.run(cmd, cwd=ROOT).returncode == 0
    assert sorted(os.listdir(str(tmp_path))) == ["r0", "r1", "r2", "w.py"]
    assert (tmp_path / "r2").read_text() == "3 2"
    assert subprocess.run(cmd + ["fail"], cwd=ROOT).returncode == 3


def test_unique_id_and_explicit_init_single_rank():
    """The mpi4py-style bootstrap: the id bytes are made on one rank and handed to init_rank."""
    import hippyflow_amd as hf
    ident = hf.NativeCollective.unique_id()
    assert len(ident) == 256 and ident != hf.NativeCollective.unique_id()
    coll = hf.NativeCollective.from_unique_id(ident, 1, 0, host_only=True)
    assert coll.size() == 1 and coll.rank() == 0
    v = np.arange(4.0)
    assert coll.allReduce(v, "avg") is v and coll.allReduce(2.5, "sum") == 2.5
    coll.barrier()
    coll.close()
    with pytest.raises(ValueError):
        hf.NativeCollective.from_unique_id(b"short", 1, 0, host_only=True)
    with pytest.raises(hf.HfmiError):
        hf.NativeCollective.from_unique_id(bytes(256), 1, 0, host_only=True)       # not made by hfmi_comm_unique_id


def test_back_to_back_communicators_never_join_a_stale_segment(tmp_path):
    """Round-2 advice: the id file is removed before any rank returns, and a rank rejects an id whose token it has
    already joined -- six communicators in a row through one file (some still open) all reduce correctly."""
    from hippyflow_amd.launch import spawn_ranks
    world = 3
    env = dict(os.environ, HFMI_COMM_TIMEOUT_S="30")
    t0 = time.time()
    assert spawn_ranks([WORKER, str(tmp_path), "back_to_back"], world, env=env, timeout=120) == 0
    assert time.time() - t0 < 60
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_array_equal(got["sums"], [world * (world + 1) / 2 + world * it for it in range(6)])
        assert "host payloads only" in str(got["why"])


def test_id_file_is_private_and_foreign_files_are_not_joined(tmp_path):
    """The id file is created 0600 with O_EXCL|O_NOFOLLOW; a reader skips files that are not regular / not 0600."""
    import ctypes as C
    import stat
    import hippyflow_amd as hf
    from hippyflow_amd import _lib as L
    path = str(tmp_path / "comm.id")
    # a world-readable file with plausible size is ignored by a non-zero rank: it times out instead of joining it
    open(path, "wb").write(hf.NativeCollective.unique_id())
    os.chmod(path, 0o644)
    os.environ["HFMI_COMM_TIMEOUT_S"] = "1"
    try:
        h = C.c_void_p()
        with pytest.raises(hf.HfmiError, match="communicator id"):
            L.call("hfmi_comm_init_from_file", None, path.encode(), 2, 1, C.byref(h))
        os.unlink(path)
        # a symlink where the temporary file would go is not followed (O_EXCL | O_NOFOLLOW): rank 0 replaces it
        victim = tmp_path / "victim"
        victim.write_text("precious")
        os.symlink(str(victim), path + ".tmp.%d" % os.getpid())
        h = C.c_void_p()
        L.call("hfmi_comm_init_from_file", None, path.encode(), 1, 0, C.byref(h))       # one rank: publishes, joins, removes
        assert victim.read_text() == "precious" and not os.path.exists(path)
        L.load().hfmi_comm_destroy(h)
    finally:
        os.environ.pop("HFMI_COMM_TIMEOUT_S", None)


def test_transport_decision_is_one_function_of_the_published_table():
    """First-contact hardening: every rank evaluates the same pure function of the same table, so a missing librccl
    (HFMI_RCCL_LIB=/nonexistent on ANY rank) sends ALL ranks to the p2p transport."""
    import ctypes as C
    from hippyflow_amd import _lib as L

    def decide(has_dev, rccl_ok, ids, force=0):
        n = len(ids)
        t = C.c_int(-9)
        why = C.create_string_buffer(200)
        arr = (C.c_char_p * n)(*[i.encode() for i in ids])
        L.call("hfmi_comm_decide_transport", n, (C.c_int * n)(*has_dev), (C.c_int * n)(*rccl_ok), arr, force, C.byref(t), why, 200)
        return t.value, why.value.decode()

    gpus = ["0000:%02x:00.0" % (5 + 8 * i) for i in range(8)]
    assert decide([1] * 8, [1] * 8, gpus)[0] == 1                                   # rccl
    t, why = decide([1] * 8, [1, 1, 1, 0, 1, 1, 1, 1], gpus)
    assert t == 2 and "rank 3" in why                                               # one rank without librccl -> everybody p2p
    t, why = decide([1, 1], [1, 1], [gpus[0], gpus[0]])
    assert t == 2 and "share a GPU" in why
    assert decide([1] * 4, [1] * 4, gpus[:4], force=1)[0] == 2
    assert decide([0, 0, 0], [0, 0, 0], ["", "", ""])[0] == 0                       # host only
    assert decide([1, 0], [1, 0], [gpus[0], ""])[0] == -1                           # inconsistent


def test_missing_librccl_is_reported_not_fatal(tmp_path):
    """HFMI_RCCL_LIB=/nonexistent: the id carries no RCCL id, nothing fails, the table says so (on a GPU box the same
    launch picks p2p on every rank -- tests/test_gpu_comm.py)."""
    env = dict(os.environ, HFMI_RCCL_LIB="/nonexistent/librccl.so", HFMI_COMM_TIMEOUT_S="30")
    from hippyflow_amd.launch import spawn_ranks
    assert spawn_ranks([WORKER, str(tmp_path), "payloads"], 2, env=env, timeout=120) == 0


def test_a_rank_killed_mid_solve_gives_one_report_and_no_hang(tmp_path):
    """a rank that dies in the middle of a run takes its peers down at once (they would otherwise sit in the
    next collective until the communicator's time-out), and the launcher's report says which rank, with every rank's stderr."""
    from hippyflow_amd.launch import spawn_ranks
    script = tmp_path / "dies.py"
    script.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import hippyflow_amd as hf\n"
        "c = hf.NativeCollective.from_env(host_only=True)\n"
        "a = c.allReduce(np.ones(4), 'sum')\n"
        "sys.stderr.write('rank %%d after the first collective\\n' %% c.rank()); sys.stderr.flush()\n"
        "if c.rank() == 1:\n"
        "    os._exit(7)\n"
        "print('{\"never\": \"printed\"}') if False else None\n"
        "c.allReduce(np.ones(4), 'sum')\n"          # the survivors wait here
        "print('{\"value\": 1}')\n" % ROOT)
    env = dict(os.environ, HFMI_COMM_TIMEOUT_S="120")
    report = {}
    t0 = time.time()
    code = spawn_ranks([str(script)], 3, env=env, timeout=100, report=report)
    assert code == 7 and time.time() - t0 < 60                      # far inside the communicator's time-out
    assert report["first_failed"] == 1 and report["codes"][1] == 7 and not report["timed_out"]
    assert set(report["stopped"]) <= {0, 2}
    assert "after the first collective" in report["stderr_tail"]["1"]
    assert "value" not in report["stdout_rank0"]                    # rank 0 never got to its line


def test_bench_multi_gpu_failure_is_one_json_error_line():
    """`python bench.py --gpus 2` where the ranks cannot run (no GPU in this container): ONE JSON line with "error", the ranks'
    exit codes and stderr tails, a non-zero exit code, no hang."""
    import json
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--quick"],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                         env=dict(os.environ, HFMI_BENCH_TIMEOUT_S="200"))
    try:
        import hippyflow_amd as hf
        if hf.device_count() > 0:
            pytest.skip("a GPU is visible here: this run succeeds")
    except Exception:
        pass
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert res.returncode != 0 and len(lines) == 1 and time.time() - t0 < 250
    line = json.loads(lines[0])
    assert line["value"] is None and "error" in line and line["n_gpus"] == 2
    assert len(line["rank_exit_codes"]) == 2 and line["communicator"] == "not reached"
    assert "needs a GPU" in "".join(line["stderr_tail"].values())
