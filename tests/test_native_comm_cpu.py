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
