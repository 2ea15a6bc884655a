"""world_size-2 gloo run of the sharding plumbing bench.py uses for N > 1 (CPU only)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_shard_streams_and_aggregate():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and r["frames"] == 4.0
    assert r["seeds"] == [[0, 1], [1000, 1001]]           # one stream per rank, disjoint
    assert r["job_time"] >= r["my_time"] and r["job_time"] >= 0.1  # max over ranks (rank 1 sleeps longer)
    assert r["kps"] > 100 and abs(r["value"] - 4.0 / r["job_time"]) < 1e-9


def test_single_process_path_needs_no_process_group():
    from fasttrack_amd import shard
    assert shard.init(0, 1) is None
    assert shard.reduce_max(None, 1.5) == 1.5 and shard.reduce_sum(None, [1, 2]) == [1.0, 2.0]
    assert shard.stream_seeds(3, 2) == [3000, 3001]


def test_gather_floats_single_process():
    from fasttrack_amd import shard
    assert shard.gather_floats(None, 2.5, 1) == [2.5]


def test_bench_refuses_a_world_that_differs_from_gpus():
    """`bench.py --gpus 8` inside a 1-rank launcher environment must fail instead of printing n_gpus: 1"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "--gpus 8 but WORLD_SIZE is 1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_gpus_n_starts_n_ranks(tmp_path, monkeypatch):
    """without a launcher environment `--gpus N` starts N ranks through torch.distributed.run (here: the command
    line it builds, with the child replaced by a recorder - there is no GPU in this container)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import pytest
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_numa_pinning_helpers(tmp_path):
    """shard.pin_to_numa_node on a fake sysfs tree: the CPUs of the device's node, shared evenly by the ranks of that node,
    intersected with the affinity the process already has; unknown nodes change nothing"""
    from fasttrack_amd import shard
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    have = sorted(os.sched_getaffinity(0))
    sysfs = tmp_path / "sys"
    (sysfs / "bus/pci/devices/0000:c1:00.0").mkdir(parents=True)
    (sysfs / "bus/pci/devices/0000:c1:00.0/numa_node").write_text("1\n")
    (sysfs / "devices/system/node/node1").mkdir(parents=True)
    (sysfs / "devices/system/node/node1/cpulist").write_text(f"{have[0]}-{have[-1]}\n")
    assert shard.numa_node_of_pci("0000:C1:00.0", str(sysfs)) == 1 and shard.numa_node_of_pci("0000:00:00.0", str(sysfs)) == -1
    try:
        assert shard.pin_to_numa_node(-1, sysfs=str(sysfs)) is None and shard.pin_to_numa_node(7, sysfs=str(sysfs)) is None
        assert sorted(os.sched_getaffinity(0)) == have
        if len(have) >= 4:
            a = shard.pin_to_numa_node(1, world=2, slot=0, sysfs=str(sysfs))
            assert a == have[:len(have) // 2] and sorted(os.sched_getaffinity(0)) == a
            os.sched_setaffinity(0, have)
            b = shard.pin_to_numa_node(1, world=2, slot=1, sysfs=str(sysfs))
            assert b == have[len(have) // 2:] and not set(a) & set(b)
    finally:
        os.sched_setaffinity(0, have)
