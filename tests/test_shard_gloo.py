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
