"""CPU test of the N>1 path: world_size 2, gloo.  Query sharding + result gather (trinity_amd/dist.py) must
reproduce the single-process answers for every query."""
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_results_match_single_process(tmp_path):
    marker = tmp_path / "ok.txt"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "dist_worker.py"), str(marker)]  # fmt: skip
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert marker.read_text().startswith("ok world=2")
