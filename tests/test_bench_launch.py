"""bench.py's N > 1 path on a CPU-only machine: the driver's launcher line (torch.distributed.run, one process per "GPU", RANK / WORLD_SIZE
from the environment), the default workload at N > 1 (cfg5), the query sharding, the REAL host planner on every rank's shard, the
per-step result gather (ResultGather.rebind + all_gather_into_tensor) and the one JSON line of rank 0 — everything up to the kernels,
which --dry-run leaves out (zeros in the result blocks, gloo instead of RCCL)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bench(nproc, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--dry-run", "--docs", "100000", "--vocab", "10000"] + extra  # fmt: skip
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_default_workload_is_the_mixed_batch():
    out = run_bench(2, ["--queries", "1000"])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["scaling"] == "weak"
    assert out["config"]["workload"].startswith("cfg5") and out["config"]["queries_per_gpu_per_step"] == 1000 and out["config"]["queries_per_step"] == 2000
    assert [b["queries"] for b in out["config"]["batches_per_step"]] == [700, 300]
    assert out["gather_check"] == {"ranks": 2, "blocks": ["counts", "docs", "scores", "topk_counts"], "equal_on_every_rank": True}
    assert out["scaling_ref"]["queries_per_step"] == 1000 and "speedup_vs_scaling_ref" in out and out["per_gpu_value"] * 2 == out["value"]
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline", "kernels_only", "end_to_end"):
        assert k in out


def test_strong_scaling_splits_the_batch():
    out = run_bench(2, ["--queries", "1000", "--scaling", "strong", "--workload", "cfg2", "--scaling-ref-steps", "0"])
    assert out["scaling"] == "strong" and out["config"]["queries_per_gpu_per_step"] == 500 and out["config"]["queries_per_step"] == 1000 and "scaling_ref" not in out
