"""bench.py's N > 1 path on a CPU-only machine: the driver's launcher line (torch.distributed.run, one process per "GPU", RANK / WORLD_SIZE
from the environment), the default workload at N > 1 (cfg5), the query sharding, the REAL host planner on every rank's shard, the
per-step result gather (ResultGather.rebind + all_gather_into_tensor) and the one JSON line of rank 0 — everything up to the kernels,
which --dry-run leaves out (zeros in the result blocks, gloo instead of RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

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


POINT_KEYS = {"workload", "scaling", "total_queries", "n_gpus", "value", "unit", "per_gpu_value", "ms_per_step", "gather_ms", "hbm_bytes_in_use", "n1", "speedup_vs_n1", "what"}


def check_scaling_point(out, n, total):
    """Every line carries the scaling curve's point it stands for, self-contained: the mixed batch, strong scaling, the same workload's one-GPU rate (n1) beside it."""
    sp = out["scaling_point"]
    assert POINT_KEYS <= set(sp), sorted(POINT_KEYS - set(sp))
    assert sp["scaling"] == "strong" and sp["n_gpus"] == n and sp["total_queries"] == total and sp["workload"].startswith("cfg5")
    assert sp["per_gpu_value"] * n == pytest.approx(sp["value"]) and set(sp["hbm_bytes_in_use"]) == {"engine_pool_in_use", "engine_pool_idle", "device_in_use", "device_total"}
    n1 = sp["n1"]
    assert n1["n_gpus"] == 1 and n1["total_queries"] == total and n1["workload"] == sp["workload"] and n1["value"] > 0
    assert sp["speedup_vs_n1"] == pytest.approx(sp["value"] / n1["value"])
    assert (sp["gather_ms"] is None) == (n == 1)
    assert "hbm_bytes_in_use" in out


def test_two_ranks_default_workload_is_the_mixed_batch_split_over_the_ranks():
    out = run_bench(2, ["--queries", "1000"])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["scaling"] == "strong"  # (north_star: ONE batch sharded over the GPUs)
    assert out["config"]["workload"].startswith("cfg5") and out["config"]["queries_per_gpu_per_step"] == 500 and out["config"]["queries_per_step"] == 1000
    assert [b["queries"] for b in out["config"]["batches_per_step"]] == [350, 150]
    assert out["gather_check"] == {"ranks": 2, "blocks": ["counts", "docs", "scores", "topk_counts"], "equal_on_every_rank": True}
    assert out["per_gpu_value"] * 2 == out["value"] and out["gather_ms"] >= 0
    check_scaling_point(out, 2, 1000)
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline", "kernels_only", "end_to_end"):
        assert k in out


def test_eight_ranks_carry_the_same_block():
    out = run_bench(8, ["--queries", "1600"])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["queries_per_gpu_per_step"] == 200
    check_scaling_point(out, 8, 1600)


def test_one_gpu_line_carries_the_mixed_batch_point_beside_the_cfg2_headline():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dry-run", "--docs", "100000", "--vocab", "10000", "--queries", "512", "--scaling-total", "800",
           "--scaling-ref-steps", "2"]  # fmt: skip
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["config"]["workload"].startswith("cfg2")  # the headline stays BASELINE.json's configs[1]
    check_scaling_point(out, 1, 800)
    assert out["scaling_point"]["speedup_vs_n1"] == 1.0


def test_weak_and_strong_scaling_on_request():
    out = run_bench(2, ["--queries", "1000", "--scaling", "strong", "--workload", "cfg2", "--scaling-ref-steps", "0"])
    assert out["scaling"] == "strong" and out["config"]["queries_per_gpu_per_step"] == 500 and out["config"]["queries_per_step"] == 1000 and out["scaling_point"]["n1"] is None
    out = run_bench(2, ["--queries", "600", "--scaling", "weak", "--scaling-ref-steps", "0"])
    assert out["scaling"] == "weak" and out["config"]["queries_per_gpu_per_step"] == 600 and out["config"]["queries_per_step"] == 1200


def test_reference_cpu_leg_runs_the_genuine_reference_on_the_sample():
    """bench.py's cpu_baseline of kind "reference": the sampled programs as query text through oracle/_ref/ref_driver's `timed` command (exec_query of the
    reference compiled from its own sources) — the match counts it reports equal the oracle's for conjunctions, unions, phrases, exclusions and
    optional parts, unscored and scored; a program without a text (matchsome) is refused, a missing driver is an error entry, not an exception."""
    import importlib.util
    from types import SimpleNamespace

    import numpy as np
    import oracle_lib as O
    import trinity_amd as T

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.program_text(O.parse_query("t1 t2")) == "t1 t2" and bench.program_text(O.parse_query("[t0, t1, t2]", some_min=2)) is None
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_driver")):
        pytest.skip("oracle/_ref/ref_driver is built only where /root/reference exists")
    D, V = 20000, 2000
    seg = T.Segment(D, V, 10, 42)
    ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
    texts = ["t0 t1", "t3 OR t5 OR t9", "t0 t1 (t2 OR t3 OR t4)", "t3 t5 NOT t1", '"t0 t1" t2', '"t4 t5 t6"', "t0 <t7>", "t1 NOT (t2 OR t3)", "t11 t400", "t1999 t3"]
    progs = [O.parse_query(t) for t in texts]
    for flags, oflag in ((1, O.FLAG_DOCUMENTS_ONLY), (2, O.FLAG_ACCUM_SCORE)):
        want = [len(ora.exec(p, oflag)[0]) for p in progs]
        r = bench.cpu_reference(seg, SimpleNamespace(flags=flags), progs, np.array(want), 5.0)
        assert "error" not in r, r
        assert r["kind"] == "reference" and r["match_counts_equal_gpu"] and r["value"] > 0 and f"first {len(progs)} queries" in r["sample"]
    assert "error" in bench.cpu_reference(seg, SimpleNamespace(flags=1), [O.parse_query("[t0, t1, t2]", some_min=2)], np.array([0]), 1.0)


def test_ranks_of_a_node_pin_their_planner_threads_to_disjoint_cpus():
    """Eight ranks started side by side (LOCAL_RANK / LOCAL_WORLD_SIZE as torch.distributed.run exports them): each one's planner pool pins its
    workers inside its own slice of the affinity mask — disjoint from every other rank's, whatever CPU the creating thread runs on (round 4's
    pool took the CPUs next to the creator's: ranks landing within 16 CPUs of each other stacked their spinning workers)."""
    ncpu = len(os.sched_getaffinity(0))
    world = 8 if ncpu >= 16 else max(1, min(4, ncpu // 2))  # (slices of two CPUs or more: a one-CPU slice gets no worker at all — below)
    seen = {}
    for r in range(world):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world))
        res = subprocess.run([sys.executable, "-c", "import json, sys; sys.path.insert(0, %r); from trinity_amd import hostplan as HP; print(json.dumps(HP.pool_cpus(16)))" % ROOT],
                             capture_output=True, text=True, timeout=120, env=env)  # fmt: skip
        assert res.returncode == 0, res.stderr[-2000:]
        seen[r] = json.loads(res.stdout.strip().splitlines()[-1])
    allowed = sorted(os.sched_getaffinity(0))
    for r, cpus in seen.items():
        lo, hi = len(allowed) * r // world, len(allowed) * (r + 1) // world
        assert cpus and set(cpus) <= set(allowed[lo:hi]) and len(set(cpus)) == len(cpus) == min(15, hi - lo), (r, cpus)
    flat = [c for cpus in seen.values() for c in cpus]
    assert len(flat) == len(set(flat))  # no CPU carries two ranks' pollers
    # a slice of ONE CPU: no worker (it would poll on the CPU the unpinned caller runs on) — the caller plans alone
    env = dict(os.environ, LOCAL_RANK="0", LOCAL_WORLD_SIZE=str(ncpu))
    res = subprocess.run([sys.executable, "-c", "import json, sys; sys.path.insert(0, %r); from trinity_amd import hostplan as HP; print(json.dumps(HP.pool_cpus(16)))" % ROOT],
                         capture_output=True, text=True, timeout=120, env=env)  # fmt: skip
    assert res.returncode == 0 and (ncpu < 2 or json.loads(res.stdout.strip().splitlines()[-1]) == []), res.stdout[-500:] + res.stderr[-1000:]
    # the CPU budget the pools are sized to: the mask capped by the container's quota, a rank's share of it under a launcher
    def budget(env):
        res = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from trinity_amd import hostplan as HP; print(HP.cpu_budget())" % ROOT],
                             capture_output=True, text=True, timeout=120, env=env)  # fmt: skip
        assert res.returncode == 0, res.stderr[-2000:]
        return int(res.stdout.strip().splitlines()[-1])

    solo_env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    whole = budget(solo_env)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    assert 1 <= whole <= ncpu and (quota is None or whole == min(ncpu, quota))
    assert budget(dict(solo_env, LOCAL_RANK="1", LOCAL_WORLD_SIZE="4")) == max(1, whole // 4)
    # without the launcher's variables: next to the creating thread, still distinct CPUs
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    res = subprocess.run([sys.executable, "-c", "import json, sys; sys.path.insert(0, %r); from trinity_amd import hostplan as HP; print(json.dumps(HP.pool_cpus(4)))" % ROOT],
                         capture_output=True, text=True, timeout=120, env=env)  # fmt: skip
    solo = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(solo) == min(3, ncpu) and len(set(solo)) == len(solo)
