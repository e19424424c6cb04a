"""The host planner's thread pool under ThreadSanitizer (no GPU): libtrinity_host.so's sources (csrc/host/plan_host.cpp -> csrc/planner.hpp, csrc/host_pool.hpp:
pinned polling workers, jobs taken by a CAS on (generation, index), four parallel passes per plan) built with -fsanitize=thread plan every workload's
batches on 8 threads in a child interpreter with the TSan runtime preloaded.  A data race between the fragments' passes or in the pool's hand-over is a
report on stderr — there must be none."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = """
import sys
sys.path.insert(0, %r)
import trinity_amd as T
from trinity_amd import hostplan as HP, workloads as W
D, V = 300_000, 30_000
for codec in (1, 2):
    seg = T.Segment(D, V, 10, 42, codec=codec)
    hi = HP.HostIndex.from_segment(seg)
    for wl in ("cfg2", "cfg3", "cfg4", "cfg5"):
        parts, _ = W.build_parts(wl, D, V, 10, 42, 4096)
        for pt in parts:
            if pt.codec != codec:
                continue
            for rep in range(3):
                p = HP.HostPlan(hi, pt.programs, pt.flags, pt.topk, threads=8, options={"frag_cache": rep & 1})  # (every other plan on recycled fragments)
                assert p.s["n_tasks"] > 0
                p.close()
print("planned ok")
"""


def _runtime():
    p = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_runtime() is None, reason="gcc's ThreadSanitizer runtime not found")
def test_planner_threads_are_race_free_under_tsan(tmp_path):
    from trinity_amd.build import HOST_SRCS

    lib = str(tmp_path / "libtrinity_host_tsan.so")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-pthread", "-fsanitize=thread", "-o", lib] + HOST_SRCS, check=True)
    env = dict(os.environ)
    env.update(LD_PRELOAD=_runtime(), TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66", TRINITY_HOST_LIB=lib)
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
    assert "planned ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ThreadSanitizer" not in r.stderr and r.returncode == 0, r.stderr[-4000:]


ORACLE_SCRIPT = """
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
import oracle_lib as O
ora = O.Index.generate(50000, 3000, 10, 42)
rng = np.random.default_rng(1)
pa = np.array([[O.tok(O.OP_TERM, int(a)), O.tok(O.OP_TERM, int(b)), O.tok(O.OP_AND, 2)] for a, b in rng.integers(0, 300, size=(400, 2))], dtype=np.uint32)
done, m, dt = ora.exec_batch_mt(pa, O.FLAG_DOCUMENTS_ONLY, 8, 30.0)
single = sum(len(ora.exec(p, O.FLAG_DOCUMENTS_ONLY)[0]) for p in pa)
assert done == len(pa) and m == single, (done, m, single)
print("batch ok")
"""


@pytest.mark.skipif(_runtime() is None, reason="gcc's ThreadSanitizer runtime not found")
def test_oracle_all_cores_leg_is_race_free_under_tsan(tmp_path):
    """The CPU baseline's all-cores leg (oracle to_exec_batch_mt: one query per thread from a shared cursor over one read-only index) built with
    ThreadSanitizer: no report, and the threads' matches add up to the single-thread total."""
    lib = str(tmp_path / "liboracle_tsan.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("trinity_oracle.c", "trinity_oracle_lucene.c", "fastpfor128.c")]
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-fsanitize=thread", "-shared", "-o", lib] + srcs + ["-lm", "-lpthread"], check=True)
    env = dict(os.environ)
    env.update(LD_PRELOAD=_runtime(), TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66", TRINITY_ORACLE_LIB=lib)
    r = subprocess.run([sys.executable, "-c", ORACLE_SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert "batch ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ThreadSanitizer" not in r.stderr and r.returncode == 0, r.stderr[-4000:]
