"""The engine's ISA, checked without a GPU: every s_barrier must be behind an `s_waitcnt lgkmcnt(0)` on every path from the wave's last LDS
store (tools/isa_barrier_check.py).  Round 4 found one the compiler left out — a barrier at the head of a task loop, reached over the back edge
from the LDS stores that publish the next task's record: the other waves read the previous record's words."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="hipcc not found")
def test_every_barrier_is_behind_an_lds_wait():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_barrier_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 barrier(s) without an LDS wait" in r.stdout


def test_the_checker_sees_a_barrier_reached_over_a_back_edge(tmp_path):
    """The checker on hand-written ISA: an LDS store at the end of a loop body, the barrier at the loop's head with no wait in between (the round-4
    bug's shape) is reported; the same loop with `s_waitcnt lgkmcnt(0)` on the back edge, or with the store before a waited barrier, is not."""
    bad = """
_Z3badv:
	s_mov_b32 s0, 0
.LBB0_1:
	s_barrier
	ds_read_b32 v1, v0
	s_waitcnt lgkmcnt(0)
	v_add_u32_e32 v1, 1, v1
	s_cmp_lt_u32 s0, 4
	s_cbranch_scc1 .LBB0_3
	s_endpgm
.LBB0_3:
	ds_write_b32 v0, v1
	s_add_i32 s0, s0, 1
	s_branch .LBB0_1
.Lfunc_end0:
"""
    good = bad.replace("\ts_add_i32 s0, s0, 1\n\ts_branch .LBB0_1", "\ts_add_i32 s0, s0, 1\n\ts_waitcnt lgkmcnt(0)\n\ts_branch .LBB0_1")
    tool = os.path.join(ROOT, "tools", "isa_barrier_check.py")
    for text, rc in ((bad, 1), (good, 0)):
        p = tmp_path / f"k{rc}.s"
        p.write_text(text)
        r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
        assert r.returncode == rc, r.stdout
        assert ("reachable from `ds_write_b32 v0, v1`" in r.stdout) == bool(rc)
