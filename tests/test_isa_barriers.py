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
