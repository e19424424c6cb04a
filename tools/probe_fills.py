#!/usr/bin/env python3
"""Who issues the multi-millisecond __amd_rocclr_fillBufferAligned / copyBuffer dispatches a rocprofv3 kernel trace of bench.py shows, and do they overlap a timed step?
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --workload cfg4 ...);  python tools/probe_fills.py OUT
Reads the per-dispatch trace (start / end timestamps, grid size), lists every fill / copy dispatch longer than 0.2 ms with the bytes its grid covers and the
engine kernels running right before and after it, and says how many of them start while an engine kernel of a STEP (k_psets, k_and, k_planes, k_phrase ...) is running."""
import csv
import glob
import os
import sys

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)))
rows.sort()
t0 = rows[0][0]
short = lambda n: n.split("(")[0].replace("void ", "")[:28]
engine = [r for r in rows if r[2].startswith("void k_") or r[2].startswith("k_")]
big = [i for i, r in enumerate(rows) if ("fillBuffer" in r[2] or "copyBuffer" in r[2]) and r[1] - r[0] > 200_000]
print(f"{len(rows)} dispatches, {len(engine)} engine kernels, {len(big)} fill / copy dispatches over 0.2 ms")
inside = 0
for i in big:
    s, e, name, grid, wg = rows[i]
    prev = next((short(rows[j][2]) for j in range(i - 1, -1, -1) if rows[j][2] != name), "-")
    nxt = next((short(rows[j][2]) for j in range(i + 1, len(rows)) if rows[j][2] != name), "-")
    over = [short(r[2]) for r in engine if r[0] < e and r[1] > s]
    inside += bool(over)
    print(f"  t={1e-6 * (s - t0):10.3f} ms  {short(name):24s} {1e-6 * (e - s):7.3f} ms  grid {grid:>10d} x wg {wg:<4d}  after {prev:28s} before {nxt:28s} overlaps {','.join(sorted(set(over))) or '-'}")
print(f"{inside} of {len(big)} overlap an engine kernel")
first_step = next((r[0] for r in engine if "k_phrase" in r[2] or "k_planes" in r[2] or "k_psets" in r[2]), None)
if first_step:
    print(f"first engine step kernel at t={1e-6 * (first_step - t0):.3f} ms; big fills / copies before it: {sum(1 for i in big if rows[i][0] < first_step)}")
