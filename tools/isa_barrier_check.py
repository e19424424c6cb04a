#!/usr/bin/env python3
"""Every s_barrier of the engine's kernels must be preceded, on EVERY path, by `s_waitcnt lgkmcnt(0)` after the wave's last LDS store / atomic —
otherwise another wave may read LDS behind the barrier before the store has landed (hazard 1: the compiler left that wait out
at a barrier reached over a loop's back edge).  Walks the control-flow graph of `hipcc -S --cuda-device-only` output backwards from each barrier.
    usage: tools/isa_barrier_check.py [file.s]     (without a file: compiles trinity_amd/csrc/trinity_hip.hip to build/isa_check.s first)"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    os.makedirs(os.path.join(root, "build"), exist_ok=True)
    path = os.path.join(root, "build", "isa_check.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value", "-o", path,
                    os.path.join(root, "trinity_amd", "csrc", "trinity_hip.hip")] + [a for a in os.environ.get("ISA_CHECK_FLAGS", "").split() if a], check=True)  # fmt: skip
LDS_WRITE = re.compile(r"^ds_(write|or|and|xor|add|sub|min|max|inc|dec|cmpst|wrxchg|append|consume)")
kernels, cur = {}, None
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = kernels.setdefault(m.group(1), [])
        cur.append(["entry", []])
        continue
    if cur is None:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        cur.append([m.group(1), []])
        continue
    t = line.strip()
    if t.startswith(".Lfunc_end"):
        cur = None
    elif t and not t.startswith((";", ".")):
        cur[-1][1].append(t.split(";")[0].strip())
bad = 0
for name, blocks in kernels.items():
    index = {b[0]: i for i, b in enumerate(blocks)}
    preds = {i: set() for i in range(len(blocks))}
    for i, (_, ins) in enumerate(blocks):
        fall = True
        for t in ins:
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in index:
                preds[index[m.group(1)]].add(i)
            if t.startswith(("s_branch", "s_endpgm", "s_setpc")):
                fall = False
        if fall and i + 1 < len(blocks):
            preds[i + 1].add(i)
    for i, (label, ins) in enumerate(blocks):
        for k, t in enumerate(ins):
            if not t.startswith("s_barrier"):
                continue
            # backwards from (i, k): a path ends at a full LDS wait; it is a finding when it meets an LDS store first
            seen, work, hit = set(), [(i, k)], None
            while work and hit is None:
                bi, upto = work.pop()
                done = False
                for t2 in reversed(blocks[bi][1][:upto]):
                    if t2.startswith("s_waitcnt") and "lgkmcnt(0)" in t2:
                        done = True
                        break
                    if LDS_WRITE.match(t2):
                        hit = (blocks[bi][0], t2)
                        break
                if done or hit:
                    continue
                for p in preds[bi]:
                    if p not in seen:
                        seen.add(p)
                        work.append((p, len(blocks[p][1])))
            if hit:
                bad += 1
                print(f"{name[:70]}: s_barrier in {label} reachable from `{hit[1]}` ({hit[0]}) without s_waitcnt lgkmcnt(0)")
print(f"{sum(len(b) for b in kernels.values())} blocks in {len(kernels)} functions; {bad} barrier(s) without an LDS wait on some path")
sys.exit(1 if bad else 0)
