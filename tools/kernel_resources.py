#!/usr/bin/env python3
"""Registers, spills, scratch, LDS and the occupancy the compiler states for every kernel of the engine (no GPU): `hipcc -Rpass-analysis=kernel-resource-usage`
on trinity_hip.hip, one row per kernel.    usage: tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "build"), exist_ok=True)
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-fPIC", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage",
                    os.path.join(root, "trinity_amd", "csrc", "trinity_hip.hip"), "-o", os.path.join(root, "build", "resources.o")], capture_output=True, text=True)  # fmt: skip
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
demangle = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch B':>9s} {'LDS B':>7s} {'waves/SIMD':>10s}")
for x, d in zip(rows, demangle):
    short = re.sub(r"\(.*", "", d).replace("void ", "")
    print(f"{short[:44]:44s} {x.get('VGPRs', '?'):>5s} {x.get('AGPRs', '?'):>5s} {x.get('TotalSGPRs', '?'):>5s} {x.get('VGPRs Spill', '?'):>6s} {x.get('SGPRs Spill', '?'):>6s} "
          f"{x.get('ScratchSize', '?'):>9s} {x.get('LDS Size', '?'):>7s} {x.get('Occupancy', '?'):>10s}")
