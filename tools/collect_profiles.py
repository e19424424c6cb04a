#!/usr/bin/env python3
"""Copy what a tools/gpu_round.sh run left under gpurun_out/<tag>/ into profiles/ (tracked): per workload the bench line, the
rocprofv3 kernel stats and the per-kernel PMC traffic, named profiles/<round>_{bench,kernel_stats,pmc}_<workload>.*, and rebuild
profiles/pmc_latest.json (what bench.py quotes as roofline.traffic, with its source).
    usage: tools/collect_profiles.py <gpurun_out tag> <round prefix, e.g. r02>"""
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
latest_path = os.path.join(dst, "pmc_latest.json")
latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
sys.path.insert(0, root)
from trinity_amd.build import kernels_stamp  # noqa: E402

stamp = kernels_stamp()  # (the tree the profiles were collected from: run this before touching the kernels again)
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
for f in sorted(os.listdir(src)):
    p = os.path.join(src, f)
    if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{f}"))
    elif f.startswith("kernel_stats_") and f.endswith(".csv"):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{f}"))
    elif f.startswith("pmc_") and f.endswith(".json") and "_FETCH_SIZE" not in f and "_WRITE_SIZE" not in f:
        name = f[4:-5]
        table = json.load(open(p))
        shutil.copy(p, os.path.join(dst, f"{rnd}_{f}"))
        bench = os.path.join(src, f"pmc_{name}_FETCH_SIZE.json")
        cfg = {}
        try:
            cfg = json.loads(open(bench).read().strip().splitlines()[-1])["config"]
        except Exception:
            pass
        latest[f"bench_{name}"] = {
            "collected": f"{rnd}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --workload {name}; profiles/{rnd}_{f}",
            "kernels_stamp": stamp, "git_head": head,
            "docs": cfg.get("docs"), "vocab": cfg.get("vocab"), "queries": cfg.get("queries_per_gpu_per_step"),
            "kernels": {k: {"traffic_bytes_per_launch": v["traffic_bytes_per_launch"], "hbm_read_bytes_per_launch_corrected": v["hbm_read_bytes_per_launch_corrected"],
                            "hbm_write_bytes_per_launch": v["hbm_write_bytes_per_launch"], "dispatches": v["dispatches"]} for k, v in table.items() if k.startswith("k_")},
        }  # fmt: skip
json.dump(latest, open(latest_path, "w"), indent=1)
print("profiles/ updated:", sorted(x for x in os.listdir(dst) if x.startswith(rnd)))
