#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/pmc_kernels.sh "<counters, space separated>" <command...>
# One rocprofv3 --pmc pass per counter (never combined with runtime/sys traces), CSV output under gpurun_out/pmc/<counter>/,
# then a per-kernel summary (sum over dispatches and per-dispatch mean) printed and written to gpurun_out/pmc/summary.json.
set -u
counters="$1"; shift
root="$(pwd)"
export TMPDIR=/tmp
mkdir -p "$root/gpurun_out/pmc"
for c in $counters; do
        out="$root/gpurun_out/pmc/$c"
        rm -rf "$out"; mkdir -p "$out"
        (cd /tmp && timeout 300 rocprofv3 --pmc "$c" --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/stdout.log" 2> "$out/stderr.log")
done
python3 - "$root/gpurun_out/pmc" $counters <<'PY'
import csv, glob, json, os, sys, collections
base, counters = sys.argv[1], sys.argv[2:]
summary = {}
for c in counters:
    per = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(base, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            k = r["Kernel_Name"].split("(")[0]
            per[k][0] += float(r["Counter_Value"]); per[k][1] += 1
    # a dispatch contributes one row per counter instance dimension; count dispatches via Dispatch_Id
    disp = collections.defaultdict(set)
    for f in glob.glob(os.path.join(base, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                disp[r["Kernel_Name"].split("(")[0]].add(r["Dispatch_Id"])
    summary[c] = {k: {"sum": v[0], "dispatches": len(disp[k]), "per_dispatch": v[0] / max(1, len(disp[k]))} for k, v in per.items()}
json.dump(summary, open(os.path.join(base, "summary.json"), "w"), indent=1)
for c in counters:
    for k, v in sorted(summary[c].items(), key=lambda kv: -kv[1]["sum"])[:6]:
        print(f"{c:28s} {k[:60]:60s} n={v['dispatches']:4d} per_dispatch={v['per_dispatch']:.4g}")
PY
