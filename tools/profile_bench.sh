#!/bin/bash
# Run on the GPU box from the repo root (gpurun):  tools/profile_bench.sh <tag>
#   1. rocprofv3 --kernel-trace --stats over `bench.py --steps 5 --warmup 2` -> gpurun_out/prof_<tag>/kernel_stats.csv + bench line
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 2 --warmup 1` -> gpurun_out/prof_<tag>/pmc_bench.json
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
tag="${1:-latest}"
root="$(pwd)"
out="$root/gpurun_out/prof_$tag"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python "$root/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 > "$out/bench_under_rocprof.json" 2> "$out/trace.err")
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
"$root/tools/pmc_kernels.sh" "FETCH_SIZE WRITE_SIZE" python "$root/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 > "$out/pmc_table.txt" 2>&1
python3 - "$root" "$out" <<'PY'
import json, sys, os
root, out = sys.argv[1], sys.argv[2]
s = json.load(open(os.path.join(root, "gpurun_out", "pmc", "summary.json")))
kern = {}
for full, v in s["FETCH_SIZE"].items():
    name = "k_and_dense" if "k_and_dense" in full else "k_and" if "k_and<" in full else None
    if not name:
        continue
    w = s["WRITE_SIZE"][full]
    rd = v["per_dispatch"] * 1024 * 2  # KiB; x2: gfx950 FETCH_SIZE correction (MI355X guide; calibration in profiles/pmc_latest.json)
    wr = w["per_dispatch"] * 1024
    kern[name] = {"FETCH_SIZE_KiB_per_launch": v["per_dispatch"], "WRITE_SIZE_KiB_per_launch": w["per_dispatch"], "dispatches": v["dispatches"],
                  "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr}
json.dump({"command": "python bench.py --steps 2 --warmup 1 --cpu-seconds 0", "docs": 10000000, "vocab": 1000000, "queries": 16384, "kernels": kern},
          open(os.path.join(out, "pmc_bench.json"), "w"), indent=1)
print(json.dumps(kern, indent=1))
PY
rm -rf "$out/trace"
head -4 "$out/kernel_stats.csv"; cat "$out/bench_under_rocprof.json"
