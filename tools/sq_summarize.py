#!/usr/bin/env python3
"""Mean per dispatch of every counter rocprofv3 --pmc collected under <dir> (tools/gpu_round.sh sq:), per kernel."""
import collections
import csv
import glob
import os
import sys

per, disp = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
for k in sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", 0.0)):
    n = max(1, len(disp[k]))
    print(f"{k:48s} n={n}")
    for c, v in sorted(per[k].items()):
        print(f"    {c:24s} {v / n:16.0f}")
