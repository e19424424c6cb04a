python -m pytest tests -m gpu -x -q -k "fused or scored or cfg3 or workload or or_and_mixed or general_trees_scored or other_similarities or optional or not_scored" 2>&1 | tail -2
WORKLOAD=cfg3 NQ=8192 RUNS=3 python tools/probe_workload.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-330
