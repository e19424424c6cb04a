mkdir -p gpurun_out/r9c
O=gpurun_out/r9c/psets.txt
for w in "cfg2 16384" "or5 1250"; do set -- $w
  echo "== $1 $2" >> $O
  WORKLOAD=$1 NQ=$2 RUNS=3 timeout 300 python tools/probe_workload.py 2>&1 | tail -2 >> $O
done
cat $O
timeout 900 python -m pytest tests -x -q -m gpu -k "docset or fullsize or parity or union or scatter or masked" 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r9c/tests.txt
