for dw in 0 4096 16384 49152; do
echo "== dense_window_cost $dw"
python bench.py --workload cfg5 --steps 4 --warmup 2 --cpu-seconds 0 --scaling-ref-steps 0 --option dense_window_cost=$dw 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print(round(j['value']), 'ms', round(j['ms_per_step'],3), 'kern', round(j['kernels_only']['ms_per_step'],3), {k:round(v['kernel_ms'],3) for k,v in r['kernels'].items()})"
done
