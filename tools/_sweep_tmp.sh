for wl in cfg1:2000 cfg3:8192 cfg2:16384; do
WORKLOAD=${wl%%:*} NQ=${wl##*:} $( [ ${wl%%:*} = cfg1 ] && echo "DOCS=100000 VOCAB=10000" ) RUNS=4 env $( [ ${wl%%:*} = cfg1 ] && echo "DOCS=100000 VOCAB=10000" ) python tools/probe_workload.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-330
done
