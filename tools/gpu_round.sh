#!/bin/bash
# One gpurun call's worth of work on the GPU box (run from the repo root):  tools/gpu_round.sh <tag> <stage>...
# Every stage writes under gpurun_out/<tag>/ and is bounded by its own timeout; a failing early stage stops the expensive ones.
set -u
tag="$1"; shift
root="$(pwd)"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
export TMPDIR=/tmp
log() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$out/stages.log"; }
for stage in "$@"; do
  log "stage $stage"
  case "$stage" in
    smoke)
      timeout -k 10 420 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; rc=$?
      tail -3 "$out/smoke.log"; log "smoke rc=$rc"; [ $rc -ne 0 ] && exit 1 ;;
    test:*)
      sel="${stage#test:}"
      timeout -k 10 1500 python -m pytest tests -m gpu -x -q -k "$sel" --timeout 900 > "$out/test_$(echo "$sel" | tr ' ' '_').log" 2>&1; rc=$?
      tail -15 "$out/test_$(echo "$sel" | tr ' ' '_').log"; log "tests[$sel] rc=$rc"; [ $rc -ne 0 ] && exit 1 ;;
    tests)
      timeout -k 10 2400 python -m pytest tests -m gpu -x -q --timeout 900 --durations=15 > "$out/tests_all.log" 2>&1; rc=$?
      tail -25 "$out/tests_all.log"; log "tests rc=$rc"; [ $rc -ne 0 ] && exit 1 ;;
    testopt:*)
      # testopt:<name>:<TRINITY_TEST_OPTIONS with ; for ,>:<-k expr> — the GPU suite (or a selection) under planner options that force one path
      name="$(echo "$stage" | cut -d: -f2)"; topts="$(echo "$stage" | cut -d: -f3 | tr ';' ',')"; sel="$(echo "$stage" | cut -d: -f4-)"
      TRINITY_TEST_OPTIONS="$topts" timeout -k 10 1500 python -m pytest tests -m gpu -x -q ${sel:+-k "$sel"} --timeout 900 > "$out/testopt_$name.log" 2>&1; rc=$?
      tail -12 "$out/testopt_$name.log"; log "testopt[$name: $topts] rc=$rc"; [ $rc -ne 0 ] && exit 1 ;;
    bench:*)
      # bench:<name>:<args with , for spaces>
      name="$(echo "$stage" | cut -d: -f2)"; args="$(echo "$stage" | cut -d: -f3- | tr ',' ' ')"
      timeout -k 10 900 python bench.py $args > "$out/bench_$name.json" 2> "$out/bench_$name.err"; rc=$?
      cat "$out/bench_$name.json"; tail -3 "$out/bench_$name.err"; log "bench $name rc=$rc" ;;
    prof:*)
      name="$(echo "$stage" | cut -d: -f2)"; args="$(echo "$stage" | cut -d: -f3- | tr ',' ' ')"
      rm -rf "$out/trace_$name"
      (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$name" -- python "$root/bench.py" $args --cpu-seconds 0 --scaling-ref-steps 0 --rotating-sets 0 --delivered-steps 0 > "$out/prof_$name.json" 2> "$out/prof_$name.err")
      find "$out/trace_$name" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_$name.csv" \;
      rm -rf "$out/trace_$name"
      head -12 "$out/kernel_stats_$name.csv"; log "prof $name done" ;;
    pmc:*)
      name="$(echo "$stage" | cut -d: -f2)"; args="$(echo "$stage" | cut -d: -f3- | tr ',' ' ')"
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -rf "$out/pmc_${name}_$ctr"
        (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_${name}_$ctr" -- python "$root/bench.py" $args --cpu-seconds 0 --scaling-ref-steps 0 --rotating-sets 0 --delivered-steps 0 > "$out/pmc_${name}_$ctr.json" 2> "$out/pmc_${name}_$ctr.err")
      done
      python3 "$root/tools/pmc_summarize.py" "$out" "$name" | tee "$out/pmc_$name.txt"; log "pmc $name done" ;;
    sq:*)
      # SQ issue / stall counters of every kernel (one pass, 8 SQ slots): where the waves' cycles go
      name="$(echo "$stage" | cut -d: -f2)"; args="$(echo "$stage" | cut -d: -f3- | tr ',' ' ')"
      rm -rf "$out/sq_$name"
      (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --pmc ${SQ_CTRS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS} --output-format csv -d "$out/sq_$name" -- python "$root/bench.py" $args --cpu-seconds 0 --scaling-ref-steps 0 --rotating-sets 0 --delivered-steps 0 > "$out/sq_$name.json" 2> "$out/sq_$name.err")
      python3 "$root/tools/sq_summarize.py" "$out/sq_$name" | tee "$out/sq_$name.txt"; rm -rf "$out/sq_$name"; log "sq $name done" ;;
    cmd:*)
      c="${stage#cmd:}"; timeout -k 10 900 bash -c "$c" > "$out/cmd_$(echo "$c" | md5sum | cut -c1-8).log" 2>&1; log "cmd rc=$?" ;;
  esac
done
log "all stages done"
