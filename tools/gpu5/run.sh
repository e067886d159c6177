#!/bin/bash
# One parameterised GPU-box job script for round 5: `gpurun -- 'bash tools/gpu5/run.sh <job> [args]'`.
# Outputs go to gpurun_out/r05/ (scratch); summaries worth keeping are copied into profiles/ by hand.
# Every rocprofv3 call puts python3 itself after `--` (no wrapper hop); --pmc passes carry --kernel-trace only.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p "$OUT"
export TMPDIR=/tmp
job="$1"; shift
case "$job" in
  tests)        # tests <label> <pytest args...>   (NOX=-q: do not stop at the first failure)
    label="$1"; shift
    timeout -k 10 1100 python3 -m pytest "$@" ${NOX:--x} -q -m gpu --durations=${DUR:-25} --durations-min=0.2 -p no:cacheprovider > "$OUT/tests_$label.log" 2>&1
    rc=$?; tail -n 40 "$OUT/tests_$label.log"; exit $rc ;;
  bench)        # bench <label> <bench.py args...>
    label="$1"; shift
    timeout -k 10 900 python3 bench.py "$@" > "$OUT/bench_$label.json" 2> "$OUT/bench_$label.err"
    rc=$?; tail -c 1500 "$OUT/bench_$label.json"; tail -n 5 "$OUT/bench_$label.err"; exit $rc ;;
  stats)        # stats <label> <bench.py args...>: rocprofv3 --kernel-trace --stats of the same command
    label="$1"; shift
    rm -rf "$OUT/prof_$label"
    timeout -k 10 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$label" -o "$label" --output-format csv -- python3 bench.py "$@" > "$OUT/stats_$label.json" 2> "$OUT/stats_$label.err"
    rc=$?; find "$OUT/prof_$label" -name '*kernel_stats.csv' | head -n 1 | xargs -r head -n 30; exit $rc ;;
  pmc)          # pmc <label> <model> <workload> <kernel substring> <bench.py args...>: FETCH_SIZE and WRITE_SIZE, one pass each
    label="$1"; model="$2"; workload="$3"; kernel="$4"; shift 4
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf "$OUT/pmc_${label}_$c"
      timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_${label}_$c" -o "$label" --output-format csv -- python3 bench.py "$@" > "$OUT/pmc_${label}_$c.json" 2> "$OUT/pmc_${label}_$c.err" || exit $?
    done
    python3 tools/profile_summary.py pmc "$OUT/pmc_${label}_FETCH_SIZE" "$OUT/pmc_${label}_WRITE_SIZE" --out "$OUT/pmc_${label}_hbm.csv" --cmd "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py $*"
    python3 tools/profile_summary.py traffic --fetch "$OUT/pmc_${label}_FETCH_SIZE" --write "$OUT/pmc_${label}_WRITE_SIZE" --kernel "$kernel" --workload "$workload" --model "$model" --measured "$(date -u +%Y-%m-%d), round 5 (tools/gpu5/run.sh pmc), one MI355X box of the pool ($(hostname)); python3 bench.py $*" --out "$OUT/pmc_traffic_${workload}_${model}.json"
    head -n 20 "$OUT/pmc_${label}_hbm.csv"; cat "$OUT/pmc_traffic_${workload}_${model}.json" ;;
  py)           # py <label> <script args...>
    label="$1"; shift
    timeout -k 10 1100 python3 "$@" > "$OUT/py_$label.log" 2>&1
    rc=$?; tail -n 60 "$OUT/py_$label.log"; exit $rc ;;
  smoke)
    timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 5 ;;
  *) echo "unknown job $job"; exit 2 ;;
esac
