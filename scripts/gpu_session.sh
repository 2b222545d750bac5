#!/bin/bash
# ONE parameterised runner for a GPU-box session (round 6; replaces the per-call scratch scripts of round 5, which are in the history at
# c12e85f).  Usage, from the repo root through gpurun:   bash scripts/gpu_session.sh <tag> <stage> [<stage> ...]
# Every stage writes under gpurun_out/<tag>/ and prints a short summary; copy what should be judged into profiles/.
#   tests            the whole -m gpu suite            tests:<expr>   pytest -k <expr> (e.g. tests:small_call)
#   smoke            __graft_entry__.smoke()
#   latency[:args]   scripts/small_call_latency.py (comma-free args after ':' are passed through, '+' separates them)
#   bench[:args]     bench.py with the given args ('+'-separated) -> bench_<n>.json (stdout line), bench_<n>_detail.json
#   trace:<name>:<bench args>   rocprofv3 --kernel-trace of ONE call (scripts/prof_bench.sh) -> <name>.txt
#   evidence[:stages] scripts/evidence.sh into gpurun_out/<tag>/evidence ('+'-separated stages: trace pmc ragged mfma c0; default all but ragged)
set -u
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
n=0
for STAGE in "$@"; do
  KIND=${STAGE%%:*}; REST=""; [[ "$STAGE" == *:* ]] && REST=${STAGE#*:}
  echo "=== stage $STAGE"
  case "$KIND" in
    tests)
      if [ -n "$REST" ]; then
        timeout 1500 python -m pytest tests -m gpu -q -rP -k "$REST" > "$OUT/pytest_${REST//[^a-zA-Z0-9_]/_}.log" 2>&1; echo "pytest -k '$REST' exit $?"
        grep -E "^(FAILED|ERROR)|passed|failed|^envelope |^transform errors|\[float32x2\]|\[bfloat16\]: vel" "$OUT/pytest_${REST//[^a-zA-Z0-9_]/_}.log" | tail -40
      else
        timeout 1700 python -m pytest tests -m gpu -q --durations=40 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?"
        grep -E "^(FAILED|ERROR)|passed|failed|s call" "$OUT/pytest_gpu.log" | tail -30
      fi ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?"; tail -4 "$OUT/smoke.log" ;;
    latency)
      timeout 900 python scripts/small_call_latency.py ${REST//+/ } > "$OUT/latency_$n.jsonl" 2> "$OUT/latency_$n.err"; echo "latency exit $?"
      cat "$OUT/latency_$n.jsonl"; tail -2 "$OUT/latency_$n.err" ;;
    bench)
      timeout 1500 python bench.py ${REST//+/ } --detail-out "$OUT/bench_${n}_detail.json" > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; echo "bench exit $?"
      wc -c "$OUT/bench_$n.json"; cat "$OUT/bench_$n.json"; tail -2 "$OUT/bench_$n.err" | cut -c1-300 ;;
    trace)
      NAME=${REST%%:*}; ARGS=""; [[ "$REST" == *:* ]] && ARGS=${REST#*:}
      HEAD=${HEAD:-24} timeout 900 bash scripts/prof_bench.sh "$OUT/$NAME" ${ARGS//+/ } ;;
    evidence) bash scripts/evidence.sh "$OUT/evidence" ${REST//+/ } ;;
    *) echo "unknown stage $STAGE" ;;
  esac
  n=$((n + 1))
done
echo "gpu_session $TAG done"
