#!/bin/bash
# One GPU-box session: parity tests, per-kernel variant sweep, the bench line, and a rocprofv3 kernel trace of the
# same bench command.  Usage (from the repo root, through gpurun):  bash scripts/gpu_run.sh <tag> [stages...]
# stages: tests sweep bench prof pmc   (default: tests sweep bench prof); PYTEST_ARGS overrides "-x -q"
set -u
TAG=${1:-run}; shift || true
STAGES=${*:-"tests sweep bench prof"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }

if has tests; then
  timeout 1500 python -m pytest tests -m gpu ${PYTEST_ARGS:--x -q} > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
  grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | tail -40
fi
if has sweep; then
  timeout 300 python scripts/kernel_bench.py > "$OUT/kernel_bench_f32.jsonl" 2>> "$OUT/kernel_bench.err"
  timeout 300 python scripts/kernel_bench.py --dtype bfloat16 --bounded 2 > "$OUT/kernel_bench_bf16.jsonl" 2>> "$OUT/kernel_bench.err"
  timeout 300 python scripts/kernel_bench.py --dtype float16 > "$OUT/kernel_bench_f16.jsonl" 2>> "$OUT/kernel_bench.err"
  ( cd scripts && timeout 200 python gemm_epi_bench.py > "../$OUT/gemm_epi_bench.jsonl" 2>> "../$OUT/kernel_bench.err" )
  cat "$OUT/kernel_bench_f32.jsonl" "$OUT/kernel_bench_bf16.jsonl"
fi
if has bench; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"; cat "$OUT/bench.json"
fi
if has prof; then
  for DT in float32 bfloat16; do
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/prof_bench_$DT" -o bench -- \
        python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --dtype $DT > "$GRAFT_REPO_ROOT/$OUT/prof_bench_$DT.log" 2>&1 )
    DB=$(find "$OUT/prof_bench_$DT" -name '*.db' | head -1)
    if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" > "$OUT/bench_${DT}_kernel_trace_stats.txt"; head -16 "$OUT/bench_${DT}_kernel_trace_stats.txt"; fi
    find "$OUT/prof_bench_$DT" -name '*.db' -delete
  done
fi
if has pmc; then
  bash scripts/pmc_passes.sh "$OUT/pmc"
fi
echo "gpu_run $TAG done"
