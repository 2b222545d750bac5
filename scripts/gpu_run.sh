#!/bin/bash
# One GPU-box session: parity tests, per-kernel variant sweep, the bench line, and a rocprofv3 kernel trace of the
# same bench command.  Usage (from the repo root, through gpurun):  bash scripts/gpu_run.sh <tag> [stages...]
# stages: tests sweep h16 bench prof pmc   (default: tests sweep bench prof); PYTEST_ARGS overrides "-x -q"
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
if has h16; then
  : > "$OUT/kernel_bench_h16.jsonl"
  for v in 0 1 2 3 5; do
    timeout 300 python scripts/kernel_bench.py --dtype bfloat16 --only gemm --h16-gemm-variant $v >> "$OUT/kernel_bench_h16.jsonl" 2>> "$OUT/kernel_bench.err"
  done
  timeout 300 python scripts/kernel_bench.py --dtype bfloat16 --only attention >> "$OUT/kernel_bench_h16.jsonl" 2>> "$OUT/kernel_bench.err"
  timeout 300 python scripts/kernel_bench.py --dtype float16 >> "$OUT/kernel_bench_h16.jsonl" 2>> "$OUT/kernel_bench.err"
  cat "$OUT/kernel_bench_h16.jsonl"; tail -5 "$OUT/kernel_bench.err"
fi
if has sweep; then
  : > "$OUT/kernel_bench.jsonl"
  for v in 2 4 8 16; do
    timeout 300 python scripts/kernel_bench.py --only gemm --gemm-variant $v >> "$OUT/kernel_bench.jsonl" 2>> "$OUT/kernel_bench.err"
  done
  for v in 1 3 5; do
    timeout 300 python scripts/kernel_bench.py --only attention --attn-variant $v >> "$OUT/kernel_bench.jsonl" 2>> "$OUT/kernel_bench.err"
  done
  timeout 300 python scripts/kernel_bench.py > "$OUT/kernel_bench_default.jsonl" 2>> "$OUT/kernel_bench.err"
  cat "$OUT/kernel_bench.jsonl"
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
  for DT in float32 bfloat16; do
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$GRAFT_REPO_ROOT/$OUT/pmc_${DT}_$c" -o pmc -- \
          python "$GRAFT_REPO_ROOT/scripts/kernel_bench.py" --dtype $DT --only attention --pmc > "$GRAFT_REPO_ROOT/$OUT/pmc_${DT}_$c.log" 2>&1 )
      DB=$(find "$OUT/pmc_${DT}_$c" -name '*.db' | head -1)
      if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" --pmc | grep -E "PMC.*attention" | sed "s/^/$DT /" >> "$OUT/pmc_traffic.txt"; fi
      find "$OUT/pmc_${DT}_$c" -name '*.db' -delete
    done
  done
  cat "$OUT/pmc_traffic.txt"
fi
echo "gpu_run $TAG done"
