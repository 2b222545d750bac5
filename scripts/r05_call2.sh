#!/bin/bash
# round 5, GPU call 2: the software-pipelined split-precision attention -- parity, A/B against the plain kernel, whole call with either
set -u
OUT=gpurun_out/r05_c2
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_x2_gpu.py -q -s -k attention > "$OUT/pytest_x2_attn.log" 2>&1; echo "x2 attention tests exit $?"
grep -E "^(FAILED|ERROR)|passed|failed|x2 attention" "$OUT/pytest_x2_attn.log" | tail -30
timeout 300 python scripts/x2_bench.py --only attn > "$OUT/x2_bench_attn.jsonl" 2> "$OUT/x2_bench.err"; echo "x2 bench exit $?"; cat "$OUT/x2_bench_attn.jsonl"; tail -3 "$OUT/x2_bench.err"
for V in 1 2; do
  timeout 600 python bench.py --dtype float32x2 --steps 2 --warmup 1 --no-ragged --gamma-scale 0 --no-cpu-baseline --no-secondary --tuning 16=$V > "$OUT/bench_x2_v$V.json" 2> "$OUT/bench_x2_v$V.err"; echo "bench v$V exit $?"
  python - "$OUT/bench_x2_v$V.json" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    r = j.get("roofline") or {}
    print({k: j.get(k) for k in ("value", "ms_per_step", "dtype")}, {k: r.get(k) for k in ("achieved", "frac", "avg_launch_ms")}, r.get("gemm"), r.get("fraction_of_step_time"))
except Exception as e:
    print("no json", e)
PY
  tail -3 "$OUT/bench_x2_v$V.err"
done
echo "r05 call 2 done"
