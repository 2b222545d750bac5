#!/bin/bash
# round 5, GPU call 1: the split-precision kernels -- parity (kernel level, golden fixtures), micro-benchmark, a short bench line
set -u
OUT=gpurun_out/r05_c1
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_x2_gpu.py -q -s > "$OUT/pytest_x2.log" 2>&1; echo "x2 tests exit $?" | tee -a "$OUT/pytest_x2.log"
grep -E "^(FAILED|ERROR)|passed|failed|x2 |velocity vs" "$OUT/pytest_x2.log" | tail -40
timeout 400 python scripts/x2_bench.py > "$OUT/x2_bench.jsonl" 2> "$OUT/x2_bench.err"; echo "x2 bench exit $?"; cat "$OUT/x2_bench.jsonl"; tail -5 "$OUT/x2_bench.err"
timeout 900 python -m pytest tests/test_sample_gpu.py -q -k "golden" > "$OUT/pytest_golden.log" 2>&1; echo "golden exit $?"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_golden.log" | tail -20
timeout 900 python -m pytest tests/test_headline_gpu.py -q -k "split or c4_geometry" > "$OUT/pytest_headline_x2.log" 2>&1; echo "headline exit $?"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_headline_x2.log" | tail -20
timeout 900 python bench.py --steps 1 --warmup 1 --no-ragged --gamma-scale 0 > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"; echo "bench exit $?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r05_c1/bench_quick.json"))
    print({k: j.get(k) for k in ("value", "ms_per_step", "points_per_s_by_mode", "instrumented_over_clean")})
    for k in ("emulated_fp32", "reduced_precision", "f16"):
        l = j.get(k) or {}
        print(k, {x: l.get(x) for x in ("value", "ms_per_step", "deviation_from_fp32_path", "parity_vs_reference_golden")}, (l.get("roofline") or {}).get("frac"), ((l.get("roofline") or {}).get("gemm")))
    print("se3", j.get("se3_vs_cpu_oracle")); print("hbm", j.get("hbm_kernels"))
    print("roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "instrumented_over_clean")})
except Exception as e:
    print("no bench json:", e)
PY
tail -5 "$OUT/bench_quick.err"
echo "r05 call 1 done"
