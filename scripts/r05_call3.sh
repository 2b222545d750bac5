#!/bin/bash
# round 5, GPU call 3: evidence for the split-precision mode (SQ counters, PMC traffic, kernel trace of one sampling call), the new
# round-5 tests (constructor switches, checkpoint script, NaN propagation), few-token latency of every mode
set -u
OUT=gpurun_out/r05_c3
mkdir -p "$OUT"
export TMPDIR=/tmp
# 1. matrix-pipe utilisation and clock of the x2 kernels on the model path (one flow step)
DTYPES="float32x2" bash scripts/mfma_util.sh "$OUT/mfma" > "$OUT/mfma_stdout.txt" 2>&1; cat "$OUT/mfma/mfma_utilisation.txt" | head -20
# 2. HBM traffic of the x2 attention kernel, model path, separate FETCH / WRITE passes
: > "$OUT/pmc_traffic_x2.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  D=$(mktemp -d /tmp/pmc.XXXXXX)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$D" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --dtype float32x2 --steps 1 --warmup 0 --flow-steps 1 \
      --no-cpu-baseline --no-secondary --no-ragged --no-profile --gamma-scale 0 > "$GRAFT_REPO_ROOT/$OUT/pmc_$c.log" 2>&1 )
  DB=$(find "$D" -name '*.db' | head -1)
  if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" --pmc | grep -E "^PMC.*(attention|gemm_h16|layernorm_x2)" | sed "s/^/x2_model_path /" >> "$OUT/pmc_traffic_x2.txt"; else echo "$c: no db" >> "$OUT/pmc_traffic_x2.txt"; fi
  rm -rf "$D"
done
cat "$OUT/pmc_traffic_x2.txt"
# 3. kernel trace of one 20-step sampling call in split precision
HEAD=16 bash scripts/prof_bench.sh "$OUT/x2_kernel_trace" --dtype float32x2
# 4. new tests
timeout 900 python -m pytest tests/test_sample_gpu.py -q -s -k "switches or checkpoint" > "$OUT/pytest_switches.log" 2>&1; echo "switch tests exit $?"
grep -E "^(FAILED|ERROR)|passed|failed|velocity" "$OUT/pytest_switches.log" | tail -20
timeout 600 python -m pytest tests/test_h16_gpu.py tests/test_x2_gpu.py -q -k "fp16_residual or attention" > "$OUT/pytest_misc.log" 2>&1; echo "misc tests exit $?"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_misc.log" | tail -10
# 5. few-token latency (configs[0] geometry: 1 pair x 2 x 1024, 10 steps) in every mode, un-instrumented
for DT in float32 float32x2 bfloat16; do
  timeout 300 python bench.py --dtype $DT --batch 1 --points 1024 --flow-steps 10 --steps 20 --warmup 5 --no-ragged --gamma-scale 0 --no-cpu-baseline --no-secondary --no-profile > "$OUT/bench_c0_$DT.json" 2> "$OUT/bench_c0_$DT.err"
  python - "$OUT/bench_c0_$DT.json" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(j["dtype"], "configs[0] geometry:", round(j["ms_per_step"], 2), "ms per call,", round(j["value"]), "points/s")
except Exception as e:
    print("no json", e)
PY
done
echo "r05 call 3 done"
