#!/bin/bash
# r02 GPU call 14: the bench lines VERDICT r01 asked for (rigidity off, fp16, bf16 headline) + rocprofv3 kernel traces of one sample call
set -u
OUT=gpurun_out/r02_c14; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python bench.py --rigidity 0 --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_f32_rigidity_off.json 2> $OUT/e1.log
timeout 300 python bench.py --dtype float16 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_f16.json 2> $OUT/e2.log
timeout 300 python bench.py --dtype bfloat16 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_bf16.json 2> $OUT/e3.log
python - <<'PY'
import json
for f in ("bench_f32_rigidity_off", "bench_f16", "bench_bf16"):
    j = json.load(open(f"gpurun_out/r02_c14/{f}.json")); r = j["roofline"]
    print(f, round(j["value"]), round(j["ms_per_step"], 1), j["dtype"], round(r["achieved"], 1), round(r["frac"], 3), r["gemm"]["tflops"], r["fraction_of_step_time"])
PY
for DT in float32 bfloat16; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/prof_bench_$DT" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --dtype $DT > "$GRAFT_REPO_ROOT/$OUT/prof_bench_$DT.log" 2>&1 )
  DB=$(find "$OUT/prof_bench_$DT" -name '*.db' | head -1)
  if [ -n "$DB" ]; then python scripts/rocpd_summary.py "$DB" > "$OUT/bench_${DT}_kernel_trace_stats.txt"; head -22 "$OUT/bench_${DT}_kernel_trace_stats.txt"; fi
  find "$OUT/prof_bench_$DT" -name '*.db' -delete
done
echo "r02 call 14 done"
