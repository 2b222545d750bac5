#!/bin/bash
# r02 GPU call 50: fp32 GEGLU epilogue: exact erff vs the packed-pair 1.5e-7 erfc (tuning key 9): ff1 time, headline, parity vs the reference golden
set -u
OUT=gpurun_out/r02_c50; mkdir -p $OUT
export TMPDIR=/tmp
for T in 0 1; do
  timeout 300 python scripts/kernel_bench.py --only gemm --tuning 9=$T 2>> $OUT/kb.err | sed "s/^/{\"geglu_fast\": $T, \"row\": /; s/$/}/" >> $OUT/kb.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_c50/kb.jsonl"):
    try: j = json.loads(l)
    except Exception: continue
    r = j["row"]; print(j["geglu_fast"], r.get("kernel", "")[:40], round(r.get("ms"), 4), round(r.get("tflops"), 1))
PY
timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1 --tuning 9=1 > $OUT/bench_f32_fast_geglu.json 2> $OUT/e.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c50/bench_f32_fast_geglu.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), r["gemm"]["tflops"], j["parity_vs_reference_golden"])
PY
echo "r02 call 50 done"
