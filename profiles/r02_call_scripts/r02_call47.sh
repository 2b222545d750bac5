#!/bin/bash
# r02 GPU call 47: final state of the round: whole GPU suite, smoke, the default bench line, the driver-flag bench (--steps 20 --warmup 5 is ~7 min: run with 5 / 1)
set -u
OUT=gpurun_out/r02_c47; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/e.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c47/bench_default.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), round(r["achieved"], 2), round(r["frac"], 4), r["gemm"]["tflops"], r["fraction_of_step_time"])
rp = j.get("reduced_precision", {})
print("bf16", round(rp.get("value", 0)), round(rp.get("ms_per_step", 0), 1), rp.get("roofline", {}).get("achieved"), rp.get("roofline", {}).get("gemm"))
print("cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("cores"))
PY
echo "r02 call 47 done"
