#!/bin/bash
# r02 GPU call 53: end of round: whole GPU suite + smoke, then bench.py with the DRIVER's flags (--gpus 1 --steps 20 --warmup 5)
set -u
OUT=gpurun_out/r02_c53; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
T0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/e.log
T1=$(date +%s)
echo "bench wall seconds: $((T1 - T0))"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02_c53/bench_driver_flags.json")); r = j["roofline"]
print(round(j["value"]), round(j["ms_per_step"], 1), j["steps"], j["warmup"], round(r["achieved"], 2), round(r["frac"], 4), r["launches"], r["gemm"]["tflops"], r["fraction_of_step_time"], j["host_call_ms_per_step"])
rp = j.get("reduced_precision", {})
print("bf16", round(rp.get("value", 0)), round(rp.get("ms_per_step", 0), 1), rp.get("roofline", {}).get("achieved"), rp.get("roofline", {}).get("fraction_of_step_time"))
print("cpu", j.get("cpu_baseline", {}).get("value"), j.get("parity_vs_reference_golden", {}).get("final_cloud_max_abs"))
PY
echo "r02 call 53 done"
